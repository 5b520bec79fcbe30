#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02k
mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_full_size_gpu.py tests/test_c_driver.py -m gpu -q -p no:cacheprovider --timeout 600 -k "stream or split or c_driver" 2>&1 | tail -3 )
for cfg in "--workload 5" "--workload 5 --variant 5" "--workload 5 --slots 7 --one-engine" "--workload 5 --slots 7 --one-engine --variant 5" "" "--variant 5" "--slots 7 --variant 5" "--slots 3 --variant 5"; do
  timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras $cfg >> $O/bench.jsonl 2>> $O/err
done
python - <<'PY'
import json,glob
for line in open("gpurun_out/r02k/bench.jsonl"):
    if not line.startswith("{"): continue
    d=json.loads(line); c=d["config"]
    print(c["workload_id"], c["peer_slots"], [e["slots"] for e in c["engines"]], c["kernel_variant"], "|", round(d["value"]/1e9,2), "G/s", round(d["ms_per_step"]*1e3,1), "us/step", round(d["roofline"]["frac"],3))
PY
grep -v amdgpu $O/err | tail -3
