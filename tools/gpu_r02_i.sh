#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02i
mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_full_size_gpu.py -m gpu -q -p no:cacheprovider --timeout 600 -k "stream or golden or recompute" 2>&1 | tail -3 )
for v in 0 2 4; do
  timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras --variant $v >> $O/bench_variants.jsonl 2>> $O/err
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --variant $v --groups 8000000 >> $O/bench_variants.jsonl 2>> $O/err
done
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline >> $O/bench_c2.jsonl 2>> $O/err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02i/bench_*.json*")):
    for line in open(f):
        if not line.startswith("{"): continue
        d=json.loads(line)
        print(f.split("/")[-1], d["config"]["kernel_variant"], d["config"]["groups_per_gpu"], round(d["value"]/1e9,2), "G/s", round(d["roofline"]["avg_launch_us"],1), "us", round(d["roofline"]["frac"],3))
        if "recompute_only" in d: print("  recompute", d["recompute_only"]["us_per_launch"], d["recompute_only"]["one_group_per_lane_us"], d["recompute_only"]["roofline"]["frac"], "ooc", d["out_of_cache"]["us_per_launch"], d["out_of_cache"]["roofline"]["frac"])
PY
grep -v amdgpu $O/err | tail -3
