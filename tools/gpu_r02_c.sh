#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02c
mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_publish_gpu.py tests/test_parity_gpu.py tests/test_scenarios.py -m gpu -q -p no:cacheprovider --timeout 600 ) > $O/gputests.log 2>&1
tail -4 $O/gputests.log
for cfg in "" "--workload 5" "--workload 5 --slots 7 --one-engine" "--slots 3"; do
  timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras $cfg >> $O/bench_configs.jsonl 2>> $O/bench_configs.err
done
for cfg in "" "--publish-every 4" "--slots 7"; do
BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline $cfg >> $O/bench_dist_ws1.jsonl 2>> $O/bench_dist.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02c/bench_*.json*")):
    for line in open(f):
        if not line.startswith("{"): continue
        d=json.loads(line)
        print(f.split("/")[-1], d["n_gpus"], d["config"]["workload_id"], d["config"]["peer_slots"], [e["slots"] for e in d["config"]["engines"]], round(d["value"]/1e9,2), "G/s", round(d["ms_per_step"]*1e3,1), "us/step", "host", d["config"]["host_issue_us_per_step"], round(d["roofline"]["frac"],3), d["config"].get("rejects_per_group"), (d["config"].get("publication") or {}).get("publications"))
PY
