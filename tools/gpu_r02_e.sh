#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02e
mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_publish_gpu.py tests/test_parity_gpu.py tests/test_full_size_gpu.py -m gpu -q -p no:cacheprovider --timeout 600 -k "publish or golden or recompute or bench or fused" ) > $O/gputests.log 2>&1
tail -4 $O/gputests.log
timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err
for cfg in "" "--publish-every 4" "--slots 7"; do
BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline $cfg >> $O/bench_dist_ws1.jsonl 2>> $O/bench_dist.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02e/bench_*.json*")):
    for line in open(f):
        if not line.startswith("{"): continue
        d=json.loads(line)
        print(f.split("/")[-1], d["config"]["peer_slots"], round(d["value"]/1e9,2), "G/s", round(d["ms_per_step"]*1e3,1), "us/step", "host", d["config"]["host_issue_us_per_step"], round(d["roofline"]["frac"],3), (d["config"].get("publication") or {}).get("publications"))
        if "recompute_only" in d: print("  recompute", d["recompute_only"]["us_per_launch"], d["recompute_only"]["one_group_per_lane_us"], d["recompute_only"]["roofline"]["frac"], "ooc", d["out_of_cache"]["us_per_launch"], d["out_of_cache"]["roofline"]["frac"])
PY
