#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02g
mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_publish_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3 )
for dbg in 0 1 4; do
RG_PUB_DEBUG=$dbg BENCH_FORCE_DIST=1 BENCH_SKIP_VERIFY=1 timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline >> $O/bench_dist_dbg.jsonl 2>> $O/bench_dist.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02g/bench_*.json*")):
    for line in open(f):
        if not line.startswith("{"): continue
        d=json.loads(line)
        print(f.split("/")[-1], d["config"]["peer_slots"], round(d["value"]/1e9,2), "G/s", round(d["ms_per_step"]*1e3,1), "us/step", "host", d["config"]["host_issue_us_per_step"], round(d["roofline"]["frac"],3), (d["config"].get("publication") or {}).get("publications"))
        if "recompute_only" in d: print("  recompute", d["recompute_only"]["us_per_launch"], d["recompute_only"]["one_group_per_lane_us"], d["recompute_only"]["roofline"]["frac"], "ooc", d["out_of_cache"]["us_per_launch"], d["out_of_cache"]["roofline"]["frac"])
PY
grep -v amdgpu $O/*.err | tail -5
