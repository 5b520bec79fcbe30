#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02d
mkdir -p $O
export TMPDIR=/tmp
for cfg in "" "--publish-every 4"; do
BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline $cfg >> $O/bench_dist_ws1.jsonl 2>> $O/bench_dist.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02d/bench_*.json*")):
    for line in open(f):
        if not line.startswith("{"): continue
        d=json.loads(line)
        print(round(d["ms_per_step"]*1e3,1), "us/step", "host", d["config"]["host_issue_us_per_step"], d["config"].get("publication"))
PY
