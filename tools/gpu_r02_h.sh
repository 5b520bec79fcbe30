#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02h
mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 ) > $O/gputests.log 2>&1
grep -E "passed|failed" $O/gputests.log
BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline >> $O/bench_dist.jsonl 2>> $O/bench_dist.err
BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --slots 7 >> $O/bench_dist.jsonl 2>> $O/bench_dist.err
BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --steps 20 --warmup 3 --groups 500000 --no-cpu-baseline >> $O/bench_dist.jsonl 2>> $O/bench_dist.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02h/bench_*.json*")):
    for line in open(f):
        if not line.startswith("{"): continue
        d=json.loads(line)
        print(f.split("/")[-1], d["n_gpus"], d["config"]["peer_slots"], round(d["value"]/1e9,2), "G/s", round(d["ms_per_step"]*1e3,1), "us/step", "host", d["config"]["host_issue_us_per_step"], round(d["roofline"]["frac"],3), (d["config"].get("publication") or {}).get("publications"))
PY
