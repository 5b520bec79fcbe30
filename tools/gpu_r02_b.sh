#!/bin/bash
# round 2, second GPU visit: gpu tests (publication path, elections in registers), config lines, dist path at world 1
# (RCCL) and with two ranks sharing the GPU (gloo transport), rocprof csv of the headline bench
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02b
mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 ) > $O/gputests.log 2>&1
tail -5 $O/gputests.log
timeout 600 python bench.py --steps 50 --warmup 5 > $O/bench_c2.json 2> $O/bench_c2.err
for cfg in "--workload 3" "--slots 7" "--slots 3" "--workload 5" "--workload 5 --slots 7 --one-engine"; do
  timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras $cfg >> $O/bench_configs.jsonl 2>> $O/bench_configs.err
done
BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline >> $O/bench_dist_ws1.jsonl 2>> $O/bench_dist.err
BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --slots 7 >> $O/bench_dist_ws1.jsonl 2>> $O/bench_dist.err
BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --steps 20 --warmup 3 --groups 500000 --no-cpu-baseline >> $O/bench_share2.jsonl 2>> $O/bench_dist.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02b/bench_*.json*")):
    for line in open(f):
        if not line.startswith("{"): continue
        d=json.loads(line)
        print(f.split("/")[-1], d["n_gpus"], d["config"]["workload_id"], d["config"]["peer_slots"], [e["slots"] for e in d["config"]["engines"]], round(d["value"]/1e9,2), "G/s", round(d["ms_per_step"]*1e3,1), "us/step", round(d["roofline"]["frac"],3), d["config"].get("rejects_per_group"), d["config"].get("publication"))
PY
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof -o r02b -- python $OLDPWD/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OLDPWD/$O/prof_bench.json 2> $OLDPWD/$O/prof_bench.err
cd $OLDPWD
ls -R $O/prof | head -20
