#!/bin/bash
# round 2, first GPU visit: full gpu test suite, headline bench, config lines, cost of the election branch, rocprof
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/r02a
mkdir -p $O
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 ) > $O/gputests.log 2>&1
tail -5 $O/gputests.log
timeout 600 python bench.py --steps 50 --warmup 5 > $O/bench_c2.json 2> $O/bench_c2.err
tail -c 600 $O/bench_c2.json
for cfg in "--workload 3" "--slots 7" "--slots 3" "--workload 5" "--workload 5 --one-engine"; do
  timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras $cfg >> $O/bench_configs.jsonl 2>> $O/bench_configs.err
done
for cfg in "" "--slots 7" "--slots 3"; do
  RG_LIB_PATH=$PWD/raft_rs_amd/libraftgroups_noelect.so timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras $cfg >> $O/bench_noelect.jsonl 2>> $O/bench_noelect.err
done
python - <<'PY'
import json
for f in ("bench_configs.jsonl","bench_noelect.jsonl"):
    for line in open("gpurun_out/r02a/"+f):
        try: d=json.loads(line)
        except Exception: continue
        print(f, d["config"]["workload_id"], d["config"]["peer_slots"], [e["slots"] for e in d["config"]["engines"]], round(d["value"]/1e9,2), "G/s", round(d["roofline"]["avg_launch_us"],1), "us", round(d["roofline"]["frac"],3), d["config"].get("rejects_per_group"))
PY
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof -o r02a -- python $OLDPWD/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OLDPWD/$O/prof_bench.json 2> $OLDPWD/$O/prof_bench.err
cd $OLDPWD
ls -R $O/prof | head -20
