#!/bin/bash
# Round 4, final evidence on build eaa62f1: the whole GPU suite, smoke(), the driver's bench command, the secondary configurations
set -u
cd $GRAFT_REPO_ROOT
R=$PWD
O=$R/gpurun_out/r04v
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
echo eaa62f1 > $O/build_commit.txt
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep "passed\|failed\|error" | tail -3 > $O/tests_gpu.txt
cat $O/tests_gpu.txt
timeout 300 python -c "import __graft_entry__ as e; e.smoke(); print('SMOKE_OK')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
T0=$(date +%s%N)
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
T1=$(date +%s%N)
echo "python bench.py: $(( (T1 - T0) / 1000000 )) ms wall" > $O/bench_n1_wall.txt
cat $O/bench_n1_wall.txt
for cfg in "--slots 3" "--groups 2000000" "--groups 2400000 --steps 30" "--groups 4000000 --steps 20" "--workload 5 --slots 7 --sorted" "--workload 5 --slots 7 --sorted --groups 8000000 --steps 12" "--workload 5 --groups 8000000 --steps 12" "--workload 5 --slots 7 --sorted --groups 100000" "--workload 5 --groups 100000" "--fuse 4" "--fuse 8" "--fuse 8 --groups 8000000 --steps 16" "--workload 5 --fuse 4"; do
  timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras $cfg >> $O/bench_other_configs.jsonl 2>> $O/bench_other.err
done
for cfg in "" "--slots 7"; do
  BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline $cfg >> $O/bench_dist_ws1.jsonl 2>> $O/bench_dist.err
done
BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 20 --warmup 3 --groups 500000 --no-cpu-baseline >> $O/bench_dist_share2.jsonl 2>> $O/bench_dist.err
ls $O
python - <<"PY" || true
import json
d=json.loads(open('gpurun_out/r04v/bench_n1.json').read().strip().splitlines()[-1])
print(d['value']/1e9, d['ms_per_step']*1e3, d['roofline']['frac'])
for k,v in d['roofline']['by_config'].items(): print(k, v.get('frac'), v.get('us'))
for f in ('gpurun_out/r04v/bench_dist_ws1.jsonl','gpurun_out/r04v/bench_dist_share2.jsonl'):
    for l in open(f):
        if not l.startswith('{"metric"'): continue
        d=json.loads(l); c=d['config']; print(f[-22:], d['n_gpus'], round(d['ms_per_step']*1e3,1), c.get('publish_every'), c.get('publish_every_auto'))
PY
