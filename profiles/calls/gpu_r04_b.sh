#!/bin/bash
# round 4, call b: (1) the new class tests; (2) where does the class-placed one launch stand against three size-class engines at
# other sizes (100 k / 1 M / 4 M / 8 M groups); (3) what the term-table push at the tail of an electing wave costs (measurement
# build -DRG_NO_PUSH); (4) HBM traffic of the class-placed layout
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04b
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "class_placed or interleaved_sizes or sorted_mixed" 2>&1 | grep -v "^E    .*match\[" | tail -30 > gpurun_out/r04b/tests.txt
cat gpurun_out/r04b/tests.txt | tail -5
O=gpurun_out/r04b/bench.jsonl; : > $O
run() { timeout 300 python bench.py --no-cpu-baseline --no-extras "$@" 2>gpurun_out/r04b/err.txt | tail -1 >> $O; }
for G in 100000 1000000 4000000 8000000; do
  S=40; [ $G -ge 4000000 ] && S=12
  run --workload 5 --slots 7 --sorted --groups $G --steps $S
  run --workload 5 --groups $G --steps $S
  run --workload 5 --slots 7 --one-engine --groups $G --steps $S
done
export RG_LIB_PATH=$GRAFT_REPO_ROOT/raft_rs_amd/libraftgroups_nopush.so
run --workload 5 --slots 7 --sorted --steps 40
run --workload 5 --steps 40
run --workload 5 --slots 7 --one-engine --steps 40
unset RG_LIB_PATH
python - <<'PY'
import json
for i,l in enumerate(open('gpurun_out/r04b/bench.jsonl')):
    try:
        d=json.loads(l); r=d['roofline']; c=d['config']
        print('%2d %8d %-58s | %.2f G/s  %.1f us  frac %.3f %s' % (i, c['groups_per_gpu'], c['workload'][75:133], d['value']/1e9, d['ms_per_step']*1e3, r['frac'], r['kernel']))
    except Exception as e: print('??', l[:200])
PY
tail -3 gpurun_out/r04b/err.txt
tools/pmc_traffic.sh "5:1000000:7:sorted" 30 --workload 5 --slots 7 --sorted > /dev/null 2>&1
cat gpurun_out/traffic_5_1000000_7_sorted.json | head -20
