#!/bin/bash
# round 4, call e: elections file their run inside become_leader (no tail loads): suites, config 5 layouts x2, steady configs, send
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04e
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^E    .*match\[" | tail -12 > gpurun_out/r04e/tests.txt
tail -4 gpurun_out/r04e/tests.txt
O=gpurun_out/r04e/bench.jsonl; : > $O
run() { timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 40 "$@" 2>gpurun_out/r04e/err.txt | tail -1 >> $O; }
for rep in 1 2; do
  run --workload 5 --slots 7 --sorted
  run --workload 5
  run --workload 5 --slots 7 --one-engine
done
run
run --slots 7
run --slots 3
run --groups 8000000 --steps 16
run --inflights 256
run --inflights 256 --fused-send
python - <<'PY'
import json
for i,l in enumerate(open('gpurun_out/r04e/bench.jsonl')):
    try:
        d=json.loads(l); r=d['roofline']; c=d['config']
        print('%2d %8d %-70s | %.2f G/s  %.1f us  frac %.3f %s' % (i, c['groups_per_gpu'], c['workload'][:70] if c['workload_id']!=5 else c['workload'][75:145], d['value']/1e9, d['ms_per_step']*1e3, r['frac'], r['kernel']))
    except Exception as e: print('??', l[:200])
PY
tail -3 gpurun_out/r04e/err.txt
tools/pmc_sq_tail.sh r04e_c5sorted 20 --workload 5 --slots 7 --sorted > /dev/null 2>&1
cat gpurun_out/pmct_r04e_c5sorted.txt | grep -v "^    SQ_INSTS_LDS\|ACTIVE_INST_LDS"
