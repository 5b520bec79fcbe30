#!/bin/bash
# round 4, call d: the class kernel with per-body kernarg reads, k_tick_send with per-phase kernarg reads (4 waves at P = 5):
# the suites that touch them, config 5 in three layouts twice, the send stage in both forms, SQ counters
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04d
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_full_size_gpu.py tests/test_sendstage_gpu.py tests/test_api_sequences_gpu.py tests/test_sparse_path_gpu.py tests/test_graph_capture_gpu.py -m gpu -x -q 2>&1 | grep -v "^E    .*match\[" | tail -12 > gpurun_out/r04d/tests.txt
tail -4 gpurun_out/r04d/tests.txt
O=gpurun_out/r04d/bench.jsonl; : > $O
run() { timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 40 "$@" 2>gpurun_out/r04d/err.txt | tail -1 >> $O; }
for rep in 1 2; do
  run --workload 5 --slots 7 --sorted
  run --workload 5
  run --inflights 256
  run --inflights 256 --fused-send
done
run --workload 5 --slots 7 --sorted --groups 8000000 --steps 12
run --workload 5 --groups 8000000 --steps 12
run --inflights 256 --fused-send --groups 8000000 --steps 12
run --inflights 256 --fused-send --slots 3
python - <<'PY'
import json
for i,l in enumerate(open('gpurun_out/r04d/bench.jsonl')):
    try:
        d=json.loads(l); r=d['roofline']; c=d['config']
        print('%2d %8d %-70s | %.2f G/s  %.1f us  frac %.3f %s' % (i, c['groups_per_gpu'], c['workload'][:70] if c['workload_id']!=5 else c['workload'][75:145], d['value']/1e9, d['ms_per_step']*1e3, r['frac'], r['kernel']))
    except Exception as e: print('??', l[:200])
PY
tail -3 gpurun_out/r04d/err.txt
tools/pmc_sq_tail.sh r04d_c5sorted 20 --workload 5 --slots 7 --sorted > /dev/null 2>&1
tools/pmc_sq_tail.sh r04d_ticksend 20 --inflights 256 --fused-send > /dev/null 2>&1
cat gpurun_out/pmct_r04d_c5sorted.txt gpurun_out/pmct_r04d_ticksend.txt | grep -v "^    SQ_INSTS_LDS\|ACTIVE_INST_LDS"
