#!/bin/bash
# round 4, call a: the running-quorum replay (RgRunningQuorum) and the class-placed one-launch tick (k_tick_classes):
# parity tests, then config 5 in its three layouts + the steady configs, then SQ counters of the c5 layouts
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04a
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_full_size_gpu.py tests/test_wire_format.py tests/test_api_sequences_gpu.py tests/test_sendstage_gpu.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r04a/tests.txt
cat gpurun_out/r04a/tests.txt
O=gpurun_out/r04a/bench.jsonl; : > $O
run() { timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 40 "$@" 2>gpurun_out/r04a/err.txt | tail -1 >> $O; }
run
run --workload 5 --slots 7 --sorted
run --workload 5
run --workload 5 --slots 7 --one-engine
run --slots 7
run --groups 8000000 --steps 20
python - <<'PY'
import json
for l in open('gpurun_out/r04a/bench.jsonl'):
    try:
        d=json.loads(l); r=d['roofline']; c=d['config']
        print('%-100s | %.2f G/s  %.1f us  frac %.3f %s' % (c['workload'][:100], d['value']/1e9, d['ms_per_step']*1e3, r['frac'], r['kernel']))
    except Exception as e: print('??', l[:200])
PY
tail -3 gpurun_out/r04a/err.txt
tools/pmc_sq_tail.sh r04a_c5sorted 20 --workload 5 --slots 7 --sorted > /dev/null 2>&1
tools/pmc_sq_tail.sh r04a_c5classes 20 --workload 5 > /dev/null 2>&1
tools/pmc_sq_tail.sh r04a_c5one 20 --workload 5 --slots 7 --one-engine > /dev/null 2>&1
cat gpurun_out/pmct_r04a_c5sorted.txt gpurun_out/pmct_r04a_c5classes.txt gpurun_out/pmct_r04a_c5one.txt | grep -v "^    SQ_INSTS_LDS\|ACTIVE_INST_LDS"
