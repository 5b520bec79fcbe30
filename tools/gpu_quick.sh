#!/bin/bash
# quick GPU check of a tick-kernel change: targeted parity tests, then the headline and the rollover configs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_full_size_gpu.py tests/test_sparse_path_gpu.py tests/test_api_sequences_gpu.py tests/test_scenarios.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/quick_tests.txt
cat gpurun_out/quick_tests.txt
O=gpurun_out/quick.jsonl; : > $O
run() { timeout 300 python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | tail -1 >> $O; }
run
run --workload 5
run --workload 5 --slots 7 --one-engine
run --slots 7
run --slots 3
run --groups 8000000 --steps 20
python - <<'PY'
import json
for l in open('gpurun_out/quick.jsonl'):
    try:
        d=json.loads(l); r=d['roofline']; c=d['config']
        print('%-62s G=%d P=%d | %.2f G/s  %.1f us  frac %.3f | %s' % (c['workload'][:62], c['groups_per_gpu'], c['peer_slots'], d['value']/1e9, d['ms_per_step']*1e3, r['frac'], c['sharding'][:40]))
    except Exception as e: print('??', l[:80])
PY
