#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03g; mkdir -p $O
( timeout 2300 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) | tee $O/gpu_tests.txt
J=$O/bench.jsonl; : > $J
run() { tag=$1; shift; echo "## $tag" >> $J; timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 5 "$@" 2>/dev/null | tail -1 >> $J; }
run "c5 size-class lane" --workload 5
run "c5 one-engine lane" --workload 5 --slots 7 --one-engine
run "c2 lane"
run "c4 shard lane" --slots 7
run "c3 joint" --workload 3
run "c2 8M lane" --groups 8000000 --steps 15
run "c5 size-class lane again" --workload 5
run "c5 fused x4" --workload 5 --fuse 4
run "c2 fused x8" --fuse 8
python - <<'PY' | tee $O/bench_summary.txt
import json
tag=None
for l in open('gpurun_out/r03g/bench.jsonl'):
    if l.startswith('##'): tag=l[2:].strip(); continue
    try:
        d=json.loads(l); r=d['roofline']
        print('%-36s | %.2f G/s  %.1f us/step  frac %.3f' % (tag, d['value']/1e9, d['ms_per_step']*1e3, r['frac']))
    except Exception as e: print('%-36s | ?? %s' % (tag, l[:80]))
PY
