#!/bin/bash
# round 3, call t: the per-launch choice of streaming (non-temporal) message loads -- parity with the variant forced on
# (RG_NT_MSGS=1), then default (chosen from the engine's footprint) vs forced off / on
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03t; mkdir -p $O
RG_NT_MSGS=1 timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_full_size_gpu.py tests/test_scenarios.py -m gpu -x -q 2>&1 | tail -4 > $O/tests_nt_on.txt
cat $O/tests_nt_on.txt
J=$O/bench.jsonl; : > $J
run() { tag=$1; shift; echo "## $tag" >> $J; timeout 300 python bench.py --no-cpu-baseline --no-extras --warmup 5 --steps 40 "$@" 2>/dev/null | tail -1 >> $J; }
for nt in default 0 1 default 0 1; do
  if [ "$nt" = default ]; then unset RG_NT_MSGS; else export RG_NT_MSGS=$nt; fi
  run "nt=$nt c2 1Mx5"
  run "nt=$nt c2 1.25Mx5" --groups 1250000
  run "nt=$nt c2 1.5Mx5" --groups 1500000
  run "nt=$nt c2 2Mx5" --groups 2000000
  run "nt=$nt c4 shard 1Mx7" --slots 7
  run "nt=$nt c5 one engine" --workload 5 --slots 7 --one-engine
  run "nt=$nt c3 1Mx5" --workload 3
done
unset RG_NT_MSGS
python - <<'PY' | tee $O/summary.txt
import json
tag=None
for l in open('gpurun_out/r03t/bench.jsonl'):
    if l.startswith('##'): tag=l[2:].strip(); continue
    try:
        d=json.loads(l); r=d['roofline']
        print('%-28s | %.2f G/s  %.1f us/step  frac %.3f' % (tag, d['value']/1e9, d['ms_per_step']*1e3, r['frac']))
    except Exception as e: print('%-28s | ?? %s' % (tag, l[:80]))
PY
