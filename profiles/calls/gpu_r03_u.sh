#!/bin/bash
# round 3, call u: streaming message loads beyond the Infinity Cache: 3 M / 4 M / 8 M groups x 5, 8 M x 7, forced off vs default (on)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03u; mkdir -p $O
J=$O/bench.jsonl; : > $J
run() { tag=$1; shift; echo "## $tag" >> $J; timeout 300 python bench.py --no-cpu-baseline --no-extras --warmup 3 "$@" 2>/dev/null | tail -1 >> $J; }
for nt in 0 default 0 default; do
  if [ "$nt" = default ]; then unset RG_NT_MSGS; else export RG_NT_MSGS=$nt; fi
  run "nt=$nt c2 3Mx5" --groups 3000000 --steps 30
  run "nt=$nt c2 4Mx5" --groups 4000000 --steps 25
  run "nt=$nt c2 8Mx5" --groups 8000000 --steps 15
  run "nt=$nt c4 8Mx7" --groups 8000000 --slots 7 --steps 12
done
unset RG_NT_MSGS
python - <<'PY' | tee $O/summary.txt
import json
tag=None
for l in open('gpurun_out/r03u/bench.jsonl'):
    if l.startswith('##'): tag=l[2:].strip(); continue
    try:
        d=json.loads(l); r=d['roofline']
        print('%-28s | %.2f G/s  %.1f us/step  frac %.3f' % (tag, d['value']/1e9, d['ms_per_step']*1e3, r['frac']))
    except Exception as e: print('%-28s | ?? %s' % (tag, l[:80]))
PY
