#!/bin/bash
# round 3, call w: non-temporal STORES of `next` and the result word (RG_OPT bit 3; opt14 = 6 | 8) on top of the streamed message
# loads, between the cache regimes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03w; mkdir -p $O
L=$GRAFT_REPO_ROOT/raft_rs_amd
J=$O/bench.jsonl; : > $J
run() { tag=$1; shift; echo "## $tag" >> $J; timeout 300 python bench.py --no-cpu-baseline --no-extras --warmup 4 "$@" 2>/dev/null | tail -1 >> $J; }
for lib in "" opt14 "" opt14; do
  if [ -n "$lib" ]; then export RG_LIB_PATH=$L/libraftgroups_$lib.so; else unset RG_LIB_PATH; fi
  run "$lib c2 1Mx5" --steps 40
  run "$lib c2 2Mx5" --groups 2000000 --steps 30
  run "$lib c2 2.5Mx5" --groups 2500000 --steps 30
  run "$lib c2 3Mx5" --groups 3000000 --steps 30
  run "$lib c4 1Mx7" --slots 7 --steps 40
  run "$lib c4 2Mx7" --slots 7 --groups 2000000 --steps 30
  run "$lib c2 8Mx5" --groups 8000000 --steps 12
done
python - <<'PY' | tee $O/summary.txt
import json
tag=None
for l in open('gpurun_out/r03w/bench.jsonl'):
    if l.startswith('##'): tag=l[2:].strip(); continue
    try:
        d=json.loads(l); r=d['roofline']
        print('%-22s | %.2f G/s  %.1f us/step  frac %.3f' % (tag, d['value']/1e9, d['ms_per_step']*1e3, r['frac']))
    except Exception as e: print('%-22s | ?? %s' % (tag, l[:80]))
PY
