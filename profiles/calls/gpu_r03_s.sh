#!/bin/bash
# round 3, call s: non-temporal loads of the read-once message columns (RG_OPT bit 0; opt7 = 6 | 1) where state + one tick's
# messages straddle the 256 MB Infinity Cache (1 M x 7: 224 + 120 MB; config 5 in one engine: 208 + 112 MB)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s2; mkdir -p $O
L=$GRAFT_REPO_ROOT/raft_rs_amd
J=$O/bench.jsonl; : > $J
run() { tag=$1; shift; echo "## $tag" >> $J; timeout 300 python bench.py --no-cpu-baseline --no-extras --warmup 5 --steps 40 "$@" 2>/dev/null | tail -1 >> $J; }
for lib in "" opt7 "" opt7; do
  if [ -n "$lib" ]; then export RG_LIB_PATH=$L/libraftgroups_$lib.so; else unset RG_LIB_PATH; fi
  run "$lib c2 1Mx5"
  run "$lib c4 shard 1Mx7" --slots 7
  run "$lib c5 one engine" --workload 5 --slots 7 --one-engine
  run "$lib c5 size classes" --workload 5
  run "$lib c2 1.5Mx5" --groups 1500000
done
python - <<'PY' | tee $O/summary.txt
import json
tag=None
for l in open('gpurun_out/r03s2/bench.jsonl'):
    if l.startswith('##'): tag=l[2:].strip(); continue
    try:
        d=json.loads(l); r=d['roofline']
        print('%-24s | %.2f G/s  %.1f us/step  frac %.3f' % (tag, d['value']/1e9, d['ms_per_step']*1e3, r['frac']))
    except Exception as e: print('%-24s | ?? %s' % (tag, l[:80]))
PY
