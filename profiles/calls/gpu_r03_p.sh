#!/bin/bash
# round 3, call p: the 7-slot tick compiled for 4 waves per SIMD (-DRG_MIN_WAVES=4: 128 VGPRs + 148 B of scratch) on the config-4
# shard and config 5
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03p2; mkdir -p $O
L=$GRAFT_REPO_ROOT/raft_rs_amd
J=$O/bench.jsonl; : > $J
run() { tag=$1; shift; echo "## $tag" >> $J; timeout 300 python bench.py --no-cpu-baseline --no-extras --warmup 5 --steps 40 "$@" 2>/dev/null | tail -1 >> $J; }
for lib in "" mw4 "" mw4; do
  if [ -n "$lib" ]; then export RG_LIB_PATH=$L/libraftgroups_$lib.so; else unset RG_LIB_PATH; fi
  run "$lib c4 shard 1Mx7" --slots 7
  run "$lib c5 size classes" --workload 5
  run "$lib c5 one engine" --workload 5 --slots 7 --one-engine
  run "$lib c4 shard 8Mx7" --slots 7 --groups 8000000 --steps 12
done
python - <<'PY' | tee $O/summary.txt
import json
tag=None
for l in open('gpurun_out/r03p2/bench.jsonl'):
    if l.startswith('##'): tag=l[2:].strip(); continue
    try:
        d=json.loads(l); r=d['roofline']
        print('%-28s | %.2f G/s  %.1f us/step  frac %.3f' % (tag, d['value']/1e9, d['ms_per_step']*1e3, r['frac']))
    except Exception as e: print('%-28s | ?? %s' % (tag, l[:80]))
PY
