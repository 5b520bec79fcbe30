#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03h; mkdir -p $O
J=$O/bench.jsonl; : > $J
run() { tag=$1; shift; echo "## $tag" >> $J; timeout 300 python bench.py --no-cpu-baseline --no-extras --warmup 5 "$@" 2>/dev/null | tail -1 >> $J; }
L=$GRAFT_REPO_ROOT/raft_rs_amd
for lib in "" blk256 "" blk256; do
  if [ -n "$lib" ]; then export RG_LIB_PATH=$L/libraftgroups_$lib.so; else unset RG_LIB_PATH; fi
  run "$lib c2 1M" --steps 40
  run "$lib c2 8M" --groups 8000000 --steps 15
  run "$lib c2 4M" --groups 4000000 --steps 20
  run "$lib c4 1M" --slots 7 --steps 40
  run "$lib c4 8M" --slots 7 --groups 8000000 --steps 12
done
python - <<'PY' | tee $O/bench_summary.txt
import json
tag=None
for l in open('gpurun_out/r03h/bench.jsonl'):
    if l.startswith('##'): tag=l[2:].strip(); continue
    try:
        d=json.loads(l); r=d['roofline']
        print('%-26s | %.2f G/s  %.1f us/step  frac %.3f' % (tag, d['value']/1e9, d['ms_per_step']*1e3, r['frac']))
    except Exception as e: print('%-26s | ?? %s' % (tag, l[:80]))
PY
