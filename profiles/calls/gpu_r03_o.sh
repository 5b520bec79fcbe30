#!/bin/bash
# round 3, call o: RG_OPT bit 5 (wave-level rewrites of matched / committed_index) on config 5, where absent peers and rare
# events leave many lane-masked lines
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03o; mkdir -p $O
L=$GRAFT_REPO_ROOT/raft_rs_amd
J=$O/bench.jsonl; : > $J
run() { tag=$1; shift; echo "## $tag" >> $J; timeout 300 python bench.py --no-cpu-baseline --no-extras --warmup 5 --steps 40 "$@" 2>/dev/null | tail -1 >> $J; }
for lib in "" opt38 "" opt38; do
  if [ -n "$lib" ]; then export RG_LIB_PATH=$L/libraftgroups_$lib.so; else unset RG_LIB_PATH; fi
  run "$lib c5 size classes" --workload 5
  run "$lib c5 one engine" --workload 5 --slots 7 --one-engine
  run "$lib c5 one engine 8M" --workload 5 --slots 7 --one-engine --groups 8000000 --steps 12
  run "$lib c5 size classes 8M" --workload 5 --groups 8000000 --steps 12
done
python - <<'PY' | tee $O/summary.txt
import json
tag=None
for l in open('gpurun_out/r03o/bench.jsonl'):
    if l.startswith('##'): tag=l[2:].strip(); continue
    try:
        d=json.loads(l); r=d['roofline']
        print('%-28s | %.2f G/s  %.1f us/step  frac %.3f' % (tag, d['value']/1e9, d['ms_per_step']*1e3, r['frac']))
    except Exception as e: print('%-28s | ?? %s' % (tag, l[:80]))
PY
