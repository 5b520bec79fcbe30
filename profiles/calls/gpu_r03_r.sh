#!/bin/bash
# round 3, call r: the send stage's work-item columns as non-temporal stores (-DRG_SEND_NT_ITEMS=1) at the cache boundary (1 M x 5:
# 360 MB of hot state against the 256 MB Infinity Cache) and beyond it
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03r; mkdir -p $O
L=$GRAFT_REPO_ROOT/raft_rs_amd
J=$O/side.jsonl; : > $J
run() { tag=$1; shift; echo "## $tag" >> $J; timeout 300 python bench.py --side tick --workload 2 --warmup 5 --steps 30 "$@" 2>$O/err.txt | tail -1 >> $J; }
for lib in "" ntitems "" ntitems; do
  if [ -n "$lib" ]; then export RG_LIB_PATH=$L/libraftgroups_$lib.so; else unset RG_LIB_PATH; fi
  run "$lib two 1Mx5" --inflights 256
  run "$lib one 1Mx5" --inflights 256 --fused-send
  run "$lib two 1Mx3" --slots 3 --inflights 256
  run "$lib one 8Mx5" --groups 8000000 --steps 10 --warmup 3 --inflights 256 --fused-send
done
unset RG_LIB_PATH
python - <<'PY' | tee $O/summary.txt
import json
tag=None
for l in open('gpurun_out/r03r/side.jsonl'):
    if l.startswith('##'): tag=l[2:].strip(); continue
    try:
        d=json.loads(l); r=d['roofline']; s=d.get('send_stage',{})
        sr=s.get('roofline',{})
        print('%-22s | %.2f G/s  %.1f us/step | tick %.1f stage %.1f | frac %.3f (%s)' % (
            tag, d['value']/1e9, d['us_per_step'], s.get('us_per_tick_median',0), s.get('us_per_stage_median',0), sr.get('frac',r['frac']), sr.get('kernel',r['kernel'])))
    except Exception as e: print('%-22s | ?? %s' % (tag, l[:100]))
PY
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench default ok:', d['value']/1e9, [ (k, v.get('error', round(v.get('us_per_step',0),1))) for k,v in d['other_configs'].items()], d['small_batch_latency'].get('error','lat ok'))"
