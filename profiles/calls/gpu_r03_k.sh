#!/bin/bash
# round 3, call k: k_tick_send with the stage's window columns prefetched into LDS by LDS-DMA (-DRG_TS_SPEC=2)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03k; mkdir -p $O
L=$GRAFT_REPO_ROOT/raft_rs_amd
RG_LIB_PATH=$L/libraftgroups_tsdma.so timeout 1200 python -m pytest tests/test_sendstage_gpu.py -m gpu -x -q 2>&1 | tail -8 > $O/tests_tsdma.txt
cat $O/tests_tsdma.txt
J=$O/side.jsonl; : > $J
run() { tag=$1; shift; echo "## $tag" >> $J; timeout 300 python bench.py --side tick --workload 2 --warmup 5 --steps 30 "$@" 2>$O/err.txt | tail -1 >> $J; }
for lib in "" tsdma "" tsdma; do
  if [ -n "$lib" ]; then export RG_LIB_PATH=$L/libraftgroups_$lib.so; [ -f $RG_LIB_PATH ] || continue; else unset RG_LIB_PATH; fi
  run "$lib one launch 1Mx5" --inflights 256 --fused-send
  run "$lib one launch 1Mx3" --slots 3 --inflights 256 --fused-send
  run "$lib one launch 1Mx7" --slots 7 --inflights 256 --fused-send
  run "$lib one launch 8Mx5" --groups 8000000 --steps 10 --warmup 3 --inflights 256 --fused-send
done
unset RG_LIB_PATH
python - <<'PY' | tee $O/summary.txt
import json
tag=None
for l in open('gpurun_out/r03k/side.jsonl'):
    if l.startswith('##'): tag=l[2:].strip(); continue
    try:
        d=json.loads(l); r=d['roofline']; s=d.get('send_stage',{})
        sr=s.get('roofline',{})
        print('%-34s | %.2f G/s  %.1f us/step | frac %.3f (%s, %.0f B/group)' % (
            tag, d['value']/1e9, d['us_per_step'], sr.get('frac',0), sr.get('kernel','-'), sr.get('bytes_per_group',0)))
    except Exception as e: print('%-34s | ?? %s' % (tag, l[:100]))
PY
