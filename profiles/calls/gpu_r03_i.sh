#!/bin/bash
# round 3, call i: rg_tick_send / rg_tick_device_send (k_tick_send: the tick and its send stage in ONE launch) --
# parity first, then two launches vs one launch at 1 M x 5 / 1 M x 7 / 8 M x 5, then the build knobs.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03i; mkdir -p $O
timeout 1500 python -m pytest tests/test_sendstage_gpu.py tests/test_abi.py -m gpu -x -q 2>&1 | tail -8 > $O/tests.txt
cat $O/tests.txt
J=$O/side.jsonl; : > $J
run() { tag=$1; shift; echo "## $tag" >> $J; timeout 300 python bench.py --side tick --workload 2 --warmup 5 --steps 30 "$@" 2>$O/err.txt | tail -1 >> $J; }
L=$GRAFT_REPO_ROOT/raft_rs_amd
for lib in "" tsw4 tso1 ""; do
  if [ -n "$lib" ]; then export RG_LIB_PATH=$L/libraftgroups_$lib.so; [ -f $RG_LIB_PATH ] || continue; else unset RG_LIB_PATH; fi
  if [ -z "$lib" ]; then run "two launches 1Mx5" --inflights 256; fi
  run "$lib one launch 1Mx5" --inflights 256 --fused-send
  if [ -z "$lib" ]; then
    run "two launches 1Mx7" --slots 7 --inflights 256
    run "one launch 1Mx7" --slots 7 --inflights 256 --fused-send
    run "two launches 1Mx3" --slots 3 --inflights 256
    run "one launch 1Mx3" --slots 3 --inflights 256 --fused-send
  fi
done
unset RG_LIB_PATH
run "two launches 8Mx5" --groups 8000000 --steps 10 --warmup 3 --inflights 256
run "one launch 8Mx5" --groups 8000000 --steps 10 --warmup 3 --inflights 256 --fused-send
run "tick only 1Mx5 (host Inflights)"
python - <<'PY' | tee $O/summary.txt
import json
tag=None
for l in open('gpurun_out/r03i/side.jsonl'):
    if l.startswith('##'): tag=l[2:].strip(); continue
    try:
        d=json.loads(l); r=d['roofline']; s=d.get('send_stage',{})
        sr=s.get('roofline',{})
        print('%-34s | %.2f G/s  %.1f us/step | tick %.1f stage %.1f | stage/fused frac %.3f (%s, %.0f B/group)' % (
            tag, d['value']/1e9, d['us_per_step'], s.get('us_per_tick_median',0), s.get('us_per_stage_median',0),
            sr.get('frac',0), sr.get('kernel','-'), sr.get('bytes_per_group',0)))
    except Exception as e: print('%-34s | ?? %s' % (tag, l[:100]))
PY
