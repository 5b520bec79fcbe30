#!/bin/bash
# round 3, call j: k_tick_send -- the whole send-stage test file, one launch vs two (again, + RG_TS_SPEC build), PMC passes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03j; mkdir -p $O
timeout 1500 python -m pytest tests/test_sendstage_gpu.py tests/test_abi.py -m gpu -x -q 2>&1 | tail -8 > $O/tests.txt
cat $O/tests.txt
J=$O/side.jsonl; : > $J
run() { tag=$1; shift; echo "## $tag" >> $J; timeout 300 python bench.py --side tick --workload 2 --warmup 5 --steps 30 "$@" 2>$O/err.txt | tail -1 >> $J; }
L=$GRAFT_REPO_ROOT/raft_rs_amd
for lib in "" tsspec "" tsspec; do
  if [ -n "$lib" ]; then export RG_LIB_PATH=$L/libraftgroups_$lib.so; [ -f $RG_LIB_PATH ] || continue; else unset RG_LIB_PATH; fi
  run "$lib one launch 1Mx5" --inflights 256 --fused-send
  run "$lib one launch 1Mx3" --slots 3 --inflights 256 --fused-send
  run "$lib one launch 8Mx5" --groups 8000000 --steps 10 --warmup 3 --inflights 256 --fused-send
done
unset RG_LIB_PATH
python - <<'PY' | tee $O/summary.txt
import json
tag=None
for l in open('gpurun_out/r03j/side.jsonl'):
    if l.startswith('##'): tag=l[2:].strip(); continue
    try:
        d=json.loads(l); r=d['roofline']; s=d.get('send_stage',{})
        sr=s.get('roofline',{})
        print('%-34s | %.2f G/s  %.1f us/step | frac %.3f (%s, %.0f B/group)' % (
            tag, d['value']/1e9, d['us_per_step'], sr.get('frac',0), sr.get('kernel','-'), sr.get('bytes_per_group',0)))
    except Exception as e: print('%-34s | ?? %s' % (tag, l[:100]))
PY
bash tools/pmc_traffic.sh 2:1000000:5:inflights:fused-send 20 --inflights 256 --fused-send > /dev/null 2>&1
bash tools/pmc_traffic.sh 2:1000000:5:inflights 20 --inflights 256 > /dev/null 2>&1
bash tools/pmc_sq.sh ticksend --inflights 256 --fused-send > /dev/null 2>&1
cp gpurun_out/traffic_2_1000000_5_inflights*.json gpurun_out/pmc_ticksend.txt $O/ 2>/dev/null
head -c 1500 $O/traffic_2_1000000_5_inflights_fused-send.json
head -30 $O/pmc_ticksend.txt
