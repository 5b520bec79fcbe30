#!/bin/bash
# round 3, call n: whole-line (wave-level) accesses in the send stage (RG_SEND_WAVE_LINES, default on; wl0 = off) and in the
# tick's own stores (RG_OPT bit 5; opt38 = on)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03n; mkdir -p $O
L=$GRAFT_REPO_ROOT/raft_rs_amd
timeout 1200 python -m pytest tests/test_sendstage_gpu.py tests/test_scenarios.py tests/test_api_sequences_gpu.py -m gpu -x -q 2>&1 | tail -6 > $O/tests.txt
cat $O/tests.txt
RG_LIB_PATH=$L/libraftgroups_opt38.so timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_sendstage_gpu.py tests/test_full_size_gpu.py -m gpu -x -q 2>&1 | tail -6 > $O/tests_opt38.txt
cat $O/tests_opt38.txt
J=$O/side.jsonl; : > $J
run() { tag=$1; shift; echo "## $tag" >> $J; timeout 300 python bench.py --side tick --workload 2 --warmup 5 --steps 30 "$@" 2>$O/err.txt | tail -1 >> $J; }
for lib in wl0 "" opt38 wl0 "" opt38; do
  if [ -n "$lib" ]; then export RG_LIB_PATH=$L/libraftgroups_$lib.so; [ -f $RG_LIB_PATH ] || continue; else unset RG_LIB_PATH; fi
  run "$lib two 1Mx5" --inflights 256
  run "$lib one 1Mx5" --inflights 256 --fused-send
  run "$lib one 1Mx3" --slots 3 --inflights 256 --fused-send
  run "$lib two 8Mx5" --groups 8000000 --steps 10 --warmup 3 --inflights 256
  run "$lib one 8Mx5" --groups 8000000 --steps 10 --warmup 3 --inflights 256 --fused-send
  if [ "$lib" != "wl0" ]; then
    run "$lib tick 1Mx5" 
    run "$lib tick 8Mx5" --groups 8000000 --steps 10 --warmup 3
    run "$lib tick 1Mx7" --slots 7
  fi
done
unset RG_LIB_PATH
python - <<'PY' | tee $O/summary.txt
import json
tag=None
for l in open('gpurun_out/r03n/side.jsonl'):
    if l.startswith('##'): tag=l[2:].strip(); continue
    try:
        d=json.loads(l); r=d['roofline']; s=d.get('send_stage',{})
        sr=s.get('roofline',{})
        print('%-18s | %.2f G/s  %.1f us/step | tick %.1f stage %.1f | frac %.3f (%s)' % (
            tag, d['value']/1e9, d['us_per_step'], s.get('us_per_tick_median',0), s.get('us_per_stage_median',0), sr.get('frac',r['frac']), sr.get('kernel',r['kernel'])))
    except Exception as e: print('%-18s | ?? %s' % (tag, l[:100]))
PY
