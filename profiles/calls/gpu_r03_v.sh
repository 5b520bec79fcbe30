#!/bin/bash
# round 3, call v: streamed message loads with the Inflights on the device (k_tick_lane by footprint, k_tick_send always) + parity
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03v; mkdir -p $O
timeout 900 python -m pytest tests/test_sendstage_gpu.py tests/test_api_sequences_gpu.py -m gpu -x -q 2>&1 | tail -3 > $O/tests.txt
cat $O/tests.txt
J=$O/side.jsonl; : > $J
run() { tag=$1; shift; echo "## $tag" >> $J; timeout 300 python bench.py --side tick --workload 2 --warmup 5 --steps 30 "$@" 2>$O/err.txt | tail -1 >> $J; }
for rep in 1 2 3; do
  run "two 1Mx5" --inflights 256
  run "one 1Mx5" --inflights 256 --fused-send
  run "two 1Mx3" --slots 3 --inflights 256
  run "one 1Mx3" --slots 3 --inflights 256 --fused-send
  run "one 8Mx5" --groups 8000000 --steps 10 --warmup 3 --inflights 256 --fused-send
done
python - <<'PY' | tee $O/summary.txt
import json
tag=None
for l in open('gpurun_out/r03v/side.jsonl'):
    if l.startswith('##'): tag=l[2:].strip(); continue
    try:
        d=json.loads(l); r=d['roofline']; s=d.get('send_stage',{})
        sr=s.get('roofline',{})
        print('%-14s | %.2f G/s  %.1f us/step | tick %.1f stage %.1f | frac %.3f (%s)' % (
            tag, d['value']/1e9, d['us_per_step'], s.get('us_per_tick_median',0), s.get('us_per_stage_median',0), sr.get('frac',r['frac']), sr.get('kernel',r['kernel'])))
    except Exception as e: print('%-14s | ?? %s' % (tag, l[:100]))
PY
