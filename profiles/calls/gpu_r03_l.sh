#!/bin/bash
# round 3, call l: the send stage's single-message case as predicate arithmetic (rg_send.h) -- parity, then both forms
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03l; mkdir -p $O
timeout 1200 python -m pytest tests/test_sendstage_gpu.py tests/test_scenarios.py tests/test_votes_and_mirror_gpu.py -m gpu -x -q 2>&1 | tail -8 > $O/tests.txt
cat $O/tests.txt
J=$O/side.jsonl; : > $J
run() { tag=$1; shift; echo "## $tag" >> $J; timeout 300 python bench.py --side tick --workload 2 --warmup 5 --steps 30 "$@" 2>$O/err.txt | tail -1 >> $J; }
for rep in 1 2; do
  run "two launches 1Mx5" --inflights 256
  run "one launch 1Mx5" --inflights 256 --fused-send
  run "two launches 1Mx3" --slots 3 --inflights 256
  run "one launch 1Mx3" --slots 3 --inflights 256 --fused-send
  run "two launches 1Mx7" --slots 7 --inflights 256
  run "one launch 1Mx7" --slots 7 --inflights 256 --fused-send
  run "two launches 8Mx5" --groups 8000000 --steps 10 --warmup 3 --inflights 256
  run "one launch 8Mx5" --groups 8000000 --steps 10 --warmup 3 --inflights 256 --fused-send
done
python - <<'PY' | tee $O/summary.txt
import json
tag=None
for l in open('gpurun_out/r03l/side.jsonl'):
    if l.startswith('##'): tag=l[2:].strip(); continue
    try:
        d=json.loads(l); r=d['roofline']; s=d.get('send_stage',{})
        sr=s.get('roofline',{})
        print('%-24s | %.2f G/s  %.1f us/step | tick %.1f stage %.1f | frac %.3f (%s, %.0f B/group)' % (
            tag, d['value']/1e9, d['us_per_step'], s.get('us_per_tick_median',0), s.get('us_per_stage_median',0), sr.get('frac',0), sr.get('kernel','-'), sr.get('bytes_per_group',0)))
    except Exception as e: print('%-24s | ?? %s' % (tag, l[:100]))
PY
bash tools/pmc_sq.sh ticksend2 --inflights 256 --fused-send > /dev/null 2>&1
cp gpurun_out/pmc_ticksend2.txt $O/
head -24 $O/pmc_ticksend2.txt
