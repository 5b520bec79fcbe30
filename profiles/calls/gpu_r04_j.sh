#!/bin/bash
# round 4, call j: the one-launch send form is bimodal from run to run (140-150 us / 163-178 us). Does it follow the size of
# the Inflights ring arena (cap 256: 10.7 GB; cap 8: 0.3 GB), i.e. address translation over a huge, barely touched allocation?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04j
O=gpurun_out/r04j/send.txt; : > $O
run() { echo -n "$* : " >> $O; timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 40 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f us' % (d['ms_per_step']*1e3))" >> $O; }
for rep in 1 2 3 4; do
  run --inflights 256 --fused-send
  run --inflights 8 --fused-send
  run --inflights 256
  run --inflights 8
done
cat $O
