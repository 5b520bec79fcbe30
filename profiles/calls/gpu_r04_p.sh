#!/bin/bash
# round 4, call p: (1) tools/microbench/nt_shape: the tick's access shape with plain / non-temporal loads and stores over engine
# sizes -- is the upper end of the all-streamed window the machine's? (2) config 5 placed by size class and the one-launch send
# form with the message columns cached instead of streamed (RG_NT_MSGS=0; the rule prices a 7-slot engine by 7 slots everywhere)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04p
( cd tools/microbench && timeout 300 ./nt_shape ) > gpurun_out/r04p/nt_shape.txt 2>&1
cat gpurun_out/r04p/nt_shape.txt
O=gpurun_out/r04p/ab.txt; : > $O
run() { echo -n "$TAG $* : " >> $O; timeout 300 python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f us frac %.3f' % (d['ms_per_step']*1e3, d['roofline']['frac']))" >> $O; }
for rep in 1 2 3; do
for M in auto 0 1; do
  if [ $M = auto ]; then unset RG_NT_MSGS; else export RG_NT_MSGS=$M; fi
  TAG="RG_NT_MSGS=$M"
  run --steps 40 --workload 5 --slots 7 --sorted
  run --steps 40 --inflights 256 --fused-send
  run --steps 40 --inflights 256
  run --steps 40 --slots 7
done
done
cat $O
