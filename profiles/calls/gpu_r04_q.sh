#!/bin/bash
# round 4, call q: the two modes of the send forms. Hypothesis: what a step re-touches (state 160 MB + window columns 100 MB at
# 1 M x 5) is the size of the Infinity Cache, so how much of it survives from launch to launch is decided by where the process'
# pages happen to lie. (1) the window columns streamed (build knob RG_SEND_NT_WIN = 1: meta, head, tail; 2: head and tail),
# 5 processes each; (2) the default build at 800 k and 1.25 M groups -- well inside / well outside the cache.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04q2
O=gpurun_out/r04q2/win.txt; : > $O
run() { echo -n "$TAG $* : " >> $O; timeout 300 python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f us frac %.3f' % (d['ms_per_step']*1e3, d['roofline']['frac']))" >> $O; }
for rep in 1 2 3 4 5; do
for L in base w1 w2; do
  if [ $L = base ]; then unset RG_LIB_PATH; else export RG_LIB_PATH=$GRAFT_REPO_ROOT/raft_rs_amd/libraftgroups_$L.so; fi
  TAG=$L
  run --steps 40 --inflights 256 --fused-send
  run --steps 40 --inflights 256
done
done
unset RG_LIB_PATH
TAG=base
for rep in 1 2 3 4 5; do
  run --steps 40 --inflights 256 --fused-send --groups 800000
  run --steps 40 --inflights 256 --fused-send --groups 1250000
done
cat $O
