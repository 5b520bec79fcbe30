#!/bin/bash
# round 4, call s: the window columns streamed (build w1: RG_SEND_NT_WIN=1) against the default build over engine sizes --
# where does it start to pay? (calls q, r: 1 M x 5, -3..-5 % in 8 of 8 pairs)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s
O=gpurun_out/r04s/win_sizes.txt; : > $O
run() { echo -n "$TAG $* : " >> $O; timeout 300 python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f us' % (d['ms_per_step']*1e3))" >> $O; }
for rep in 1 2; do
for G in 500000 700000 850000 1000000 1250000 2000000 4000000; do
for L in base w1; do
  if [ $L = base ]; then unset RG_LIB_PATH; else export RG_LIB_PATH=$GRAFT_REPO_ROOT/raft_rs_amd/libraftgroups_$L.so; fi
  TAG=$L
  run --steps 100 --inflights 256 --fused-send --groups $G
  run --steps 100 --inflights 256 --groups $G
done
done
done
for L in base w1; do
  if [ $L = base ]; then unset RG_LIB_PATH; else export RG_LIB_PATH=$GRAFT_REPO_ROOT/raft_rs_amd/libraftgroups_$L.so; fi
  TAG=$L
  run --steps 100 --inflights 256 --fused-send --slots 3
  run --steps 100 --inflights 256 --fused-send --slots 7
done
cat $O
