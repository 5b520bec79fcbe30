#!/bin/bash
# round 4, call t: the fused kernel (T ticks per launch) with its message sets and per-tick result columns streamed past the
# Infinity Cache (build fnt: -DRG_FUSED_NT=1) against the default build: 8 x 88 MB of messages per launch at 1 M groups went
# through the cache that should hold the 160 MB of state
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04t2
O=gpurun_out/r04t2/fused_nt.txt; : > $O
run() { echo -n "$TAG $* : " >> $O; timeout 300 python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f us per tick  %.2f G evals/s' % (d['ms_per_step']*1e3, d['value']/1e9))" >> $O; }
for rep in 1 2 3; do
for L in base fnt; do
  if [ $L = base ]; then unset RG_LIB_PATH; else export RG_LIB_PATH=$GRAFT_REPO_ROOT/raft_rs_amd/libraftgroups_$L.so; fi
  TAG=$L
  run --steps 64 --fuse 8
  run --steps 64 --fuse 4
  run --steps 64 --fuse 2
  run --steps 32 --fuse 8 --groups 2000000
  run --steps 16 --fuse 8 --groups 8000000
  run --steps 64 --fuse 4 --workload 5
  run --steps 64 --fuse 8 --slots 7
done
done
cat $O
