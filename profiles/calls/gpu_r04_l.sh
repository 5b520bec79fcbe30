#!/bin/bash
# round 4, call l: beyond the Infinity Cache (8 M groups) everything is read once per launch -- the state columns as
# non-temporal loads (-DRG_NT_STATE: where the engine streams the messages, NTM), non-temporal stores of every state column
# (RG_OPT bit 4), both; against the default build, same box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04l
O=gpurun_out/r04l/nt.txt; : > $O
run() { echo -n "$TAG $* : " >> $O; timeout 300 python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f us frac %.3f' % (d['ms_per_step']*1e3, d['roofline']['frac']))" >> $O; }
for rep in 1 2; do
for L in base nts o22 nts22; do
  if [ $L = base ]; then unset RG_LIB_PATH; else export RG_LIB_PATH=$GRAFT_REPO_ROOT/raft_rs_amd/libraftgroups_$L.so; fi
  TAG=$L
  run --steps 16 --groups 8000000
  run --steps 16 --groups 8000000 --slots 7
  run --steps 30 --groups 3000000
  run --steps 40 --groups 2000000
done
done
cat $O
