#!/bin/bash
# round 4, call i: launch order of the class kernel once more -- more parts, and the largest class dealt first
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04i
O=gpurun_out/r04i/bench.txt; : > $O
run() { echo -n "$TAG : " >> $O; timeout 300 python bench.py --no-cpu-baseline --no-extras --workload 5 --slots 7 --sorted --steps 40 2>gpurun_out/r04i/err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f us frac %.3f' % (d['ms_per_step']*1e3, d['roofline']['frac']))" >> $O; }
for rep in 1 2 3; do
  unset RG_LIB_PATH
  for W in 3 4 9 12 30 300; do export RG_CLASS_WAYS=$W; TAG="ways=$W"; run; done
  export RG_LIB_PATH=$GRAFT_REPO_ROOT/raft_rs_amd/libraftgroups_rev.so
  for W in 3 9; do export RG_CLASS_WAYS=$W; TAG="reversed ways=$W"; run; done
done
cat $O
