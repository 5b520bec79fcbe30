#!/bin/bash
# round 4, call f: launch order of the class kernel (the shard dealt out in 1 / 2 / 3 / 6 parts) and the message columns
# streamed or not, config 5 placed by size class, 1 M and 8 M groups; class tests first
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04f
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_full_size_gpu.py -m gpu -x -q 2>&1 | grep -v "^E    .*match\[" | tail -5 > gpurun_out/r04f/tests.txt
tail -2 gpurun_out/r04f/tests.txt
O=gpurun_out/r04f/bench.txt; : > $O
run() { echo -n "$TAG $* : " >> $O; timeout 300 python bench.py --no-cpu-baseline --no-extras --workload 5 --slots 7 --sorted "$@" 2>gpurun_out/r04f/err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f us frac %.3f' % (d['ms_per_step']*1e3, d['roofline']['frac']))" >> $O; }
for rep in 1 2; do
for W in 1 2 3 6; do
  export RG_CLASS_WAYS=$W; TAG="ways=$W"; run --steps 40
done
done
unset RG_CLASS_WAYS
for W in 1 3; do
  export RG_CLASS_WAYS=$W RG_NT_MSGS=0; TAG="ways=$W nt=0"; run --steps 40
  export RG_NT_MSGS=1; TAG="ways=$W nt=1"; run --steps 40
  unset RG_NT_MSGS
  TAG="ways=$W 8M"; run --steps 12 --groups 8000000
  TAG="ways=$W 100k"; run --steps 40 --groups 100000
done
cat $O; tail -2 gpurun_out/r04f/err.txt
