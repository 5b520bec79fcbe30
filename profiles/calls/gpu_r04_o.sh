#!/bin/bash
# round 4, call o: the all-streamed regime far beyond the cache -- where does it stop paying? (16 M groups lost 7 % in call n)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04o
O=gpurun_out/r04o/nt.txt; : > $O
run() { echo -n "$TAG $* : " >> $O; timeout 300 python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f us frac %.3f' % (d['ms_per_step']*1e3, d['roofline']['frac']))" >> $O; }
for rep in 1 2; do
for G in 6000000 8000000 10000000 12000000 16000000 24000000; do
for M in 0 1; do
  export RG_NT_ALL=$M
  TAG="RG_NT_ALL=$M"
  run --steps 12 --warmup 3 --groups $G
done
done
done
cat $O
