#!/bin/bash
# round 4, call m: the all-streamed regime as the engine picks it (state alone > 1.5 x the Infinity Cache) against RG_NT_ALL=0,
# around the threshold and far beyond it; its parity tests first
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04m
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_full_size_gpu.py -m gpu -x -q 2>&1 | grep "passed\|failed"
O=gpurun_out/r04m/nt.txt; : > $O
run() { echo -n "$TAG $* : " >> $O; timeout 300 python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f us frac %.3f' % (d['ms_per_step']*1e3, d['roofline']['frac']))" >> $O; }
for rep in 1 2; do
for M in auto 0 1; do
  if [ $M = auto ]; then unset RG_NT_ALL; else export RG_NT_ALL=$M; fi
  TAG="RG_NT_ALL=$M"
  run --steps 40 --groups 2000000
  run --steps 30 --groups 2400000
  run --steps 30 --groups 2800000
  run --steps 30 --groups 3200000
  run --steps 20 --groups 4000000
  run --steps 16 --groups 8000000
  run --steps 16 --groups 8000000 --slots 7
  run --steps 16 --groups 16000000
  run --steps 12 --groups 8000000 --workload 5 --slots 7 --sorted
  run --steps 40 --groups 2000000 --slots 7
done
done
cat $O
