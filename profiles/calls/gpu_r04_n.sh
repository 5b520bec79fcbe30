#!/bin/bash
# round 4, call n: with the non-temporal stores really in the code (rg_st<NT>: the flag is a template argument now) -- the
# all-streamed regime around its threshold and far beyond, and the one-launch send form (k_tick_send had lost its NT item stores)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04n
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_sendstage_gpu.py -m gpu -x -q 2>&1 | grep "passed\|failed"
O=gpurun_out/r04n/nt.txt; : > $O
run() { echo -n "$TAG $* : " >> $O; timeout 300 python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f us frac %.3f' % (d['ms_per_step']*1e3, d['roofline']['frac']))" >> $O; }
for rep in 1 2; do
for M in 0 1; do
  export RG_NT_ALL=$M
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
unset RG_NT_ALL
TAG=send
for rep in 1 2 3 4; do run --steps 40 --inflights 256 --fused-send; run --steps 40 --inflights 256; done
cat $O
