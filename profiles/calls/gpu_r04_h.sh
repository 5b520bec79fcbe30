#!/bin/bash
# round 4, call h: the single-tick lane kernels with opaque cell offsets (rg_u32o) and FOUR waves per SIMD asked for
# (-DRG_MIN_WAVES=4: the 7-slot bodies fit 128 VGPRs with 28 B of scratch per lane instead of 162 VGPRs / 3 waves) against the
# default build, same box: does the fourth wave pay for 24 spilled registers?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04h
O=gpurun_out/r04h/bench.txt; : > $O
run() { echo -n "$TAG $* : " >> $O; timeout 300 python bench.py --no-cpu-baseline --no-extras "$@" 2>gpurun_out/r04h/err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f us frac %.3f' % (d['ms_per_step']*1e3, d['roofline']['frac']))" >> $O; }
for rep in 1 2; do
for L in base o4; do
  if [ $L = base ]; then unset RG_LIB_PATH; else export RG_LIB_PATH=$GRAFT_REPO_ROOT/raft_rs_amd/libraftgroups_$L.so; fi
  TAG=$L
  run --steps 60 --slots 7
  run --steps 40 --workload 5 --slots 7 --sorted
  run --steps 40 --workload 5
  run --steps 40 --workload 5 --slots 7 --one-engine
  run --steps 16 --groups 8000000 --slots 7
  run --steps 60
done
done
cat $O
export RG_LIB_PATH=$GRAFT_REPO_ROOT/raft_rs_amd/libraftgroups_o4.so
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_full_size_gpu.py -m gpu -x -q 2>&1 | grep "passed\|failed"
