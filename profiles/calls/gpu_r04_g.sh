#!/bin/bash
# round 4, call g: -DRG_AT_OPAQUE (the 32-bit cell offset made opaque right before every access, so that `base + zext(offset)`
# stays next to the access and is selected as SGPR-base + VGPR-offset addressing instead of a 64-bit VGPR address pair:
# lane<5> 123 -> 101 VGPRs, lane<3> 90 -> 69, lane<7> 162 -> 135, fused<5> 156 -> 115) against the default build, same box
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04g
O=gpurun_out/r04g/bench.txt; : > $O
run() { echo -n "$TAG $* : " >> $O; timeout 300 python bench.py --no-cpu-baseline --no-extras "$@" 2>gpurun_out/r04g/err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f us frac %.3f' % (d['ms_per_step']*1e3, d['roofline']['frac']))" >> $O; }
for rep in 1 2; do
for L in base atop; do
  if [ $L = atop ]; then export RG_LIB_PATH=$GRAFT_REPO_ROOT/raft_rs_amd/libraftgroups_atop.so; else unset RG_LIB_PATH; fi
  TAG=$L
  run --steps 60
  run --steps 60 --slots 7
  run --steps 60 --slots 3
  run --steps 60 --workload 3
  run --steps 16 --groups 8000000
  run --steps 40 --groups 2000000
  run --steps 40 --workload 5 --slots 7 --sorted
  run --steps 40 --workload 5
  run --steps 40 --fuse 4
  run --steps 40 --fuse 8
  run --steps 40 --inflights 256
  run --steps 40 --inflights 256 --fused-send
done
done
cat $O
export RG_LIB_PATH=$GRAFT_REPO_ROOT/raft_rs_amd/libraftgroups_atop.so
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_full_size_gpu.py tests/test_sendstage_gpu.py tests/test_sparse_path_gpu.py -m gpu -x -q 2>&1 | tail -4
