#!/bin/bash
# round 3, GPU call C: what each KIND of rare group costs a wave (measurement build; only that kind's lanes stay, in place)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03c
L=$GRAFT_REPO_ROOT/raft_rs_amd
export RG_LIB_PATH=$L/libraftgroups_cptm.so
for m in 3 4 5 6; do
BENCH_MEASURE_DROP=$m tools/pmc_sq_tail.sh c5one_kind$m 20 --workload 5 --slots 7 --one-engine --variant 5 > /dev/null 2>&1
done
cat gpurun_out/pmct_c5one_kind*.txt > gpurun_out/r03c/all.txt
grep -A6 "^# \|SQ_WAVE_CYCLES" gpurun_out/r03c/all.txt | grep "^#\|INSTS_VALU\|INSTS_SALU\|WAVE_CYCLES"
