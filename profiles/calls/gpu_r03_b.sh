#!/bin/bash
# round 3, GPU call B: where the compact variant's instructions go -- natural waves vs the designated wave (measurement build)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03b
L=$GRAFT_REPO_ROOT/raft_rs_amd
export RG_LIB_PATH=$L/libraftgroups_cptm.so
tools/pmc_sq_tail.sh c4_compact 20 --slots 7 --variant 5 > /dev/null 2>&1
tools/pmc_sq_tail.sh c4_lane 20 --slots 7 > /dev/null 2>&1
tools/pmc_sq_tail.sh c5one_compact 20 --workload 5 --slots 7 --one-engine --variant 5 > /dev/null 2>&1
BENCH_MEASURE_DROP=1 tools/pmc_sq_tail.sh c5one_compact_drop_rare 20 --workload 5 --slots 7 --one-engine --variant 5 > /dev/null 2>&1
BENCH_MEASURE_DROP=2 tools/pmc_sq_tail.sh c5one_compact_drop_steady 20 --workload 5 --slots 7 --one-engine --variant 5 > /dev/null 2>&1
tools/pmc_sq_tail.sh c5one_lane 20 --workload 5 --slots 7 --one-engine > /dev/null 2>&1
cat gpurun_out/pmct_*.txt > gpurun_out/r03b/all.txt
for v in "" 1 2; do echo "drop=$v"; BENCH_MEASURE_DROP=$v timeout 200 python bench.py --no-cpu-baseline --no-extras --steps 30 --workload 5 --slots 7 --one-engine --variant 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step']*1e3)"; done | tee gpurun_out/r03b/timing.txt
cat gpurun_out/r03b/all.txt
