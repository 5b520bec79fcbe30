#!/bin/bash
# Round 3, final evidence on the final build: the whole GPU suite, smoke(), the driver's bench command, rocprofv3 stats of the
# headline command, PMC traffic passes of the send-stage configurations (their kernels changed last).
set -u
cd $GRAFT_REPO_ROOT
R=$PWD
O=$R/gpurun_out/r03z
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $O/tests_gpu.txt
cat $O/tests_gpu.txt
timeout 300 python -c "import __graft_entry__ as e; e.smoke(); print('SMOKE_OK')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
T0=$(date +%s%N)
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
T1=$(date +%s%N)
echo "python bench.py: $(( (T1 - T0) / 1000000 )) ms wall" > $O/bench_n1_wall.txt
cat $O/bench_n1_wall.txt
cd /tmp
CMD="python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -o s -- $CMD > $O/prof_stats.json 2> $O/prof_stats.err
CMDS="python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras --inflights 256 --fused-send"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats_ts -o s -- $CMDS > /dev/null 2> $O/prof_stats_ts.err
cd $R
cp $(find $O/prof_stats -name "*kernel_stats.csv" | head -1) $O/r03_final_kernel_stats.csv
cp $(find $O/prof_stats_ts -name "*kernel_stats.csv" | head -1) $O/r03_tick_send_kernel_stats.csv
rm -rf $O/prof_stats $O/prof_stats_ts
tools/pmc_traffic.sh "2:1000000:5:inflights" 30 --inflights 256 > /dev/null 2>&1
tools/pmc_traffic.sh "2:1000000:5:inflights:fused-send" 30 --inflights 256 --fused-send > /dev/null 2>&1
cp gpurun_out/traffic_*.json $O/
head -12 $O/r03_final_kernel_stats.csv
ls $O
