#!/bin/bash
# Round 4, PMC / rocprofv3 passes on the LAST build (1c83110: the send stage streams its window columns; the tick kernels are those of 89b1f7b).
# Was:
# the 8 M-group tick changed). Was: PMC / rocprofv3 passes re-run on the last build (1c83110) for the kernels that changed after gpu_r04_profiles.sh
# (elections file their run inside become_leader; the class kernel's launch-order table), and the send-stage soak
# (k_tick_send was restructured into phases this round)
set -u
cd $GRAFT_REPO_ROOT
R=$PWD
O=$R/gpurun_out/r04t
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
echo 1c83110 > $O/build_commit.txt
cd /tmp
CMD="python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -o s -- $CMD > $O/prof_stats.json 2> $O/prof_stats.err
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -o f -- $CMD > /dev/null 2> $O/prof_fetch.err
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof_write -o w -- $CMD > /dev/null 2> $O/prof_write.err
CMD5="python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras --workload 5 --slots 7 --sorted"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats_c5 -o s -- $CMD5 > /dev/null 2> $O/prof_stats_c5.err
CMDS="python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras --inflights 256 --fused-send"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats_ts -o s -- $CMDS > /dev/null 2> $O/prof_stats_ts.err
CMD8="python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras --groups 8000000"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats_8m -o s -- $CMD8 > /dev/null 2> $O/prof_stats_8m.err
cd $R
python tools/summarize_prof.py --tag r04 --stats $O/prof_stats --fetch $O/prof_fetch --write $O/prof_write --last 50 --out $O \
  --note "python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras (1 M groups x 5 peers, config 2), build 1c83110 (the last of round 4)" > /dev/null
cp $(find $O/prof_stats -name "*kernel_stats.csv" | head -1) $O/r04_kernel_stats.csv
cp $(find $O/prof_stats_c5 -name "*kernel_stats.csv" | head -1) $O/r04_c5_kernel_stats.csv
cp $(find $O/prof_stats_ts -name "*kernel_stats.csv" | head -1) $O/r04_tick_send_kernel_stats.csv
cp $(find $O/prof_stats_8m -name "*kernel_stats.csv" | head -1) $O/r04_8m_kernel_stats.csv
rm -rf $O/prof_stats $O/prof_fetch $O/prof_write $O/prof_stats_c5 $O/prof_stats_ts $O/prof_stats_8m
tools/pmc_traffic.sh "2:1000000:5" 30 > /dev/null 2>&1
tools/pmc_traffic.sh "3:1000000:5" 30 --workload 3 > /dev/null 2>&1
tools/pmc_traffic.sh "2:1000000:7" 30 --slots 7 > /dev/null 2>&1
tools/pmc_traffic.sh "5:1000000:7:sorted" 30 --workload 5 --slots 7 --sorted > /dev/null 2>&1
tools/pmc_traffic.sh "5:1000000:7" 30 --workload 5 --slots 7 > /dev/null 2>&1
tools/pmc_traffic.sh "5:1000000:7:one-engine" 30 --workload 5 --slots 7 --one-engine > /dev/null 2>&1
tools/pmc_traffic.sh "2:8000000:5" 12 --groups 8000000 > /dev/null 2>&1
tools/pmc_traffic.sh "2:1000000:5:inflights" 30 --inflights 256 > /dev/null 2>&1
tools/pmc_traffic.sh "2:1000000:5:inflights:fused-send" 30 --inflights 256 --fused-send > /dev/null 2>&1
cp gpurun_out/traffic_*.json $O/
timeout 400 python tools/soak_send_gpu.py 120 30000 > $O/soak_send.txt 2>&1
tail -5 $O/soak_send.txt
ls $O; head -3 $O/r04_kernel_stats.csv; head -3 $O/r04_c5_kernel_stats.csv
