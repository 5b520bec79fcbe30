#!/bin/bash
# Round 3 evidence run on the final build: the driver's command, rocprofv3 stats of the headline command, PMC traffic passes of
# every configuration the bench line reports, the N>1 code path at world size 1 and with two ranks sharing the GPU.
set -u
cd $GRAFT_REPO_ROOT
R=$PWD
O=$R/gpurun_out/r03p
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
T0=$(date +%s%N)
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
T1=$(date +%s%N)
echo "python bench.py: $(( (T1 - T0) / 1000000 )) ms wall" > $O/bench_n1_wall.txt
for cfg in "--slots 3" "--groups 4000000 --steps 20" "--variant 2" "--variant 4" "--variant 5" "--split 2" "--fuse 4" "--fuse 8" "--workload 5 --fuse 4" "--workload 5 --variant 5"; do
  timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras $cfg >> $O/bench_other_configs.jsonl 2>> $O/bench_other.err
done
for cfg in "" "--slots 7"; do
  BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline $cfg >> $O/bench_dist_ws1.jsonl 2>> $O/bench_dist.err
done
BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --steps 20 --warmup 3 --groups 500000 --no-cpu-baseline >> $O/bench_dist_share2.jsonl 2>> $O/bench_dist.err
cd /tmp
CMD="python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -o s -- $CMD > $O/prof_stats.json 2> $O/prof_stats.err
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -o f -- $CMD > /dev/null 2> $O/prof_fetch.err
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof_write -o w -- $CMD > /dev/null 2> $O/prof_write.err
CMD5="python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras --workload 5"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats_c5 -o s -- $CMD5 > /dev/null 2> $O/prof_stats_c5.err
cd $R
python tools/summarize_prof.py --tag r03 --stats $O/prof_stats --fetch $O/prof_fetch --write $O/prof_write --last 50 --out $O \
  --note "python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras (1 M groups x 5 peers, config 2), final build of round 3" > /dev/null
cp $(find $O/prof_stats -name "*kernel_stats.csv" | head -1) $O/r03_kernel_stats.csv
cp $(find $O/prof_stats_c5 -name "*kernel_stats.csv" | head -1) $O/r03_c5_kernel_stats.csv
rm -rf $O/prof_stats $O/prof_fetch $O/prof_write $O/prof_stats_c5
tools/pmc_traffic.sh "2:1000000:5" 30 > /dev/null 2>&1
tools/pmc_traffic.sh "3:1000000:5" 30 --workload 3 > /dev/null 2>&1
tools/pmc_traffic.sh "2:1000000:7" 30 --slots 7 > /dev/null 2>&1
tools/pmc_traffic.sh "5:1000000:7" 30 --workload 5 --slots 7 > /dev/null 2>&1
tools/pmc_traffic.sh "5:1000000:7:one-engine" 30 --workload 5 --slots 7 --one-engine > /dev/null 2>&1
tools/pmc_traffic.sh "2:8000000:5" 12 --groups 8000000 > /dev/null 2>&1
tools/pmc_traffic.sh "2:1000000:5:inflights" 30 --inflights 256 > /dev/null 2>&1
tools/pmc_traffic.sh "recompute:1000000:5" 30 --side recompute > /dev/null 2>&1
tools/pmc_traffic.sh "recompute:8000000:5" 20 --side recompute --groups 8000000 > /dev/null 2>&1
cp gpurun_out/traffic_*.json $O/
python tools/bench_flush_latency.py > $O/flush_latency.txt 2>&1
ls $O
