#!/bin/bash
# Round 2 evidence run: every bench line DESIGN.md quotes + rocprofv3 stats and the PMC passes of the headline command.
set -u
cd "$(dirname "$0")/.."
R=$PWD
O=$R/gpurun_out/r02p
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python bench.py --steps 50 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err
for cfg in "--workload 3" "--slots 7" "--slots 3" "--workload 5" "--workload 5 --slots 7 --one-engine" "--groups 4000000 --steps 20" "--groups 8000000 --steps 20" "--variant 2" "--variant 4" "--split 2" "--fuse 4" "--fuse 8" "--inflights 256"; do
  timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras $cfg >> $O/bench_other_configs.jsonl 2>> $O/bench_other.err
done
for cfg in "" "--slots 7" "--workload 5"; do
  BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline $cfg >> $O/bench_dist_ws1.jsonl 2>> $O/bench_dist.err
done
BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --steps 20 --warmup 3 --groups 500000 --no-cpu-baseline >> $O/bench_dist_share2.jsonl 2>> $O/bench_dist.err
python tools/bench_send.py > $O/send_stage.txt 2>&1
python tools/bench_flush_latency.py > $O/flush_latency.txt 2>&1
cd /tmp
CMD="python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -o s -- $CMD > $O/prof_stats.json 2> $O/prof_stats.err
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -o f -- $CMD > /dev/null 2> $O/prof_fetch.err
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof_write -o w -- $CMD > /dev/null 2> $O/prof_write.err
CMD5="python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras --workload 5"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats_c5 -o s -- $CMD5 > /dev/null 2> $O/prof_stats_c5.err
cd $R
# SQ instruction mix / wait breakdown and HBM traffic of the steady 7-slot shard against config 5 in one 7-slot engine
bash tools/pmc_sq.sh c4 --slots 7 > /dev/null 2>&1
bash tools/pmc_sq.sh c5one --workload 5 --slots 7 --one-engine > /dev/null 2>&1
bash tools/pmc_tcc.sh c4 --slots 7 > /dev/null 2>&1
bash tools/pmc_tcc.sh c5one --workload 5 --slots 7 --one-engine > /dev/null 2>&1
CONFIGS=8:0 PMC_CMD="python $R/tools/bench_send.py" bash tools/pmc_sq.sh send > /dev/null 2>&1
CONFIGS=8:0 PMC_CMD="python $R/tools/bench_send.py" bash tools/pmc_tcc.sh send > /dev/null 2>&1
cp gpurun_out/pmc_*.txt gpurun_out/tcc_*.txt $O/ 2>/dev/null
rm -rf gpurun_out/pmc_c4 gpurun_out/pmc_c5one gpurun_out/pmc_send gpurun_out/tcc_c4 gpurun_out/tcc_c5one gpurun_out/tcc_send
ls -R $O | head -60
