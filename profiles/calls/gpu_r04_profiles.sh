#!/bin/bash
# Round 4 evidence run (build df6703a): the driver's command, rocprofv3 stats of the headline command and of config 5 placed by
# size class, PMC traffic passes of every configuration the bench line reports whose kernel changed this round.
set -u
cd $GRAFT_REPO_ROOT
R=$PWD
O=$R/gpurun_out/r04p
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
echo df6703a > $O/build_commit.txt
T0=$(date +%s%N)
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
T1=$(date +%s%N)
echo "python bench.py: $(( (T1 - T0) / 1000000 )) ms wall" > $O/bench_n1_wall.txt
cat $O/bench_n1_wall.txt
for cfg in "--slots 3" "--groups 2000000" "--workload 5 --slots 7 --sorted" "--workload 5 --slots 7 --sorted --groups 8000000 --steps 12" "--workload 5 --groups 8000000 --steps 12" "--workload 5 --slots 7 --sorted --groups 100000" "--workload 5 --groups 100000" "--fuse 4" "--fuse 8" "--workload 5 --fuse 4"; do
  timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras $cfg >> $O/bench_other_configs.jsonl 2>> $O/bench_other.err
done
cd /tmp
CMD="python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -o s -- $CMD > $O/prof_stats.json 2> $O/prof_stats.err
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -o f -- $CMD > /dev/null 2> $O/prof_fetch.err
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof_write -o w -- $CMD > /dev/null 2> $O/prof_write.err
CMD5="python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras --workload 5 --slots 7 --sorted"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats_c5 -o s -- $CMD5 > /dev/null 2> $O/prof_stats_c5.err
CMDS="python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras --inflights 256 --fused-send"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats_ts -o s -- $CMDS > /dev/null 2> $O/prof_stats_ts.err
cd $R
python tools/summarize_prof.py --tag r04 --stats $O/prof_stats --fetch $O/prof_fetch --write $O/prof_write --last 50 --out $O \
  --note "python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras (1 M groups x 5 peers, config 2), build df6703a of round 4" > /dev/null
cp $(find $O/prof_stats -name "*kernel_stats.csv" | head -1) $O/r04_kernel_stats.csv
cp $(find $O/prof_stats_c5 -name "*kernel_stats.csv" | head -1) $O/r04_c5_kernel_stats.csv
cp $(find $O/prof_stats_ts -name "*kernel_stats.csv" | head -1) $O/r04_tick_send_kernel_stats.csv
rm -rf $O/prof_stats $O/prof_fetch $O/prof_write $O/prof_stats_c5 $O/prof_stats_ts
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
ls $O; head -5 $O/r04_kernel_stats.csv; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04p/bench_n1.json').read().strip().splitlines()[-1])
print(json.dumps(d['roofline'].get('by_config'), indent=0)[:3000])
PY
