#!/bin/bash
# round 4, call w: rocprofv3 kernel stats of the 2.4 M-group tick on the last build -- which kernel runs there (k_tick_split)
set -u
cd $GRAFT_REPO_ROOT
R=$PWD
O=$R/gpurun_out/r04w2
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o s -- python $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-extras --groups 2400000 > $O/bench.json 2> $O/err.txt
cp $(find $O/p -name "*kernel_stats.csv" | head -1) $O/r04_2m4_kernel_stats.csv
rm -rf $O/p
head -4 $O/r04_2m4_kernel_stats.csv | cut -c1-160
