cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r06_pubprof
( cd /tmp && BENCH_FORCE_DIST=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pp -o pp --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 5 --repeats 1 --no-cpu-baseline --no-extras --no-publish-compare > $GRAFT_REPO_ROOT/gpurun_out/r06_pubprof/bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/r06_pubprof/err.txt )
f=$(find /tmp/pp -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r06_pubprof/kernel_stats.csv; head -14 $f | cut -c1-150
t=$(find /tmp/pp -name "*kernel_trace.csv" | head -1); python - $t <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last 40 kernel events: name, start offset, duration, stream/queue
t0=int(rows[-60]['Start_Timestamp'])
for r in rows[-60:-20]:
    print('%-46s start %8.1f us  dur %6.1f us  q %s' % (r['Kernel_Name'][:46], (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, r.get('Queue_Id','?')))
PY
