cd $GRAFT_REPO_ROOT
O=gpurun_out/others.jsonl; : > $O
run() { timeout 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 >> $O; }
run --workload 3
run --workload 5
run --slots 7
run --slots 3
run --workload 5 --slots 7 --one-engine
run --groups 4000000 --steps 20
run --groups 8000000 --steps 20
run --variant 2
run --split 2
run --fuse 4
run --fuse 8
run --groups 8000000 --steps 24 --fuse 8
BENCH_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 >> $O
python - <<'PY'
import json
for l in open('gpurun_out/others.jsonl'):
    try:
        d=json.loads(l); r=d['roofline']; c=d['config']
        print('%-62s G=%d P=%d fuse=%d | %.2f G/s  %.1f us  frac %.3f | %s' % (c['workload'][:62], c['groups_per_gpu'], c['peer_slots'], c['ticks_per_launch'], d['value']/1e9, d['ms_per_step']*1e3, r['frac'], c['sharding'][:40]))
    except Exception as e: print('??', l[:80])
PY
