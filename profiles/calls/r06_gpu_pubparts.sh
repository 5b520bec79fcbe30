# What each part of a publication costs at world size 1 (experiment build pubdbg; bits: 2 = no slice reset, 4 = no exchange): the
# replica is WRONG with them -- bench.py's verification is expected to complain, only the timing is read.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_pubparts; mkdir -p $O
for rep in 1 2; do for dbg in 0 2 4 6; do
  BENCH_SKIP_VERIFY=1 RG_PUB_DEBUG=$dbg RG_LIB_PATH=$GRAFT_REPO_ROOT/raft_rs_amd/libraftgroups_pubdbg.so BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras --no-publish-compare 2>$O/err_$dbg.txt | python -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
try:
    d=json.loads(t[-1]); print('RG_PUB_DEBUG=$dbg', '%.2f us per tick + publication' % (d['ms_per_step']*1e3))
except Exception as e: print('RG_PUB_DEBUG=$dbg', 'no line', e)" | tee -a $O/parts.txt
done; done
tail -n 3 $O/err_6.txt
