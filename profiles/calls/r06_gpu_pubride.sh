# A/B in one call: the publication's event on the tick's dispatch packet (default build) against hipEventRecord behind the tick
# (noride = -DRG_NO_PUB_RIDE), the N > 1 path of bench.py at world size 1 (RCCL), config 2 and the config-4 shard.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_pubride; mkdir -p $O
timeout 900 python -m pytest tests/test_publish_gpu.py -m gpu -x -q 2>&1 | tail -4 | tee $O/tests.txt
for rep in 1 2 3; do for lib in default noride; do for cfg in "" "--slots 7"; do
  L=raft_rs_amd/libraftgroups.so; [ $lib = noride ] && L=raft_rs_amd/libraftgroups_noride.so
  RG_LIB_PATH=$GRAFT_REPO_ROOT/$L BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-extras --no-publish-compare $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', '$cfg', '%.2f us per tick + publication  %.2f G evals/s' % (d['ms_per_step']*1e3, d['value']/1e9))" | tee -a $O/ab.txt
done; done; done
