#!/bin/bash
# round 4, call r: the two modes of the send forms again. Call q refuted the cache-edge hypothesis (800 k groups cost the same
# per group as 1 M in the slow mode, every process of that box was slow) and call p's modes came in runs (fast x3, slow x4, fast):
# a device clock / power state? Sample rocm-smi's clocks and power while the bench runs.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04r2
O=gpurun_out/r04r2/clocks.txt; : > $O
rocm-smi --showclocks --showpower 2>/dev/null | head -40 >> $O
sample() { while true; do rocm-smi --showclocks --showpower --json 2>/dev/null | python -c "
import json,sys
try:
    d=json.load(sys.stdin); c=d[sorted(d)[0]]
    print(' '.join('%s=%s' % (k.split('(')[0].strip().replace(' ','_'), str(v).split(')')[0].strip('(')) for k,v in c.items() if any(t in k.lower() for t in ('sclk','mclk','fclk','socclk','power'))))
except Exception as e: print('?', e)
" >> $1; sleep 0.4; done; }
run() {
  S=gpurun_out/r04r2/samples.tmp; : > $S
  sample $S & SP=$!
  R=$(timeout 300 python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f us' % (d['ms_per_step']*1e3))")
  kill $SP 2>/dev/null; wait $SP 2>/dev/null
  echo "$TAG $* : $R" >> $O
  # the samples taken while the GPU was busy are the last ones before the bench exits
  tail -4 $S | sed 's/^/      /' >> $O
}
for rep in 1 2 3 4; do
for L in base w1; do
  if [ $L = base ]; then unset RG_LIB_PATH; else export RG_LIB_PATH=$GRAFT_REPO_ROOT/raft_rs_amd/libraftgroups_$L.so; fi
  TAG=$L
  run --steps 400 --inflights 256 --fused-send
done
unset RG_LIB_PATH; TAG=base
run --steps 400
done
cat $O
