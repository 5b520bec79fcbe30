#!/bin/bash
# round 3, GPU call D: the paused-ack shortcut (RgTick::paused_acks) in the lane / compact kernels
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03d; mkdir -p $O
( timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q 2>&1 | tail -3 ) > $O/tests_parity.txt; cat $O/tests_parity.txt
J=$O/bench.jsonl; : > $J
run() { tag=$1; shift; echo "## $tag" >> $J; timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 5 "$@" 2>/dev/null | tail -1 >> $J; }
L=$GRAFT_REPO_ROOT/raft_rs_amd
for lib in "" pac; do
  if [ -n "$lib" ]; then export RG_LIB_PATH=$L/libraftgroups_$lib.so; fi
  run "$lib c5 size-class lane" --workload 5
  run "$lib c5 size-class compact" --workload 5 --variant 5
  run "$lib c5 one-engine lane" --workload 5 --slots 7 --one-engine
  run "$lib c5 one-engine compact" --workload 5 --slots 7 --one-engine --variant 5
  run "$lib c2 lane"
  run "$lib c4 shard lane" --slots 7
  run "$lib c2 8M lane" --groups 8000000 --steps 15
done
unset RG_LIB_PATH
python - <<'PY' | tee $O/bench_summary.txt
import json
tag=None
for l in open('gpurun_out/r03d/bench.jsonl'):
    if l.startswith('##'): tag=l[2:].strip(); continue
    try:
        d=json.loads(l); r=d['roofline']
        print('%-36s | %.2f G/s  %.1f us/step  frac %.3f' % (tag, d['value']/1e9, d['ms_per_step']*1e3, r['frac']))
    except Exception as e: print('%-36s | ?? %s' % (tag, l[:80]))
PY
tools/pmc_sq_tail.sh r03d_c5one_lane 20 --workload 5 --slots 7 --one-engine > /dev/null 2>&1
grep "INSTS_VALU\|INSTS_SALU\|WAVE_CYCLES\|^#" gpurun_out/pmct_r03d_c5one_lane.txt
