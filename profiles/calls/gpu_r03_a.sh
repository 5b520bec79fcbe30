#!/bin/bash
# round 3, GPU call A: parity of the compact variant + the 64-bit-offset instantiations, config-5 timings of the compact
# variant (default / CAP 32 / CAP 64 / fixed designated wave) against the lane kernel, the stream-count sweep, SQ counters
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r03a; mkdir -p $O
( timeout 1500 python -m pytest tests/test_parity_gpu.py -m gpu -x -q 2>&1 | tail -5 ) > $O/tests_parity.txt
( timeout 1500 python -m pytest tests/test_full_size_gpu.py -m gpu -x -q -k "compact or config5" 2>&1 | tail -5 ) > $O/tests_full.txt
cat $O/tests_parity.txt $O/tests_full.txt
J=$O/bench.jsonl; : > $J
run() { tag=$1; shift; echo "## $tag" >> $J; timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 5 "$@" 2>/dev/null | tail -1 >> $J; }
L=$GRAFT_REPO_ROOT/raft_rs_amd
run "c5 size-class lane" --workload 5
run "c5 size-class compact" --workload 5 --variant 5
RG_LIB_PATH=$L/libraftgroups_cpt32.so run "c5 size-class compact cap32" --workload 5 --variant 5
RG_LIB_PATH=$L/libraftgroups_cpt64.so run "c5 size-class compact cap64" --workload 5 --variant 5
RG_LIB_PATH=$L/libraftgroups_cptw0.so run "c5 size-class compact fixed-wd" --workload 5 --variant 5
run "c5 one-engine lane" --workload 5 --slots 7 --one-engine
run "c5 one-engine compact" --workload 5 --slots 7 --one-engine --variant 5
RG_LIB_PATH=$L/libraftgroups_cpt64.so run "c5 one-engine compact cap64" --workload 5 --slots 7 --one-engine --variant 5
run "c2 lane" 
run "c2 compact" --variant 5
run "c4 shard lane" --slots 7
run "c4 shard compact" --slots 7 --variant 5
run "c2 8M lane" --groups 8000000 --steps 15
run "c2 8M compact" --groups 8000000 --steps 15 --variant 5
run "c5 size-class lane again" --workload 5
run "c5 size-class compact again" --workload 5 --variant 5
python - <<'PY' | tee $O/bench_summary.txt
import json
tag=None
for l in open('gpurun_out/r03a/bench.jsonl'):
    if l.startswith('##'): tag=l[2:].strip(); continue
    try:
        d=json.loads(l); r=d['roofline']; c=d['config']
        print('%-36s | %.2f G/s  %.1f us/step  kernel-avg %.1f us  frac %.3f' % (tag, d['value']/1e9, d['ms_per_step']*1e3, r['avg_launch_us'], r['frac']))
    except Exception as e: print('%-36s | ?? %s' % (tag, l[:80]))
PY
( cd tools/microbench && timeout 300 ./stream_sweep ) | tee $O/stream_sweep.txt
tools/pmc_sq.sh r03a_c5one_compact --workload 5 --slots 7 --one-engine --variant 5 > /dev/null 2>&1
tools/pmc_sq.sh r03a_c5one_lane --workload 5 --slots 7 --one-engine > /dev/null 2>&1
head -12 gpurun_out/pmc_r03a_c5one_compact.txt; sed -n 20,32p gpurun_out/pmc_r03a_c5one_compact.txt
