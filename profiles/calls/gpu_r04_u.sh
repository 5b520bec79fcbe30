#!/bin/bash
# round 4, call u: partial residency (k_tick_split) -- the first R MB worth of groups keep their state in the Infinity Cache,
# the rest of the engine is streamed; against all-streamed (R = 0) and plain accesses, over engine sizes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04u
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "partly_resident or everything_streamed" 2>&1 | tail -2 > gpurun_out/r04u/tests.txt
cat gpurun_out/r04u/tests.txt
O=gpurun_out/r04u/resident.txt; : > $O
run() { echo -n "$TAG $* : " >> $O; timeout 300 python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f us frac %.3f' % (d['ms_per_step']*1e3, d['roofline']['frac']))" >> $O; }
nt() { export RG_NT_ALL=1; export RG_NT_RESIDENT_MB=$1; TAG="streamed, resident $1 MB"; }
plain() { export RG_NT_ALL=0; unset RG_NT_RESIDENT_MB; TAG="plain"; }
for rep in 1 2; do for R in 0 128 176 224; do nt $R; run --steps 16 --groups 8000000; done; done
for R in 0 128 176 224; do nt $R; run --steps 20 --groups 4000000; done
plain; run --steps 30 --groups 2400000
for R in 0 176 224; do nt $R; run --steps 30 --groups 2400000; done
plain; run --steps 30 --groups 2000000
for R in 176 224; do nt $R; run --steps 30 --groups 2000000; done
plain; run --steps 12 --groups 16000000
for R in 0 176; do nt $R; run --steps 12 --groups 16000000; done
cat $O
