#!/bin/bash
# round 4, call v: the resident range only while the engine is the process' only live one (g_live_engines) -- the parity tests
# of both streamed forms, 2.4 M x 5 (the range in use) and config 5 at 8 M as three engines (no range: 724 us before there was one)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04v2
O=gpurun_out/r04v2/live.txt; : > $O
timeout 300 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "partly_resident or everything_streamed or workload_stream" 2>&1 | tail -1 >> $O
run() { echo -n "$* : " >> $O; timeout 300 python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.1f us frac %.3f' % (d['ms_per_step']*1e3, d['roofline']['frac']))" >> $O; }
run --steps 30 --groups 2400000
run --steps 12 --workload 5 --groups 8000000
cat $O
