#!/bin/bash
# Round 3, last GPU seconds: the progress-event tests on the build that honours RG_CFG_PRESENT and has the dense form; the
# event micro-benchmark.
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ev
mkdir -p $O
timeout 150 python -m pytest tests/test_sendstage_gpu.py tests/test_scenarios.py tests/test_cpp_host.py -m gpu -q -k "progress_event or report_ or unreachable or snapshot_failure or cpp" 2>&1 | tail -12 | tee $O/tests_events.txt
timeout 100 python tools/bench_events.py 2>&1 | tail -8 | tee $O/bench_events.txt
