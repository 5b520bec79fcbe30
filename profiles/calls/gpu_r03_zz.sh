#!/bin/bash
# Round 3, the call after the last (6.6 GPU-minutes left): rg_progress_events / rg_report_* and the restated reference tests
# that came with them first, then the whole GPU suite.
set -u
cd $GRAFT_REPO_ROOT
R=$PWD
O=$R/gpurun_out/r03zzz
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_cpp_host.py tests/test_scenarios.py tests/test_sendstage_gpu.py -m gpu -q -k "cpp or unreachable or snapshot_failure or progress_events or report_" 2>&1 | tail -30 > $O/tests_new.txt
cat $O/tests_new.txt
timeout 330 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/tests_gpu.txt
cat $O/tests_gpu.txt
