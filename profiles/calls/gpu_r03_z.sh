#!/bin/bash
# Round 3, last call (10 GPU-minutes left): the whole GPU suite on the final build (incl. the two new reference tests on the
# built messages, tests/test_cpp_host.py), smoke(), the driver's bench command.
set -u
cd $GRAFT_REPO_ROOT
R=$PWD
O=$R/gpurun_out/r03zz
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 120 python -m pytest tests/test_cpp_host.py -m gpu -q 2>&1 | tail -25 > $O/tests_cpp.txt
cat $O/tests_cpp.txt
timeout 420 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/tests_gpu.txt
cat $O/tests_gpu.txt
timeout 120 python -c "import __graft_entry__ as e; e.smoke(); print('SMOKE_OK')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
T0=$(date +%s%N)
timeout 300 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
T1=$(date +%s%N)
echo "python bench.py: $(( (T1 - T0) / 1000000 )) ms wall" > $O/bench_n1_wall.txt
cat $O/bench_n1_wall.txt
head -c 600 $O/bench_n1.json
