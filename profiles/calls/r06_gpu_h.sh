cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06_i; O=gpurun_out/r06_i
timeout 1500 python -m pytest tests/test_sendstage_gpu.py tests/test_placement_gpu.py tests/test_api_sequences_gpu.py -m gpu -x -q 2>&1 | tail -15 | cut -c1-300 > $O/tests.txt; tail -3 $O/tests.txt
timeout 900 python -m pytest tests/test_full_size_gpu.py -m gpu -x -q -k "send_stage" 2>&1 | tail -5 | cut -c1-300 > $O/tests_full.txt; tail -2 $O/tests_full.txt
for i in 1 2; do
python tools/sweep_libs.py --libs cw2,default --configs "send1:--inflights 256 --fused-send|send2:--inflights 256" 2>&1 | tee -a $O/sweep_windows4.txt
done
