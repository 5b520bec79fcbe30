cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06_f; O=gpurun_out/r06_f
timeout 1500 python -m pytest tests/test_sendstage_gpu.py tests/test_placement_gpu.py tests/test_api_sequences_gpu.py -m gpu -x -q 2>&1 | tail -15 | cut -c1-300 > $O/tests.txt; tail -3 $O/tests.txt
timeout 900 python -m pytest tests/test_full_size_gpu.py -m gpu -x -q -k "send_stage" 2>&1 | tail -5 | cut -c1-300 > $O/tests_full.txt; tail -2 $O/tests_full.txt
for i in 1 2; do
python tools/sweep_libs.py --libs cw2,default --configs "send1:--inflights 256 --fused-send|send2:--inflights 256" 2>&1 | tee -a $O/sweep_windows.txt
done
python tools/sweep_libs.py --libs default --configs "c5_7:--workload 5 --size-class-engines --c5-sizes 7:333334|c5_44:--workload 5 --size-class-engines --c5-sizes 4:666668|c5_5:--workload 5 --size-class-engines --c5-sizes 5:333333|c5_3:--workload 5 --size-class-engines --c5-sizes 3:333333" 2>&1 | tee -a $O/c5_pair_emulation2.txt
bash tools/pmc_traffic.sh 2:1000000:5:inflights:fused-send 20 --inflights 256 --fused-send > $O/pmc_fused.txt 2>&1; tail -12 $O/pmc_fused.txt
bash tools/pmc_traffic.sh 2:1000000:5:inflights 20 --inflights 256 > $O/pmc_two.txt 2>&1; tail -4 $O/pmc_two.txt
