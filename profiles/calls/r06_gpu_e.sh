cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06_e; O=gpurun_out/r06_e
timeout 1500 python -m pytest tests/test_placement_gpu.py tests/test_publish_gpu.py tests/test_scenarios.py tests/test_sendstage_gpu.py tests/test_sparse_path_gpu.py tests/test_votes_and_mirror_gpu.py -m gpu -x -q 2>&1 | tail -15 | cut -c1-300 > $O/tests.txt; tail -3 $O/tests.txt
python tools/sweep_libs.py --libs default,noring --configs "send1:--inflights 256 --fused-send|send2:--inflights 256" 2>&1 | tee $O/sweep_noring.txt
for i in 1 2; do
python tools/sweep_libs.py --libs default --configs "c5_357:--workload 5 --size-class-engines|c5_3544:--workload 5 --size-class-engines --c5-sizes 3:333333,5:333333,4:666668|c5_one:--workload 5" 2>&1 | tee -a $O/c5_pair_emulation.txt
done
RG_LIB_PATH=$GRAFT_REPO_ROOT/raft_rs_amd/libraftgroups_noring.so bash tools/pmc_traffic.sh noring_fused 20 --inflights 256 --fused-send > $O/pmc_noring.txt 2>&1; tail -12 $O/pmc_noring.txt
