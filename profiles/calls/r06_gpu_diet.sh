cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06_diet
for i in 1 2 3; do python tools/sweep_libs.py --libs default,diet --configs "send1:--inflights 256 --fused-send|send2:--inflights 256" 2>&1 | tee -a gpurun_out/r06_diet/sweep_diet.txt; done
