# SQ counters behind profiles/r06_c5_lane_pair.txt: the 7-peer class of config 5 as it runs today (7-slot body) and as twice as
# many 4-peer groups (what two lanes per group could reach at best)
cd $GRAFT_REPO_ROOT
bash tools/pmc_sq.sh r06_c5_7slot --workload 5 --size-class-engines --c5-sizes 7:333334 > /dev/null 2>&1
bash tools/pmc_sq.sh r06_c5_4slot_x2 --workload 5 --size-class-engines --c5-sizes 4:666668 > /dev/null 2>&1
grep -v "^#" gpurun_out/pmc_r06_c5_7slot.txt | head -20; grep -v "^#" gpurun_out/pmc_r06_c5_4slot_x2.txt | head -20
