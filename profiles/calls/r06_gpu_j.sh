cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06_j; O=gpurun_out/r06_j
for i in 1 2 3; do
python tools/sweep_libs.py --libs w20,default,cw2 --configs "send1:--inflights 256 --fused-send|send2:--inflights 256" 2>&1 | tee -a $O/sweep_w20.txt
done
export TMPDIR=/tmp; cd /tmp
for lib in w20 default; do
L=$GRAFT_REPO_ROOT/raft_rs_amd/libraftgroups_$lib.so; [ $lib = default ] && L=$GRAFT_REPO_ROOT/raft_rs_amd/libraftgroups.so
RG_LIB_PATH=$L timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$lib -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 3 --repeats 1 --no-cpu-baseline --no-extras --inflights 256 > /dev/null 2>&1
f=$(find /tmp/p_$lib -name "*kernel_stats.csv" | head -1); echo "== $lib"; head -4 $f | cut -c1-150
RG_LIB_PATH=$L timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d /tmp/q_$lib -o q -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-extras --inflights 256 > /dev/null 2>&1
python - "$lib" <<'PY'
import csv,glob,sys,collections
lib=sys.argv[1]; acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"/tmp/q_{lib}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0]
        if "k_send_dense" in k or "k_tick_lane" in k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,d in acc.items(): print(lib, k[:40], {c: round(sum(v[-10:])/len(v[-10:])) for c,v in d.items()})
PY
done 2>&1 | tee $GRAFT_REPO_ROOT/$O/sq_w20.txt
