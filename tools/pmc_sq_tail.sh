#!/bin/bash
# SQ counters of the tick kernel over the LAST <steps> dispatches only (the timed replay of bench.py; the recording pass
# before it launches the same kernel): usage tools/pmc_sq_tail.sh <tag> <steps> <bench args...>  -> gpurun_out/pmct_<tag>.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; STEPS=$2; shift 2
O=$R/gpurun_out/pmct_$TAG
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
CMD="python $R/bench.py --steps $STEPS --warmup 3 --no-cpu-baseline --no-extras $*"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS --output-format csv -d $O/a -o a -- $CMD > /dev/null 2> $O/a.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --output-format csv -d $O/b -o b -- $CMD > /dev/null 2> $O/b.err
python - "$O" "$TAG" "$STEPS" "$*" <<'PY' > $R/gpurun_out/pmct_$TAG.txt
import csv, glob, sys, collections
O, tag, steps, args = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
print(f"# {tag}: bench.py --steps {steps} {args}  (rocprofv3 --pmc; tick kernels only; averages over the last {steps} dispatches per kernel = the timed replay)")
for sub in ("a", "b"):
    rows = collections.defaultdict(lambda: collections.defaultdict(dict))  # kernel -> dispatch -> counter -> value
    for f in glob.glob(f"{O}/{sub}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            if "k_tick" not in k: continue
            d = rows[k][int(row["Dispatch_Id"])]
            d[row["Counter_Name"]] = d.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    for k in sorted(rows):
        ids = sorted(rows[k])[-steps:]
        print(f"{k[:70]:70s} dispatches {len(rows[k])}, averaged {len(ids)}")
        names = sorted({c for i in ids for c in rows[k][i]})
        for c in names:
            v = sum(rows[k][i].get(c, 0.0) for i in ids) / len(ids)
            print(f"    {c:24s} {v:16.1f}")
PY
cat $R/gpurun_out/pmct_$TAG.txt
