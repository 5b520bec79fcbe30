#!/bin/bash
# SQ counter passes (rocprofv3 --pmc, kernel-trace only) of bench.py configurations: instruction mix and where wave cycles go.
# usage: tools/pmc_sq.sh <tag> <bench args...>      -> gpurun_out/pmc_<tag>.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
O=$R/gpurun_out/pmc_$TAG
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
CMD=${PMC_CMD:-"python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras $*"}
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d $O/a -o a -- $CMD > /dev/null 2> $O/a.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --output-format csv -d $O/b -o b -- $CMD > /dev/null 2> $O/b.err
python - "$O" "$TAG" "$*" <<'PY' > $R/gpurun_out/pmc_$TAG.txt
import csv, glob, sys, collections
O, tag, args = sys.argv[1], sys.argv[2], sys.argv[3]
print(f"# {tag}: bench.py {args}  (rocprofv3 --pmc, averages per launch of each kernel)")
for sub in ("a", "b"):
    files = glob.glob(f"{O}/{sub}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for f in files:
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k].add(row["Dispatch_Id"])
    for k in sorted(acc, key=lambda k: -sum(acc[k].values()))[:4]:
        print(f"{k[:70]:70s} launches {len(n[k])}")
        for c, v in sorted(acc[k].items()):
            print(f"    {c:24s} {v/len(n[k]):16.1f}")
PY
cat $R/gpurun_out/pmc_$TAG.txt
