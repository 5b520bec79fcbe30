#!/bin/bash
# HBM traffic per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; gfx950: reads = 2 x FETCH_SIZE x 32 B... see
# tools/summarize_prof.py for the unit handling) of a bench.py configuration.
# usage: tools/pmc_tcc.sh <tag> <bench args...>      -> gpurun_out/tcc_<tag>.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
O=$R/gpurun_out/tcc_$TAG
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
CMD=${PMC_CMD:-"python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras $*"}
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f -o f -- $CMD > /dev/null 2> $O/f.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/w -o w -- $CMD > /dev/null 2> $O/w.err
python - "$O" "$TAG" "$*" <<'PY' > $R/gpurun_out/tcc_$TAG.txt
import csv, glob, sys, collections
O, tag, args = sys.argv[1], sys.argv[2], sys.argv[3]
print(f"# {tag}: bench.py {args}  (averages per launch; FETCH_SIZE/WRITE_SIZE in KiB as rocprofv3 reports them; gfx950 reads = 2 x FETCH_SIZE)")
tot = {}
for sub in ("f", "w"):
    acc = collections.defaultdict(float); n = collections.defaultdict(set)
    for f in glob.glob(f"{O}/{sub}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            acc[(k, row["Counter_Name"])] += float(row["Counter_Value"]); n[k].add(row["Dispatch_Id"])
    for (k, c), v in sorted(acc.items(), key=lambda kv: -kv[1])[:3]:
        per = v / len(n[k])
        mb = per * 1024 / 1e6 * (2 if c == "FETCH_SIZE" else 1)
        print(f"{k[:60]:60s} {c:12s} {per:14.1f} KiB/launch -> {mb:9.1f} MB/launch   launches {len(n[k])}")
PY
cat $R/gpurun_out/tcc_$TAG.txt
