#!/bin/bash
set -u
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r02j
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for name in c4 c5; do
  if [ $name = c4 ]; then cfg="--slots 7"; else cfg="--workload 5 --slots 7 --one-engine"; fi
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/pmc_$name -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras $cfg > $O/pmc_$name.json 2> $O/pmc_$name.err
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
for name in ("c4","c5"):
    f = glob.glob(f"gpurun_out/r02j/pmc_{name}/**/*counter_collection.csv", recursive=True)
    if not f: print(name, "no csv", glob.glob(f"gpurun_out/r02j/pmc_{name}/**/*", recursive=True)[:5]); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        if "k_tick_lane" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(name, k, {c: round(sum(v[-10:])/len(v[-10:])) for c, v in d.items()})
PY
