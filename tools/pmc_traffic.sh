#!/bin/bash
# HBM traffic per STEP of a bench.py configuration: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (kernel-trace
# only), averaged over the LAST <steps> dispatches of every tick / send / recompute kernel (the timed replay), corrected as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (reads = 2 x FETCH_SIZE x 1024 B, writes = WRITE_SIZE x 1024 B).
# usage: tools/pmc_traffic.sh <key> <steps> <bench args...>   -> gpurun_out/traffic_<tag>.json  (key = profiles/traffic.json key)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
KEY=$1; STEPS=$2; shift 2
TAG=$(echo "$KEY" | tr ':' '_')
O=$R/gpurun_out/traffic_$TAG
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
CMD="python $R/bench.py --steps $STEPS --warmup 3 --repeats 1 --no-cpu-baseline --no-extras $*"
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f -o f -- $CMD > /dev/null 2> $O/f.err
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/w -o w -- $CMD > /dev/null 2> $O/w.err
python - "$O" "$KEY" "$STEPS" "$*" <<'PY' > $R/gpurun_out/traffic_$TAG.json
import csv, glob, sys, collections, json, re
O, key, steps, args = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
res = {"key": key, "command": f"bench.py --steps {steps} --warmup 3 --repeats 1 --no-cpu-baseline --no-extras {args}", "kernels": {}}
for sub, name in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
    rows = collections.defaultdict(dict)
    for f in glob.glob(f"{O}/{sub}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            if not re.search(r"k_tick|k_tick_send|k_send_dense|k_send_appends|k_recompute", k) or row["Counter_Name"] != name:
                continue
            rows[k][int(row["Dispatch_Id"])] = rows[k].get(int(row["Dispatch_Id"]), 0.0) + float(row["Counter_Value"])
    for k, d in rows.items():
        ids = sorted(d)[-steps:]
        v = sum(d[i] for i in ids) / len(ids) * 1024 * (2 if name == "FETCH_SIZE" else 1)
        res["kernels"].setdefault(k, {})["read" if name == "FETCH_SIZE" else "write"] = v
        res["kernels"][k]["launches_averaged"] = len(ids)
res["read"] = sum(k.get("read", 0) for k in res["kernels"].values())
res["write"] = sum(k.get("write", 0) for k in res["kernels"].values())
res["bytes"] = res["read"] + res["write"]
print(json.dumps(res, indent=1))
PY
rm -rf $O
cat $R/gpurun_out/traffic_$TAG.json | head -30
