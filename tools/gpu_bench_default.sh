#!/bin/bash
# the driver's end-of-round command, with a clock around it, and a digest of the line it prints
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-bench_default}; mkdir -p $O
T0=$(date +%s%N)
python bench.py > $O/bench_default.json 2> $O/bench_default.err
T1=$(date +%s%N)
echo "bench.py wall: $(( (T1 - T0) / 1000000 )) ms" | tee $O/wall.txt
tail -3 $O/bench_default.err
python - "$O" <<'PY'
import json, sys
O = sys.argv[1]
raw = open(O + "/bench_default.json").read()
line = json.loads(raw.strip().splitlines()[-1])
print("stdout: %d line(s), %d bytes; roofline.frac %.4f; cpu_baseline %s" % (raw.count("\n"), len(raw), line["roofline"]["frac"], (line.get("cpu_baseline") or {}).get("value")))
d = json.load(open("gpurun_out/bench_full.json"))  # the nested result (the line itself carries scalars only)
import shutil; shutil.copy("gpurun_out/bench_full.json", O + "/bench_full.json")
def show(name, o):
    r = o["roofline"]
    print("%-34s %8.2f G/s %8.1f us  frac %.3f  %s  traffic %s" % (name, o["value"] / 1e9, o.get("us_per_step", d["ms_per_step"] * 1e3), r["frac"], r["regime"], r.get("traffic")))
show("headline", d)
for k in ("recompute_only", "recompute_only_out_of_cache", "out_of_cache"):
    show(k, d[k])
for k, v in d["other_configs"].items():
    show(k, v)
    if "send_stage" in v:
        s = v["send_stage"]
        if "us_per_tick_median" in s:
            print("    send stage: tick %.1f us stage %.1f us frac %.3f bytes/group %.0f" % (s["us_per_tick_median"], s["us_per_stage_median"], s["roofline"]["frac"], s["roofline"]["bytes_per_group"]))
        else:
            print("    tick + send stage, one launch: %.1f us frac %.3f bytes/group %.0f" % (s["us_per_step_median"], s["roofline"]["frac"], s["roofline"]["bytes_per_group"]))
print(d["small_batch_latency"])
c = d["cpu_baseline"]
print({k: c[k] for k in ("value", "cores", "value_1core", "soa_value", "soa_value_1core", "soa_cores")})
print(c["config1"])
PY
