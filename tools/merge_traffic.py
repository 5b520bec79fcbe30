#!/usr/bin/env python3
"""Fold the per-configuration PMC results of tools/pmc_traffic.sh (gpurun_out/.../traffic_*.json) into the two tracked files:
profiles/<round>_traffic_passes.json (every pass as measured, per kernel) and profiles/traffic.json (bytes per step by bench.py's
key, the `roofline.traffic` of the bench line, with the build and the call script they came from).

usage: tools/merge_traffic.py <dir with traffic_*.json> <round tag, e.g. r04> <build commit> <call script>"""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src, tag, commit, script = sys.argv[1:5]
    sys.path.insert(0, ROOT)
    import bench
    csrc = bench.csrc_sha16()  # (run this with the tree the passes were measured on checked out)
    passes_path = os.path.join(ROOT, "profiles", f"{tag}_traffic_passes.json")
    passes = json.load(open(passes_path)) if os.path.exists(passes_path) else {}
    table_path = os.path.join(ROOT, "profiles", "traffic.json")
    table = json.load(open(table_path))
    for f in sorted(glob.glob(os.path.join(src, "traffic_*.json"))):
        d = json.load(open(f))
        if not d.get("kernels"):
            print("no kernels in", f, "(pass failed?) -- skipped")
            continue
        key, note = d["key"], ""
        kernels = d["kernels"]
        if key.startswith("recompute:"):  # (`--side recompute` runs three ticks of the stream first: only the recompute launches count)
            kernels = {k: v for k, v in kernels.items() if k.startswith("k_recompute")}
            note = "; k_recompute launches only"
        if key.endswith(":fused-send"):  # the recording pass of the bench runs the two-launch kernels as well: only the timed ones
            kernels = {k: v for k, v in kernels.items() if k.startswith("k_tick_send")}
            note = "; the k_tick_send launches of the timed replay only"
        d["build"] = commit
        passes[key] = d
        b = sum(v.get("read", 0) + v.get("write", 0) for v in kernels.values())
        table[key] = {"bytes": b, "source": f"profiles/{tag}_traffic_passes.json ({script}; {d['command']}{note})", "commit": commit,
                      "csrc_sha16": csrc}
        print(f"{key:45s} {b / 1e6:9.1f} MB")
    json.dump(passes, open(passes_path, "w"), indent=1)
    json.dump(table, open(table_path, "w"), indent=1)


if __name__ == "__main__":
    main()
