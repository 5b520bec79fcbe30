#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + separate FETCH_SIZE / WRITE_SIZE PMC passes) into
the summary files kept under profiles/.

    tools/summarize_prof.py --tag r01 --stats gpurun_out/prof_stats --fetch gpurun_out/prof_fetch \
        --write gpurun_out/prof_write [--kernel k_tick_lane]

HBM traffic per launch follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and
WRITE_SIZE are in KiB-units of 1024 B... (counter value x 1024 bytes); on gfx950 FETCH_SIZE reports
HALF of the bytes of a coalesced streaming read, so reads = 2 x FETCH_SIZE x 1024; WRITE_SIZE is
taken at face value (uncalibrated per the guide; it matches the expected store bytes of this kernel
within ~10%).
"""
import argparse
import collections
import csv
import glob
import json
import os


def find(d, suffix):
    m = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    return m[0] if m else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    ap.add_argument("--stats")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--kernel", default="k_tick_lane")
    ap.add_argument("--out", default="profiles")
    ap.add_argument("--note", default="")
    ap.add_argument("--last", type=int, default=0, help="also average the LAST N launches of --kernel (the timed region)")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    summary = {"tag": a.tag, "kernel": a.kernel, "note": a.note}
    lines = [f"# rocprofv3 summary {a.tag}", "", a.note, ""]
    if a.stats:
        ks = find(a.stats, "kernel_stats.csv")
        rows = list(csv.DictReader(open(ks)))
        lines += ["## rocprofv3 --kernel-trace --stats (kernel_stats.csv)", "",
                  "| kernel | calls | avg us | min us | max us | % |", "|---|---|---|---|---|---|"]
        for r in rows:
            lines.append(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.2f} | "
                         f"{float(r['MinNs'])/1e3:.2f} | {float(r['MaxNs'])/1e3:.2f} | {r['Percentage']} |")
            if a.kernel in r["Name"]:
                summary["avg_launch_us"] = float(r["AverageNs"]) / 1e3
                summary["calls"] = int(r["Calls"])
        lines.append("")
        kt = find(a.stats, "kernel_trace.csv")
        if a.last and kt:
            d = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(kt))
                 if a.kernel in r["Kernel_Name"]]
            d.sort()
            tail = d[-a.last:]
            avg = sum(e - s for s, e in tail) / len(tail) / 1e3
            span = (tail[-1][1] - tail[0][0]) / len(tail) / 1e3
            summary["timed_region_avg_launch_us"] = avg
            summary["timed_region_span_per_launch_us"] = span
            lines += [f"Timed region (last {len(tail)} `{a.kernel}` launches of kernel_trace.csv): average duration "
                      f"**{avg:.2f} us**, start-to-end span per launch {span:.2f} us (bench.py's HIP-event figure is the span).", ""]
    for name, d in (("FETCH_SIZE", a.fetch), ("WRITE_SIZE", a.write)):
        if not d:
            continue
        cc = find(d, "counter_collection.csv")
        vals = collections.defaultdict(list)
        for r in csv.DictReader(open(cc)):
            if r["Counter_Name"] == name:
                vals[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        lines += [f"## rocprofv3 --pmc {name} (separate pass)", "", "| kernel | launches | mean | min | max |",
                  "|---|---|---|---|---|"]
        for k, v in vals.items():
            lines.append(f"| `{k[:70]}` | {len(v)} | {sum(v)/len(v):.1f} | {min(v):.1f} | {max(v):.1f} |")
            if a.kernel in k:
                summary[name] = sum(v) / len(v)
        lines.append("")
    if "FETCH_SIZE" in summary and "WRITE_SIZE" in summary:
        rd = 2.0 * summary["FETCH_SIZE"] * 1024
        wr = summary["WRITE_SIZE"] * 1024
        summary["hbm_read_bytes_per_launch"] = rd
        summary["hbm_write_bytes_per_launch"] = wr
        summary["hbm_bytes_per_launch"] = rd + wr
        lines += ["## HBM traffic per launch (gfx950 correction: reads = 2 x FETCH_SIZE)", "",
                  f"reads {rd/1e6:.1f} MB + writes {wr/1e6:.1f} MB = **{(rd+wr)/1e6:.1f} MB** per `{a.kernel}` launch", ""]
    with open(os.path.join(a.out, f"{a.tag}_rocprof_summary.md"), "w") as f:
        f.write("\n".join(lines))
    with open(os.path.join(a.out, f"{a.tag}_rocprof_summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    print(json.dumps(summary))


if __name__ == "__main__":
    main()
