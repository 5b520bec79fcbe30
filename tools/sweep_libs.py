#!/usr/bin/env python3
"""One build per row, one bench configuration per column: us per tick (median [min-max] of bench.py's repeats) and the roofline
fraction, for the experiment builds of `python -m raft_rs_amd.build --exp <name> ...` (libraftgroups_<name>.so) next to the
default library.

    python tools/sweep_libs.py --libs default,r4,w4 --configs "c5s:--workload 5 --slots 7 --sorted|c4:--slots 7" [--groups 1000000]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", default="default")
    ap.add_argument("--configs", required=True, help="name:bench args|name:bench args ...")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--repeats", type=int, default=3)
    args = ap.parse_args()
    configs = [c.split(":", 1) for c in args.configs.split("|")]
    for name, bargs in configs:
        for lib in args.libs.split(","):
            path = os.path.join(ROOT, "raft_rs_amd", "libraftgroups.so" if lib == "default" else f"libraftgroups_{lib}.so")
            if not os.path.exists(path):
                print(f"{name:12s} {lib:10s} missing {path}")
                continue
            env = dict(os.environ, RG_LIB_PATH=path)
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(args.steps), "--warmup", "3", "--repeats", str(args.repeats),
                   "--no-cpu-baseline", "--no-extras"] + bargs.split()
            try:
                out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
                d = json.loads(out.stdout.strip().splitlines()[-1])
                r = d["roofline"]
                print(f"{name:12s} {lib:10s} {r['avg_launch_us']:8.1f} us [{r.get('avg_launch_us_min', 0):.1f}-{r.get('avg_launch_us_max', 0):.1f}]  "
                      f"frac {r['frac']:.3f}  {d['value'] / 1e9:6.2f} G evals/s  kernel {r['kernel']}", flush=True)
            except Exception as e:  # noqa: BLE001
                print(f"{name:12s} {lib:10s} FAILED {type(e).__name__}: {e}", (out.stderr[-300:] if 'out' in dir() else ''), flush=True)


if __name__ == "__main__":
    main()
