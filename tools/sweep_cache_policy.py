#!/usr/bin/env python3
"""Where do the cache-policy windows of rg_create lie at 3 and 7 peer slots? (the windows 1.3 x / 1.5 x /
2.5 x / 7.5 x the Infinity Cache were measured at 5 slots only and keyed on bytes of state.)

For every slot count and every engine size (state = mult x 256 MiB), the steady config-2 stream under each explicit policy
(rg_config.cache_policy: plain, stream_msgs, stream_all, resident with the default 176 MB range), us per tick as the median of
3 repeats of K ticks, and what RG_CACHE_AUTO picks. One engine alive at a time.

    python tools/sweep_cache_policy.py [--slots 3,7] [--mults 0.8,1.1,...] > profiles/r05_cache_policy_sweep.txt
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def measure(rg, torch, G, P, policy, steps, reps=3):
    eng = rg.Engine(G, P, cache_policy=policy)
    s = torch.cuda.current_stream()
    eng.set_stream(s.cuda_stream)
    eng.workload_init(2)
    T = 3 + steps
    # one recorded tick set is enough: replaying the same messages is idempotent traffic-wise (acks <= match are stale but
    # move the same bytes) -- but to keep the work identical to bench.py, record T ticks like it does when memory allows
    cols = [torch.empty((T, P, eng.stride), dtype=torch.int64, device="cuda") for _ in range(4)]
    flags = torch.empty((T, G, 8), dtype=torch.uint8, device="cuda")
    eng.checkpoint()
    for t in range(T):
        eng.workload_gen(2, t, *[c[t].data_ptr() for c in cols], flags[t].data_ptr())
        eng.tick_device(*[c[t].data_ptr() for c in cols], flags[t].data_ptr())
    eng.sync()
    us = []
    for _ in range(reps):
        eng.restore()
        for t in range(3):
            eng.tick_device(*[c[t].data_ptr() for c in cols], flags[t].data_ptr())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(s)
        for t in range(3, T):
            eng.tick_device(*[c[t].data_ptr() for c in cols], flags[t].data_ptr())
        e1.record(s)
        torch.cuda.synchronize()
        us.append(e0.elapsed_time(e1) * 1e3 / steps)
    info = eng.device_info()
    eng.close()
    del cols, flags
    torch.cuda.empty_cache()
    return float(np.median(us)), min(us), max(us), info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slots", default="3,7")
    ap.add_argument("--mults", default="0.8,1.1,1.3,1.5,2.0,2.5,3.5,5.0,7.5,9.0")
    ap.add_argument("--groups", default="", help="explicit engine sizes (groups) instead of --mults")
    ap.add_argument("--policies", default="plain,stream_msgs,stream_all,resident")
    args = ap.parse_args()
    import torch
    import raft_rs_amd as rg
    torch.cuda.set_stream(torch.cuda.Stream())
    mall = 256 * 2**20
    print("# state = mult x 256 MiB; us per tick: median [min-max] of 3 x K ticks; * = what RG_CACHE_AUTO picks; bytes/eval = 9P+58(P-0.2)+37")
    for P in [int(x) for x in args.slots.split(",")]:
        per_group = 24 * P + 40
        sizes = ([int(x) for x in args.groups.split(",")] if args.groups else
                 [int(float(x) * mall / per_group) // 256 * 256 for x in args.mults.split(",")])
        for G in sizes:
            mult = G * per_group / mall
            steps = max(6, min(30, int(3e6 * 30 / G)))
            auto = rg.Engine(G, P)
            pick = auto.device_info()["cache_policy"]
            auto.close()
            row = []
            for name, pol in (("plain", rg.CACHE.PLAIN), ("stream_msgs", rg.CACHE.STREAM_MSGS), ("stream_all", rg.CACHE.STREAM_ALL),
                              ("resident", rg.CACHE.RESIDENT)):
                if name not in args.policies.split(","):
                    continue
                if name == "resident" and G * per_group <= 176 * 2**20 + 256 * per_group:
                    row.append(f"{name}: -")
                    continue
                med, lo, hi, info = measure(rg, torch, G, P, pol, steps)
                row.append(f"{'*' if name == pick else ''}{name}: {med:7.1f} [{lo:.1f}-{hi:.1f}]")
            alg = (9 * P + 58 * (P - 0.2) + 37) * G
            print(f"P={P} mult={mult:4.1f} G={G:9d} state={G * per_group / 2**20:7.0f} MiB alg={alg / 1e6:7.0f} MB K={steps:2d} | " + " | ".join(row), flush=True)


if __name__ == "__main__":
    main()
