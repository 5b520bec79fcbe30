#!/usr/bin/env python3
"""Measure the sparse path (rg_ingest + rg_tick_ingested, host records -> results) and the
recompute-only kernel on one MI355X. Prints a small table; numbers go into DESIGN.md / profiles/."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
import raft_rs_amd as rg  # noqa: E402
from raft_rs_amd.engine import WIRE_DTYPE  # noqa: E402

G, P = 1_000_000, 5
eng = rg.Engine(G, P)
eng.workload_init(rg.WL_MAJORITY)
st = eng.read_state()
rng = np.random.default_rng(1)
print(f"sparse path, {G} groups x {P} peers (host records in, compact results out; includes H2D/D2H and syncs)")
for frac in (0.001, 0.01, 0.05, 0.2):
    n_g = int(G * frac)
    times_i, times_t = [], []
    for rep in range(6):
        groups = rng.choice(G, size=n_g, replace=False).astype(np.uint64)
        recs = np.zeros(n_g * 4, dtype=WIRE_DTYPE)
        k = 0
        for p in range(1, 5):
            sl = recs[k:k + n_g]
            sl["group"] = groups
            sl["slot"] = p
            sl["flags"] = rg.MF.VALID | rg.MF.SENT
            sl["index"] = np.minimum(st["term_hi"][groups], st["match"][p, groups] + rep + 1)
            sl["commit"] = np.minimum(st["commit"][groups], sl["index"])
            k += n_g
        t0 = time.perf_counter()
        dup = eng.ingest(recs)
        t1 = time.perf_counter()
        n = eng.tick_ingested()
        t2 = time.perf_counter()
        assert dup == 0 and n == n_g
        times_i.append(t1 - t0)
        times_t.append(t2 - t1)
    # the same work through rg_ingest_tick (one host<->device round trip, results cached on the host)
    times_c = []
    for rep in range(6, 12):
        groups = rng.choice(G, size=n_g, replace=False).astype(np.uint64)
        k = 0
        for p in range(1, 5):
            sl = recs[k:k + n_g]
            sl["group"] = groups
            sl["index"] = np.minimum(st["term_hi"][groups], st["match"][p, groups] + rep + 1)
            sl["commit"] = np.minimum(st["commit"][groups], sl["index"])
            k += n_g
        t0 = time.perf_counter()
        n, dup = eng.ingest_tick(recs)
        eng.ingested_results()
        times_c.append(time.perf_counter() - t0)
        assert dup == 0 and n == n_g
    tc = np.median(times_c[1:])
    ti, tt = np.median(times_i[1:]), np.median(times_t[1:])
    print(f"  {frac*100:5.1f}% of groups touched: {len(recs):8d} records  ingest {ti*1e6:8.1f} us  "
          f"tick {tt*1e6:7.1f} us  -> {len(recs)/(ti+tt)/1e6:7.1f} M msgs/s, {n_g/(ti+tt)/1e6:6.2f} M group-evals/s"
          f"  | rg_ingest_tick + results {tc*1e6:8.1f} us -> {len(recs)/tc/1e6:7.1f} M msgs/s")

# recompute-only (Raft::maybe_commit for every group, no messages): B0 = 8P+37 bytes per group
coop = rg.Engine(G, P, variant=rg.VARIANT_COOP)
coop.workload_init(rg.WL_MAJORITY)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
for _ in range(5):
    eng.recompute()
e0.record()
K = 50
for _ in range(K):
    eng.recompute()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / K
b0 = (8 * P + 37) * G
print(f"recompute-only k_recompute<{P}>: {us:.1f} us per sweep, {G/us/1e3:.2f} G recomputes/s, "
      f"algorithmic {b0/us/1e3:.0f} GB/s ({b0/us/1e3/8000*100:.1f}% of 8 TB/s; B0 = {8*P+37} B/group)")
coop.set_stream(torch.cuda.current_stream().cuda_stream)
for _ in range(5):
    coop.recompute()
e0.record()
for _ in range(K):
    coop.recompute()
e1.record()
torch.cuda.synchronize()
usc = e0.elapsed_time(e1) * 1e3 / K
print(f"recompute-only, wave-cooperative variant (8 lanes/group, shuffle rank-select): {usc:.1f} us per sweep, "
      f"{G/usc/1e3:.2f} G recomputes/s ({b0/usc/1e3/8000*100:.1f}% of 8 TB/s)")
coop.close()
eng.close()
