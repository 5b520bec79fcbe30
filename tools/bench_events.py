#!/usr/bin/env python3
"""rg_progress_events / rg_report_unreachable: what RawNode::report_unreachable costs through the engine. A lost connection to
one peer is an MsgUnreachable for that peer in EVERY group a node leads: a batch of G records (one per group, same slot).
1 M resident groups x 5 peers, the bench's steady-state workload (every follower in Replicate)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
import raft_rs_amd as rg  # noqa: E402
from raft_rs_amd import engine as E  # noqa: E402

G, P = 1_000_000, 5
eng = rg.Engine(G, P)
eng.workload_init(rg.WL_MAJORITY)
eng.checkpoint()
st = eng.read_state()
self_slot = ((st["cfg"] >> 16) & 7).astype(np.int64)
for n in (1, 100, 10_000, G):
    lat = []
    for rep in range(7):
        eng.restore()
        groups = np.arange(n, dtype=np.uint64) if n == G else np.sort(np.random.default_rng(rep).choice(G, size=n, replace=False)).astype(np.uint64)
        ev = np.zeros(n, dtype=E.PROGRESS_EVENT_DTYPE)
        ev["group"], ev["kind"] = groups, E.EV_UNREACHABLE
        ev["slot"] = (self_slot[groups.astype(np.int64)] + 1) % P  # a follower of every group
        eng.sync()
        t0 = time.perf_counter()
        eng.progress_events(ev)  # H2D copy of the records + one launch + synchronisation
        lat.append(time.perf_counter() - t0)
    got = eng.read_state()
    s = ev["slot"].astype(np.int64)
    g = groups.astype(np.int64)
    was_rep = (st["pflags"][g, s] & 3) == 1
    assert was_rep.mean() > 0.5 and ((got["pflags"][g, s] & 3) == 0)[was_rep].all()
    assert (got["next"][s, g] == got["match"][s, g] + 1)[was_rep].all()
    print(f"rg_progress_events, {n:8d} MsgUnreachable: median {np.median(lat[2:]) * 1e6:10.1f} us "
          f"({n / np.median(lat[2:]) / 1e6:8.2f} M events/s; {16 * n / 1e6:.2f} MB of records over PCIe)")
# the dense form: one byte per group instead of a 16-byte record
lat = []
slot1 = (((self_slot + 1) % P) + 1).astype(np.uint8)
for rep in range(7):
    eng.restore()
    eng.sync()
    t0 = time.perf_counter()
    eng.progress_event_dense(E.EV_UNREACHABLE, slot1)
    lat.append(time.perf_counter() - t0)
got2 = eng.read_state()
assert (got2["pflags"] == got["pflags"]).all() and (got2["next"] == got["next"]).all()  # (the last record run covered all G groups)
print(f"rg_progress_event_dense, {G:8d} MsgUnreachable: median {np.median(lat[2:]) * 1e6:10.1f} us "
      f"({G / np.median(lat[2:]) / 1e6:8.2f} M events/s; {G / 1e6:.2f} MB over PCIe)")
# one call by peer id through the mirror
for gi in range(64):
    eng.set_peers(gi, [1, 2, 3, 4, 5], 4)
eng.restore()
lat = []
for gi in range(64):
    to = int((self_slot[gi] + 1) % P) + 1
    t0 = time.perf_counter()
    eng.report_unreachable(gi, to)
    lat.append(time.perf_counter() - t0)
print(f"rg_report_unreachable (one group, by peer id): median {np.median(lat[8:]) * 1e6:.1f} us")
eng.close()
