#!/usr/bin/env python3
"""Latency of one RawNode::step-style round trip through the host mirror for SMALL batches: k groups get one
MsgAppendResponse each (rg_step), then rg_flush + rg_ingested_results. 1 M resident groups x 5 peers."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
import raft_rs_amd as rg  # noqa: E402

G, P = 1_000_000, 5
eng = rg.Engine(G, P)
eng.workload_init(rg.WL_MAJORITY)
st = eng.read_state()
for g in range(G):
    pass
ids = [1, 2, 3, 4, 5]
t0 = time.perf_counter()
L, h = eng.L, eng.h
import ctypes as C  # noqa: E402
arr = (C.c_uint64 * 5)(*ids)
for g in range(G):
    L.rg_set_peers(h, g, arr, 5, 4)
print(f"rg_set_peers x {G}: {time.perf_counter()-t0:.2f} s")
rng = np.random.default_rng(5)
print("k groups touched: median latency of k x rg_step + rg_flush + rg_ingested_results (host wall clock)")
for k in (1, 10, 100, 1000, 4000, 10000):
    lat_step, lat_flush, lat_res = [], [], []
    for rep in range(30):
        groups = rng.choice(G, size=k, replace=False)
        idx = np.minimum(st["term_hi"][groups], st["match"][1, groups] + rep + 1)
        t0 = time.perf_counter()
        for g, i in zip(groups.tolist(), idx.tolist()):
            eng.step(g, 2, 4, i)
        t1 = time.perf_counter()
        eng.flush()
        t2 = time.perf_counter()
        gr, commit, out = eng.ingested_results()
        t3 = time.perf_counter()
        assert len(gr) == k
        lat_step.append(t1 - t0)
        lat_flush.append(t2 - t1)
        lat_res.append(t3 - t2)
    f = lambda a: float(np.median(a[5:])) * 1e6
    print(f"  k={k:6d}: steps {f(lat_step):9.1f} us (python loop)  flush {f(lat_flush):8.1f} us  results {f(lat_res):7.1f} us")
# the same loop with the resident mailbox workgroup serving the small flushes (no launch, no synchronisation per flush)
eng.mailbox_start()
print("with rg_mailbox_start (flushes of <= 256 records are served by the resident workgroup):")
for k in (1, 10, 100, 1000):
    lat_flush, lat_res = [], []
    for rep in range(60):
        groups = rng.choice(G, size=k, replace=False)
        idx = np.minimum(st["term_hi"][groups], st["match"][1, groups] + rep + 40)
        for g, i in zip(groups.tolist(), idx.tolist()):
            eng.step(g, 2, 4, i)
        t1 = time.perf_counter()
        eng.flush()
        t2 = time.perf_counter()
        gr, commit, out = eng.ingested_results()
        t3 = time.perf_counter()
        assert len(gr) == k
        lat_flush.append(t2 - t1)
        lat_res.append(t3 - t2)
    f = lambda a: float(np.median(a[10:])) * 1e6
    print(f"  k={k:6d}: flush {f(lat_flush):8.1f} us  results {f(lat_res):7.1f} us   (p90 flush {float(np.percentile(lat_flush[10:], 90))*1e6:.1f} us)")
eng.mailbox_stop()
eng.close()

# the same round trip with the Inflights on the device: + rg_send_appends + rg_send_items
eng = rg.Engine(G, P, max_inflight=8)
eng.workload_init(rg.WL_MAJORITY)
L, h = eng.L, eng.h
for g in range(G):
    L.rg_set_peers(h, g, arr, 5, 4)
print("with device Inflights: k x rg_step, then rg_flush + rg_send_appends + rg_send_items + rg_ingested_results"
      " | the same through rg_flush_send")
for k in (1, 100, 1000):
    lat, lat1 = [], []
    for rep in range(30):
        groups = rng.choice(G, size=k, replace=False)
        idx = np.minimum(st["term_hi"][groups], st["match"][1, groups] + rep + 1)
        for g, i in zip(groups.tolist(), idx.tolist()):
            eng.step(g, 2, 4, i)
        t1 = time.perf_counter()
        eng.flush()
        eng.send_appends()
        items = eng.send_items()
        gr, commit, out = eng.ingested_results()
        lat.append(time.perf_counter() - t1)
        groups = rng.choice(G, size=k, replace=False)
        idx = np.minimum(st["term_hi"][groups], st["match"][2, groups] + rep + 1)
        for g, i in zip(groups.tolist(), idx.tolist()):
            eng.step(g, 3, 4, i)
        t1 = time.perf_counter()
        eng.flush_send()
        items = eng.send_items()
        gr, commit, out = eng.ingested_results()
        lat1.append(time.perf_counter() - t1)
    print(f"  k={k:6d}: {float(np.median(lat[5:]))*1e6:8.1f} us | {float(np.median(lat1[5:]))*1e6:8.1f} us "
          f"({len(items)} work items in the last round)")
print("... and rg_flush_send with rg_mailbox_start (the resident workgroup runs the tick AND the send stage of a request):")
eng.mailbox_start()
for k in (1, 10, 50):
    lat1 = []
    for rep in range(40):
        groups = rng.choice(G, size=k, replace=False)
        idx = np.minimum(st["term_hi"][groups], st["match"][3, groups] + rep + 1)
        for g, i in zip(groups.tolist(), idx.tolist()):
            eng.step(g, 4, 4, i)
        t1 = time.perf_counter()
        eng.flush_send()
        items = eng.send_items()
        gr, commit, out = eng.ingested_results()
        lat1.append(time.perf_counter() - t1)
    print(f"  k={k:6d}: {float(np.median(lat1[5:]))*1e6:8.1f} us ({len(items)} work items in the last round; "
          f"{eng.mailbox_stats()[0]} flushes served so far)")
eng.mailbox_stop()
eng.close()
