#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer boundary: rg_tick(host message columns) for 1 M groups x 5 peers, with
pageable and with pinned (page-locked) caller buffers; and the same for rg_results (D2H of commit + out).
Numbers go into DESIGN.md section 4 (they are never bench.py's `value`)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import raft_rs_amd as rg  # noqa: E402
from raft_rs_amd import engine as E  # noqa: E402

G, P = 1_000_000, 5
eng = rg.Engine(G, P)
eng.workload_init(rg.WL_MAJORITY)
st = eng.read_state()
mb = rg.MsgBuffers(G, P, eng.stride)
E.workload_gen_host(st, mb, rg.WL_MAJORITY, 0)
mb.m_flags &= np.uint8(0xEF)  # no SENT: the same tick can be replayed from a checkpoint
eng.checkpoint()
bytes_in = mb.m_index.nbytes * 2 + mb.m_flags.nbytes  # m_hint / m_rs are not passed (no rejects in this stream)


class View:
    pass


def run(label, bufs, reps=12):
    v = View()
    v.m_index, v.m_commit, v.m_flags = bufs
    v.m_hint = v.m_rs = v.m_logterm = None
    ts = []
    for _ in range(reps):
        eng.restore()
        eng.sync()
        t0 = time.perf_counter()
        eng.tick(v)  # H2D copies + kernel + sync
        ts.append(time.perf_counter() - t0)
    t = float(np.median(ts[2:]))
    print(f"  rg_tick, {label:9s} caller buffers: {t*1e3:7.3f} ms per tick = {G/t/1e9:5.2f} G evals/s "
          f"({bytes_in/t/1e9:5.1f} GB/s over PCIe, {bytes_in/1e6:.0f} MB in)")


print(f"host-fed dense tick, {G} groups x {P} peers (message columns cross PCIe every tick)")
run("pageable", (mb.m_index, mb.m_commit, mb.m_flags))
pin = [torch.from_numpy(a).pin_memory() for a in (mb.m_index.view(np.int64), mb.m_commit.view(np.int64), mb.m_flags)]
run("pinned", tuple(t.numpy() if t.dtype == torch.uint8 else t.numpy().view(np.uint64) for t in pin))
for label, alloc in (("pageable", lambda n, dt: np.empty(n, dtype=dt)),
                     ("pinned", lambda n, dt: torch.empty(n, dtype=getattr(torch, np.dtype(dt).name.replace("uint", "int"))).pin_memory().numpy().view(dt))):
    commit, out = alloc(G, np.uint64), alloc(G, np.uint32)
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        eng._check(eng.L.rg_results(eng.h, commit.ctypes.data, out.ctypes.data))
        ts.append(time.perf_counter() - t0)
    t = float(np.median(ts[2:]))
    print(f"  rg_results, {label:9s} caller buffers: {t*1e3:7.3f} ms ({12*G/t/1e9:5.1f} GB/s, 12 MB out)")
eng.close()
