#!/usr/bin/env python3
"""Long differential soak on a GPU box: engine (C ABI) vs oracle over many ticks. Not part of the test
suite (minutes of CPU oracle time); run ad hoc: python tools/soak_gpu.py [ticks] [groups]."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401
import fuzz  # noqa: E402
import oracle_lib as O  # noqa: E402
import raft_rs_amd as rg  # noqa: E402
from raft_rs_amd import engine as E  # noqa: E402

ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 40000


def check(eng, cl, st, gout, what):
    got = eng.read_state()
    cl.store_soa(st)
    d = fuzz.diff_states(st, got, st["n_groups"], st["n_slots"])
    assert not d, (what, d[:5])
    assert (got["out"] == gout).all(), what


t0 = time.time()
for wl, P in ((5, 7), (3, 5), (2, 3)):
    eng = rg.Engine(G, P)
    eng.workload_init(wl)
    st = eng.read_state()
    cl = O.Cluster(G)
    cl.load_soa(st, term=4)
    mb = rg.MsgBuffers(G, P, eng.stride)
    gout = np.zeros(G, dtype=np.uint32)
    for t in range(ticks):
        E.workload_gen_host(st, mb, wl, t)
        eng.tick(mb)
        cl.tick_soa_mt(mb.as_dict(), gout, 32)
        cl.store_soa(st)
        if t % 100 == 99 or t == ticks - 1:
            check(eng, cl, st, gout, f"workload {wl} P={P} tick {t}")
    print(f"workload {wl} P={P}: {ticks} ticks x {G} groups OK ({time.time()-t0:.0f} s)", flush=True)
    eng.close()

# random streams with term tables, log terms, heartbeats, malformed acks
rng = np.random.default_rng(2026)
for P in (3, 5, 8):
    TERM = 9
    st = O.add_term_table(O.alloc_state(G, P))
    st["cfg"][:] = fuzz.random_cfg(rng, G, P, missing_progress_frac=0.05)
    fuzz.random_state(rng, st, small_values=True)
    fuzz.random_term_table(rng, st, TERM)
    eng = rg.Engine(G, P)
    eng.load_state(st)
    cl = O.Cluster(G)
    cl.load_soa(st, term=TERM)
    msgs = O.alloc_msgs(G, P)
    mb = rg.MsgBuffers(G, P, eng.stride)
    gout = np.zeros(G, dtype=np.uint32)
    n = max(20, ticks // 10)
    for t in range(n):
        cl.store_soa(st)
        if t % 7 == 6:
            fuzz.garbage_msgs(rng, st, msgs)
        else:
            fuzz.random_msgs(rng, st, msgs, malformed_p=0.01, logterm_max=TERM)
        for k in ("m_index", "m_commit", "m_hint", "m_rs", "m_logterm", "m_flags"):
            getattr(mb, k)[...] = msgs[k]
        eng.tick(mb)
        cl.tick_soa_mt(msgs, gout, 32)
        check(eng, cl, st, gout, f"random P={P} tick {t}")
    print(f"random streams P={P}: {n} ticks x {G} groups OK ({time.time()-t0:.0f} s)", flush=True)
    eng.close()
print("SOAK_OK")
