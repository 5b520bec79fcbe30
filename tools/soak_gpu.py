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
import hosthints  # noqa: E402
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
# (5, 7, True) / (5, 8, True): BASELINE config 5 with its groups placed by replica-set size class -- ONE launch per tick, k_tick_classes
for wl, P, placed in ((5, 7, True), (5, 8, True), (5, 7, False), (3, 5, False), (2, 3, False)):
    eng = rg.Engine(G, P)
    eng.workload_init(wl, sorted_classes=placed)
    assert bool(eng.size_classes()) == placed
    st = eng.read_state()
    cl = O.Cluster(G)
    cl.load_soa(st, term=4)
    mb = rg.MsgBuffers(G, P, eng.stride)
    gout = np.zeros(G, dtype=np.uint32)
    for t in range(ticks):
        E.workload_gen_host(st, mb, wl, t, sorted_classes=placed)
        eng.tick(mb)
        cl.tick_soa_mt(mb.as_dict(), gout, 32)
        cl.store_soa(st)
        if t % 100 == 99 or t == ticks - 1:
            check(eng, cl, st, gout, f"workload {wl} P={P} tick {t}")
    print(f"workload {wl} P={P}{' placed by size class (k_tick_classes)' if placed else ''}: {ticks} ticks x {G} groups OK ({time.time()-t0:.0f} s)", flush=True)
    eng.close()

# random streams with term tables, log terms, heartbeats, elections, malformed acks and garbage ticks; histories outgrow the
# engine's term-run table, so rejects come back as RG_OUT_HOST_HINT and are settled against the oracle's COMPLETE log
# (tests/hosthints.py); (8, True): replica sets of 3 / 5 / 8 in contiguous ranges of one engine
rng = np.random.default_rng(2026)
hinted = 0
for P, placed in ((3, False), (5, False), (8, False), (8, True)):
    TERM = 9
    st = O.add_term_table(O.alloc_state(G, P))
    if placed:
        st["cfg"][:] = fuzz.class_placed_cfg(rng, [(G // 3 + 7, 3), (G // 3 - 30, 5), (G - 2 * (G // 3) + 23, 8)], P, missing_progress_frac=0.05)
    else:
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
            fuzz.random_msgs(rng, st, msgs, malformed_p=0.01, logterm_max=TERM + t, elect_p=0.1, elect_term=TERM + 1 + t)
        for k in ("m_index", "m_commit", "m_hint", "m_rs", "m_logterm", "m_flags"):
            getattr(mb, k)[...] = msgs[k]
        eng.tick(mb)
        cl.tick_soa_mt(msgs, gout, 32)
        hinted += hosthints.settle_engine(eng, cl, msgs)  # (nothing to do where no result word carries the bit)
        check(eng, cl, st, gout, f"random P={P} tick {t}")
    print(f"random streams P={P}{' placed by size class' if placed else ''}: {n} ticks x {G} groups OK, {hinted} rejects handed back and settled so far ({time.time()-t0:.0f} s)", flush=True)
    eng.close()
print("SOAK_OK")
