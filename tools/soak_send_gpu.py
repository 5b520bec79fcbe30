#!/usr/bin/env python3
"""Long differential soak of the send stage on a GPU box: engine (tick + rg_send_appends through the C ABI) vs the
oracle (message-at-a-time ticks + maybe_send_append) on the synthetic stream with the host's SENT events removed; every
other tick runs as ONE launch (rg_tick_send).
Ad hoc: python tools/soak_send_gpu.py [ticks] [groups]."""
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
import sendstage  # noqa: E402
import raft_rs_amd as rg  # noqa: E402
from raft_rs_amd import engine as E  # noqa: E402

ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 300
G = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
rng = np.random.default_rng(77)
t0 = time.time()
for wl, P, cap, max_entries in ((2, 5, 4, 3), (5, 7, 256, 0), (3, 5, 2, 1)):
    eng = rg.Engine(G, P, max_inflight=cap)
    eng.workload_init(wl)
    st = eng.read_state()
    O.add_term_table(st)  # zero table: nothing is compacted (dummy index 0), as in the engine's cold columns
    st["cur_term"][:] = 4
    cl = O.Cluster(G)
    cl.load_soa(st, term=4, max_inflight=cap)
    cl.set_own_inflights(True)
    mb = rg.MsgBuffers(G, P, eng.stride)
    gout = np.zeros(G, dtype=np.uint32)
    n_items = n_full = deepest = 0
    for t in range(ticks):
        E.workload_gen_host(st, mb, wl, t)
        mb.m_flags &= np.uint8(0xE7)  # no SENT, no host INS_FULL: the device owns the send path
        skip = t % 3 == 2
        if t % 2:  # every other tick as ONE launch (rg_tick_send: k_tick_send)
            eng.tick_send(mb, max_entries, skip_bcast_commit=skip)
        else:
            eng.tick(mb)
            eng.send_appends(max_entries, skip_bcast_commit=skip)
        cl.tick_soa_mt(mb.as_dict(), gout, 32)
        items = eng.send_items()
        omsgs = cl.send_stage_soa(gout, max_entries, capacity=G * P * max(4, min(cap, 64)), skip_bcast_commit=skip)
        cl.store_soa(st)
        n_items += len(items)
        if t % 25 == 24 or t == ticks - 1:
            _, out = eng.results()
            assert (out == gout).all(), (wl, t)
            sendstage.compare_items(items, omsgs)
            got = eng.read_state()
            d = fuzz.diff_states(st, got, G, P)
            assert not d, (wl, t, d[:5])
            meta, ring = eng.read_inflights()
            cnt = (meta[:, :G] >> 16)
            n_full += int((cnt == cap).sum())
            deepest = max(deepest, int(cnt.max()))
            sample = rng.choice(G, size=min(G, 3000), replace=False) if t != ticks - 1 else np.arange(G)
            present = (st["cfg"] >> 24) & 0xff
            for g in sample:
                for p in range(P):
                    if (int(present[g]) >> p) & 1 and (int(st["pflags"][g, p]) & 3) == O.REPLICATE:
                        want = cl.ins_contents(int(g), p + 1)
                        assert sendstage.ring_contents(meta, ring, int(g), p, cap) == want, (wl, t, g, p)
        else:
            assert len(items) == len(sendstage.coalesce_oracle(omsgs)), (wl, t)
    print(f"workload {wl} P={P} cap={cap} max_entries={max_entries}: {ticks} ticks x {G} groups OK, {n_items} work "
          f"items, deepest window {deepest}, {n_full} full windows at the checks ({time.time()-t0:.0f} s)", flush=True)
    eng.close()
print("SOAK_SEND_OK")
