#!/usr/bin/env python3
"""Where does the host time of a tick + publication go? Per-call host microseconds, null stream vs an explicit stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import raft_rs_amd as rg
from raft_rs_amd import engine as E

def run(use_null, G=1_000_000, P=5, T=40, publish=True):
    stream = torch.cuda.current_stream() if use_null else torch.cuda.Stream()
    eng = rg.Engine(G, P)
    eng.set_stream(stream.cuda_stream)
    eng.workload_init(2)
    cols = [torch.zeros((P, eng.stride), dtype=torch.int64, device="cuda") for _ in range(4)]
    flags = torch.zeros((G, 8), dtype=torch.uint8, device="cuda")
    eng.workload_gen(2, 0, *[c.data_ptr() for c in cols], flags.data_ptr())
    if publish:
        eng.comm_init(0, 1, unique_id=E.comm_unique_id())
    torch.cuda.synchronize()
    ptrs = [c.data_ptr() for c in cols] + [flags.data_ptr()]
    tt, tp = [], []
    t_all = time.perf_counter()
    for t in range(T):
        a = time.perf_counter(); eng.tick_device(*ptrs); b = time.perf_counter()
        if publish: eng.publish_commit()
        c = time.perf_counter()
        tt.append(b - a); tp.append(c - b)
    issue = time.perf_counter() - t_all
    torch.cuda.synchronize()
    total = time.perf_counter() - t_all
    st = eng.publish_stats() if publish else {}
    print(f"null={use_null} publish={publish}: tick call {np.median(tt)*1e6:.1f} us (max {max(tt)*1e6:.0f}), publish call {np.median(tp)*1e6:.1f} us, "
          f"issue {issue/T*1e6:.1f} us/step, total {total/T*1e6:.1f} us/step", {k: round(v / max(1, st.get('publications', 1)), 2) for k, v in st.items() if k.startswith('host_us')})
    eng.close()

for use_null in (True, False):
    for publish in (False, True):
        run(use_null, publish=publish)
