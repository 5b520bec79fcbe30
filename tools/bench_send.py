#!/usr/bin/env python3
"""Measure the send stage (rg_send_appends: device Inflights + maybe_send_append decisions) next to the tick it
follows, 1 M groups x 5 peers on one MI355X, on the synthetic stream with the host's SENT events removed (the
stage applies Progress::update_state itself). Numbers go into DESIGN.md / profiles/."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import raft_rs_amd as rg  # noqa: E402

G = int(os.environ.get("G", 1_000_000))
P = int(os.environ.get("P", 5))
K = 40
print(f"send stage after every tick, {G} groups x {P} peers, {K} timed ticks")
for cap, max_entries in ((8, 0), (256, 0), (256, 4)):
    eng = rg.Engine(G, P, max_inflight=cap)
    eng.workload_init(rg.WL_MAJORITY)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    cols = [torch.zeros((P, eng.stride), dtype=torch.int64, device="cuda") for _ in range(4)]
    flags = torch.zeros((G, 8), dtype=torch.uint8, device="cuda")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3 * (K + 5))]
    t_tick = t_send = 0.0
    items = 0
    for t in range(K + 5):
        eng.workload_gen(rg.WL_MAJORITY, t, *[c.data_ptr() for c in cols], flags.data_ptr())
        flags &= 0xEF  # no RG_MF_SENT: the device sends
        a, b, c = ev[3 * t:3 * t + 3]
        a.record()
        eng.tick_device(*[c_.data_ptr() for c_ in cols], flags.data_ptr())
        b.record()
        eng.send_appends(max_entries)
        c.record()
        if t == K + 4:
            items = len(eng.send_items())
    torch.cuda.synchronize()
    for t in range(5, K + 5):
        a, b, c = ev[3 * t:3 * t + 3]
        t_tick += a.elapsed_time(b)
        t_send += b.elapsed_time(c)
    t_tick, t_send = t_tick / K * 1e3, t_send / K * 1e3
    full = int((torch.from_numpy(eng.read_column(rg.COL.PFLAGS)) & 0x10).ne(0).sum())
    print(f"  cap {cap:3d} max_entries {max_entries}: tick {t_tick:7.1f} us  send stage {t_send:7.1f} us  "
          f"-> {G / (t_tick + t_send):7.1f} M group-evals/s incl. sends; {items} work items in the last stage, "
          f"{full} full windows")
    eng.close()
