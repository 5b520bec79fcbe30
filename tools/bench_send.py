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
W = int(os.environ.get("WARMUP", 20))  # untimed ticks (clock ramp, first-touch of the ring arena)
torch.cuda.set_stream(torch.cuda.Stream())
print(f"send stage after every tick, {G} groups x {P} peers, {K} timed ticks")
# cap:max_entries[:max_bytes] -- a third field switches to RG_SEND_BYTES (byte-accurate max_size_per_msg over synthetic
# entry sizes of 40..540 bytes kept in a 64-entry window per group)
CONFIGS = [tuple(int(x) for x in c.split(":")) for c in os.environ.get("CONFIGS", "8:0,256:0,256:4,256:0:700").split(",")]
for cfg_ in CONFIGS:
    cap, max_entries, max_bytes = cfg_[0], cfg_[1], (cfg_[2] if len(cfg_) > 2 else None)
    eng = rg.Engine(G, P, max_inflight=cap)
    if max_bytes is not None:
        eng.log_sizes_enable(64)
    eng.workload_init(rg.WL_MAJORITY)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)  # (an explicit stream: set below, before the loop)
    cols = [torch.zeros((P, eng.stride), dtype=torch.int64, device="cuda") for _ in range(4)]
    flags = torch.zeros((G, 8), dtype=torch.uint8, device="cuda")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4 * (K + W))]
    t_tick = t_send = 0.0
    items = 0
    for t in range(K + W):
        eng.workload_gen(rg.WL_MAJORITY, t, *[c.data_ptr() for c in cols], flags.data_ptr())
        flags &= 0xEF  # no RG_MF_SENT: the device sends
        a, b, b2, c = ev[4 * t:4 * t + 4]
        a.record()
        eng.tick_device(*[c_.data_ptr() for c_ in cols], flags.data_ptr())
        b.record()
        if max_bytes is not None:
            eng.workload_sizes(0x5eed, 40, 500)  # the sizes of the entries this tick appended (not part of the stage)
        b2.record()
        eng.send_appends(max_entries, max_bytes=max_bytes)
        c.record()
        if t == K + W - 1:
            items = len(eng.send_items())
    torch.cuda.synchronize()
    ticks, sends = [], []
    for t in range(W, K + W):
        a, b, b2, c = ev[4 * t:4 * t + 4]
        ticks.append(a.elapsed_time(b))
        sends.append(b2.elapsed_time(c))
    # medians: single ticks are disturbed by the generator's launches in between and by clock changes
    t_tick, t_send = sorted(ticks)[K // 2] * 1e3, sorted(sends)[K // 2] * 1e3
    full = int((torch.from_numpy(eng.read_column(rg.COL.PFLAGS)) & 0x10).ne(0).sum())
    # The stage's own algorithmic bytes: per group out 4 + cfg 4 + last_index 8 + first_index 8 +
    # flag row 8 r + 8 w = 40; per peer in the work set (a send request, an Inflights effect, or a broadcast): window
    # meta 4 r + 4 w, oldest / newest inflight 16 r + 16 w, next 8 r + 8 w, pending snapshot request 8 r, matched 8 r
    # = 72; work items as peer-major columns: the n | kind cell of every peer (4 B) + prev / last index of every item
    # (16 B). Ring words are not counted (windows of <= 2 never touch the ring).
    import numpy as np
    _, out = eng.results()
    cfg = eng.read_column(rg.COL.CFG)
    present, self_slot = (cfg >> 24) & 0xff, (cfg >> 16) & 7
    bcast = (out & 0x9) != 0  # CHANGED (should_bcast_commit: skip_bcast_commit is off here) or APPENDED
    work = ((out >> 8) | (out >> 16) | (out >> 24)) & 0xff
    work = np.where(bcast, work | present, work) & present & ~(1 << self_slot)
    n_work = int(sum(((work >> p) & 1).sum() for p in range(8)))
    nbytes = 40 * G + 72 * n_work + 4 * P * G + 16 * items  # work items as columns: a 4-B cell per peer + 16 B per item
    gbs = nbytes / (t_send * 1e-6) / 1e9
    lim = f"max_entries {max_entries}" if max_bytes is None else f"max_bytes {max_bytes}"
    print(f"  cap {cap:3d} {lim}: tick {t_tick:7.1f} us  send stage {t_send:7.1f} us  "
          f"-> {G / (t_tick + t_send):7.1f} M group-evals/s incl. sends; {items} work items in the last stage, "
          f"{full} full windows; stage byte model {nbytes / 1e6:.0f} MB ({nbytes / G:.0f} B/group) -> {gbs:.0f} GB/s = "
          f"{gbs / 8000:.3f} of 8 TB/s")
    eng.close()
