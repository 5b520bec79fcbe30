#!/usr/bin/env python3
"""Can a stream of rg_tick_device launches be captured into a hipGraph and replayed? (torch.cuda.graph drives
hipStreamBeginCapture / hipGraphLaunch; the engine runs on the capture stream.) Small shards, where a tick is shorter than a
launch: K recorded ticks eagerly vs as one graph replay. Usage: python tools/probe_graph.py [groups] [ticks] [classes]
(`classes`: BASELINE config 5 placed by replica-set size class in a 7-slot engine -- the captured launches are k_tick_classes)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raft_rs_amd as rg  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 32
PLACED = len(sys.argv) > 3 and sys.argv[3] == "classes"
FUSED_LT = len(sys.argv) > 3 and sys.argv[3] == "fused-logterm"  # rg_tick_device_fused, four ticks per call, every other tick with a log-term column
P, WL = (7, 5) if PLACED else (5, 2)
side = torch.cuda.Stream()
eng = rg.Engine(G, P)
eng.set_stream(side.cuda_stream)
eng.workload_init(WL, sorted_classes=PLACED)
cols = [torch.empty((K, P, eng.stride), dtype=torch.int64, device="cuda") for _ in range(4)]
flags = torch.empty((K, G, 8), dtype=torch.uint8, device="cuda")


def ptrs(t):
    return [c[t].data_ptr() for c in cols] + [flags[t].data_ptr()]


eng.checkpoint()
for t in range(K):  # record the stream from the evolving state
    eng.workload_gen(WL, t, *ptrs(t), sorted_classes=PLACED)
    eng.tick_device(*ptrs(t))
eng.sync()
ref = eng.results()
ref_state = eng.read_state()


logterm = torch.zeros((P, eng.stride), dtype=torch.int64, device="cuda")  # (no reject of this stream carries a log term: the column only has to be there)
out_t = torch.zeros((4, G), dtype=torch.int32, device="cuda")


def eager():
    if FUSED_LT:
        # a log-term tick inside a fused call runs as a single launch behind its pre-pass, and since round 6 the call asks that
        # pre-pass whether it left a reject to the host -- a host wait that must NOT happen while the stream is being captured
        for t0 in range(0, K, 4):
            ticks = [ptrs(t) + ([logterm.data_ptr()] if t % 2 else []) for t in range(t0, min(t0 + 4, K))]
            assert eng.tick_device_fused(ticks, out_t.data_ptr()) == len(ticks)
        return
    for t in range(K):
        eng.tick_device(*ptrs(t))


def timed(fn, reps=20):
    best = []
    for _ in range(reps):
        eng.restore()
        eng.sync()
        t0 = time.perf_counter()
        fn()
        eng.sync()
        best.append(time.perf_counter() - t0)
    return float(np.median(best)) * 1e6


us_eager = timed(eager)
c, o = eng.results()
assert np.array_equal(c, ref[0]) and np.array_equal(o, ref[1])

eng.restore()
eng.sync()
if PLACED:
    # rg_restore brought RG_COL_CFG back, so the size classes are re-derived by the next dense tick -- a synchronising step that
    # cannot run inside a capture (a captured tick of a stale engine takes the plain kernel): ask for them first
    assert [q for _, _, q in eng.size_classes()] == [3, 5, 7]
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph, stream=side):
    eager()
graph.replay()
torch.cuda.synchronize()
c, o = eng.results()
assert np.array_equal(c, ref[0]) and np.array_equal(o, ref[1]), "graph replay differs from the eager run"


def replay():
    with torch.cuda.stream(side):
        graph.replay()


us_graph = timed(replay)
c, o = eng.results()
assert np.array_equal(c, ref[0]) and np.array_equal(o, ref[1])
st = eng.read_state()
for k in ("match", "next", "pr_commit", "commit", "pflags"):
    assert np.array_equal(st[k], ref_state[k]), k
print(f"{G} groups x {P} {'slots, placed by size class' if PLACED else 'peers'}, {K} ticks: eager {us_eager:.1f} us ({us_eager / K:.2f} us per tick), "
      f"one hipGraph replay {us_graph:.1f} us ({us_graph / K:.2f} us per tick); results identical")
print("GRAPH_OK")
