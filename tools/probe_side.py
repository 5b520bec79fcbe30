import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import raft_rs_amd as rg
def run(G, explicit, T=15, W=3):
    if explicit: torch.cuda.set_stream(torch.cuda.Stream())
    stream = torch.cuda.current_stream()
    eng = rg.Engine(G, 5); eng.set_stream(stream.cuda_stream); eng.workload_init(2)
    cols = [torch.empty((T, 5, eng.stride), dtype=torch.int64, device="cuda") for _ in range(4)]
    flags = torch.empty((T, G, 8), dtype=torch.uint8, device="cuda")
    eng.checkpoint()
    faults = []
    for t in range(T):
        ptrs = [c[t].data_ptr() for c in cols] + [flags[t].data_ptr()]
        eng.workload_gen(2, t, *ptrs); eng.msg_stats(flags[t].data_ptr()); eng.tick_device(*ptrs)
        faults.append(eng.result_counts()[1])
    ref, _ = eng.results()
    eng.restore()
    f2 = []
    for t in range(T):
        eng.tick_device(*([c[t].data_ptr() for c in cols] + [flags[t].data_ptr()]))
        f2.append(eng.result_counts()[1])
    got, _ = eng.results()
    print("G", G, "explicit", explicit, "record faults", sum(faults), "replay faults", f2, "same", np.array_equal(ref, got))
    eng.close()
for G in (1_000_000, 8_000_000):
    for explicit in (False, True):
        run(G, explicit)
