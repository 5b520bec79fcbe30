"""GPU: publication of commit indices behind the C ABI (rg_comm_init / rg_publish_commit / rg_published_commit).

The exchange itself is ncclAllGather (RCCL, world size 1 on this single-GPU box) or a host-provided all-gather
(two / three ranks sharing the GPU, gloo moving the slices): what is under test is everything around it -- the byte
the tick kernels fuse into their store path, accumulation over several ticks, the exact-value list, the loss ->
full-snapshot protocol, the lazily updated replica."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _device_msgs(torch, eng, n_slots):
    cols = [torch.zeros((n_slots, eng.stride), dtype=torch.int64, device="cuda") for _ in range(4)]
    flags = torch.zeros((eng.n_groups, 8), dtype=torch.uint8, device="cuda")
    return cols, flags


def test_comm_warmup_and_the_census_of_an_engine_without_a_communicator(rg):
    """rg_comm_warmup loads RCCL and runs a one-rank communicator through its life (a host calls it at start-up, before any step
    a timeout bounds); an engine without a communicator reports transport "none" and no RCCL ranks."""
    from raft_rs_amd import engine as E
    E.comm_warmup()
    E.comm_warmup()  # (idempotent: the second call finds the library mapped)
    eng = rg.Engine(1000, 3)
    assert eng.comm_info() == {"rank": 0, "world": 0, "transport": "none", "in_process": False, "rccl_ranks": 0, "rccl_rank": 0}
    eng.close()


@pytest.mark.parametrize("workload,n_slots", [(2, 5), (5, 7)])
def test_rccl_world1_replica_follows_every_tick(rg, workload, n_slots):
    """ncclAllGather at world size 1: after every tick the replica equals the commit column; the slice is ~1 B/group."""
    import torch
    from raft_rs_amd import engine as E
    G = 50_000 + 3
    eng = rg.Engine(G, n_slots)
    eng.workload_init(workload)
    eng.comm_init(0, 1, unique_id=E.comm_unique_id(), ring_ticks=4)
    assert np.array_equal(eng.published_commit(0), eng.read_column(rg.COL.COMMIT)), "rg_comm_init publishes a full snapshot"
    cols, flags = _device_msgs(torch, eng, n_slots)
    for t in range(11):  # more than two rings' worth
        eng.workload_gen(workload, t, *[c.data_ptr() for c in cols], flags.data_ptr())
        eng.tick_device(*[c.data_ptr() for c in cols], flags.data_ptr())
        eng.publish_commit()
        if t % 3 == 2 or t == 10:  # reading the replica is allowed at any time, not only when the ring is full
            assert np.array_equal(eng.published_commit(0), eng.read_column(rg.COL.COMMIT)), t
    st = eng.publish_stats()
    assert st["publications"] == 12 and st["full_publications"] == 1
    # every one of the 11 publications came right behind its dense tick: the "slice complete" event rode on the tick's own
    # dispatch packet (no event packet in the engine's queue: tools/microbench/pub_signal.hip) -- and the replica was exact
    assert st["events_on_tick_packets"] == 11, st
    assert st["bytes_per_rank_delta"] < 1.1 * G + 4096 and st["bytes_per_rank_full"] >= 8 * G
    # what the COMMUNICATOR says about itself (ncclCommCount / ncclCommUserRank), not what the engine was told
    ci = eng.comm_info()
    assert ci == {"rank": 0, "world": 1, "transport": "rccl", "in_process": False, "rccl_ranks": 1, "rccl_rank": 0}
    ptr, stride = eng.published_commit_ptr()
    assert ptr and stride >= G
    eng.comm_destroy()
    eng.close()


def test_the_one_launch_tick_and_send_stage_publishes_like_the_tick(rg):
    """rg_tick_device_send (k_tick_send) carries the tick's publication byte in its store path as well: with the Inflights on
    the device the replica follows every tick, whichever form the tick takes."""
    import torch
    from raft_rs_amd import engine as E
    G, P = 30_000 + 11, 5
    eng = rg.Engine(G, P, max_inflight=8)
    eng.workload_init(2)
    eng.comm_init(0, 1, unique_id=E.comm_unique_id(), ring_ticks=4)
    cols, flags = _device_msgs(torch, eng, P)
    moved = 0
    for t in range(9):
        eng.workload_gen(2, t, *[c.data_ptr() for c in cols], flags.data_ptr())
        flags &= 0xE7  # the device owns the send path: no SENT / INS_FULL events from the host
        before = eng.read_column(rg.COL.COMMIT).copy()
        if t % 2:
            eng.tick_device_send(*[c.data_ptr() for c in cols], flags.data_ptr())
        else:
            eng.tick_device(*[c.data_ptr() for c in cols], flags.data_ptr())
            eng.send_appends()
        eng.publish_commit()
        now = eng.read_column(rg.COL.COMMIT)
        assert np.array_equal(eng.published_commit(0), now), t
        moved += int((now != before).sum())
    assert moved > G
    assert eng.publish_stats()["full_publications"] == 1
    eng.comm_destroy()
    eng.close()


def test_ticks_accumulate_between_publications_and_other_commit_paths(rg):
    """Any cadence: several dense ticks, a sparse tick (rg_ingest_tick) and rg_recompute between two publications."""
    import torch
    from raft_rs_amd import engine as E
    from raft_rs_amd.engine import WIRE_DTYPE
    G, P = 20_000, 5
    eng = rg.Engine(G, P)
    eng.workload_init(2)
    eng.comm_init(0, 1, unique_id=E.comm_unique_id())
    cols, flags = _device_msgs(torch, eng, P)
    for t in range(5):
        eng.workload_gen(2, t, *[c.data_ptr() for c in cols], flags.data_ptr())
        eng.tick_device(*[c.data_ptr() for c in cols], flags.data_ptr())
        if t == 2:
            eng.publish_commit()
    # a sparse tick: the leaders of 100 groups append 40 entries and every follower acks them
    st = eng.read_state()
    groups = np.arange(0, G, G // 100)[:100]
    rec = np.zeros(len(groups) * P, dtype=WIRE_DTYPE)
    k = 0
    for g in groups:
        new_last = int(st["term_hi"][g]) + 40
        for p in range(P):
            rec[k] = (g, new_last, new_last if p == 0 else int(st["commit"][g]), 0, 0, 0, p,
                      (rg.MF.APPEND | rg.MF.VALID) if p == 0 else rg.MF.VALID, 0)
            k += 1
    # (the followers' acks of index new_last need the append first: slot 0 is processed before them)
    n, dup = eng.ingest_tick(rec)
    assert n == len(groups) and dup == 0
    # a recompute after a membership change that lowers the quorum: group 1 drops to a single voter
    eng.set_config(1, rg.cfg_make(0b00001, present=0b11111))
    eng.recompute()
    eng.publish_commit()
    commit = eng.read_column(rg.COL.COMMIT)
    assert (commit[groups] == st["term_hi"][groups] + 40).all()
    assert np.array_equal(eng.published_commit(0), commit)
    # a fused launch (4 ticks of the stream in one launch) between two publications: its total advance is published
    T = 4
    tcols = [[torch.zeros((P, eng.stride), dtype=torch.int64, device="cuda") for _ in range(4)] for _ in range(T)]
    tflags = [torch.zeros((G, 8), dtype=torch.uint8, device="cuda") for _ in range(T)]
    ref = rg.Engine(G, P)  # the same four ticks one at a time: what the fused launch must equal
    ref.load_state(eng.read_state())
    for t in range(T):
        ref.workload_gen(2, 20 + t, *[c.data_ptr() for c in tcols[t]], tflags[t].data_ptr())
        ref.tick_device(*[c.data_ptr() for c in tcols[t]], tflags[t].data_ptr())
    out_t = torch.zeros((T, G), dtype=torch.int32, device="cuda")
    eng.tick_device_fused([[c.data_ptr() for c in tcols[t]] + [tflags[t].data_ptr()] for t in range(T)], out_t.data_ptr())
    eng.publish_commit()
    commit2 = eng.read_column(rg.COL.COMMIT)
    assert np.array_equal(commit2, ref.read_column(rg.COL.COMMIT)) and (commit2 > commit).sum() > G // 2
    assert np.array_equal(eng.published_commit(0), commit2)
    # only the first publication followed a dense tick directly; behind a recompute / a fused launch the event is recorded
    assert eng.publish_stats()["events_on_tick_packets"] == 1
    ref.close()
    eng.close()  # (rg_destroy tears the communicator down)


def test_saturated_bytes_list_overflow_and_rollback_resynchronise(rg):
    """Advances >= 255 go through the exact-value list; a list too short for them marks the slice lost: the ranks
    notice when they fold the slices into their replicas (check points: every ring_ticks-th publication, the same
    numbers everywhere) and the check point after that publishes a full snapshot; rg_restore takes the same road."""
    from raft_rs_amd import engine as E
    G, P, cap, ring = 3000, 3, 8, 2
    st = O.alloc_state(G, P)
    st["cfg"][:] = rg.cfg_make(0b111, self_slot=0)
    st["term_lo"][:] = 1
    st["term_hi"][:] = 1 << 40
    st["match"][0, :G] = 1 << 40
    st["next"][:, :G] = 1
    st["pflags"][:, :P] = 1
    eng = rg.Engine(G, P)
    eng.load_state(st)
    eng.comm_init(0, 1, unique_id=E.comm_unique_id(), overflow_slots=cap, ring_ticks=ring)   # publication 0 (full)
    mb = rg.MsgBuffers(G, P, eng.stride)

    def ack_all(index):  # both followers acknowledge index[g]: the commit index of every group becomes index[g]
        mb.clear()
        for p in (1, 2):
            mb.m_flags[:, p] = rg.MF.VALID
            mb.m_index[p, :G] = index
        eng.tick(mb)

    target = np.full(G, 100, dtype=np.uint64)
    target[:5] = [255, 256, 1000, 1 << 33, 254]  # 255 still fits the byte; 256, 1000 and 2^33 spill into the list
    ack_all(target)
    eng.publish_commit()                                                                      # publication 1
    assert np.array_equal(eng.published_commit(0), target)
    assert eng.publish_stats()["full_publications"] == 1
    # more saturated groups than the list holds
    target2 = target + np.uint64(7)
    target2[100:100 + 3 * cap] += np.uint64(5000)
    ack_all(target2)
    eng.publish_commit()                                                                      # 2: lost
    assert not np.array_equal(eng.published_commit(0), target2), "the replica is knowingly inexact now"
    for k in range(1, 5):                                                                     # 3, 4 (seen), 5, 6 (full)
        assert eng.publish_stats()["full_publications"] == 1
        ack_all(target2 + np.uint64(k))
        eng.publish_commit()
    assert eng.publish_stats()["full_publications"] == 2
    assert np.array_equal(eng.published_commit(0), target2 + np.uint64(4))
    # rollback: the published advances no longer describe the column
    eng.checkpoint()
    ack_all(target2 + np.uint64(50))
    eng.publish_commit()                                                                      # 7
    eng.restore()
    for k in range(6):                                                                        # 8 (announces) .. 13
        ack_all(target2 + np.uint64(5 + k))
        eng.publish_commit()
    assert eng.publish_stats()["full_publications"] == 3
    assert np.array_equal(eng.published_commit(0), target2 + np.uint64(10))
    # an explicit full publication (what a host does right after it reloads state)
    eng.publish_commit(full=True)
    assert eng.publish_stats()["full_publications"] == 4
    assert np.array_equal(eng.published_commit(0), target2 + np.uint64(10))
    eng.close()


WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["RG_ROOT"]); sys.path.insert(0, os.path.join(os.environ["RG_ROOT"], "tests"))
import raft_rs_amd as rg

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)  # every rank shares the box's one GPU
G, P, WL, TICKS = 40_000, 5, rg.WL_MIXED if os.environ.get("RG_WL") == "5" else rg.WL_MAJORITY, 9

class Dev:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}

calls = []
def allgather(dev_send, dev_recv, nbytes, stream):
    """The host-provided transport of rg_comm_config: gloo moves the slices (RCCL refuses two ranks on one GPU)."""
    torch.cuda.synchronize()
    send = torch.as_tensor(Dev(dev_send, nbytes), device="cuda").cpu()
    out = torch.empty(world * nbytes, dtype=torch.uint8)
    dist.all_gather_into_tensor(out, send)
    torch.as_tensor(Dev(dev_recv, world * nbytes), device="cuda").copy_(out)
    torch.cuda.synchronize()
    calls.append(nbytes)
    return 0

eng = rg.Engine(G, P)
eng.workload_init(WL, first_group=rank * G)
eng.comm_init(rank, world, transport=allgather, ring_ticks=4)
cols = [torch.zeros((P, eng.stride), dtype=torch.int64, device="cuda") for _ in range(4)]
flags = torch.zeros((G, 8), dtype=torch.uint8, device="cuda")
for t in range(TICKS):
    eng.workload_gen(WL, t, *[c.data_ptr() for c in cols], flags.data_ptr(), first_group=rank * G)
    eng.tick_device(*[c.data_ptr() for c in cols], flags.data_ptr())
    eng.publish_commit()
mine = eng.read_column(rg.COL.COMMIT)
rep = eng.published_commit()              # [world][G], from this rank's replica
assert np.array_equal(rep[rank], mine)
# every rank's replica holds every rank's column
allc = [torch.empty(G, dtype=torch.int64) for _ in range(world)]
dist.all_gather(allc, torch.from_numpy(mine.view(np.int64)))
for r in range(world):
    assert np.array_equal(rep[r], allc[r].numpy().view(np.uint64)), (rank, r)
st = eng.publish_stats()
assert calls[0] >= 8 * G and all(c == st["bytes_per_rank_delta"] for c in calls[1:]), calls
assert st["publications"] == TICKS + 1 and st["full_publications"] == 1
if rank == 0:
    # ... and equals one unsharded engine over all groups
    whole = rg.Engine(world * G, P)
    whole.workload_init(WL)
    wc = [torch.zeros((P, whole.stride), dtype=torch.int64, device="cuda") for _ in range(4)]
    wf = torch.zeros((world * G, 8), dtype=torch.uint8, device="cuda")
    for t in range(TICKS):
        whole.workload_gen(WL, t, *[c.data_ptr() for c in wc], wf.data_ptr())
        whole.tick_device(*[c.data_ptr() for c in wc], wf.data_ptr())
    assert np.array_equal(rep.reshape(-1), whole.read_column(rg.COL.COMMIT))
    print("PUBLISH_OK", world, st["bytes_per_rank_delta"])
dist.barrier()
eng.close()
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world,workload", [(2, "2"), (3, "5")])
def test_ranks_sharing_the_gpu_publish_through_the_c_entry_point(rg, tmp_path, world, workload):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, RG_ROOT=ROOT, RG_WL=workload)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(29560 + world), str(script)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    assert f"PUBLISH_OK {world}" in r.stdout, r.stdout[-3000:]


WORKER8 = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["RG_ROOT"]); sys.path.insert(0, os.path.join(os.environ["RG_ROOT"], "tests"))
import raft_rs_amd as rg

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)  # every rank shares the box's one GPU
# BASELINE configs[3]'s shape: 7-peer groups sharded over 8 ranks, >= 128 k groups per rank, a publication after EVERY tick
G, P, WL, RING, TICKS, LOSER = 131_072 + 5, 7, rg.WL_MAJORITY, 4, 19, 5

class Dev:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}

calls, sends, recvs = [], set(), set()
DELTA, FULL = rg.engine.pub_bytes_per_rank(G), None
def allgather(dev_send, dev_recv, nbytes, stream):
    # What the library hands its all-gather is, argument for argument, what ncclAllGather(sendbuff, recvbuff, sendcount,
    # ncclUint8, comm, stream) takes when this callback is not installed (rg_pub_allgather): the N-rank SHAPE of the exchange is
    # checked here, so that RCCL itself is the only thing this box cannot run at N > 1. One slice of `nbytes` in, `world` slices
    # out, rank r's at recvbuff + r * nbytes (all_gather_into_tensor below has that very contract); the two ranges are
    # disjoint device allocations of the engine, whole cache lines apart; the count is one of the two slice sizes.
    assert nbytes == DELTA or nbytes % 8 == 0 and nbytes >= 8 * G, (nbytes, DELTA)
    assert dev_send % 256 == 0 and dev_recv % 256 == 0, (hex(dev_send), hex(dev_recv))
    assert dev_send + nbytes <= dev_recv or dev_recv + world * nbytes <= dev_send
    sends.add((dev_send, nbytes)); recvs.add((dev_recv, nbytes))
    torch.cuda.synchronize()
    send = torch.as_tensor(Dev(dev_send, nbytes), device="cuda").cpu()
    out = torch.empty(world * nbytes, dtype=torch.uint8)
    dist.all_gather_into_tensor(out, send)
    torch.as_tensor(Dev(dev_recv, world * nbytes), device="cuda").copy_(out)
    torch.cuda.synchronize()
    calls.append(nbytes)
    return 0

eng = rg.Engine(G, P)
eng.workload_init(WL, first_group=rank * G)
eng.comm_init(rank, world, transport=allgather, ring_ticks=RING)
cols = [torch.zeros((P, eng.stride), dtype=torch.int64, device="cuda") for _ in range(4)]
flags = torch.zeros((G, 8), dtype=torch.uint8, device="cuda")
exact_at = []
for t in range(TICKS):
    if rank == LOSER and t == 3:
        eng.checkpoint()
    if rank == LOSER and t == 6:
        eng.restore()          # a rollback on ONE rank: its published advances no longer describe its column
    eng.workload_gen(WL, t, *[c.data_ptr() for c in cols], flags.data_ptr(), first_group=rank * G)
    eng.tick_device(*[c.data_ptr() for c in cols], flags.data_ptr())
    eng.publish_commit()
    # every rank's replica against every rank's column, after every tick (reading folds the ring: any cadence is legal)
    mine = eng.read_column(rg.COL.COMMIT)
    allc = [torch.empty(G, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(allc, torch.from_numpy(mine.view(np.int64)))
    rep = eng.published_commit()
    ok = all(np.array_equal(rep[r], allc[r].numpy().view(np.uint64)) for r in range(world))
    exact_at.append(ok)
    others_ok = all(np.array_equal(rep[r], allc[r].numpy().view(np.uint64)) for r in range(world) if r != LOSER)
    assert others_ok, (rank, t)  # only the rank that rolled back may be inexact, and only for a while
st = eng.publish_stats()
# exact before the rollback, inexact for at most 2 x ring_ticks publications after it, exact again from then on
assert all(exact_at[:6]), exact_at
assert not all(exact_at[6:6 + 2 * RING + 1]) and all(exact_at[6 + 2 * RING + 1:]), exact_at
assert st["full_publications"] == 2, st          # rg_comm_init's and the resynchronisation
assert calls.count(st["bytes_per_rank_full"]) == 2 and calls.count(st["bytes_per_rank_delta"]) == TICKS - 1, (calls, st)
assert st["bytes_per_rank_delta"] < 1.1 * G + 4096 and st["bytes_per_rank_delta"] == DELTA
# a collective: every rank passed the SAME count at the SAME call -- also across the loss / full-snapshot protocol, which each
# rank decides on its own from the gathered headers
every = [None] * world
dist.all_gather_object(every, calls)
assert all(c == every[0] for c in every), every
# the engine rotates 4 send slices and RING gathered slots per slice size (+ one full-size pair)
assert len({p for p, n in sends if n == DELTA}) <= 4 and len({p for p, n in recvs if n == DELTA}) <= RING, (len(sends), len(recvs))
if rank == 0:
    print("PUBLISH8_OK", world, st["bytes_per_rank_delta"], st["bytes_per_rank_full"], exact_at.count(False))
dist.barrier()
eng.close()
dist.destroy_process_group()
'''


def test_eight_ranks_of_the_config4_shape_publish_every_tick_with_one_rollback(rg, tmp_path):
    """BASELINE configs[3]'s exchange at its real rank count, on the one GPU this box has: 8 ranks x 131 k groups x 7
    peers through rg_publish_commit (transport callback: gloo moves the slices), a publication after every tick, the
    ring of 4 wrapping several times, and ONE rank rolling back (rg_restore) in the middle -- its slice is announced lost
    in the gathered headers, every rank decides at the same check point that the next one carries full 8 B/group
    snapshots, and all replicas are exact again within 2 x ring_ticks publications."""
    script = tmp_path / "worker8.py"
    script.write_text(WORKER8)
    env = dict(os.environ, RG_ROOT=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8",
           "--master-addr", "127.0.0.1", "--master-port", "29578", str(script)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:]
    assert "PUBLISH8_OK 8" in r.stdout, r.stdout[-3000:]


def test_bench_line_of_eight_ranks_sharing_the_gpu(rg):
    """`bench.py --gpus 8 --slots 7` as the driver launches it (torch.distributed.run, one rank per GPU) -- here with
    BENCH_SHARE_GPU=1 (all ranks on GPU 0, gloo moving the slices): the N = 8 line's JSON end to end."""
    env = dict(os.environ, BENCH_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
           "--master-port", "29579", os.path.join(ROOT, "bench.py"), "--gpus", "8", "--slots", "7", "--groups", "131072",
           "--steps", "6", "--warmup", "2", "--publish-every", "1"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    # ONE line on stdout: what gloo and RCCL print while eight ranks connect goes to stderr (bench.py keeps fd 1 for the JSON)
    assert len([ln for ln in r.stdout.splitlines() if ln.strip()]) == 1, r.stdout[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 8 and line["steps"] == 6 and line["scaling"] == "weak" and line["unit"] == "group-evals/s"
    assert line["value"] == pytest.approx(8 * 131072 * 6 / (line["ms_per_step"] * 6 / 1e3), rel=1e-6)
    assert len(r.stdout.encode()) <= 6000  # (what the driver's record keeps of a run's stdout holds the whole line)
    cfg = line["config"]
    assert "configs[3]" in cfg["workload"] and "sharded over 8 GPUs" in cfg["workload"] and cfg["peer_slots"] == 7
    assert not any(isinstance(v, (dict, list)) for v in cfg.values())  # scalars only; the nested statistics are in `full`
    assert cfg["pub_publications"] >= 6 + 2 and cfg["pub_bytes_per_rank_delta"] < 1.1 * 131072 + 4096 <= cfg["pub_bytes_per_rank_full"]
    assert cfg["publication_mode"].startswith("delta") and cfg["pub_compare_mode"].startswith("raw") and cfg["pub_compare_value"] > 0
    # the census of the exchange: eight ranks sharing ONE GPU cannot be RCCL ranks (RCCL refuses two ranks on a device) and the
    # line says so -- on an 8-GPU node the same keys read transport "rccl", rccl_ranks 8, rccl_distinct_ranks 8
    assert cfg["transport"] == "callback" and cfg["rccl_ranks"] == 0 and cfg["rccl_engines"] == 0 and cfg["rccl_distinct_ranks"] == 0
    full = json.load(open(os.path.join(ROOT, line["full"])))
    pub = full["config"]["publication"]
    assert pub["publications"] == cfg["pub_publications"]
    assert full["config"]["publication_compare"]["bytes_per_rank_per_publication"] == pub["bytes_per_rank_full"]
    assert line["roofline"]["regime"] in ("infinity-cache", "hbm") and line["cpu_baseline"] is None


def test_bench_starts_its_own_ranks_with_strong_scaling_and_an_automatic_cadence(rg):
    """The PLAIN command line -- `python bench.py --gpus 2 ...`, no launcher, what the driver runs at N = 1 -- starts its own
    ranks (torch.distributed.run underneath, rendezvous on 127.0.0.1 at a free port); `--total-groups` splits one population
    into equal contiguous ranges (strong scaling) and `--publish-every auto` picks the cadence from a measured exchange
    (rank 0 decides, the control plane broadcasts). Two ranks share the GPU (BENCH_SHARE_GPU=1: gloo moves the slices)."""
    env = dict(os.environ, BENCH_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--total-groups", "262144", "--steps", "6", "--warmup", "2",
           "--publish-every", "auto", "--no-publish-compare"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert len([ln for ln in r.stdout.splitlines() if ln.strip()]) == 1, r.stdout[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["groups_per_gpu"] == 131072
    assert line["value"] == pytest.approx(262144 * 6 / (line["ms_per_step"] * 6 / 1e3), rel=1e-6)
    cfg = line["config"]
    E_, tick_us, exchange_us = cfg["publish_every"], cfg["publish_auto_tick_us"], cfg["publish_auto_exchange_us"]
    assert 1 <= E_ <= 32 and tick_us > 0 and exchange_us > 0
    # (through the host transport an exchange is far slower than a tick: the rule must have picked a cadence above 1)
    assert abs(E_ - min(32, int(np.ceil(exchange_us / tick_us)))) <= 1  # (the note rounds to 0.01 us)
    assert cfg["pub_publications"] >= 1 + 6 // E_
    full = json.load(open(os.path.join(ROOT, line["full"])))
    assert full["config"]["publish_every_auto"]["publish_every"] == E_
    # an uneven split is refused, loudly
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--total-groups", "262145", "--steps", "2",
                          "--warmup", "1"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, cwd=ROOT)
    assert bad.returncode != 0 and "does not divide" in bad.stderr


# ---------------------------------------------------------------------------------------------------------------
# several ranks in ONE process (one host process driving every shard, like the reference's embedding drives many
# RawNodes from one thread): (1) one thread for all engines -- rg_comm_init_all / rg_publish_commit_all; (2) one thread per
# engine -- rg_comm_init / rg_publish_commit as a rank of its own process would. Three engines share this box's one GPU, so the
# exchange is the in-process device-to-device transport (1) or a transport callback between the threads (2); with a GPU per
# engine the same calls go through RCCL (grouped in form 1).
# ---------------------------------------------------------------------------------------------------------------
def _unsharded_commit(rg, torch, world, n, n_slots, workload, ticks):
    """The same global groups [0, world * n) in ONE engine: what every replica must end up holding."""
    eng = rg.Engine(world * n, n_slots)
    eng.workload_init(workload, first_group=0)
    cols, flags = _device_msgs(torch, eng, n_slots)
    per_tick = []
    for t in range(ticks):
        eng.workload_gen(workload, t, *[c.data_ptr() for c in cols], flags.data_ptr(), first_group=0)
        eng.tick_device(*[c.data_ptr() for c in cols], flags.data_ptr())
        per_tick.append(eng.read_column(rg.COL.COMMIT).reshape(world, n).copy())
    eng.close()
    return per_tick


@pytest.mark.parametrize("workload,n_slots", [(2, 5), (5, 7)])
def test_three_engines_one_thread_publish_together(rg, workload, n_slots):
    """rg_comm_init_all + rg_publish_commit_all: three shards of one process on one GPU, published after every tick by the one
    thread that also ticks them; every replica equals all three commit columns after every publication, and the unsharded run."""
    import torch
    from raft_rs_amd import engine as E
    world, n, ticks = 3, 20_224, 9
    want = _unsharded_commit(rg, torch, world, n, n_slots, workload, ticks)
    engs = [rg.Engine(n, n_slots) for _ in range(world)]
    assert [e.device_info()["engines_on_device"] for e in engs] == [1, 2, 3]
    for r, e in enumerate(engs):
        e.workload_init(workload, first_group=r * n)
    E.comm_init_all(engs, ring_ticks=4)
    assert [e.comm_info() for e in engs] == [{"rank": r, "world": 3, "transport": "local", "in_process": True, "rccl_ranks": 0,
                                              "rccl_rank": 0} for r in range(3)]
    with pytest.raises(rg.EngineError) as ei:
        engs[1].publish_commit()
    assert ei.value.code == E.ERR["STATE"] and "rg_publish_commit_all" in str(ei.value)
    bufs = [_device_msgs(torch, e, n_slots) for e in engs]
    for t in range(ticks):
        for r, (e, (cols, flags)) in enumerate(zip(engs, bufs)):
            e.workload_gen(workload, t, *[c.data_ptr() for c in cols], flags.data_ptr(), first_group=r * n)
            e.tick_device(*[c.data_ptr() for c in cols], flags.data_ptr())
        E.publish_commit_all(engs, full=(t == 6))  # (a forced full snapshot in between: all ranks take the same form)
        for e in engs:
            rep = e.published_commit()
            assert np.array_equal(rep, want[t]), (t, e.comm_rank)
    for r, e in enumerate(engs):
        assert np.array_equal(e.read_column(rg.COL.COMMIT), want[-1][r])
        st = e.publish_stats()
        assert st["publications"] == ticks + 1 and st["full_publications"] == 2 and st["bytes_per_rank_delta"] < 1.2 * n + 4096
        e.comm_destroy()
        e.close()


def test_publish_all_refuses_what_is_not_one_communicator(rg):
    from raft_rs_amd import engine as E
    a, b, c = rg.Engine(1000, 3), rg.Engine(1000, 3), rg.Engine(1200, 3)
    for e in (a, b, c):
        e.workload_init(2)
    with pytest.raises(rg.EngineError) as ei:  # unequal shards
        E.comm_init_all([a, b, c])
    assert ei.value.code == E.ERR["INVALID_ARG"]
    with pytest.raises(rg.EngineError) as ei:  # RCCL cannot put two ranks on one device
        E.comm_init_all([a, b], transport=E.COMM_ALL_RCCL)
    assert ei.value.code == E.ERR["INVALID_ARG"] and "share" in str(ei.value)
    with pytest.raises(rg.EngineError) as ei:  # never initialised
        E.publish_commit_all([a, b])
    assert ei.value.code == E.ERR["STATE"]
    E.comm_init_all([a, b])
    with pytest.raises(rg.EngineError) as ei:  # the ranks of a communicator are published together, in rank order
        E.publish_commit_all([b, a])
    assert ei.value.code == E.ERR["STATE"]
    E.publish_commit_all([a, b])
    assert np.array_equal(a.published_commit(1), b.read_column(rg.COL.COMMIT))
    for e in (a, b, c):
        e.close()


def test_three_engines_one_thread_each(rg):
    """Thread-per-engine: three threads of ONE process, each with its own engine on the shared GPU, each calling rg_comm_init /
    rg_tick_device / rg_publish_commit on its own handle -- concurrently, like three ranks -- with a transport callback that
    meets the other threads at a barrier (what RCCL's collective does between ranks). Nothing but the handles' own state and
    the callback's meeting point is shared; every replica ends up equal to the unsharded run."""
    import threading
    import torch
    world, n, n_slots, ticks = 3, 12_032, 5, 8
    want = _unsharded_commit(rg, torch, world, n, n_slots, 2, ticks)
    barrier = threading.Barrier(world)
    sends, errors, out = [None] * world, [], [None] * world

    class Dev:
        def __init__(self, ptr, nbytes):
            self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}

    def run(rank):
        try:
            torch.cuda.set_device(0)
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                eng = rg.Engine(n, n_slots)
                eng.set_stream(stream.cuda_stream)
                eng.workload_init(2, first_group=rank * n)

                def allgather(dev_send, dev_recv, nbytes, hip_stream):
                    torch.cuda.synchronize()  # (a host transport: the slice is complete before it is handed over)
                    sends[rank] = (dev_send, nbytes)
                    barrier.wait()
                    recv = torch.as_tensor(Dev(dev_recv, world * nbytes), device="cuda")
                    for r in range(world):
                        assert sends[r][1] == nbytes
                        recv[r * nbytes:(r + 1) * nbytes].copy_(torch.as_tensor(Dev(sends[r][0], nbytes), device="cuda"))
                    torch.cuda.synchronize()
                    barrier.wait()  # nobody resets its slice while somebody still reads it
                    return 0

                eng.comm_init(rank, world, transport=allgather, ring_ticks=3)
                cols, flags = _device_msgs(torch, eng, n_slots)
                seen = []
                for t in range(ticks):
                    eng.workload_gen(2, t, *[c.data_ptr() for c in cols], flags.data_ptr(), first_group=rank * n)
                    eng.tick_device(*[c.data_ptr() for c in cols], flags.data_ptr())
                    eng.publish_commit()
                    seen.append(eng.published_commit())
                out[rank] = seen
                barrier.wait()
                eng.comm_destroy()
                eng.close()
        except Exception as e:  # noqa: BLE001
            errors.append((rank, repr(e)))
            barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
    assert not errors, errors
    for rank in range(world):
        for t in range(ticks):
            assert np.array_equal(out[rank][t], want[t]), (rank, t)
