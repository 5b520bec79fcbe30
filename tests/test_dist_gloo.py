"""CPU multi-process test of the N>1 path: 2 and 3 ranks over gloo. The per-shard ticks run on the CPU
oracle here (the HIP engine needs a GPU); what is under test is the product's sharding arithmetic
(raft_rs_amd/sharding.py), the shard-local stream generation and the commit-index publication ENCODING of the C ABI
(host twins rg_pub_accumulate_host / rg_pub_apply_host; tests/test_publish_gpu.py drives rg_publish_commit itself)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["RG_ROOT"]); sys.path.insert(0, os.path.join(os.environ["RG_ROOT"], "tests"))
import oracle_lib as O
from raft_rs_amd import engine as E
from raft_rs_amd import sharding as S

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
G, P, WL, TICKS = 1536, 5, E.WL_MAJORITY, 4
sh = S.weak_shard(rank, world, G)
assert sh.first_group == rank * G and sh.n_groups == G
# strong shards: disjoint, contiguous, covering
tot = 10_001
ss = [S.strong_shard(r, world, tot) for r in range(world)]
assert ss[0].first_group == 0 and sum(s.n_groups for s in ss) == tot
assert all(ss[i].first_group + ss[i].n_groups == ss[i + 1].first_group for i in range(world - 1))

def run(first, n):
    st = O.alloc_state(n, P)
    E.workload_init_host(st, WL, first_group=first)
    cl = O.Cluster(n); cl.load_soa(st, term=4)
    msgs = E.MsgBuffers(n, P, st["stride"]); gout = np.zeros(n, dtype=np.uint32)
    commits = [st["commit"].copy()]  # [0] = before the first tick
    for t in range(TICKS):
        E.workload_gen_host(st, msgs, WL, t, first_group=first)
        cl.tick_soa(msgs.as_dict(), gout); cl.store_soa(st)
        commits.append(st["commit"].copy())
    return commits

mine = run(sh.first_group, sh.n_groups)
init, mine = mine[0], mine[1:]

# The product's publication encoding (include/raftgroups.h "multi-GPU"; the host twins of what the tick kernels write and
# the replica kernels add), with gloo standing in for ncclAllGather: a full snapshot first, then ~1 B/group per
# publication -- one publication per tick, then one for two ticks (the ticks accumulate).
bpr, stride = E.pub_bytes_per_rank(G), (G + 255) // 256 * 256
replica = np.zeros((world, stride), dtype=np.uint64)
full = np.zeros(stride, dtype=np.uint64); full[:G] = init
dist.all_gather_into_tensor(torch.from_numpy(replica.view(np.int64).reshape(-1)), torch.from_numpy(full.view(np.int64)))
def publish(old, new):
    sl = np.zeros(bpr, dtype=np.uint8)
    E.pub_accumulate_host(old, new, sl)
    gathered = np.zeros(world * bpr, dtype=np.uint8)
    dist.all_gather_into_tensor(torch.from_numpy(gathered), torch.from_numpy(sl))
    assert E.pub_apply_host(G, world, gathered, replica) == 0
    return bpr
prev = init
for t in (0, 1):
    publish(prev, mine[t]); prev = mine[t]
    assert (replica[rank, :G] == mine[t]).all()
publish(prev, mine[3])  # ticks 2 and 3 in one publication
assert bpr < 1.1 * G + 2048, "about one byte per group instead of eight"
if rank == 0:
    # the union of the shards must equal one unsharded run over all groups
    whole = run(0, world * G)
    got = replica[:, :G].reshape(-1)
    assert (got == whole[-1]).all(), "sharded, delta-published commit indices differ from the unsharded run"
    print("DIST_OK", world, int(got.sum()))
dist.barrier()
dist.destroy_process_group()
'''


import pytest


@pytest.mark.parametrize("world", [2, 3])  # (3: shards of a group count that does not divide, an odd gather)
def test_ranks_shard_and_publish_commit(tmp_path, world):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, RG_ROOT=ROOT, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(29531 + world), str(script)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
    assert f"DIST_OK {world}" in r.stdout, r.stdout[-3000:]


def test_publication_encoding_saturation_list_and_loss():
    """The host twins of the publication encoding: advances accumulate over an interval, a saturated byte spills its
    excess into the exact-value list (additive, so order never matters), a full list marks the slice lost."""
    import numpy as np
    from raft_rs_amd import engine as E
    G, cap = 1000, 4
    bpr, stride = E.pub_bytes_per_rank(G, cap), 1024
    rng = np.random.default_rng(5)
    c0 = rng.integers(0, 1 << 40, size=G).astype(np.uint64)
    steps = [c0]
    for k in range(3):
        adv = rng.integers(0, 60, size=G).astype(np.uint64)
        adv[7] = [10, 250, 1 << 33][k]        # saturates in the second step, a huge jump in the third
        adv[500] = [300, 0, 5][k]             # saturates at once
        steps.append(steps[-1] + adv)
    sl = np.zeros(bpr, dtype=np.uint8)
    for a, b in zip(steps, steps[1:]):        # three ticks, ONE publication
        E.pub_accumulate_host(a, b, sl, cap)
    replica = np.zeros((1, stride), dtype=np.uint64)
    replica[0, :G] = c0
    assert E.pub_apply_host(G, 1, sl, replica, cap) == 0
    assert (replica[0, :G] == steps[-1]).all()
    n_list = int(sl[:4].view(np.uint32)[0])
    assert 3 <= n_list <= cap, n_list
    # more saturated groups than the list holds: the slice says so, the receivers resynchronise
    sl = np.zeros(bpr, dtype=np.uint8)
    E.pub_accumulate_host(c0, c0 + np.uint64(1000), sl, cap)
    replica[0, :G] = c0
    assert E.pub_apply_host(G, 1, sl, replica, cap) == 1
    import pytest
    with pytest.raises(E.EngineError):
        E.pub_accumulate_host(c0 + np.uint64(1), c0, sl, cap)  # a commit index never decreases


def test_publication_encoding_property():
    """Hypothesis: whatever the advances, the cadence and the list capacity, accumulate -> gather -> apply reproduces the
    commit column exactly unless a slice reports itself lost -- and a lost slice is always reported."""
    import numpy as np
    from hypothesis import given, settings, strategies as st
    from raft_rs_amd import engine as E

    @settings(max_examples=60, deadline=None)
    @given(st.integers(1, 700), st.integers(1, 40), st.integers(1, 5), st.integers(0, 2 ** 32 - 1),
           st.sampled_from([3, 60, 300, 70000, 1 << 40]))
    def run(G, cap, ticks_per_pub, seed, big):
        rng = np.random.default_rng(seed)
        stride = (G + 255) // 256 * 256
        bpr = E.pub_bytes_per_rank(G, cap)
        commit = rng.integers(0, 1 << 50, size=G).astype(np.uint64)
        replica = np.zeros((1, stride), dtype=np.uint64)
        replica[0, :G] = commit
        for _ in range(4):  # four publications of `ticks_per_pub` ticks each
            sl = np.zeros(bpr, dtype=np.uint8)
            for _ in range(ticks_per_pub):
                adv = rng.integers(0, 90, size=G).astype(np.uint64)
                jump = rng.random(G) < 0.02
                adv[jump] = rng.integers(0, big, size=int(jump.sum())).astype(np.uint64)
                new = commit + adv
                E.pub_accumulate_host(commit, new, sl, cap)
                commit = new
            n_list, flags = sl[:8].view(np.uint32)
            lost = E.pub_apply_host(G, 1, sl, replica, cap)
            assert lost == int(n_list > cap or (flags & 1))
            if lost:
                replica[0, :G] = commit  # what the full snapshot that follows does
            assert (replica[0, :G] == commit).all()
            assert (replica[0, G:] == 0).all()
    run()
