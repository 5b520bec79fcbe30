"""CPU multi-process test of the N>1 path: 2 ranks over gloo. The per-shard ticks run on the CPU
oracle here (the HIP engine needs a GPU); what is under test is the product's sharding arithmetic,
the shard-local stream generation and the commit-index publication (raft_rs_amd/sharding.py)."""
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
    commits = []
    for t in range(TICKS):
        E.workload_gen_host(st, msgs, WL, t, first_group=first)
        cl.tick_soa(msgs.as_dict(), gout); cl.store_soa(st)
        commits.append(st["commit"].copy())
    return commits

mine = run(sh.first_group, sh.n_groups)
pub = S.CommitPublisher(dist, G, world, "cpu")
for t in range(TICKS):
    b = pub.publish(t, torch.from_numpy(mine[t].view(np.int64)))
    got = pub.result(b).numpy().view(np.uint64)
    assert (got[rank] == mine[t]).all()
if rank == 0:
    # the union of the shards must equal one unsharded run over all groups
    whole = run(0, world * G)
    got = pub.result((TICKS - 1) & 1).numpy().view(np.uint64).reshape(-1)
    assert (got == whole[-1]).all(), "sharded commit indices differ from the unsharded run"
    print("DIST_OK", world, int(got.sum()))
dist.barrier()
dist.destroy_process_group()
'''


def test_two_ranks_shard_and_publish_commit(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, RG_ROOT=ROOT, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "DIST_OK 2" in r.stdout, r.stdout[-3000:]
