"""Multi-GPU sharding of raft groups: which groups a rank holds.

Raft groups are independent, so the path shards by disjoint contiguous group ranges with NO data-path collective.
The single exchange step -- the publication of commit indices -- lives behind the C ABI (rg_comm_init /
rg_publish_commit in include/raftgroups.h: ncclAllGather over xGMI of the ~1 B/group slices the ticks produce).
"""
from dataclasses import dataclass


@dataclass(frozen=True)
class Shard:
    rank: int
    world: int
    first_group: int  # global index of this rank's first group
    n_groups: int     # groups held by this rank


def weak_shard(rank, world, groups_per_rank):
    """Weak scaling (bench.py): every rank holds `groups_per_rank` groups; global ids are rank-major."""
    assert 0 <= rank < world
    return Shard(rank, world, rank * groups_per_rank, groups_per_rank)


def strong_shard(rank, world, total_groups):
    """Strong scaling: [r*G/W, (r+1)*G/W) -- ranges are disjoint, contiguous and cover [0, G)."""
    assert 0 <= rank < world
    lo = rank * total_groups // world
    hi = (rank + 1) * total_groups // world
    return Shard(rank, world, lo, hi - lo)
