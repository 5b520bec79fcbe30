"""Multi-GPU sharding of raft groups and publication of commit indices.

Raft groups are independent, so the path shards by disjoint contiguous group ranges with NO
data-path collective; the single exchange step is an all-gather that publishes every shard's
commit_idx column to all ranks (RCCL over xGMI on GPUs: backend "nccl"; gloo on CPU for tests).
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


class CommitPublisher:
    """Double-buffered all-gather of a shard's commit column.

    publish(i, commit) snapshots `commit` (a 1-D int64 tensor aliasing the engine's commit column)
    into staging buffer i&1 on the compute stream and all-gathers it on a side stream, so tick i+1
    overlaps the exchange of tick i. On CPU tensors (gloo) the same calls run synchronously.
    Requires equal shard sizes (all_gather_into_tensor); weak_shard guarantees that.
    """

    def __init__(self, dist, n_groups, world, device):
        import torch
        self.torch, self.dist, self.world, self.n = torch, dist, world, n_groups
        self.cuda = torch.device(device).type == "cuda"
        self.stage = [torch.empty(n_groups, dtype=torch.int64, device=device) for _ in range(2)]
        self.gathered = [torch.empty(world * n_groups, dtype=torch.int64, device=device) for _ in range(2)]
        if self.cuda:
            self.side = torch.cuda.Stream()
            self.ready = [torch.cuda.Event() for _ in range(2)]
            self.done = [torch.cuda.Event() for _ in range(2)]
            self.pending = [False, False]  # done[b] recorded and not yet waited for

    def publish(self, i, commit, stream=None):
        """`stream`: the stream the tick kernels of this shard run on (default: torch's current stream)."""
        b = i & 1
        if not self.cuda:
            self.stage[b].copy_(commit)
            self.dist.all_gather_into_tensor(self.gathered[b], self.stage[b])
            return b
        torch = self.torch
        main = stream if stream is not None else torch.cuda.current_stream()
        if self.pending[b]:
            main.wait_event(self.done[b])  # the previous gather out of this staging buffer has finished
        with torch.cuda.stream(main):
            self.stage[b].copy_(commit, non_blocking=True)
        self.ready[b].record(main)
        with torch.cuda.stream(self.side):
            self.side.wait_event(self.ready[b])
            self.dist.all_gather_into_tensor(self.gathered[b], self.stage[b])
            self.done[b].record(self.side)
        self.pending[b] = True
        return b

    def join(self, stream=None):
        """Make `stream` (default: the current stream) wait for every outstanding gather (needed before
        the end of a HIP-graph capture, and before reading `result`)."""
        if not self.cuda:
            return
        main = stream if stream is not None else self.torch.cuda.current_stream()
        for b in range(2):
            if self.pending[b]:
                main.wait_event(self.done[b])
                self.pending[b] = False

    def result(self, b):
        """[world, n_groups] view of the gathered commit indices of buffer b (sync first on CUDA)."""
        return self.gathered[b].view(self.world, self.n)
