"""rg_plan_placement (pure host arithmetic, no GPU): from a shard's cfg words, the permutation that places its groups by
replica-set size class -- what lets a host whose groups arrive interleaved (or drift there through conf changes,
ProgressTracker::apply_conf, src/tracker.rs:380-397) reach the layout the one-launch class kernel runs."""
import numpy as np
import pytest

import oracle_lib as O
from raft_rs_amd import engine as E


def _body(cfg, P):
    """the smallest body k_tick_classes<P> has for the slots a cfg word names (3, 5, 7 below P; P)"""
    tr = (cfg >> 20) & 0xf
    m = ((cfg >> 24) & 0xff) | (cfg & 0xff) | ((cfg >> 8) & 0xff) | (1 << ((cfg >> 16) & 7)) | ((1 << (tr - 1)) if tr else 0)
    k = int(m).bit_length()
    for q in (3, 5, 7):
        if P > q and k <= q:
            return q
    return P


def test_config5_interleaved_becomes_the_sorted_layout():
    """Config 5's population (3 / 5 / 7 peers by global id mod 3) loaded in id order: the plan is exactly the placement the
    synthetic generator calls RG_WL_PLACE_SORTED -- all ids 0 mod 3 first, then 1 mod 3, then 2 mod 3, each ascending."""
    G, P = 100_003, 7
    st = O.alloc_state(G, P)
    E.workload_init_host(st, 5)
    perm, classes = E.plan_placement(st["cfg"], P)
    n0, n1 = (G + 2) // 3, (G + 1) // 3
    want = np.concatenate([np.arange(0, G, 3), np.arange(1, G, 3), np.arange(2, G, 3)]).astype(np.uint64)
    assert np.array_equal(perm, want)
    # the ranges the engine will derive: whole blocks of 64; a block that straddles a boundary belongs to the larger class
    b0, b1 = n0 // 64 * 64, (n0 + n1) // 64 * 64
    assert classes == [(0, b0, 3), (b0, b1 - b0, 5), (b1, G - b1, 7)]
    srt = O.alloc_state(G, P)
    E.workload_init_host(srt, 5, sorted_classes=True)
    for k in ("cfg", "commit", "term_hi"):
        assert np.array_equal(srt[k], st[k][perm.astype(np.int64)]), k
    assert np.array_equal(srt["match"][:, :G], st["match"][:, perm.astype(np.int64)])


@pytest.mark.parametrize("P", [3, 4, 5, 6, 7, 8])
def test_plan_is_a_stable_sort_by_class_for_any_membership(P):
    import fuzz
    rng = np.random.default_rng(40 + P)
    G = 20_011
    cfg = fuzz.random_cfg(rng, G, P, missing_progress_frac=0.3)
    # conf changes that shrank some groups to the lowest slots (what a 3-replica group in a 7-slot engine looks like)
    small = rng.random(G) < 0.4
    cfg[small] = E.cfg_make(0b011, 0, 0, present=0b111)
    perm, classes = E.plan_placement(cfg, P)
    assert np.array_equal(np.sort(perm), np.arange(G, dtype=np.uint64))
    bodies = np.array([_body(int(c), P) for c in cfg])
    placed = bodies[perm.astype(np.int64)]
    assert (np.diff(placed) >= 0).all()  # classes ascend along the shard
    for q in np.unique(bodies):  # ... and inside a class the groups keep their order
        assert (np.diff(perm[placed == q].astype(np.int64)) > 0).all()
    # the planned ranges are what a per-block maximum over the placed column gives
    want, b = [], 0
    blocks = [int(placed[i:i + 64].max()) for i in range(0, G, 64)]
    while b < len(blocks):
        e = b
        while e < len(blocks) and blocks[e] == blocks[b]:
            e += 1
        want.append((b * 64, min(e * 64, G) - b * 64, blocks[b]))
        b = e
    assert classes == want
    assert sum(n for _, n, _ in classes) == G


def test_plan_refuses_nonsense():
    cfg = np.zeros(10, dtype=np.uint32)
    with pytest.raises(E.EngineError) as ei:
        E.plan_placement(cfg, 9)
    assert ei.value.code == E.ERR["INVALID_ARG"]
    with pytest.raises(E.EngineError):
        E.plan_placement(cfg[:0], 5)
