"""GPU parity at BASELINE.json's FULL sizes: every group of every 1 M-group configuration against the oracle
(all host cores, ro_tick_soa_mt), every column and the result word, several ticks of the device-generated stream
the bench replays -- plus the reference's 80 commit / group-commit golden vectors through the engine's kernels.

  config 2  1 000 000 x 5, majority                         (bench.py default)
  config 3  1 000 000 x 5 slots, joint {0,1,2}&&{1,2,3} + learners
  config 4  one rank's shard of 8 M x 7: 1 000 000 x 7, majority
  config 5  1 000 000 mixed 3/5/7 + leader-term rollover, both layouts: one 7-slot engine, and one engine per
            replica-set size (what bench.py --workload 5 runs)
"""
import json
import os

import numpy as np
import pytest

import fuzz
import oracle_lib as O

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
TERM = 5  # RG_WL_TERM0 of the generator
MSG_KEYS = ("m_index", "m_commit", "m_hint", "m_rs")


def _run_full_size(rg, workload, n_groups, n_slots, ticks, first_group=0, fixed_peers=0, variant=0, placed=False):
    import torch
    threads = os.cpu_count() or 8
    eng = rg.Engine(n_groups, n_slots, variant=variant)
    eng.workload_init(workload, first_group=first_group, fixed_peers=fixed_peers, sorted_classes=placed)
    if placed:  # three ranges, one launch per tick (k_tick_classes)
        assert [q for _, _, q in eng.size_classes()] == [3, 5, 7]
    st = eng.read_state()
    cl = O.Cluster(n_groups)
    cl.load_soa(st, term=TERM)
    dev = [torch.zeros((n_slots, eng.stride), dtype=torch.int64, device="cuda") for _ in range(4)]
    dflags = torch.zeros((n_groups, 8), dtype=torch.uint8, device="cuda")
    msgs = {"n_groups": n_groups, "n_slots": n_slots, "stride": eng.stride}
    gout = np.zeros(n_groups, dtype=np.uint32)
    seen = {"changed": 0, "rejects": 0, "elections": 0, "valid": 0}
    for t in range(ticks):
        eng.workload_gen(workload, t, *[d.data_ptr() for d in dev], dflags.data_ptr(), first_group=first_group,
                         fixed_peers=fixed_peers, sorted_classes=placed)
        eng.sync()
        for k, d in zip(MSG_KEYS, dev):
            msgs[k] = np.ascontiguousarray(d.cpu().numpy().view(np.uint64))
        msgs["m_flags"] = np.ascontiguousarray(dflags.cpu().numpy())
        eng.tick_device(*[d.data_ptr() for d in dev], dflags.data_ptr())
        stepped = cl.tick_soa_mt(msgs, gout, threads)
        got = eng.read_state()
        cl.store_soa(st)
        # ALL groups, all columns: a stride / tail / tile bug anywhere in the 1 M groups fails here
        for k in fuzz.STATE_KEYS:
            if st[k].ndim == 2 and k != "pflags":
                same = np.array_equal(st[k][:, :n_groups], got[k][:, :n_groups])
            else:
                same = np.array_equal(st[k], got[k])
            if not same:
                diffs = fuzz.diff_states(st, got, n_groups, n_slots, keys=(k,))
                # (cells of slots without a Progress are not state: diff_states masks them)
                assert not diffs, f"workload {workload} {n_groups}x{n_slots} tick {t}: " + "; ".join(diffs[:8])
        bad = np.nonzero(got["out"] != gout)[0]
        assert bad.size == 0, (workload, t, bad[:5], [hex(x) for x in got["out"][bad[:5]]], [hex(x) for x in gout[bad[:5]]])
        assert not (gout & 2).any(), "well-formed stream: no fault"
        seen["changed"] += int((gout & 1).sum())
        seen["valid"] += int(stepped)
        seen["rejects"] += int(((msgs["m_flags"][:, 1:] & 3) == 3).sum())
        seen["elections"] += int(((gout & 0x10) != 0).sum())
    commit, out = eng.results()
    assert np.array_equal(commit, st["commit"]) and np.array_equal(out, gout)
    eng.close()
    return seen


@pytest.mark.parametrize("workload,n_slots,name", [(2, 5, "config 2"), (3, 5, "config 3"), (2, 7, "config 4 shard")])
def test_one_million_groups_match_the_oracle(rg, workload, n_slots, name):
    seen = _run_full_size(rg, workload, 1_000_000, n_slots, ticks=4)
    assert seen["changed"] > 2_500_000 and seen["valid"] > 4 * 1_000_000 * (n_slots - 1) * 0.9, (name, seen)


def test_config5_one_engine_matches_the_oracle(rg):
    """1 M groups of 3 / 5 / 7 peers interleaved in one 7-slot engine, elections on every tick."""
    seen = _run_full_size(rg, 5, 1_000_000, 7, ticks=5)
    assert seen["elections"] > 5 * 1_000_000 / 32 * 0.9 and seen["rejects"] > 300_000, seen


def test_config5_placed_by_size_class_one_launch_matches_the_oracle(rg):
    """bench.py --workload 5 --slots 7 --sorted: the same 1 M groups placed by replica-set size class in ONE 7-slot engine,
    one launch per tick whose blocks run the tick instantiated for 3, 5 or 7 slots (k_tick_classes): every group, every
    column, five ticks of the rollover stream."""
    seen = _run_full_size(rg, 5, 1_000_000, 7, ticks=5, placed=True)
    assert seen["elections"] > 5 * 1_000_000 / 32 * 0.9 and seen["rejects"] > 300_000, seen


def test_config5_size_class_engines_match_the_oracle(rg):
    """bench.py --workload 5: one engine per replica-set size, global group ids as the bench assigns them."""
    G = 1_000_000
    first, total = 0, {"elections": 0, "rejects": 0}
    for slots, n in ((3, G // 3), (5, G // 3), (7, G - 2 * (G // 3))):
        seen = _run_full_size(rg, 5, n, slots, ticks=5, first_group=first, fixed_peers=slots)
        first += n
        for k in total:
            total[k] += seen[k]
    assert total["elections"] > 5 * G / 32 * 0.9 and total["rejects"] > 300_000, total


@pytest.mark.parametrize("layout", ["one-engine", "size-classes"])
def test_config5_compact_variant_full_size(rg, layout):
    """RG_VARIANT_COMPACT (rare groups gathered into one wave per workgroup through LDS) over config 5 at full size:
    the groups change lanes inside the kernel, the results must not."""
    G = 1_000_000
    if layout == "one-engine":
        seen = _run_full_size(rg, 5, G, 7, ticks=4, variant=5)
    else:
        first, seen = 0, {"elections": 0, "rejects": 0}
        for slots, n in ((3, G // 3), (5, G // 3), (7, G - 2 * (G // 3))):
            part = _run_full_size(rg, 5, n, slots, ticks=4, first_group=first, fixed_peers=slots, variant=5)
            first += n
            for k in seen:
                seen[k] += part[k]
    assert seen["elections"] > 4 * G / 32 * 0.9 and seen["rejects"] > 200_000, seen


def test_config2_compact_variant_full_size(rg):
    """... and over the steady stream, where no group changes lanes."""
    _run_full_size(rg, 2, 1_000_000, 5, ticks=2, variant=5)


def test_config2_lds_variant_full_size(rg):
    """The LDS-staged kernel over the same 1 M x 5 stream (two ticks)."""
    _run_full_size(rg, 2, 1_000_000, 5, ticks=2, variant=2)


# ---------------------------------------------------------------------------------------------------------------
# the reference's commit golden vectors (src/quorum/testdata/{majority_commit,joint_commit,joint_group_commit}.txt)
# ---------------------------------------------------------------------------------------------------------------
VEC = json.load(open(os.path.join(HERE, "golden", "quorum_vectors.json"), encoding="utf-8"))


def commit_cases():
    from test_oracle_golden import build_case, parse_result
    cases = []
    for fname, gc in (("majority_commit.txt", False), ("joint_commit.txt", False), ("joint_group_commit.txt", True)):
        for case in VEC[fname]:
            ids, idsj, joint, look = build_case(case["args"])
            cases.append((f"{fname}:{case['line']}", ids, idsj, look, gc, parse_result(case["result"])))
    return cases


@pytest.mark.parametrize("variant", [0, 1, 3])
def test_commit_golden_vectors_on_the_gpu(rg, variant):
    """All 16 + 50 + 14 `committed` / `group_committed` cases, one raft group per case, through
    rg_maximal_committed_index (variant 0: k_recompute2, two groups per lane; 1: k_recompute, one group per lane;
    3: the wave-cooperative rank select, for the cases without group commit) and through rg_recompute (Raft::maybe_commit: the same index, gated by the log)."""
    cases = commit_cases()
    assert len(cases) == 80
    if variant == 3:
        cases = [c for c in cases if not c[4]]  # the cooperative kernel has no group-commit path
    G, P = len(cases), 8
    st = O.alloc_state(G, P)
    for g, (_, ids, idsj, look, gc, _) in enumerate(cases):
        slot = {pid: k for k, pid in enumerate(sorted(set(ids) | set(idsj)))}
        m = lambda s: sum(1 << slot[i] for i in s)
        # a voter without an acked index has no Progress: acked_index() -> None -> {0, 0} (majority.rs:80-82)
        st["cfg"][g] = rg.cfg_make(m(ids), m(idsj), 0, group_commit=gc, present=m(look.keys()))
        for pid, (idx, gid) in look.items():
            st["match"][slot[pid], g] = idx
            st["next"][slot[pid], g] = idx + 1
            st["gid"][slot[pid], g] = gid
        st["term_lo"][g], st["term_hi"][g] = 1, 1 << 40  # every index is of the leader's term
    eng = rg.Engine(G, P, variant=variant)
    eng.load_state(st)
    mci, used = eng.maximal_committed_index(with_flag=True)
    for g, (name, *_rest, want) in enumerate(cases):
        assert int(mci[g]) == want, f"{name}: engine {int(mci[g])}, reference {want}"
    # joint symmetry (datadriven_test.rs:176-181): swap the two majorities
    cfg2 = (((st["cfg"] & 0xff) << 8) | ((st["cfg"] >> 8) & 0xff) | (st["cfg"] & 0xffff0000)).astype(np.uint32)
    eng.load_column(rg.COL.CFG, cfg2)
    assert np.array_equal(eng.maximal_committed_index(), mci)
    eng.load_column(rg.COL.CFG, st["cfg"])
    # Raft::maybe_commit on the same groups: commit = mci where the log holds it (mci <= last_index), else unchanged
    eng.recompute()
    commit, out = eng.results()
    for g, (name, *_rest, want) in enumerate(cases):
        exp = want if 0 < want <= (1 << 40) else 0
        assert int(commit[g]) == exp and bool(out[g] & 1) == (exp > 0), name
    eng.close()
