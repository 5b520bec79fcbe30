"""GPU: rg_permute_groups -- every column of the engine follows a permutation of the shard's groups -- and the road from a
shard loaded with its replica-set sizes interleaved to the one-launch class kernel (rg_plan_placement + rg_permute_groups).
Membership is the host's to change at any time (ProgressTracker::apply_conf, src/tracker.rs:380-397); the engine offers the
re-placement, the oracle checks that nothing but positions changed."""
import numpy as np
import pytest

import fuzz
import oracle_lib as O
import sendstage

pytestmark = pytest.mark.gpu
TERM = 5


def _all_columns(rg, eng):
    return {name: eng.read_column(c) for c, name in enumerate(rg.COL.NAMES)}


def _permuted(cols, perm, G):
    p = perm.astype(np.int64)
    out = {}
    for k, v in cols.items():
        if k == "pflags":
            out[k] = v[p]
        elif v.ndim == 2:
            w = v.copy()
            w[:, :G] = v[:, p]
            out[k] = w
        else:
            w = v.copy()
            w[:G] = v[p]
            out[k] = w
    return out


def test_config5_loaded_interleaved_then_placed_runs_one_launch(rg):
    """Config 5 with its sizes interleaved (group id order), a few ticks in that layout (the plain kernel), then
    rg_plan_placement + rg_permute_groups: every column is the old one under the permutation, the engine derives three size
    classes, the next ticks run k_tick_classes -- and five ticks of the rollover stream match the oracle, which was handed the
    ORIGINAL state permuted on the host (numpy), not anything read back from the engine."""
    import torch
    from raft_rs_amd import engine as E
    G, P = 200_000 + 77, 7
    eng = rg.Engine(G, P)
    eng.workload_init(5)
    assert eng.size_classes() == []
    cols = [torch.zeros((P, eng.stride), dtype=torch.int64, device="cuda") for _ in range(4)]
    flags = torch.zeros((G, 8), dtype=torch.uint8, device="cuda")
    for t in range(3):  # (elections of these ticks put runs into the term tables: cold columns with content to move)
        eng.workload_gen(5, t, *[c.data_ptr() for c in cols], flags.data_ptr())
        eng.tick_device(*[c.data_ptr() for c in cols], flags.data_ptr())
    assert eng.device_info()["last_tick_kernel"] == "k_tick_lane"
    before = _all_columns(rg, eng)
    assert before["run_count"].any() and before["cur_term"].max() > TERM
    perm, planned = eng.place_by_size_class()
    assert [q for _, _, q in planned] == [3, 5, 7] and eng.size_classes() == planned
    after = _all_columns(rg, eng)
    want = _permuted(before, perm, G)
    for k in want:
        assert np.array_equal(after[k], want[k]), k
    # from here on the shard IS a class-placed one: the generator's sorted placement names the same groups at the same places
    st = {"n_groups": G, "n_slots": P, "stride": eng.stride, **{k: want[k] for k in rg.COL.NAMES[:12]}}
    st = O.add_term_table(st)
    for k in ("run_first", "run_term", "dummy_index", "dummy_term", "cur_term"):
        st[k][...] = want[k]
    cl = O.Cluster(G)
    cl.load_soa(st, term=TERM)
    mb = rg.MsgBuffers(G, P, eng.stride)
    gout = np.zeros(G, dtype=np.uint32)
    elections = 0
    for t in range(3, 8):
        cl.store_soa(st)
        E.workload_gen_host(st, mb, 5, t, sorted_classes=True)
        eng.workload_gen(5, t, *[c.data_ptr() for c in cols], flags.data_ptr(), sorted_classes=True)
        eng.sync()
        # (the device generator, run on the permuted engine, and its host twin, run on the oracle's state, write the same events;
        #  cells of peers a group does not have are not written by either and keep what the buffers held)
        assert np.array_equal(flags.cpu().numpy(), mb.m_flags)
        eng.tick_device(*[c.data_ptr() for c in cols], flags.data_ptr())
        cl.tick_soa(mb.as_dict(), gout)
        got = eng.read_state()
        cl.store_soa(st)
        diffs = fuzz.diff_states(st, got, G, P)
        assert not diffs, (t, diffs[:8])
        assert np.array_equal(got["out"], gout), t
        elections += int(((gout & 0x10) != 0).sum())
    assert eng.device_info()["last_tick_kernel"] == "k_tick_classes" and elections > 5 * G / 32 * 0.8
    eng.close()


@pytest.mark.parametrize("n_slots,cap", [(7, 4), (5, 256)])
def test_windows_and_work_travel_with_their_groups(rg, n_slots, cap):
    """Device Inflights: two engines take the same random stream, B in the order the groups arrived, A re-placed in the middle
    of the run (any membership: random cfg words, learners, joint configurations). After every tick A's state, windows, rings
    and work items are B's under the permutation -- and B is checked against the oracle as everywhere else."""
    rng = np.random.default_rng(8800 + n_slots)
    G = 5000 + 29
    st = O.add_term_table(O.alloc_state(G, n_slots))
    st["cfg"][:] = fuzz.random_cfg(rng, G, n_slots, missing_progress_frac=0.3)
    fuzz.random_state(rng, st, small_values=True)
    fuzz.random_term_table(rng, st, term=6)
    A, B = rg.Engine(G, n_slots, max_inflight=cap), rg.Engine(G, n_slots, max_inflight=cap)
    A.load_state(st)
    B.load_state(st)
    cl = O.Cluster(G)
    cl.load_soa(st, term=6, max_inflight=cap)
    cl.set_own_inflights(True)
    msgs = O.alloc_msgs(G, n_slots)
    mbA, mbB = rg.MsgBuffers(G, n_slots, A.stride), rg.MsgBuffers(G, n_slots, B.stride)
    gout = np.zeros(G, dtype=np.uint32)
    perm = np.arange(G, dtype=np.int64)
    items_seen = 0
    for t in range(8):
        if t == 4:
            p, planned = A.place_by_size_class()
            perm = p.astype(np.int64)
            assert len(planned) >= 2 and A.size_classes() in (planned, [])  # ([]: engines with device Inflights run k_tick_send / the plain kernel)
        cl.store_soa(st)
        fuzz.random_msgs(rng, st, msgs, sent_p=0.0, heartbeat_p=0.2)
        sendstage.prepare_msgs(msgs)
        for k in ("m_index", "m_commit", "m_hint", "m_rs"):
            getattr(mbB, k)[...] = msgs[k]
            getattr(mbA, k)[:, :G] = msgs[k][:, perm]
        mbB.m_flags[...] = msgs["m_flags"]
        mbA.m_flags[...] = msgs["m_flags"][perm]
        if t % 2:
            A.tick_send(mbA, 2)
            B.tick_send(mbB, 2)
        else:
            for e, mb in ((A, mbA), (B, mbB)):
                e.tick(mb)
                e.send_appends(2)
        cl.tick_soa(msgs, gout)
        itB = sendstage.compare_items(B.send_items(), cl.send_stage_soa(gout, 2))
        itA = sendstage.items_dict(A.send_items())
        inv = np.empty(G, dtype=np.int64)
        inv[perm] = np.arange(G)
        assert itA == {(int(inv[g]), s): v for (g, s), v in itB.items()}, t
        items_seen += len(itB)
        sb, sa = B.read_state(), A.read_state()
        cl.store_soa(st)
        assert not fuzz.diff_states(st, sb, G, n_slots), t
        for k in rg.COL.NAMES[:12]:
            x = sb[k][perm] if (sb[k].ndim == 1 or k == "pflags") else sb[k][:, perm]
            y = sa[k] if (sa[k].ndim == 1 or k == "pflags") else sa[k][:, :G]
            assert np.array_equal(x, y[:G] if y.ndim == 1 else y), (t, k)
        metaB, ringB = B.read_inflights()
        metaA, ringA = A.read_inflights()
        sendstage.compare_rings(cl, metaB, ringB, st, cap)
        for g in range(0, G, 7):
            for s in range(n_slots):
                assert sendstage.ring_contents(metaA, ringA, g, s, cap) == sendstage.ring_contents(metaB, ringB, int(perm[g]), s, cap), (t, g, s)
    assert items_seen > 2000
    A.close()
    B.close()


def test_the_mirror_follows_and_bad_calls_are_refused(rg):
    """The host mirror's peer-id / term tables move with the groups (rg_step maps Message.from through them); a queue of steps,
    a non-permutation and a second use of one position are refused with nothing changed."""
    from raft_rs_amd import engine as E
    G, P = 3000, 5
    eng = rg.Engine(G, P)
    eng.workload_init(2)
    for g in range(G):
        eng.set_peers(g, [1000 * (g + 1) + s for s in range(P)], TERM)
    commit0 = eng.read_column(rg.COL.COMMIT)
    hi = eng.read_column(rg.COL.TERM_HI)
    bad = np.arange(G, dtype=np.uint64)
    bad[5] = 6
    with pytest.raises(rg.EngineError) as ei:
        eng.permute_groups(bad)
    assert ei.value.code == E.ERR["INVALID_ARG"] and "not a permutation" in str(ei.value)
    eng.step(7, 1000 * 8 + 1, TERM, int(hi[7]))
    with pytest.raises(rg.EngineError) as ei:
        eng.permute_groups(np.arange(G, dtype=np.uint64)[::-1].copy())
    assert ei.value.code == E.ERR["SLOT_BUSY"]
    eng.flush()
    perm = np.random.default_rng(3).permutation(G).astype(np.uint64)
    eng.checkpoint()
    eng.permute_groups(perm)
    with pytest.raises(rg.EngineError):
        eng.restore()  # (the checkpoint imaged the old placement: dropped)
    assert np.array_equal(eng.read_column(rg.COL.TERM_HI), hi[perm.astype(np.int64)])
    # position i now holds old group perm[i]: its peers' ids are the OLD group's ids
    for i in (0, 17, G - 1):
        old = int(perm[i])
        for s in range(1, P):
            eng.step(i, 1000 * (old + 1) + s, TERM, int(hi[old]))
        with pytest.raises(rg.EngineError) as ei:
            eng.step(i, 1000 * (i + 1) + 1 if i != old else 999, TERM, 1)
        assert ei.value.code == E.ERR["STEP_PEER_NOT_FOUND"]
    eng.flush()
    groups, commit, out = eng.ingested_results()
    assert sorted(groups.tolist()) == [0, 17, G - 1]
    for g, c in zip(groups.tolist(), commit.tolist()):
        assert c == int(hi[int(perm[g])]) >= int(commit0[int(perm[g])])  # every follower acked last_index: committed
    eng.close()
