"""GPU: the sparse path (rg_ingest -> rg_tick_ingested) against the oracle: wire-order records for a
small random subset of the groups, bit-exact state afterwards, results only for the touched groups."""
import numpy as np
import pytest

import fuzz
import oracle_lib as O

pytestmark = pytest.mark.gpu


def records_from_msgs(rg, msgs, groups, rng, n_slots):
    from raft_rs_amd.engine import WIRE_DTYPE
    recs = []
    for g in groups:
        for p in range(n_slots):
            f = int(msgs["m_flags"][g, p])
            if f:
                recs.append((g, msgs["m_index"][p, g], msgs["m_commit"][p, g], msgs["m_hint"][p, g],
                             msgs["m_rs"][p, g], 0, p, f, 0))
    arr = np.array(recs, dtype=WIRE_DTYPE)
    rng.shuffle(arr)  # wire order is arbitrary across cells
    return arr


@pytest.mark.parametrize("n_slots", [3, 5, 7])
def test_sparse_ticks_match_oracle(rg, n_slots):
    rng = np.random.default_rng(500 + n_slots)
    G = 20000
    st = O.alloc_state(G, n_slots)
    st["cfg"][:] = fuzz.random_cfg(rng, G, n_slots)
    fuzz.random_state(rng, st)
    eng = rg.Engine(G, n_slots)
    eng.load_state(st)
    cl = O.Cluster(G)
    cl.load_soa(st, term=4)
    msgs = O.alloc_msgs(G, n_slots)
    gout = np.zeros(G, dtype=np.uint32)
    mb = rg.MsgBuffers(G, n_slots, eng.stride)
    for t in range(6):
        cl.store_soa(st)
        fuzz.random_msgs(rng, st, msgs)
        if t == 3:  # a dense tick in between (RG_COL_OUT then holds dense results)
            for k in ("m_index", "m_commit", "m_hint", "m_rs", "m_flags"):
                getattr(mb, k)[...] = msgs[k]
            eng.tick(mb)
            cl.tick_soa(msgs, gout)
        else:
            touched = np.sort(rng.choice(G, size=G // 40, replace=False))
            keep = np.zeros(G, dtype=bool)
            keep[touched] = True
            msgs["m_flags"][~keep] = 0
            recs = records_from_msgs(rg, msgs, touched, rng, n_slots)
            # split the batch in two ingests to exercise accumulation
            half = len(recs) // 2
            assert eng.ingest(recs[:half]) == 0
            assert eng.ingest(recs[half:]) == 0
            n = eng.tick_ingested()
            with_events = np.nonzero(msgs["m_flags"].any(axis=1))[0]
            assert n == len(with_events)
            cl.tick_soa(msgs, gout)
            groups, commit, out = eng.ingested_results()
            order = np.argsort(groups)
            assert (groups[order] == with_events).all()
            cl.store_soa(st)
            assert (commit[order] == st["commit"][with_events]).all()
            assert (out[order] == gout[with_events]).all()
        got = eng.read_state()
        cl.store_soa(st)
        diffs = fuzz.diff_states(st, got, G, n_slots)
        assert not diffs, (t, diffs[:5])
        assert (got["out"] == gout).all(), f"tick {t}: RG_COL_OUT must be zero for untouched groups"
    eng.close()


def test_duplicate_and_malformed_records_are_dropped_and_counted(rg):
    from raft_rs_amd.engine import WIRE_DTYPE
    G, P = 1000, 3
    eng = rg.Engine(G, P)
    eng.workload_init(rg.WL_MAJORITY)
    st = eng.read_state()
    hi = int(st["term_hi"][7])
    V = rg.MF.VALID
    recs = np.array([(7, hi, 0, 0, 0, 0, 1, V, 0), (7, hi - 1, 0, 0, 0, 0, 1, V, 0),  # same cell twice
                     (G + 5, 1, 0, 0, 0, 0, 0, V, 0), (3, 1, 0, 0, 0, 0, 9, V, 0),    # bad group / slot
                     (9, 1, 0, 0, 0, 0, 2, 0, 0)], dtype=WIRE_DTYPE)                    # no event bits
    assert eng.ingest(recs) == 4
    assert eng.tick_ingested() == 1
    groups, commit, out = eng.ingested_results()
    assert list(groups) == [7]
    got = eng.read_state()
    assert got["match"][1, 7] in (hi, max(hi - 1, st["match"][1, 7]))
    assert eng.tick_ingested() == 0  # nothing pending
    eng.close()


def test_flush_uses_the_sparse_path_for_few_groups(rg):
    G, P, TERM = 4096, 3, 5
    eng = rg.Engine(G, P)
    eng.workload_init(rg.WL_MAJORITY)
    st = eng.read_state()
    cl = O.Cluster(G)
    cl.load_soa(st, term=TERM)
    for g in range(G):
        eng.set_peers(g, [11, 12, 13], TERM)
    rng = np.random.default_rng(9)
    for rnd in range(3):
        touched = rng.choice(G, size=50, replace=False)
        want = {}
        for g in touched:
            g = int(g)
            pr = cl.pr(g, 2)
            idx = min(cl.last_index(g), pr.matched + int(rng.integers(1, 5)))
            eng.mark_sent(g, 12)
            eng.step(g, 12, TERM, idx, commit=min(idx, cl.committed(g)))
            O.lib().ro_progress_update_state(pr, cl.last_index(g))
            o = cl.step(g, 2, idx, commit=min(idx, cl.committed(g)), ins_full=0)
            want[g] = (cl.committed(g), (1 if o.commit_changed else 0) | (int(o.send_append) << 9) |
                       (int(o.send_more) << 17) | (int(o.free_to) << 25))
        eng.flush()
        groups, commit, out = eng.ingested_results()
        assert sorted(groups.tolist()) == sorted(want)
        for g, c, o in zip(groups, commit, out):
            assert (int(c), int(o)) == want[int(g)], (rnd, g)
    got = eng.read_state()
    ref = O.alloc_state(G, P)
    for k in ("cfg", "term_lo"):
        ref[k][...] = st[k]
    cl.store_soa(ref)
    assert not fuzz.diff_states(ref, got, G, P, keys=("match", "next", "pr_commit", "pflags", "commit"))
    eng.close()


def test_device_resident_records(rg):
    """rg_ingest_device: the same records from device memory give the same result as from host memory."""
    import torch
    rng = np.random.default_rng(31)
    G, P = 10000, 5
    engs = [rg.Engine(G, P) for _ in range(2)]
    for e in engs:
        e.workload_init(rg.WL_MAJORITY)
    st = engs[0].read_state()
    msgs = O.alloc_msgs(G, P)
    fuzz.random_msgs(rng, st, msgs)
    touched = np.sort(rng.choice(G, size=500, replace=False))
    recs = records_from_msgs(rg, msgs, touched, rng, P)
    assert engs[0].ingest(recs) == 0
    dev = torch.from_numpy(recs.view(np.uint8).copy()).cuda()
    engs[1].ingest_device(dev.data_ptr(), len(recs))
    engs[1].ingest_device(dev.data_ptr(), 3)  # the first three records again: duplicates
    engs[1].sync()
    assert engs[1].ingested_duplicates() == 3
    n0, n1 = engs[0].tick_ingested(), engs[1].tick_ingested()
    assert n0 == n1 > 0
    a, b = engs[0].read_state(), engs[1].read_state()
    for k in fuzz.STATE_KEYS + ("out",):
        assert (a[k] == b[k]).all(), k
    for e in engs:
        e.close()


def test_recompute_between_sparse_ticks_leaves_no_stale_results(rg):
    """rg_recompute rewrites every group's result word; the next sparse tick must clear all of them again."""
    G, P = 4096, 3
    eng = rg.Engine(G, P)
    st = O.alloc_state(G, P, stride=eng.stride)
    st["match"][0, :G], st["match"][1, :G], st["match"][2, :G] = 9, 9, 5
    st["next"][:, :G] = 10
    st["pflags"][:, :P] = rg.PF.REPLICATE
    st["commit"][:], st["term_lo"][:], st["term_hi"][:] = 5, 1, 9
    st["cfg"][:] = rg.cfg_make(0b111, self_slot=0)
    eng.load_state(st)
    from raft_rs_amd.engine import WIRE_DTYPE
    rec = np.zeros(1, dtype=WIRE_DTYPE)
    rec["group"], rec["slot"], rec["flags"], rec["index"], rec["commit"] = 7, 2, rg.MF.VALID, 6, 5
    assert eng.ingest(rec) == 0 and eng.tick_ingested() == 1
    eng.recompute()  # post_conf_change style: every group commits 9
    commit, out = eng.results()
    changed = (out & rg.OUT.CHANGED) != 0
    assert (commit == 9).all() and changed.sum() == G - 1 and not changed[7], "group 7 committed in the sparse tick"
    rec["group"], rec["index"] = 11, 7
    assert eng.ingest(rec) == 0 and eng.tick_ingested() == 1
    out = eng.read_column(rg.COL.OUT)
    assert out[11] != 0 and np.count_nonzero(out) == 1, "stale RG_OUT_CHANGED words of the recompute survived"
    eng.close()


def test_ingest_tick_is_one_round_trip_with_identical_results(rg):
    """rg_ingest_tick == rg_ingest + rg_tick_ingested (+ rg_ingested_results from the host copy): same state, same
    results, duplicates of the whole window reported, log-term rejects resolved on the way."""
    from raft_rs_amd.engine import WIRE_DTYPE
    rng = np.random.default_rng(911)
    G, P, TERM = 9000, 5, 6
    st = O.add_term_table(O.alloc_state(G, P))
    st["cfg"][:] = fuzz.random_cfg(rng, G, P)
    fuzz.random_state(rng, st, small_values=True)
    fuzz.random_term_table(rng, st, TERM)
    a, b = rg.Engine(G, P), rg.Engine(G, P)
    a.load_state(st)
    b.load_state(st)
    cl = O.Cluster(G)
    cl.load_soa(st, term=TERM)
    msgs = O.alloc_msgs(G, P)
    gout = np.zeros(G, dtype=np.uint32)
    for t in range(5):
        cl.store_soa(st)
        fuzz.random_msgs(rng, st, msgs, logterm_max=TERM)
        touched = np.sort(rng.choice(G, size=[3, 40, 900, 1, 2000][t], replace=False))
        keep = np.zeros(G, dtype=bool)
        keep[touched] = True
        msgs["m_flags"][~keep] = 0
        recs = []
        for g in touched:
            for p in range(P):
                f = int(msgs["m_flags"][g, p])
                if f:
                    recs.append((g, msgs["m_index"][p, g], msgs["m_commit"][p, g], msgs["m_hint"][p, g],
                                 msgs["m_rs"][p, g], msgs["m_logterm"][p, g], p, f, 0))
        recs = np.array(recs, dtype=WIRE_DTYPE)
        rng.shuffle(recs)
        dup_rec = recs[:1].copy() if t == 2 and len(recs) else recs[:0]
        n, dup = a.ingest_tick(np.concatenate([recs, dup_rec]))
        assert dup == len(dup_rec)
        assert b.ingest(recs) == 0
        assert b.tick_ingested() == n == int(msgs["m_flags"].any(axis=1).sum())
        cl.tick_soa(msgs, gout)
        ga, ca, oa = a.ingested_results()
        gb, cb, ob = b.ingested_results()
        ia, ib = np.argsort(ga), np.argsort(gb)
        assert (ga[ia] == gb[ib]).all() and (ca[ia] == cb[ib]).all() and (oa[ia] == ob[ib]).all()
        assert (oa[ia] == gout[ga[ia]]).all()
        sa, sb = a.read_state(), b.read_state()
        cl.store_soa(st)
        assert not fuzz.diff_states(st, sa, G, P) and not fuzz.diff_states(st, sb, G, P)
        assert (sa["out"] == gout).all() and (sb["out"] == gout).all()
    assert a.ingest_tick(np.zeros(0, dtype=WIRE_DTYPE)) == (0, 0)
    a.close()
    b.close()


def test_mailbox_serves_small_flushes_with_identical_results(rg):
    """rg_mailbox_start: small rg_ingest_tick batches are served by the resident workgroup (no launch per flush) with the
    results and state of the launch path and of the oracle -- across other entry points cutting in (the mailbox leaves and
    comes back), large batches (launch path), idle time-outs, log-term rejects, and rg_mailbox_stop."""
    import time
    from raft_rs_amd.engine import WIRE_DTYPE
    rng = np.random.default_rng(912)
    G, P, TERM = 9000, 5, 6
    st = O.add_term_table(O.alloc_state(G, P))
    st["cfg"][:] = fuzz.random_cfg(rng, G, P)
    fuzz.random_state(rng, st, small_values=True)
    fuzz.random_term_table(rng, st, TERM)
    a, b = rg.Engine(G, P), rg.Engine(G, P)
    a.load_state(st)
    b.load_state(st)
    a.mailbox_start(idle_timeout_us=300)
    cl = O.Cluster(G)
    cl.load_soa(st, term=TERM)
    msgs = O.alloc_msgs(G, P)
    gout = np.zeros(G, dtype=np.uint32)
    sizes = [3, 1, 40, 12, 900, 5, 7, 2, 30, 1, 2500, 4, 9]
    for t, k in enumerate(sizes):
        cl.store_soa(st)
        fuzz.random_msgs(rng, st, msgs, logterm_max=TERM if t % 3 == 0 else 0)
        touched = np.sort(rng.choice(G, size=k, replace=False))
        keep = np.zeros(G, dtype=bool)
        keep[touched] = True
        msgs["m_flags"][~keep] = 0
        recs = []
        for g in touched:
            for p in range(P):
                f = int(msgs["m_flags"][g, p])
                if f:
                    recs.append((g, msgs["m_index"][p, g], msgs["m_commit"][p, g], msgs["m_hint"][p, g],
                                 msgs["m_rs"][p, g], msgs["m_logterm"][p, g], p, f, 0))
        recs = np.array(recs, dtype=WIRE_DTYPE)
        rng.shuffle(recs)
        n, dup = a.ingest_tick(recs)
        nb, dupb = b.ingest_tick(recs)
        assert (n, dup) == (nb, dupb) == (int(msgs["m_flags"].any(axis=1).sum()), 0)
        cl.tick_soa(msgs, gout)
        ga, ca, oa = a.ingested_results()
        gb, cb, ob = b.ingested_results()
        ia, ib = np.argsort(ga), np.argsort(gb)
        assert (ga[ia] == gb[ib]).all() and (ca[ia] == cb[ib]).all() and (oa[ia] == ob[ib]).all(), t
        assert (oa[ia] == gout[ga[ia]]).all(), t
        if t in (3, 8):    # another entry point cuts in: the resident workgroup leaves, the next flush brings it back
            sa = a.read_state()
            cl.store_soa(st)
            assert not fuzz.diff_states(st, sa, G, P), t
            assert (sa["out"] == gout).all()
        if t == 5:         # idle for longer than the time-out: the workgroup has left by itself
            time.sleep(0.01)
        if t == 9:
            a.mailbox_stop()
            a.mailbox_start()
    sa, sb = a.read_state(), b.read_state()
    cl.store_soa(st)
    assert not fuzz.diff_states(st, sa, G, P) and not fuzz.diff_states(st, sb, G, P)
    served, launches = a.mailbox_stats()
    # every batch of <= 256 records that followed a sparse tick went through the mailbox (not the first one -- it follows
    # the dense load -- nor the ones right after another entry point rewrote the result words densely)
    assert served >= 6 and 3 <= launches <= served, (served, launches)
    a.mailbox_stop()
    n, dup = a.ingest_tick(recs[:0])
    assert (n, dup) == (0, 0)
    a.close()
    b.close()


def test_mailbox_with_the_host_mirror(rg):
    """rg_step ... rg_flush through the mailbox: what a latency-bound RawNode::step loop does."""
    G, P = 5000, 5
    a, b = rg.Engine(G, P), rg.Engine(G, P)
    for e in (a, b):
        e.workload_init(rg.WL_MAJORITY)
        for g in range(0, 64):
            e.set_peers(g, [1, 2, 3, 4, 5], 4)
    a.mailbox_start()
    st = a.read_state()
    for rep in range(12):
        for e in (a, b):
            for g in range(rep % 5, 64, 5):
                e.step(g, 2 + rep % 3, 4, int(min(st["term_hi"][g], st["match"][1 + rep % 3, g] + rep + 1)))
            e.flush()
        ga, ca, oa = a.ingested_results()
        gb, cb, ob = b.ingested_results()
        ia, ib = np.argsort(ga), np.argsort(gb)
        assert len(ga) and (ga[ia] == gb[ib]).all() and (ca[ia] == cb[ib]).all() and (oa[ia] == ob[ib]).all(), rep
    assert a.mailbox_stats()[0] >= 10
    sa, sb = a.read_state(), b.read_state()
    assert not fuzz.diff_states(sa, sb, G, P)
    a.close()
    b.close()
