"""GPU: random SEQUENCES of entry points against the oracle -- dense ticks, sparse ticks (three-call and one-call),
the RawNode::step mirror with dense and sparse flushes, rg_recompute, checkpoint/restore -- checking the state
columns, RG_COL_OUT and the compact results after every step. Catches host-side bookkeeping slips (which result
words are stale, which caches are valid) that single-path tests cannot see."""
import os

import numpy as np
import pytest

import fuzz
import oracle_lib as O

pytestmark = pytest.mark.gpu
TERM = 5


def records(msgs, groups, P, rng):
    from raft_rs_amd.engine import WIRE_DTYPE
    recs = []
    for g in groups:
        for p in range(P):
            f = int(msgs["m_flags"][g, p])
            if f:
                recs.append((g, msgs["m_index"][p, g], msgs["m_commit"][p, g], msgs["m_hint"][p, g],
                             msgs["m_rs"][p, g], 0, p, f, 0))
    arr = np.array(recs, dtype=WIRE_DTYPE)
    rng.shuffle(arr)
    return arr


def mirror_steps(rg, eng, msgs, groups, P, self_slot):
    """Feed one tick's events of `groups` through the message-at-a-time mirror."""
    MF = rg.MF
    for g in groups:
        g = int(g)
        for p in range(P):
            f = int(msgs["m_flags"][g, p])
            if not f:
                continue
            if p == self_slot[g]:
                if f & MF.APPEND:
                    eng.local_append(g, int(msgs["m_commit"][p, g]))
                if f & MF.VALID:
                    eng.local_persisted(g, int(msgs["m_index"][p, g]))
                continue
            if f & MF.SENT:
                eng.mark_sent(g, p + 1)
            if f & MF.HEARTBEAT:
                eng.step_heartbeat_response(g, p + 1, TERM, int(msgs["m_commit"][p, g]), bool(f & MF.INS_FULL))
            elif f & MF.VALID:
                eng.step(g, p + 1, TERM, int(msgs["m_index"][p, g]), commit=int(msgs["m_commit"][p, g]),
                         reject=bool(f & MF.REJECT), reject_hint=int(msgs["m_hint"][p, g]),
                         request_snapshot=int(msgs["m_rs"][p, g]) if f & MF.HAS_RS else 0, ins_full=bool(f & MF.INS_FULL))


def clean_for_mirror(msgs, P, self_slot):
    """The mirror has no way to express meaningless combinations the raw columns allow: keep what it can say."""
    f = msgs["m_flags"]
    G = f.shape[0]
    for p in range(P):
        col = f[:, p]
        is_self = self_slot == p
        col[is_self] &= 0x21  # VALID | APPEND on the leader's own slot
        hb = (col & 0x40) != 0
        col[hb & ~is_self] &= 0x40 | 0x10 | 0x08  # HEARTBEAT (+SENT, INS_FULL)
        rest = ~hb & ~is_self
        only_mod = rest & ((col & 0x01) == 0)
        col[only_mod] &= 0x10  # without VALID only SENT means anything
        f[:, p] = col
    # request_snapshot value 0 with HAS_RS cannot be expressed either
    for p in range(P):
        has = (f[:, p] & 0x04) != 0
        zero = msgs["m_rs"][p, :G] == 0
        f[has & zero, p] &= np.uint8(~0x04 & 0xff)


def local_messages(rg, eng, cl, rng, G, P, mirror_ok=True):
    """RawNode::report_unreachable / report_snapshot between two steps of a sequence, through one of the three forms
    (records, one byte per group, the mirror by peer id), to the engine and to the oracle alike."""
    E = rg.engine
    form = rng.choice(["records", "dense", "mirror"] if mirror_ok else ["records", "dense"])
    if form == "records":
        events = fuzz.random_progress_events(rng, G, P, int(rng.integers(1, 60)))
        eng.progress_events(events)
    elif form == "dense":
        kind = int(rng.integers(1, 4))
        slot1 = np.where(rng.random(G) < 0.1, rng.integers(1, P + 1, size=G), 0).astype(np.uint8)
        eng.progress_event_dense(kind, slot1)
        events = [(int(g), int(slot1[g]) - 1, kind) for g in np.nonzero(slot1)[0]]
    else:
        events = []
        for _ in range(int(rng.integers(1, 12))):
            g, s, kind = int(rng.integers(0, G)), int(rng.integers(0, P)), int(rng.integers(1, 4))
            if kind == E.EV_UNREACHABLE:
                eng.report_unreachable(g, s + 1)
            else:
                eng.report_snapshot(g, s + 1, kind == E.EV_SNAPSHOT_FAILURE)
            events.append((g, s, kind))
    for g, s, kind in events:
        if g >= G or s >= P:
            continue
        if kind == 1:
            cl.L.ro_handle_unreachable(cl.h, g, s + 1)
        else:
            cl.L.ro_handle_snapshot_status(cl.h, g, s + 1, kind == 3)
    return form


# (RG_SOAK_SEEDS=n adds n more seeded cases here as well, P = 1..8 by the seed: an ad hoc soak, see below)
@pytest.mark.parametrize("seed,P", [(1, 3), (2, 5), (3, 7)] + [(200 + i, 1 + i % 8) for i in range(int(os.environ.get("RG_SOAK_SEEDS", "0")))])
def test_random_api_sequences_match_the_oracle(rg, seed, P):
    rng = np.random.default_rng(4200 + seed)
    G = 1500
    st = O.alloc_state(G, P)
    st["cfg"][:] = fuzz.random_cfg(rng, G, P)
    fuzz.random_state(rng, st, small_values=True)
    self_slot = ((st["cfg"] >> 16) & 7).astype(np.int64)
    eng = rg.Engine(G, P)
    eng.load_state(st)
    for g in range(G):
        eng.set_peers(g, list(range(1, P + 1)), TERM)
    cl = O.Cluster(G)
    cl.load_soa(st, term=TERM)
    msgs = O.alloc_msgs(G, P)
    del msgs["m_logterm"]
    mb = rg.MsgBuffers(G, P, eng.stride)
    gout = np.zeros(G, dtype=np.uint32)
    ckpt = None
    ops_seen = set()
    for step in range(70):
        cl.store_soa(st)
        op = rng.choice(["dense", "sparse3", "sparse1", "mirror_sparse", "mirror_dense", "recompute", "checkpoint",
                         "restore"], p=[0.15, 0.17, 0.17, 0.17, 0.1, 0.08, 0.08, 0.08])
        if op == "restore" and ckpt is None:
            op = "checkpoint"
        ops_seen.add(op)
        touched = None
        if op == "checkpoint":
            eng.checkpoint()
            ckpt = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in st.items()}
            continue
        if op == "restore":
            eng.restore()
            st = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in ckpt.items()}
            cl = O.Cluster(G)
            cl.load_soa(st, term=TERM)
            got = eng.read_state()
            assert not fuzz.diff_states(st, got, G, P), (step, op)
            continue
        if rng.random() < 0.3:  # local messages between the steps: MsgUnreachable / MsgSnapStatus
            ops_seen.add("local:" + local_messages(rg, eng, cl, rng, G, P))
            cl.store_soa(st)  # (the tick's messages are generated for the state the events left)
        if op == "recompute":
            eng.recompute()
            for g in range(G):
                gout[g] = 1 if cl.maybe_commit(g) else 0
        else:
            fuzz.random_msgs(rng, st, msgs)
            if op in ("sparse3", "sparse1", "mirror_sparse"):
                touched = np.sort(rng.choice(G, size=int(rng.integers(1, G // 3)), replace=False))
                keep = np.zeros(G, dtype=bool)
                keep[touched] = True
                msgs["m_flags"][~keep] = 0
            if op.startswith("mirror"):
                clean_for_mirror(msgs, P, self_slot)
            if op == "dense":
                for k in ("m_index", "m_commit", "m_hint", "m_rs", "m_flags"):
                    getattr(mb, k)[...] = msgs[k]
                eng.tick(mb)
            elif op == "sparse3":
                assert eng.ingest(records(msgs, touched, P, rng)) == 0
                eng.tick_ingested()
            elif op == "sparse1":
                n, dup = eng.ingest_tick(records(msgs, touched, P, rng))
                assert dup == 0
            else:
                mirror_steps(rg, eng, msgs, touched if touched is not None else range(G), P, self_slot)
                eng.flush()
            gout[:] = 0
            cl.tick_soa(msgs, gout)
        got = eng.read_state()
        cl.store_soa(st)
        diffs = fuzz.diff_states(st, got, G, P)
        assert not diffs, (step, op, diffs[:5])
        assert (got["out"] == gout).all(), (step, op, np.nonzero(got["out"] != gout)[0][:5])
        commit, out = eng.results()
        assert (commit == st["commit"]).all() and (out == gout).all(), (step, op)
        if op in ("sparse3", "sparse1", "mirror_sparse", "mirror_dense"):
            with_events = np.nonzero(msgs["m_flags"].any(axis=1))[0]
            groups, c2, o2 = eng.ingested_results()
            order = np.argsort(groups)
            assert (groups[order] == with_events).all(), (step, op)
            assert (c2[order] == st["commit"][with_events]).all() and (o2[order] == gout[with_events]).all(), (step, op)
    assert len(ops_seen) >= 9 and any(o.startswith("local:") for o in ops_seen), ops_seen
    eng.close()


# RG_SOAK_SEEDS=n adds n more seeded cases (P, window depth and mailbox use derived from the seed): an ad hoc soak
_SEND_CASES = [(11, 3, 2, False), (12, 5, 4, False), (13, 5, 3, True)] + [
    (100 + i, 2 + i % 7, 1 + (i * 5) % 9, i % 2 == 0) for i in range(int(os.environ.get("RG_SOAK_SEEDS", "0")))]


@pytest.mark.parametrize("seed,P,cap,mailbox", _SEND_CASES)
def test_random_api_sequences_with_the_send_stage(rg, seed, P, cap, mailbox):
    """The same idea with the Inflights on the device: after every kind of tick the send stage (separately or inside
    rg_flush_send) must produce the oracle's send decisions, Progress columns and window contents."""
    import sendstage
    from test_sendstage_gpu import apply_snapshots
    rng = np.random.default_rng(4300 + seed)
    G = 1200
    st = O.add_term_table(O.alloc_state(G, P))
    st["cfg"][:] = fuzz.random_cfg(rng, G, P)
    fuzz.random_state(rng, st, small_values=True)
    fuzz.random_term_table(rng, st, TERM)
    sendstage.mark_pending_conf(rng, st)
    self_slot = ((st["cfg"] >> 16) & 7).astype(np.int64)
    eng = rg.Engine(G, P, max_inflight=cap)
    eng.load_state(st)
    for g in range(G):
        eng.set_peers(g, list(range(1, P + 1)), TERM)
    if mailbox:  # small rg_flush_send batches go through the resident workgroup; everything else makes it step aside
        eng.mailbox_start()
    cl = O.Cluster(G)
    cl.load_soa(st, term=TERM, max_inflight=cap)
    cl.set_own_inflights(True)
    msgs = O.alloc_msgs(G, P)
    msgs["m_logterm"][...] = 0
    mb = rg.MsgBuffers(G, P, eng.stride)
    gout = np.zeros(G, dtype=np.uint32)
    n_items = 0
    ops_seen = set()
    for step in range(70):
        cl.store_soa(st)
        op = rng.choice(["dense", "dense_send", "sparse3", "sparse1", "mirror_sparse", "mirror_flush_send", "mirror_dense",
                         "mirror_small_flush_send", "recompute"])
        ops_seen.add(op)
        max_entries, skip = int(rng.integers(0, 4)), bool(rng.integers(0, 2))
        staged = False
        touched = None
        if rng.random() < 0.3:  # local messages between the steps (a state change resets the device window)
            ops_seen.add("local:" + local_messages(rg, eng, cl, rng, G, P))
            cl.store_soa(st)
        if op == "recompute":
            eng.recompute()
            for g in range(G):
                gout[g] = 1 if cl.maybe_commit(g) else 0
        else:
            fuzz.random_msgs(rng, st, msgs, sent_p=0.0, heartbeat_p=0.2)
            sendstage.prepare_msgs(msgs)
            if op not in ("dense", "dense_send", "mirror_dense"):
                # (mirror_small_flush_send: few enough records for the ONE-launch flush with the stage inside, k_flush_small_send)
                hi = 40 if op == "mirror_small_flush_send" else G // 3
                touched = np.sort(rng.choice(G, size=int(rng.integers(1, hi)), replace=False))
                keep = np.zeros(G, dtype=bool)
                keep[touched] = True
                msgs["m_flags"][~keep] = 0
            if op.startswith("mirror"):
                clean_for_mirror(msgs, P, self_slot)
            if op in ("dense", "dense_send"):
                for k in ("m_index", "m_commit", "m_hint", "m_rs", "m_flags"):
                    getattr(mb, k)[...] = msgs[k]
                if op == "dense_send":  # the tick and its stage as ONE launch (rg_tick_send)
                    eng.tick_send(mb, max_entries, skip_bcast_commit=skip)
                    staged = True
                else:
                    eng.tick(mb)
            elif op == "sparse3":
                assert eng.ingest(records(msgs, touched, P, rng)) == 0
                eng.tick_ingested()
            elif op == "sparse1":
                assert eng.ingest_tick(records(msgs, touched, P, rng))[1] == 0
            else:
                mirror_steps(rg, eng, msgs, touched if touched is not None else range(G), P, self_slot)
                if op in ("mirror_flush_send", "mirror_small_flush_send") or (op == "mirror_dense" and rng.random() < 0.5):
                    eng.flush_send(max_entries, skip_bcast_commit=skip)
                    staged = True
                else:
                    eng.flush()
            gout[:] = 0
            cl.tick_soa(msgs, gout)
        if not staged:
            eng.send_appends(max_entries, skip_bcast_commit=skip)
        items = sendstage.compare_items(eng.send_items(), cl.send_stage_soa(gout, max_entries, skip_bcast_commit=skip))
        n_items += len(items)
        apply_snapshots(rg, eng, cl, st, items)
        got = eng.read_state()
        cl.store_soa(st)
        diffs = fuzz.diff_states(st, got, G, P)
        assert not diffs, (step, op, diffs[:5])
        meta, ring = eng.read_inflights()
        sendstage.compare_rings(cl, meta, ring, st, cap)
    assert n_items > 2000 and {"dense_send", "mirror_small_flush_send"} <= ops_seen, (n_items, ops_seen)
    assert any(o.startswith("local:") for o in ops_seen), ops_seen
    if mailbox:
        assert eng.mailbox_stats()[0] > 0, "no flush was served by the resident workgroup"
    eng.close()
