"""GPU: the send stage -- Inflights rings in HBM and the maybe_send_append decision
(k_send_appends) -- against the oracle's message-at-a-time restatement of src/raft.rs:773-819 /
src/tracker/inflights.rs: identical work items, Progress columns and ring contents after every tick."""
import numpy as np
import pytest

import fuzz
import hosthints
import oracle_lib as O
import sendstage

pytestmark = pytest.mark.gpu


def apply_snapshots(rg, eng, cl, st, items):
    """The host's half of a snapshot send on both sides: Progress::become_snapshot(index) -- on the engine through
    rg_write_cells, whose state change must also reset the device ring."""
    cells = []
    flags = None
    for (g, p), (kind, prev, last, n) in items.items():
        if kind != O.SEND_SNAPSHOT:
            continue
        if flags is None:
            flags = eng.read_column(rg.COL.PFLAGS)
        sidx = int(st["commit"][g])
        import ctypes as C
        cl.L.ro_progress_become_snapshot(C.byref(cl.pr(g, p + 1)), sidx)
        # INS_FULL is engine-owned: rg_write_cells must ignore whatever the caller passes for it
        cells.append({"group": g, "slot": p, "pend_snap": sidx,
                      "pflags": (int(flags[g, p]) & ~0x7) | O.SNAPSHOT | 0x10})
    if cells:
        eng.write_cells(cells)
    return len(cells)


def check(rg, eng, cl, st, cap, what):
    got = eng.read_state()
    cl.store_soa(st)
    diffs = fuzz.diff_states(st, got, st["n_groups"], st["n_slots"])
    assert not diffs, f"{what}: state differs from the oracle:\n" + "\n".join(diffs[:10])
    meta, ring = eng.read_inflights()
    sendstage.compare_rings(cl, meta, ring, st, cap)


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("n_slots,cap,max_entries", [(3, 2, 1), (5, 3, 2), (5, 256, 0), (7, 4, 0), (8, 1, 3), (1, 2, 0), (2, 3, 1),
                                                    (4, 2, 2), (6, 5, 0)])
def test_send_stage_matches_oracle(rg, n_slots, cap, max_entries, fused):
    """fused: rg_tick_send -- the tick and its stage as ONE launch (k_tick_send) -- must leave exactly what rg_tick followed
    by rg_send_appends leaves: result words, work items, Progress columns, rings."""
    rng = np.random.default_rng(7700 + 31 * n_slots + cap)
    G = 6000 + 13
    st = O.add_term_table(O.alloc_state(G, n_slots))
    st["cfg"][:] = fuzz.random_cfg(rng, G, n_slots, missing_progress_frac=0.05)
    fuzz.random_state(rng, st, small_values=True)
    fuzz.random_term_table(rng, st, term=6)
    sendstage.mark_pending_conf(rng, st)
    eng = rg.Engine(G, n_slots, max_inflight=cap)
    eng.load_state(st)
    cl = O.Cluster(G)
    cl.load_soa(st, term=6, max_inflight=cap)
    cl.set_own_inflights(True)
    msgs = O.alloc_msgs(G, n_slots)
    mb = rg.MsgBuffers(G, n_slots, eng.stride)
    gout = np.zeros(G, dtype=np.uint32)
    seen = {"items": 0, "multi": 0, "snap": 0}
    for t in range(8):
        cl.store_soa(st)
        fuzz.random_msgs(rng, st, msgs, sent_p=0.0, heartbeat_p=0.2)
        sendstage.prepare_msgs(msgs)
        for k in ("m_index", "m_commit", "m_hint", "m_rs", "m_flags"):
            getattr(mb, k)[...] = msgs[k]
        skip = t % 2 == 1  # Config::skip_bcast_commit on every other tick
        if fused:
            eng.tick_send(mb, max_entries, skip_bcast_commit=skip)
        else:
            eng.tick(mb)
        cl.tick_soa(msgs, gout)
        _, out = eng.results()
        assert (out == gout).all(), (t, np.nonzero(out != gout)[0][:5])
        if not fused:
            eng.send_appends(max_entries, skip_bcast_commit=skip)
        items = sendstage.compare_items(eng.send_items(), cl.send_stage_soa(gout, max_entries, skip_bcast_commit=skip))
        seen["snap"] += apply_snapshots(rg, eng, cl, st, items)
        check(rg, eng, cl, st, cap, f"P={n_slots} cap={cap} tick {t}")
        seen["items"] += len(items)
        seen["multi"] += sum(1 for v in items.values() if v[3] > 1)
    if n_slots > 1:  # (a single-voter group has nobody to send to)
        assert seen["items"] > 1000 and seen["snap"] > 0, seen
    if max_entries and cap > 1:
        assert seen["multi"] > 0, seen
    eng.close()


@pytest.mark.parametrize("n_slots,cap,max_entries", [(3, 2, 1), (5, 256, 0), (7, 4, 0)])
def test_send_stage_with_everything_streamed(rg, n_slots, cap, max_entries):
    """rg_config.cache_policy = RG_CACHE_STREAM_ALL on an engine with device Inflights: the one-launch form runs
    k_tick_send<.., NTS = true> (the tick's state columns streamed, loads and stores), the two-launch form the streamed lane kernel
    in front of k_send_dense -- same work items, columns and windows as ever."""
    from raft_rs_amd import engine as E
    with E.config_defaults(cache_policy=E.CACHE.STREAM_ALL):
        eng = rg.Engine(64, n_slots, max_inflight=cap)
        assert eng.device_info()["cache_policy"] == "stream_all"
        eng.close()
        test_send_stage_matches_oracle(rg, n_slots, cap, max_entries, True)
        test_send_stage_matches_oracle(rg, n_slots, cap, max_entries, False)
    with pytest.raises(rg.EngineError):
        rg.Engine(64, n_slots, max_inflight=cap, cache_policy=E.CACHE.RESIDENT)


@pytest.mark.parametrize("form", ["resolve_before_stage", "resolve_after_stage", "one_launch"])
@pytest.mark.parametrize("n_slots,cap,max_entries", [(3, 2, 1), (5, 4, 0), (7, 3, 2)])
def test_send_stage_waits_for_host_hints(rg, n_slots, cap, max_entries, form):
    """Device Inflights and a log deeper than the term-run table: a reject the tick leaves to the host (RG_OUT_HOST_HINT) is
    processed by the reference BEFORE the group's other sends of the tick (raft.rs:1719 inside the step, the sends after it), so
    the group's whole send stage waits for rg_resolve_host_hints -- called before rg_send_appends, after it, or after the
    one-launch form rg_tick_send, whose stage skips the group and which the resolve call then completes. Work items, Progress
    columns and every ring against the oracle after every tick."""
    rng = np.random.default_rng(8800 + 31 * n_slots + cap)
    G, TERM = 4000, 30
    st = O.add_term_table(O.alloc_state(G, n_slots))
    st["cfg"][:] = fuzz.random_cfg(rng, G, n_slots, missing_progress_frac=0.03)
    fuzz.random_state(rng, st, probe_frac=0.5, base=200)
    fuzz.random_term_table(rng, st, TERM, min_runs=O.TERM_RUNS)
    sendstage.mark_pending_conf(rng, st)
    eng = rg.Engine(G, n_slots, max_inflight=cap)
    eng.load_state(st)
    cl = O.Cluster(G)
    cl.load_soa(st, term=TERM, max_inflight=cap)
    cl.set_own_inflights(True)
    msgs = O.alloc_msgs(G, n_slots)
    mb = rg.MsgBuffers(G, n_slots, eng.stride)
    gout = np.zeros(G, dtype=np.uint32)
    seen = {"items": 0, "settled": 0, "late_items": 0}
    for t in range(8):
        cl.store_soa(st)
        fuzz.random_msgs(rng, st, msgs, valid_p=0.8, reject_p=0.5, rs_p=0.05, sent_p=0.0, heartbeat_p=0.1, logterm_max=TERM + t,
                         elect_p=0.5, elect_term=TERM + 1 + t)
        hosthints.spread_reject_hints(rng, st, msgs, TERM + 1 + t)
        sendstage.prepare_msgs(msgs)
        for k in ("m_index", "m_commit", "m_hint", "m_rs", "m_logterm", "m_flags"):
            getattr(mb, k)[...] = msgs[k]
        skip = t % 2 == 1
        if form == "one_launch":
            eng.tick_send(mb, max_entries, skip_bcast_commit=skip)
        else:
            eng.tick(mb)
        cl.tick_soa(msgs, gout)
        out0 = eng.read_column(rg.COL.OUT)
        hinted = set(np.nonzero(out0 & hosthints.OUT_HOST_HINT)[0].tolist())
        if form == "resolve_after_stage":
            eng.send_appends(max_entries, skip_bcast_commit=skip)
        if form != "resolve_before_stage":
            early = eng.send_items()
            assert not hinted & set(int(g) for g in early["group"]), "the stage served a group that waits for its host hint"
        seen["settled"] += hosthints.settle_engine(eng, cl, msgs)
        _, out = eng.results()
        assert (out == gout).all(), (t, np.nonzero(out != gout)[0][:5])
        if form == "resolve_before_stage":
            eng.send_appends(max_entries, skip_bcast_commit=skip)
        got = eng.send_items()
        seen["late_items"] += sum(1 for g in got["group"] if int(g) in hinted)
        items = sendstage.compare_items(got, cl.send_stage_soa(gout, max_entries, skip_bcast_commit=skip))
        apply_snapshots(rg, eng, cl, st, items)
        check(rg, eng, cl, st, cap, f"{form} P={n_slots} cap={cap} tick {t}")
        seen["items"] += len(items)
    assert seen["settled"] > 100 and seen["late_items"] > 100 and seen["items"] > 1000, seen
    eng.close()


def test_unanswered_host_hints_refuse_the_next_step(rg):
    """Device Inflights: until EVERY reject a tick left to the host (RG_OUT_HOST_HINT) has been answered, nothing that starts the
    next step is accepted (RG_ERR_STATE, device state untouched) -- the group's send requests are served by
    rg_resolve_host_hints alone, so a host that moved on would leave `next` and the windows behind the reference's. The
    Inflights effects of the tick are applied by the stage either way; a group's bit falls with the LAST of its slots: answering
    one slot of a two-slot group releases nothing, answering a slot twice changes nothing."""
    from raft_rs_amd.engine import EngineError, ERR, COL
    rng = np.random.default_rng(9911)
    G, P, cap, TERM = 3000, 5, 4, 30
    st = O.add_term_table(O.alloc_state(G, P))
    st["cfg"][:] = fuzz.random_cfg(rng, G, P, missing_progress_frac=0.03)
    fuzz.random_state(rng, st, probe_frac=0.5, base=200)
    fuzz.random_term_table(rng, st, TERM, min_runs=O.TERM_RUNS)
    eng = rg.Engine(G, P, max_inflight=cap)
    eng.load_state(st)
    cl = O.Cluster(G)
    cl.load_soa(st, term=TERM, max_inflight=cap)
    cl.set_own_inflights(True)
    msgs = O.alloc_msgs(G, P)
    mb = rg.MsgBuffers(G, P, eng.stride)
    gout = np.zeros(G, dtype=np.uint32)
    seen = {"refused": 0, "partial_groups": 0, "settled": 0}
    for t in range(6):
        cl.store_soa(st)
        fuzz.random_msgs(rng, st, msgs, valid_p=0.8, reject_p=0.6, rs_p=0.05, sent_p=0.0, heartbeat_p=0.1, logterm_max=TERM + t,
                         elect_p=0.5, elect_term=TERM + 1 + t)
        hosthints.spread_reject_hints(rng, st, msgs, TERM + 1 + t)
        sendstage.prepare_msgs(msgs)
        for k in ("m_index", "m_commit", "m_hint", "m_rs", "m_logterm", "m_flags"):
            getattr(mb, k)[...] = msgs[k]
        if t % 2:
            eng.tick_send(mb, 0)
        else:
            eng.tick(mb)
            eng.send_appends(0)
        cl.tick_soa(msgs, gout)
        out0 = eng.read_column(COL.OUT)
        hh = eng.read_column(COL.HOST_HINT)
        hinted = np.nonzero(out0 & hosthints.OUT_HOST_HINT)[0]
        if hinted.size:
            before = eng.read_state()
            meta0, ring0 = eng.read_inflights()
            for call in (lambda: eng.tick(mb), lambda: eng.tick_send(mb, 0),
                         lambda: eng.progress_events([(int(hinted[0]), 1, 1)]), lambda: eng.recompute()):
                with pytest.raises(EngineError) as e:
                    call()
                assert e.value.code == ERR["STATE"] and "rg_resolve_host_hints" in str(e.value)
                seen["refused"] += 1
            assert not fuzz.diff_states(before, eng.read_state(), G, P) and np.array_equal(before["out"], eng.read_column(COL.OUT))
            meta1, ring1 = eng.read_inflights()
            assert np.array_equal(meta0, meta1) and np.array_equal(ring0, ring1)

        def resolve(recs):
            first, rest, have = [], [], set()
            for r in recs:  # one slot of every group now, the others afterwards
                (rest if r[0] in have else first).append(r)
                have.add(r[0])
            assert eng.resolve_host_hints(first).all()
            if rest:
                waiting = {r[0] for r in rest}
                seen["partial_groups"] += len(waiting)
                out1 = eng.read_column(COL.OUT)
                for g in have:
                    assert bool(out1[g] & hosthints.OUT_HOST_HINT) == (g in waiting), g
                assert not eng.resolve_host_hints(first).any(), "answered slots do not wait any more"
                with pytest.raises(EngineError) as e:
                    eng.tick(mb)
                assert e.value.code == ERR["STATE"]
                assert {int(r["group"]) for r in eng.host_hints()} == waiting
                assert eng.resolve_host_hints(rest).all()
            return eng.read_column(COL.OUT)

        merged, n = hosthints.settle(cl, msgs, out0, hh, resolve=resolve)
        seen["settled"] += n
        assert (merged == gout).all(), (t, np.nonzero(merged != gout)[0][:5])
        items = sendstage.compare_items(eng.send_items(), cl.send_stage_soa(gout, 0))
        apply_snapshots(rg, eng, cl, st, items)
        check(rg, eng, cl, st, cap, f"tick {t}")
    assert seen["refused"] >= 12 and seen["partial_groups"] > 5 and seen["settled"] > 50, seen
    eng.close()


def test_send_stage_after_sparse_ticks(rg):
    """rg_ingest -> rg_tick_ingested -> rg_send_appends: the stage walks the touched groups only."""
    from test_sparse_path_gpu import records_from_msgs
    rng = np.random.default_rng(7801)
    G, P, cap = 20000, 5, 3
    st = O.add_term_table(O.alloc_state(G, P))
    st["cfg"][:] = fuzz.random_cfg(rng, G, P)
    fuzz.random_state(rng, st, small_values=True)
    fuzz.random_term_table(rng, st, term=6)
    eng = rg.Engine(G, P, max_inflight=cap)
    eng.load_state(st)
    cl = O.Cluster(G)
    cl.load_soa(st, term=6, max_inflight=cap)
    cl.set_own_inflights(True)
    msgs = O.alloc_msgs(G, P)
    gout = np.zeros(G, dtype=np.uint32)
    total = 0
    for t in range(5):
        cl.store_soa(st)
        fuzz.random_msgs(rng, st, msgs, sent_p=0.0, heartbeat_p=0.2)
        sendstage.prepare_msgs(msgs)
        touched = np.sort(rng.choice(G, size=G // 25, replace=False))
        keep = np.zeros(G, dtype=bool)
        keep[touched] = True
        msgs["m_flags"][~keep] = 0
        assert eng.ingest(records_from_msgs(rg, msgs, touched, rng, P)) == 0
        eng.tick_ingested()
        gout[:] = 0
        cl.tick_soa(msgs, gout)
        eng.send_appends(2)
        items = sendstage.compare_items(eng.send_items(), cl.send_stage_soa(gout, 2))
        assert all(keep[g] for g, _ in items)
        apply_snapshots(rg, eng, cl, st, items)
        check(rg, eng, cl, st, cap, f"sparse tick {t}")
        total += len(items)
    assert total > 500
    eng.close()


def test_send_stage_call_sequence_and_checkpoint(rg):
    from raft_rs_amd.engine import EngineError, ERR
    G, P, cap = 300, 3, 2
    host = rg.Engine(G, P)  # Inflights stay with the host
    with pytest.raises(EngineError) as e:
        host.send_appends()
    assert e.value.code == ERR["STATE"]
    with pytest.raises(EngineError) as e:
        host.tick_send(rg.MsgBuffers(G, P, host.stride))  # the one-launch form needs the device Inflights as well
    assert e.value.code == ERR["STATE"]
    host.close()
    eng = rg.Engine(G, P, max_inflight=cap)
    eng.workload_init(2)
    with pytest.raises(EngineError) as e:
        eng.tick_send(rg.MsgBuffers(G, P, eng.stride), max_bytes=100)  # RG_SEND_BYTES without the entry sizes
    assert e.value.code == ERR["STATE"]
    with pytest.raises(EngineError) as e:
        eng.send_appends()  # no tick yet
    assert e.value.code == ERR["STATE"]
    mb = rg.MsgBuffers(G, P, eng.stride)
    hi = eng.read_column(rg.COL.TERM_HI)
    mb.m_commit[0, :G] = hi + 3  # every leader (slot 0) proposes 3 entries: bcast_append = ONE message per peer
    mb.m_flags[:, 0] = rg.MF.APPEND
    eng.tick(mb)
    eng.send_appends(1)
    items = eng.send_items()
    assert len(items) == G * (P - 1) and (items["n_msgs"] == 1).all()
    assert (items["last_index"] == items["prev_index"] + 1).all(), "max_entries_per_msg = 1"
    assert not (eng.read_column(rg.COL.PFLAGS)[:, :P] & rg.PF.INS_FULL).any()
    mb.m_commit[0, :G] = hi + 4  # a second proposal fills the window of 2
    eng.tick(mb)
    eng.checkpoint()
    eng.send_appends(1)
    assert len(eng.send_items()) == G * (P - 1)
    meta1, ring1 = eng.read_inflights()
    assert (eng.read_column(rg.COL.PFLAGS)[:, 1:P] & rg.PF.INS_FULL).all()
    with pytest.raises(EngineError) as e:
        eng.send_appends(1)  # the stage already ran for this tick
    assert e.value.code == ERR["STATE"]
    mb.m_commit[0, :G] = hi + 5  # full windows: a proposal sends nothing
    eng.tick(mb)
    eng.send_appends(1)
    assert len(eng.send_items()) == 0
    mb.m_commit[0, :G] = hi + 6
    eng.tick_send(mb, 1)  # a tick and its stage in one launch: the stage has run when the call returns
    assert len(eng.send_items()) == 0
    with pytest.raises(EngineError) as e:
        eng.send_appends(1)
    assert e.value.code == ERR["STATE"]
    eng.restore()  # back to "second tick done, its stage not run": rings and flags are part of the checkpoint
    meta0, _ = eng.read_inflights()
    assert ((meta0[1:P, :G] >> 16) == 1).all() and (meta0[0, :G] == 0).all()
    assert not (eng.read_column(rg.COL.PFLAGS)[:, :P] & rg.PF.INS_FULL).any()
    eng.send_appends(1)  # replays identically
    m2, r2 = eng.read_inflights()
    assert (m2 == meta1).all() and (r2 == ring1).all()
    eng.load_inflights(meta0, ring1)
    m3, _ = eng.read_inflights()
    assert (m3 == meta0).all()
    import torch
    with pytest.raises(EngineError) as e:  # fused launches cannot interleave the send stage
        d = torch.zeros(8, dtype=torch.int64, device="cuda")
        eng.tick_device_fused([(d.data_ptr(),) * 5], d.data_ptr())
    assert e.value.code == ERR["STATE"]
    eng.close()


@pytest.mark.parametrize("one_call", [False, True, "mailbox"])
def test_mirror_steps_with_device_inflights(rg, one_call):
    """RawNode::step mirror (rg_step / rg_local_append / rg_flush) followed by the send stage: proposals fill the
    window of every follower, an ack moves it forward and the backlog goes out in one MsgAppend. one_call: the
    same through rg_flush_send (tick + stage + results + items in one round trip on the sparse path)."""
    G, P, cap = 64, 3, 4
    eng = rg.Engine(G, P, max_inflight=cap)
    st = O.alloc_state(G, P, stride=eng.stride)
    st["match"][:, :G], st["next"][:, :G] = 10, 11
    st["pr_commit"][:, :G] = 10
    st["pflags"][:, :P] = rg.PF.REPLICATE | rg.PF.RECENT_ACTIVE
    st["commit"][:], st["term_lo"][:], st["term_hi"][:] = 10, 1, 10
    st["cfg"][:] = rg.cfg_make(0b111, self_slot=0)
    eng.load_state(st)
    for g in range(G):
        eng.set_peers(g, [1, 2, 3], term=5)
    if one_call == "mailbox":  # small rg_flush_send batches are answered by the resident workgroup, stage included
        eng.mailbox_start()
    for r in range(1, 7):  # six proposals of one entry each
        for g in range(G):
            eng.local_append(g, 10 + r)
        if one_call:
            eng.flush_send()  # every group is dirty: the dense path
        else:
            eng.flush()
            eng.send_appends()
        items = eng.send_items()
        if r <= cap:
            assert len(items) == G * 2 and (items["prev_index"] == 9 + r).all() and (items["last_index"] == 10 + r).all()
        else:
            assert len(items) == 0, "full windows: Progress::is_paused"
    assert eng.inflights(3, 1) == [11, 12, 13, 14] and eng.inflights(3, 2) == [11, 12, 13, 14]
    assert (eng.read_column(rg.COL.PFLAGS)[:, 1:3] & rg.PF.INS_FULL).all()
    for g in range(0, G, 2):  # peer 2 of every other group acks index 12
        eng.step(g, from_=2, term=5, index=12)
    if one_call:
        eng.flush_send()  # < 50 % of the groups: the sparse path, one round trip
        with pytest.raises(rg.EngineError):
            eng.send_appends()  # the stage of this tick has already run
    else:
        eng.flush()
        eng.send_appends()
    items = eng.send_items()
    groups, commit, out = eng.ingested_results()
    assert sorted(groups.tolist()) == list(range(0, G, 2)) and (commit == 10).all()
    assert len(items) == G // 2 and (items["slot"] == 1).all() and (items["group"] % 2 == 0).all()
    assert (items["prev_index"] == 14).all() and (items["last_index"] == 16).all() and (items["n_msgs"] == 1).all()
    meta, ring = eng.read_inflights()
    assert eng.inflights(4, 1, meta, ring) == [13, 14, 16] and eng.inflights(5, 1, meta, ring) == [11, 12, 13, 14]
    nxt = eng.read_column(rg.COL.NEXT)
    assert nxt[1, 4] == 17 and nxt[1, 5] == 15 and nxt[2, 4] == 15
    if one_call:  # a flush with nothing queued: no groups, no items
        eng.flush_send()
        assert len(eng.send_items()) == 0 and len(eng.ingested_results()[0]) == 0
    if one_call == "mailbox":
        # one more round through the resident workgroup: peer 3 of a few groups acks 14 -> its window empties, 15..16 go out
        before = eng.mailbox_stats()[0]
        for g in (1, 3, 5):
            eng.step(g, from_=3, term=5, index=14)
        eng.flush_send()
        assert eng.mailbox_stats()[0] == before + 1, "the flush must have been served by the mailbox"
        items = eng.send_items()
        assert sorted(items["group"].tolist()) == [1, 3, 5] and (items["slot"] == 2).all()
        assert (items["prev_index"] == 14).all() and (items["last_index"] == 16).all()
        assert eng.inflights(1, 2) == [16] and int(eng.read_column(rg.COL.NEXT)[2, 1]) == 17
        eng.mailbox_stop()
    eng.close()


@pytest.mark.parametrize("mailbox", [False, True])
def test_small_flush_send_with_the_byte_limit(rg, mailbox):
    """rg_flush_send(RG_SEND_BYTES) for a handful of groups: the one-launch flush (k_flush_small_send) and the resident
    mailbox workgroup carry the byte limit and the flags with the request. 100-byte entries, max_size_per_msg = 250:
    util::limit_size keeps two entries per MsgAppend; a proposal's bcast_append is ONE send_append per peer (raft.rs:850-857),
    so two of the five new entries go out now and the rest waits for the acks."""
    G, P, cap = 64, 3, 8
    eng = rg.Engine(G, P, max_inflight=cap)
    st = O.alloc_state(G, P, stride=eng.stride)
    st["match"][:, :G], st["next"][:, :G] = 10, 11
    st["pr_commit"][:, :G] = 10
    st["pflags"][:, :P] = rg.PF.REPLICATE | rg.PF.RECENT_ACTIVE
    st["commit"][:], st["term_lo"][:], st["term_hi"][:] = 10, 1, 10
    st["cfg"][:] = rg.cfg_make(0b111, self_slot=0)
    eng.load_state(st)
    eng.log_sizes_enable(16)
    for g in range(G):
        eng.set_peers(g, [1, 2, 3], term=5)
    if mailbox:
        eng.mailbox_start()
    touched = [3, 17, 40]
    recs = np.zeros(len(touched) * 6, dtype=rg.engine.LOG_SIZE_DTYPE)
    k = 0
    for g in touched:  # cumulative sizes of entries 10 (the base of the sums) .. 15
        for i in range(10, 16):
            recs[k] = (g, i, 100 * (i - 9))
            k += 1
    eng.log_sizes_write(recs)
    for g in touched:
        eng.local_append(g, 15)
    served0 = eng.mailbox_stats()[0]
    eng.flush_send(max_bytes=250)  # (the first sparse flush of an engine takes the launch path: k_flush_small_send)
    items = eng.send_items()
    assert len(items) == 2 * len(touched) and sorted(set(items["group"].tolist())) == touched
    assert (items["prev_index"] == 10).all() and (items["last_index"] == 12).all() and (items["n_msgs"] == 1).all()
    assert eng.inflights(17, 1) == [12] and int(eng.read_column(rg.COL.NEXT)[2, 40]) == 13
    # NO_LIMIT in bytes (UINT64_MAX) fits the request word as well: one message for the next proposal
    recs = np.zeros(len(touched) * 2, dtype=rg.engine.LOG_SIZE_DTYPE)
    for j, g in enumerate(touched):
        recs[2 * j], recs[2 * j + 1] = (g, 16, 700), (g, 17, 800)
    eng.log_sizes_write(recs)
    for g in touched:
        eng.local_append(g, 17)
    eng.flush_send(max_bytes=O.U64_MAX)
    items = eng.send_items()
    assert len(items) == 2 * len(touched) and (items["n_msgs"] == 1).all()
    assert (items["prev_index"] == 12).all() and (items["last_index"] == 17).all()
    if mailbox:  # ... and the second one was answered by the resident workgroup
        assert eng.mailbox_stats()[0] == served0 + 1
        eng.mailbox_stop()
    eng.close()


def test_send_stage_serves_the_broadcast_after_recompute(rg):
    """post_conf_change (raft.rs:2618-2634): maybe_commit() on the new quorum, then bcast_append."""
    G, P, cap = 512, 3, 8
    eng = rg.Engine(G, P, max_inflight=cap)
    st = O.alloc_state(G, P, stride=eng.stride)
    st["match"][0, :G], st["match"][1, :G], st["match"][2, :G] = 9, 9, 5
    st["next"][0, :G], st["next"][1, :G], st["next"][2, :G] = 10, 10, 6
    st["pflags"][:, :P] = rg.PF.REPLICATE | rg.PF.RECENT_ACTIVE
    st["commit"][:], st["term_lo"][:], st["term_hi"][:] = 5, 1, 9
    st["cfg"][:] = rg.cfg_make(0b111, self_slot=0)
    eng.load_state(st)
    eng.recompute()
    eng.send_appends()
    items = np.sort(eng.send_items(), order=["group", "slot"])
    assert len(items) == 2 * G
    a, b = items[0::2], items[1::2]
    assert (a["slot"] == 1).all() and (a["prev_index"] == 9).all() and (a["last_index"] == 9).all(), "empty append: new commit"
    assert (b["slot"] == 2).all() and (b["prev_index"] == 5).all() and (b["last_index"] == 9).all()
    assert eng.inflights(5, 2) == [9] and eng.inflights(5, 1) == []
    eng.close()


def test_loaded_full_window_pauses_and_frees_on_heartbeat(rg):
    """rg_load_inflights / rg_load_column(PFLAGS) re-derive the engine-owned RG_PF_INS_FULL bit: a Replicate peer whose
    LOADED window is full must read is_paused() == true in the very next tick (raft.rs:1724, `else if old_paused
    {send_append}` :1749-1751) and a heartbeat response must free its first inflight (raft.rs:1791-1798)."""
    G, P, cap = 64, 3, 4
    st = O.alloc_state(G, P)
    st["cfg"][:] = rg.cfg_make(0b111, self_slot=0)
    st["term_lo"][:] = 60  # entries below 60 are of older terms: a quorum index of 51 does not pass maybe_commit
    st["term_hi"][:] = 100
    st["commit"][:] = 50
    st["match"][0, :G] = 100
    st["match"][1:, :G] = 50
    st["next"][0, :G] = 101
    st["next"][1:, :G] = 59  # four messages of two entries each in flight: 52, 54, 56, 58
    st["pflags"][:, :P] = 1 | 8  # Replicate, recent_active (no INS_FULL bit: the host does not own it)
    eng = rg.Engine(G, P, max_inflight=cap)
    eng.load_state(st)
    meta = np.zeros((P, eng.stride), dtype=np.uint32)
    ring = np.zeros((G, P, cap), dtype=np.uint64)
    meta[1:, :G] = 1 | (cap << 16)  # start 1, count == cap
    for k in range(cap):
        ring[:, 1:, (1 + k) % cap] = 52 + 2 * k
    eng.load_inflights(meta, ring)
    assert (eng.read_column(rg.COL.PFLAGS)[:, 1:P] & rg.PF.INS_FULL).all()
    assert not (eng.read_column(rg.COL.PFLAGS)[:, 0] & rg.PF.INS_FULL).any()
    # a wholesale flag load (checkpoint restore by columns) may carry garbage in that bit: re-derived again
    eng.load_column(rg.COL.PFLAGS, st["pflags"])
    assert (eng.read_column(rg.COL.PFLAGS)[:, 1:P] & rg.PF.INS_FULL).all()
    mb = rg.MsgBuffers(G, P, eng.stride)
    mb.m_flags[:, 1] = rg.MF.HEARTBEAT  # peer 2: heartbeat response on a full window -> free_first_one
    mb.m_commit[1, :G] = 50
    mb.m_flags[:, 2] = rg.MF.VALID      # peer 3: ack 51 cannot commit (older term); it was paused -> send_append
    mb.m_index[2, :G] = 51
    mb.m_commit[2, :G] = 50
    eng.tick(mb)
    _, out = eng.results()
    assert (rg.OUT.free_to(out[0]) & 0b010) and (rg.OUT.send_append(out[0]) & 0b100), hex(int(out[0]))
    assert (out == out[0]).all()
    eng.send_appends(2)
    assert eng.inflights(0, 1)[0] == 54  # 52 was freed (and the window refilled up to the cap)
    eng.close()


def test_skipped_send_stage_is_settled_before_the_next_tick(rg):
    """A host that skips rg_send_appends after a tick drops that tick's send requests, but the tick's effects on the
    device Inflights (free_to, reset on leaving Replicate) are applied before RG_COL_OUT is overwritten."""
    G, P, cap = 32, 3, 4
    st = O.alloc_state(G, P)
    st["cfg"][:] = rg.cfg_make(0b111, self_slot=0)
    st["term_lo"][:] = 1
    st["term_hi"][:] = 100
    st["commit"][:] = 50
    st["match"][0, :G] = 100
    st["match"][1:, :G] = 50
    st["next"][0, :G] = 101
    st["next"][1:, :G] = 59
    st["pflags"][:, :P] = 1 | 8
    eng = rg.Engine(G, P, max_inflight=cap)
    eng.load_state(st)
    meta = np.zeros((P, eng.stride), dtype=np.uint32)
    ring = np.zeros((G, P, cap), dtype=np.uint64)
    meta[1:, :G] = cap << 16
    for k in range(cap):
        ring[:, 1:, k] = 52 + 2 * k
    eng.load_inflights(meta, ring)
    mb = rg.MsgBuffers(G, P, eng.stride)
    mb.m_flags[:, 1] = rg.MF.VALID  # ack 56 frees 52, 54, 56
    mb.m_index[1, :G] = 56
    mb.m_flags[:, 2] = rg.MF.VALID | rg.MF.REJECT  # a real reject: Replicate -> Probe, ins.reset()
    mb.m_index[2, :G] = 58
    mb.m_hint[2, :G] = 50
    eng.tick(mb)          # ... and NO send stage
    mb.clear()
    eng.tick(mb)          # an empty tick overwrites RG_COL_OUT
    assert eng.inflights(0, 1) == [58] and eng.inflights(0, 2) == []
    assert not (eng.read_column(rg.COL.PFLAGS)[:, 1:P] & rg.PF.INS_FULL).any()
    assert int(eng.read_column(rg.COL.NEXT)[1, 0]) == 59  # nothing was sent for the skipped stage
    eng.close()


def test_dense_stage_work_item_columns_equal_the_compact_list(rg):
    """A stage over every group writes its work items as peer-major columns (rg_send_columns); the compact list
    rg_send_items materialises from them holds exactly the non-empty cells. A sparse stage leaves the columns alone."""
    import torch
    G, P, cap = 30_000 + 7, 5, 8
    eng = rg.Engine(G, P, max_inflight=cap)
    eng.workload_init(2)
    cols = [torch.zeros((P, eng.stride), dtype=torch.int64, device="cuda") for _ in range(4)]
    flags = torch.zeros((G, 8), dtype=torch.uint8, device="cuda")

    class Dev:
        def __init__(self, ptr, n, typestr):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (int(ptr), False), "version": 2}

    for t in range(4):
        eng.workload_gen(2, t, *[c.data_ptr() for c in cols], flags.data_ptr())
        flags &= 0xEF  # no RG_MF_SENT: the device sends
        if t >= 2:  # the same through rg_tick_device_send (one launch)
            eng.tick_device_send(*[c.data_ptr() for c in cols], flags.data_ptr(), max_entries_per_msg=2 if t % 2 else 0)
        else:
            eng.tick_device(*[c.data_ptr() for c in cols], flags.data_ptr())
            eng.send_appends(2 if t % 2 else 0)
        pp, pl, pn = eng.send_columns()
        n_cells = P * eng.stride
        eng.sync()
        prev = torch.as_tensor(Dev(pp, n_cells, "<i8"), device="cuda").cpu().numpy().view(np.uint64).reshape(P, -1)
        last = torch.as_tensor(Dev(pl, n_cells, "<i8"), device="cuda").cpu().numpy().view(np.uint64).reshape(P, -1)
        nk = torch.as_tensor(Dev(pn, n_cells, "<i4"), device="cuda").cpu().numpy().view(np.uint32).reshape(P, -1)
        tail = torch.as_tensor(Dev(eng.send_tail_column(), n_cells, "<i8"), device="cuda").cpu().numpy().view(np.uint64).reshape(P, -1)
        items = eng.send_items()
        assert len(items) == int((nk[:, :G] != 0).sum()) and len(items) > 3 * G
        g, s = items["group"].astype(np.int64), items["slot"].astype(np.int64)
        # an item whose last_index is the window's new newest inflight does not store it twice (bit 31 of the n / kind word)
        # (bit 31 of the n / kind word; an empty MsgAppend's last_index is its prev_index: bit 30)
        in_tail, is_prev = (nk[s, g] >> 31).astype(bool), ((nk[s, g] >> 30) & 1).astype(bool)
        assert not (in_tail & is_prev).any()
        assert in_tail.sum() > 0.8 * len(items) and (in_tail | is_prev).sum() > 0.97 * len(items), (int(in_tail.sum()), int(is_prev.sum()), len(items))
        want_last = np.where(in_tail, tail[s, g], np.where(is_prev, prev[s, g], last[s, g]))
        assert (prev[s, g] == items["prev_index"]).all() and (want_last == items["last_index"]).all()
        assert ((nk[s, g] & 0xffff) == items["n_msgs"]).all() and (((nk[s, g] >> 16) & 0x3fff) == items["kind"]).all()
        assert len(set(zip(g.tolist(), s.tolist()))) == len(items)
    from raft_rs_amd.engine import EngineError, ERR
    rec = np.zeros(1, dtype=rg.engine.WIRE_DTYPE)
    rec[0] = (5, int(eng.read_column(rg.COL.MATCH)[1, 5]), 0, 0, 0, 0, 1, rg.MF.VALID, 0)
    assert eng.ingest_tick(rec) == (1, 0)
    eng.send_appends(0)  # a sparse stage: compact list only
    with pytest.raises(EngineError) as e:
        eng.send_columns()
    assert e.value.code == ERR["STATE"]
    eng.close()


# ---- byte-accurate Config::max_size_per_msg (RG_SEND_BYTES) ------------------------------------------------
def size_records(rg, cum, lo, hi):
    """One rg_log_size record per entry in (lo[g], hi[g]] of every group: what a host writes when its leaders append."""
    n = (hi - lo).astype(np.int64)
    g = np.repeat(np.arange(len(lo), dtype=np.uint64), n)
    first = np.repeat(lo.astype(np.int64) + 1, n)
    off = np.arange(int(n.sum()), dtype=np.int64) - np.repeat(np.cumsum(n) - n, n)
    recs = np.zeros(len(g), dtype=rg.engine.LOG_SIZE_DTYPE)
    recs["group"], recs["index"] = g, (first + off).astype(np.uint64)
    recs["cum_bytes"] = cum[g.astype(np.int64), first + off]
    return recs


@pytest.mark.parametrize("n_slots,cap,window,max_bytes", [(3, 4, 8, 900), (5, 256, 64, 1500), (5, 3, 16, 0),
                                                         (7, 8, 32, 2**32 + 5), (5, 16, 64, O.U64_MAX)])
@pytest.mark.parametrize("fused", [False, True])
def test_send_stage_byte_limit_matches_oracle(rg, n_slots, cap, window, max_bytes, fused):
    """rg_send_appends(RG_SEND_BYTES): util::limit_size over the entry sizes the host wrote to the device
    (rg_log_sizes_write) against the oracle's literal restatement; RG_SEND_HOST peers served through rg_update_state."""
    rng = np.random.default_rng(7900 + 31 * n_slots + cap + window)
    G = 5000 + 7
    st = O.add_term_table(O.alloc_state(G, n_slots))
    st["cfg"][:] = fuzz.random_cfg(rng, G, n_slots, missing_progress_frac=0.05)
    fuzz.random_state(rng, st, small_values=True)
    fuzz.random_term_table(rng, st, term=6)
    eng = rg.Engine(G, n_slots, max_inflight=cap)
    eng.load_state(st)
    eng.log_sizes_enable(window)
    cl = O.Cluster(G)
    cl.load_soa(st, term=6, max_inflight=cap)
    cl.set_own_inflights(True)
    cl.set_limit_bytes(True)
    ticks = 8
    n_index = int(st["term_hi"].max()) + 64 * ticks + 64
    sizes, cum = sendstage.entry_sizes(rng, G, n_index)
    for g in range(G):
        cl.append_entry_sizes(g, 1, sizes[g, 1:])
    # the log as it stands: every entry up to last_index (index 0 = the base of the cumulative sums)
    written = st["term_hi"].copy()
    eng.log_sizes_write(size_records(rg, cum, np.maximum(written.astype(np.int64) - window, -1) , written.astype(np.int64)))
    msgs = O.alloc_msgs(G, n_slots)
    mb = rg.MsgBuffers(G, n_slots, eng.stride)
    gout = np.zeros(G, dtype=np.uint32)
    seen = {"items": 0, "multi": 0, "host": 0}
    for t in range(ticks):
        cl.store_soa(st)
        fuzz.random_msgs(rng, st, msgs, sent_p=0.0, heartbeat_p=0.2)
        sendstage.prepare_msgs(msgs)
        for k in ("m_index", "m_commit", "m_hint", "m_rs", "m_flags"):
            getattr(mb, k)[...] = msgs[k]
        if not fused:
            eng.tick(mb)
        cl.tick_soa(msgs, gout)
        cl.store_soa(st)  # (last_index after the tick's appends)
        hi = st["term_hi"].astype(np.int64)
        assert int(hi.max()) < n_index
        # (fused: the size records of what this tick's local-append events announce are written BEFORE the launch)
        eng.log_sizes_write(size_records(rg, cum, written.astype(np.int64), hi))
        written = st["term_hi"].copy()
        if fused:
            eng.tick_send(mb, max_bytes=max_bytes)
        _, out = eng.results()
        assert (out == gout).all(), (t, np.nonzero(out != gout)[0][:5])
        if not fused:
            eng.send_appends(max_bytes=max_bytes)
        items, omsgs, served = sendstage.split_host_items(eng.send_items(), cl.send_stage_soa(gout, max_bytes))
        got = sendstage.compare_items(items, omsgs)
        if served:
            sent = np.zeros(sum(len(v) for v in served.values()), dtype=rg.engine.SENT_MSG_DTYPE)
            i = 0
            for (g, p), lasts in served.items():
                for last in lasts:
                    sent[i] = (g, last, p, 0)
                    i += 1
            eng.update_state(sent)
        apply_snapshots(rg, eng, cl, st, got)
        check(rg, eng, cl, st, cap, f"bytes P={n_slots} cap={cap} W={window} tick {t}")
        seen["items"] += len(got)
        seen["multi"] += sum(1 for v in got.values() if v[3] > 1)
        seen["host"] += len(served)
    assert seen["items"] > 1000, seen
    if window <= 16:
        assert seen["host"] > 0, seen
    if max_bytes < 2000 and cap > 1:
        assert seen["multi"] > 0, seen
    eng.close()


def test_workload_sizes_and_dense_byte_stage(rg):
    """rg_workload_sizes (the benchmark's synthetic entry sizes) is what its formula says -- checked through the dense
    byte-limited stage against the oracle fed with the same sizes -- and RG_SEND_BYTES needs the table."""
    G, P, cap, window, seed = 30011, 5, 8, 64, 0x5eed
    eng = rg.Engine(G, P, max_inflight=cap)
    eng.workload_init(rg.WL_MAJORITY)
    with pytest.raises(rg.EngineError):
        eng.send_appends(max_bytes=100)
    eng.log_sizes_enable(window)
    with pytest.raises(rg.EngineError):
        eng.log_sizes_enable(window)
    st = eng.read_state()
    for col in (rg.COL.RUN_FIRST, rg.COL.RUN_TERM, rg.COL.DUMMY_INDEX, rg.COL.DUMMY_TERM, rg.COL.CUR_TERM):
        st[rg.COL.NAMES[col]] = eng.read_column(col)  # the log as the generator left it: nothing compacted
    cl = O.Cluster(G)
    cl.load_soa(st, max_inflight=cap)
    cl.set_own_inflights(True)
    cl.set_limit_bytes(True)

    def splitmix(x):
        x = (x + np.uint64(0x9E3779B97F4A7C15))
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))

    def sizes_of(g, first, n):  # min_bytes 40, spread 500
        idx = np.arange(first, first + n, dtype=np.uint64)
        with np.errstate(over="ignore"):
            h = splitmix(np.uint64(seed) ^ (np.uint64(0x517e) << np.uint64(40)) ^ (np.uint64(g) << np.uint64(3)) ^ idx)
        return (40 + (h % np.uint64(501))).astype(np.uint32)

    import torch
    cols = [torch.zeros((P, eng.stride), dtype=torch.int64, device="cuda") for _ in range(4)]
    flags = torch.zeros((G, 8), dtype=torch.uint8, device="cuda")
    msgs = O.alloc_msgs(G, P)
    gout = np.zeros(G, dtype=np.uint32)
    fed = np.zeros(G, dtype=np.int64)  # entries the oracle has sizes for, per group
    for t in range(4):
        eng.workload_gen(rg.WL_MAJORITY, t, *[c.data_ptr() for c in cols], flags.data_ptr())
        flags &= 0xEF  # no RG_MF_SENT: the device sends
        for k, c in zip(("m_index", "m_commit", "m_hint", "m_rs"), cols):
            msgs[k][...] = c.cpu().numpy().view(np.uint64)[:, :msgs[k].shape[1]]
        msgs["m_flags"][...] = flags.cpu().numpy()
        eng.tick_device(*[c.data_ptr() for c in cols], flags.data_ptr())
        cl.tick_soa(msgs, gout)
        _, out = eng.results()
        bad = np.nonzero(out != gout)[0]
        assert bad.size == 0, (t, bad.size, bad[:4], [hex(x) for x in out[bad[:4]]], [hex(x) for x in gout[bad[:4]]],
                               msgs["m_flags"][bad[:2]], st["pflags"][bad[:2]])
        eng.workload_sizes(seed, 40, 500)
        cl.store_soa(st)
        hi = st["term_hi"].astype(np.int64)
        for g in range(G):
            first = max(1, int(hi[g]) - 4 * window) if fed[g] == 0 else int(fed[g]) + 1  # (followers lag < 40 entries)
            if fed[g] == 0:
                cl.append_entry_sizes(g, first, sizes_of(g, first, int(hi[g]) - first + 1))
            elif int(hi[g]) >= first:
                cl.append_entry_sizes(g, first, sizes_of(g, first, int(hi[g]) - first + 1))
            fed[g] = hi[g]
        eng.send_appends(max_bytes=700)
        items, omsgs, served = sendstage.split_host_items(eng.send_items(), cl.send_stage_soa(gout, 700))
        assert not served  # every follower of this stream is within the window
        try:
            got = sendstage.compare_items(items, omsgs)
        except AssertionError as e:
            g = e.args[0][0][0]
            raise AssertionError((t, e.args[0], "engine items", [tuple(i) for i in items[items["group"] == g]], "oracle msgs",
                                  [tuple(m) for m in omsgs[omsgs["group"] == g]], "out", hex(gout[g]), "hi", int(hi[g]),
                                  "next", st["next"][:, g], "match", st["match"][:, g], "pflags", st["pflags"][g],
                                  "fed", int(fed[g])))
        assert sum(1 for v in got.values() if v[3] > 1) > 100  # 700 bytes split most broadcasts into several messages
        check(rg, eng, cl, st, cap, f"workload sizes tick {t}")
    eng.close()


def test_new_entry_points_refuse_what_they_cannot_do(rg):
    """Error behaviour of the byte-limit / mailbox / update_state entry points: wrong engine mode, bad arguments."""
    from raft_rs_amd.engine import EngineError, ERR
    plain = rg.Engine(300, 3)                   # Inflights with the host
    dev = rg.Engine(300, 3, max_inflight=4)     # Inflights on the device
    for eng in (plain, dev):
        eng.workload_init(rg.WL_MAJORITY)
    with pytest.raises(EngineError) as e:
        plain.log_sizes_enable(16)              # no send stage without device Inflights
    assert e.value.code == ERR["STATE"]
    with pytest.raises(EngineError) as e:
        plain.update_state(np.zeros(1, dtype=rg.engine.SENT_MSG_DTYPE))
    assert e.value.code == ERR["STATE"]
    for bad in (0, 4, 24, 8192):                # a power of two in 8..4096
        with pytest.raises(EngineError) as e:
            dev.log_sizes_enable(bad)
        assert e.value.code == ERR["INVALID_ARG"]
    with pytest.raises(EngineError) as e:
        dev.log_sizes_write(np.zeros(1, dtype=rg.engine.LOG_SIZE_DTYPE))  # not enabled yet
    assert e.value.code == ERR["STATE"]
    dev.log_sizes_enable(8)
    dev.log_sizes_write(np.zeros(0, dtype=rg.engine.LOG_SIZE_DTYPE))      # empty batches are fine
    dev.update_state(np.zeros(0, dtype=rg.engine.SENT_MSG_DTYPE))
    recs = np.zeros(2, dtype=rg.engine.LOG_SIZE_DTYPE)
    recs["group"] = [5, 10**9]                  # a record for a group this engine does not hold is ignored
    dev.log_sizes_write(recs)
    sent = np.zeros(2, dtype=rg.engine.SENT_MSG_DTYPE)
    sent["group"], sent["slot"] = [10**9, 3], [1, 99]  # unknown group / slot: ignored
    dev.update_state(sent)
    dev.sync()
    dev.mailbox_start()                         # (device Inflights: serves rg_flush_send, stage included)
    dev.mailbox_stop()
    plain.mailbox_start()
    assert plain.mailbox_stats() == (0, 0)
    plain.mailbox_stop()
    plain.mailbox_stop()                        # idempotent
    plain.close()
    dev.close()


# ---- RawNode::report_unreachable / report_snapshot applied on the device (rg_progress_events) --------------------------
def _events_to_oracle(cl, events, G, P):
    for g, s, kind in events:
        if g >= G or s >= P:
            continue
        if kind == 1:
            cl.L.ro_handle_unreachable(cl.h, g, s + 1)
        else:
            cl.L.ro_handle_snapshot_status(cl.h, g, s + 1, kind == 3)


@pytest.mark.parametrize("n_slots,cap", [(3, 0), (5, 3), (7, 256), (8, 1)])
def test_progress_events_match_oracle(rg, n_slots, cap):
    """MsgUnreachable / MsgSnapStatus between ticks: handle_unreachable (raft.rs:1931-1954) and handle_snapshot_status
    (:1891-1929) applied to the cells in place, runs of several events on one cell, against the oracle stepping the same
    local messages one by one -- Progress columns, the engine-owned flag bits and (cap > 0) the Inflights windows, which a
    state change resets; the ticks and send stages in between run on what the events left."""
    rng = np.random.default_rng(9100 + 17 * n_slots + cap)
    G = 5000 + 7
    st = O.add_term_table(O.alloc_state(G, n_slots))
    st["cfg"][:] = fuzz.random_cfg(rng, G, n_slots, missing_progress_frac=0.05)
    fuzz.random_state(rng, st, small_values=True, snapshot_frac=0.2)
    fuzz.random_term_table(rng, st, term=6)
    eng = rg.Engine(G, n_slots, max_inflight=cap)
    eng.load_state(st)
    cl = O.Cluster(G)
    cl.load_soa(st, term=6, max_inflight=cap)
    if cap:
        cl.set_own_inflights(True)
    msgs = O.alloc_msgs(G, n_slots)
    mb = rg.MsgBuffers(G, n_slots, eng.stride)
    gout = np.zeros(G, dtype=np.uint32)
    n_changed = n_items = 0
    for t in range(5):
        cl.store_soa(st)
        before = {k: st[k].copy() for k in ("next", "pflags")}
        events = fuzz.random_progress_events(rng, G, n_slots, 2500)
        eng.progress_events(events)
        _events_to_oracle(cl, events, G, n_slots)
        cl.store_soa(st)
        n_changed += int((before["next"] != st["next"]).sum() + (before["pflags"] != st["pflags"]).sum())
        got = eng.read_state()
        diffs = fuzz.diff_states(st, got, G, n_slots)
        assert not diffs, f"events {t}: " + "\n".join(diffs[:10])
        if cap:
            meta, ring = eng.read_inflights()
            sendstage.compare_rings(cl, meta, ring, st, cap)
        # a tick (and its send stage) on top of what the events left
        fuzz.random_msgs(rng, st, msgs, sent_p=0.0 if cap else 0.2, heartbeat_p=0.2)
        if cap:
            sendstage.prepare_msgs(msgs)
        for k in ("m_index", "m_commit", "m_hint", "m_rs", "m_flags"):
            getattr(mb, k)[...] = msgs[k]
        eng.tick(mb)
        cl.tick_soa(msgs, gout)
        _, out = eng.results()
        assert (out == gout).all(), (t, np.nonzero(out != gout)[0][:5])
        if cap:
            eng.send_appends(2)
            items = sendstage.compare_items(eng.send_items(), cl.send_stage_soa(gout, 2))
            apply_snapshots(rg, eng, cl, st, items)
            check(rg, eng, cl, st, cap, f"P={n_slots} cap={cap} tick {t}")
            n_items += len(items)
        else:
            cl.store_soa(st)
            diffs = fuzz.diff_states(st, eng.read_state(), G, n_slots)
            assert not diffs, f"tick {t}: " + "\n".join(diffs[:10])
    assert n_changed > 1000, n_changed
    assert not cap or n_slots == 1 or n_items > 500
    with pytest.raises(rg.EngineError) as e:
        eng.progress_events([(0, 0, 4)])
    assert e.value.code == -1
    eng.close()


def test_report_unreachable_and_snapshot_through_the_mirror(rg):
    """rg_report_unreachable / rg_report_snapshot: RawNode::report_unreachable(id) / report_snapshot(id, status)
    (raw_node.rs:692-709) by peer id. An id without a Progress is ignored; while the group has queued traffic the call is
    refused (local messages apply in call order: flush first)."""
    G, P, cap = 8, 3, 4
    eng = rg.Engine(G, P, max_inflight=cap)
    st = O.alloc_state(G, P, stride=eng.stride)
    st["match"][:, :G], st["next"][:, :G] = 10, 14
    st["pflags"][:, :P] = rg.PF.REPLICATE | rg.PF.RECENT_ACTIVE
    st["commit"][:], st["term_lo"][:], st["term_hi"][:] = 10, 1, 13
    st["cfg"][:] = rg.cfg_make(0b111, self_slot=0)
    st["pflags"][5, 2] = O.SNAPSHOT  # group 5, peer 33: a snapshot (index 20) is on its way
    st["pend_snap"][2, 5], st["next"][2, 5], st["pend_rs"][2, 5] = 20, 11, 20
    eng.load_state(st)
    for g in range(G):
        eng.set_peers(g, [11, 22, 33], term=5)
    eng.report_unreachable(2, 22)
    eng.report_unreachable(2, 99)  # "no progress available": ignored
    eng.report_snapshot(5, 33, failure=False)
    eng.report_snapshot(5, 22, failure=True)  # not in Snapshot: ignored
    got = eng.read_state()
    assert int(got["pflags"][2, 1]) & 7 == O.PROBE and int(got["next"][1, 2]) == 11
    f = int(got["pflags"][5, 2])
    assert f & 3 == O.PROBE and f & 4 and not f & 0xC0, hex(f)  # Probe, paused, both pending bits gone
    assert int(got["next"][2, 5]) == 21 and int(got["pend_snap"][2, 5]) == 0 and int(got["pend_rs"][2, 5]) == 0
    assert int(got["pflags"][5, 1]) & 3 == O.REPLICATE and int(got["next"][1, 5]) == 14
    untouched = [g for g in range(G) if g not in (2, 5)]
    assert (got["next"][:, untouched] == 14).all()
    eng.step(3, 22, 5, 12)  # queued, not flushed
    with pytest.raises(rg.EngineError) as e:
        eng.report_unreachable(3, 33)
    assert e.value.code == rg.engine.ERR["SLOT_BUSY"]
    eng.flush()
    eng.report_unreachable(3, 33)
    assert int(eng.read_column(rg.COL.PFLAGS)[3, 2]) & 3 == O.PROBE
    eng.close()


@pytest.mark.parametrize("cap", [0, 4])
def test_dense_progress_event_equals_the_record_form(rg, cap):
    """rg_progress_event_dense (one kind of event for every group that names a slot: 1 B per group instead of a 16 B
    record) leaves exactly what rg_progress_events leaves for the same events -- and what the oracle leaves."""
    rng = np.random.default_rng(9300 + cap)
    G, P = 30000 + 11, 5
    st = O.add_term_table(O.alloc_state(G, P))
    st["cfg"][:] = fuzz.random_cfg(rng, G, P, missing_progress_frac=0.05)
    fuzz.random_state(rng, st, small_values=True, snapshot_frac=0.3)
    fuzz.random_term_table(rng, st, term=6)
    a, b = rg.Engine(G, P, max_inflight=cap), rg.Engine(G, P, max_inflight=cap)
    cl = O.Cluster(G)
    cl.load_soa(st, term=6, max_inflight=cap)
    for eng in (a, b):
        eng.load_state(st)
    orig = a.read_state()
    present = ((st["cfg"] >> 24) & 0xff).astype(np.int64)
    absent = np.array([[not (present[g] >> p) & 1 for g in range(G)] for p in range(P)])  # [P][G]
    assert absent.sum() > 1000
    for kind in (1, 2, 3):
        slot1 = rng.integers(0, P + 3, size=G).astype(np.uint8)  # 0 = none, 1..P = a slot, P+1..P+2 = no such slot
        slot1[rng.random(G) < 0.3] = 0
        a.progress_event_dense(kind, slot1)
        g = np.nonzero(slot1)[0]
        b.progress_events([(int(x), int(slot1[x]) - 1, kind) for x in g])
        _events_to_oracle(cl, [(int(x), int(slot1[x]) - 1, kind) for x in g], G, P)
        sa, sb = a.read_state(), b.read_state()
        assert not fuzz.diff_states(sa, sb, G, P), kind
        # a slot without a Progress is left alone ("no progress available")
        assert (sa["next"][:, :G][absent] == orig["next"][:, :G][absent]).all()
        assert (sa["pflags"][:, :P].T[absent] == orig["pflags"][:, :P].T[absent]).all()
        cl.store_soa(st)
        diffs = fuzz.diff_states(st, sa, G, P)
        assert not diffs, (kind, diffs[:6])
        if cap:
            ma, ra = a.read_inflights()
            mb, rb = b.read_inflights()
            assert (ma == mb).all()
    with pytest.raises(rg.EngineError):
        a.progress_event_dense(0, np.zeros(G, dtype=np.uint8))
    a.close()
    b.close()
