"""GPU parity tests proper: the HIP engine (through the C ABI) vs the CPU oracle, bit-exact.

Every comparison is over ALL state columns (match, next, pr_commit, pending_snapshot,
pending_request_snapshot, pflags, commit, term_hi) and the per-group result word.
"""
import os

import numpy as np
import pytest

import fuzz
import hosthints
import oracle_lib as O

pytestmark = pytest.mark.gpu

TERM = 7


def oracle_from_state(st):
    cl = O.Cluster(st["n_groups"])
    cl.load_soa(st, term=TERM)
    return cl


def assert_same(eng, cl, st_ref, gout_ref, what):
    got = eng.read_state()
    cl.store_soa(st_ref)
    diffs = fuzz.diff_states(st_ref, got, st_ref["n_groups"], st_ref["n_slots"])
    assert not diffs, f"{what}: state differs from the oracle:\n" + "\n".join(diffs[:10])
    bad = np.nonzero(got["out"] != gout_ref)[0]
    assert bad.size == 0, (f"{what}: out word differs at g={bad[:5]}: engine "
                           f"{[hex(x) for x in got['out'][bad[:5]]]} oracle {[hex(x) for x in gout_ref[bad[:5]]]}")


@pytest.mark.parametrize("variant", [1, 2, 4, 5])
@pytest.mark.parametrize("workload,n_slots", [(2, 3), (2, 5), (3, 5), (5, 7), (2, 7)])
def test_workload_stream_matches_oracle(rg, variant, workload, n_slots):
    G, ticks = 20000 + 77, 6
    eng = rg.Engine(G, n_slots, variant=variant)
    eng.workload_init(workload)
    st = eng.read_state()
    cl = oracle_from_state(st)
    msgs = rg.MsgBuffers(G, n_slots, eng.stride)
    gout = np.zeros(G, dtype=np.uint32)
    from raft_rs_amd import engine as E
    for t in range(ticks):
        # messages come from the host twin of the generator, run on the ORACLE's state: if engine and
        # oracle ever diverged the streams would too, and the comparison below would catch it
        cl.store_soa(st)
        E.workload_gen_host(st, msgs, workload, t)
        eng.tick(msgs)
        cl.tick_soa(msgs.as_dict(), gout)
        assert_same(eng, cl, st, gout, f"workload {workload} P={n_slots} variant={variant} tick {t}")
    commit, out = eng.results()
    assert (commit == st["commit"]).all()
    eng.close()


@pytest.mark.parametrize("P,placed", [(5, False), (7, True)])
def test_device_generator_equals_host_generator(rg, P, placed):
    """The synthetic stream on the device (what bench.py records) and its host twin (what the parity tests feed the oracle),
    interleaved sizes and groups placed by size class (RG_WL_PLACE_SORTED), the initial state through the host twin as well."""
    import torch
    G = 5000
    eng = rg.Engine(G, P)
    eng.workload_init(5, sorted_classes=placed)
    st = eng.read_state()
    st_h = O.alloc_state(G, P)
    from raft_rs_amd import engine as E0
    E0.workload_init_host(st_h, 5, sorted_classes=placed)
    assert not fuzz.diff_states(st_h, st, G, P), "rg_workload_init_host builds the state rg_workload_init builds"
    dev = [torch.zeros((P, eng.stride), dtype=torch.int64, device="cuda") for _ in range(4)]
    dflags = torch.zeros((G, 8), dtype=torch.uint8, device="cuda")
    from raft_rs_amd import engine as E
    host = rg.MsgBuffers(G, P, eng.stride)
    for t in range(3):
        eng.workload_gen(5, t, *[d.data_ptr() for d in dev], dflags.data_ptr(), sorted_classes=placed)
        eng.sync()
        E.workload_gen_host(st, host, 5, t, sorted_classes=placed)
        present = (st["cfg"] >> 24) & 0xff
        for name, d in zip(("m_index", "m_commit", "m_hint", "m_rs"), dev):
            got = d.cpu().numpy().view(np.uint64)
            for p in range(P):
                sel = ((present >> p) & 1) == 1
                assert (got[p, :G][sel] == getattr(host, name)[p, :G][sel]).all(), (name, t, p)
        assert (dflags.cpu().numpy() == host.m_flags).all()
        eng.tick_device(*[d.data_ptr() for d in dev], dflags.data_ptr())
        st = eng.read_state()
    eng.close()


@pytest.mark.parametrize("variant", [1, 2, 4, 5])
@pytest.mark.parametrize("n_slots", [1, 2, 3, 4, 5, 6, 7, 8])
def test_random_streams_match_oracle(rg, variant, n_slots):
    rng = np.random.default_rng(1000 + n_slots)
    G = 4096 + 13
    st = O.alloc_state(G, n_slots)
    st["cfg"][:] = fuzz.random_cfg(rng, G, n_slots, missing_progress_frac=0.05)
    fuzz.random_state(rng, st, small_values=(n_slots % 2 == 0))
    eng = rg.Engine(G, n_slots, variant=variant)
    eng.load_state(st)
    cl = oracle_from_state(st)
    msgs = O.alloc_msgs(G, n_slots)
    gout = np.zeros(G, dtype=np.uint32)
    mb = rg.MsgBuffers(G, n_slots, eng.stride)
    for t in range(5):
        cl.store_soa(st)
        fuzz.random_msgs(rng, st, msgs, malformed_p=0.02 if t == 3 else 0.0)
        for k in ("m_index", "m_commit", "m_hint", "m_rs", "m_flags"):
            getattr(mb, k)[...] = msgs[k]
        eng.tick(mb)
        cl.tick_soa(msgs, gout)
        assert_same(eng, cl, st, gout, f"random P={n_slots} variant={variant} tick {t}")
    eng.close()


@pytest.mark.parametrize("n_groups,n_slots", [(1, 3), (63, 5), (64, 5), (65, 7), (257, 8)])
def test_tiny_and_ragged_group_counts(rg, n_groups, n_slots):
    """Edge sizes: a single group, one short of / exactly / one past a wave, one past a 256 tile; indices
    near the top of the u64 range."""
    rng = np.random.default_rng(n_groups * 10 + n_slots)
    for variant in (1, 2, 5):
        st = O.alloc_state(n_groups, n_slots)
        st["cfg"][:] = fuzz.random_cfg(rng, n_groups, n_slots)
        fuzz.random_state(rng, st, small_values=True, base=2 ** 62)
        eng = rg.Engine(n_groups, n_slots, variant=variant)
        eng.load_state(st)
        cl = oracle_from_state(st)
        msgs = O.alloc_msgs(n_groups, n_slots)
        gout = np.zeros(n_groups, dtype=np.uint32)
        mb = rg.MsgBuffers(n_groups, n_slots, eng.stride)
        for t in range(4):
            cl.store_soa(st)
            fuzz.random_msgs(rng, st, msgs)
            for k in ("m_index", "m_commit", "m_hint", "m_rs", "m_flags"):
                getattr(mb, k)[...] = msgs[k]
            eng.tick(mb)
            cl.tick_soa(msgs, gout)
            assert_same(eng, cl, st, gout, f"G={n_groups} P={n_slots} variant={variant} tick {t}")
        eng.close()


def test_empty_tick_changes_nothing(rg):
    G, P = 5000, 5
    eng = rg.Engine(G, P)
    eng.workload_init(2)
    before = eng.read_state()
    eng.tick(rg.MsgBuffers(G, P, eng.stride))  # no events at all
    after = eng.read_state()
    for k in fuzz.STATE_KEYS:
        assert (before[k] == after[k]).all(), k
    assert (after["out"] == 0).all()
    eng.close()


@pytest.mark.parametrize("n_slots", [3, 5, 7])
def test_group_commit_streams_match_oracle(rg, n_slots):
    rng = np.random.default_rng(77 + n_slots)
    G = 3000
    st = O.alloc_state(G, n_slots)
    st["cfg"][:] = fuzz.random_cfg(rng, G, n_slots, group_commit_frac=0.6)
    fuzz.random_state(rng, st, small_values=True, with_gids=True)
    eng = rg.Engine(G, n_slots)
    eng.load_state(st)
    cl = oracle_from_state(st)
    # maximal_committed_index incl. the group-commit flag
    mci, used = eng.maximal_committed_index(with_flag=True)
    for g in range(G):
        v, f = cl.mci(g)
        assert mci[g] == v and bool(used[g]) == f, (g, mci[g], v, used[g], f)
    msgs = O.alloc_msgs(G, n_slots)
    gout = np.zeros(G, dtype=np.uint32)
    mb = rg.MsgBuffers(G, n_slots, eng.stride)
    for t in range(4):
        cl.store_soa(st)
        fuzz.random_msgs(rng, st, msgs)
        for k in ("m_index", "m_commit", "m_hint", "m_rs", "m_flags"):
            getattr(mb, k)[...] = msgs[k]
        eng.tick(mb)
        cl.tick_soa(msgs, gout)
        assert_same(eng, cl, st, gout, f"group-commit P={n_slots} tick {t}")
    eng.close()


@pytest.mark.parametrize("variant,P", [(1, 5), (3, 5), (3, 8), (3, 3)])
def test_recompute_matches_oracle_maybe_commit(rg, variant, P):
    rng = np.random.default_rng(5)
    G = 5000 + 7
    st = O.alloc_state(G, P)
    st["cfg"][:] = fuzz.random_cfg(rng, G, P)
    fuzz.random_state(rng, st)
    st["commit"][:] = st["commit"] // 2  # leave room to commit
    eng = rg.Engine(G, P, variant=variant)  # 3 = wave-cooperative rank select (8 lanes per group)
    eng.load_state(st)
    mci = eng.maximal_committed_index()
    cl = oracle_from_state(st)
    assert (mci == np.array([cl.mci(g)[0] for g in range(G)], dtype=np.uint64)).all()
    eng.recompute()
    gout = np.array([1 if cl.maybe_commit(g) else 0 for g in range(G)], dtype=np.uint32)
    assert_same(eng, cl, st, gout, "recompute")
    eng.recompute()  # idempotent: nothing left to commit
    _, out = eng.results()
    assert (out == 0).all()
    eng.close()


def test_checkpoint_restore_replays_identically(rg):
    G, P = 30000, 5
    eng = rg.Engine(G, P)
    eng.workload_init(2)
    eng.checkpoint()
    from raft_rs_amd import engine as E
    st0 = eng.read_state()
    msgs = rg.MsgBuffers(G, P, eng.stride)
    E.workload_gen_host(st0, msgs, 2, 0)
    eng.tick(msgs)
    a = eng.read_state()
    eng.restore()
    b0 = eng.read_state()
    for k in fuzz.STATE_KEYS:
        assert (b0[k] == st0[k]).all(), k
    eng.tick(msgs)
    b = eng.read_state()
    for k in fuzz.STATE_KEYS + ("out",):
        assert (a[k] == b[k]).all(), k
    eng.close()


def test_full_size_properties_1m_groups(rg):
    """BASELINE config 2 size (1M x 5): properties that need no oracle pass over 1M groups, plus an
    oracle check on a 50k-group slice and lane-vs-LDS variant equality on everything."""
    import torch
    G, P, ticks = 1_000_000, 5, 4
    engs = [rg.Engine(G, P, variant=v) for v in (1, 2)]
    for e in engs:
        e.workload_init(2)
    st = engs[0].read_state()
    n_or = 50_000
    sub = O.alloc_state(n_or, P)
    for k in ("match", "next", "pr_commit", "pend_snap", "pend_rs", "gid"):
        sub[k][:, :n_or] = st[k][:, :n_or]
    for k in ("pflags", "commit", "term_lo", "term_hi", "cfg"):
        sub[k][...] = st[k][:n_or]
    cl = oracle_from_state(sub)
    dev = [torch.zeros((P, engs[0].stride), dtype=torch.int64, device="cuda") for _ in range(4)]
    dflags = torch.zeros((G, 8), dtype=torch.uint8, device="cuda")
    prev_commit = st["commit"].copy()
    gout = np.zeros(n_or, dtype=np.uint32)
    for t in range(ticks):
        engs[0].workload_gen(2, t, *[d.data_ptr() for d in dev], dflags.data_ptr())
        engs[0].sync()
        for e in engs:
            e.tick_device(*[d.data_ptr() for d in dev], dflags.data_ptr())
        s0, s1 = engs[0].read_state(), engs[1].read_state()
        for k in fuzz.STATE_KEYS + ("out",):
            assert (s0[k] == s1[k]).all(), f"lane and LDS variants differ in {k} at tick {t}"
        assert (s0["commit"] >= prev_commit).all()
        assert (s0["commit"] <= s0["term_hi"]).all()
        assert ((s0["out"] >> 1) & 1).sum() == 0, "well-formed stream raises no fault"
        changed = (s0["out"] & 1) == 1
        assert ((s0["commit"] > prev_commit) == changed).all()
        # the committed index is acked by a majority: count(match >= commit) >= 3 where it changed
        cnt = (s0["match"][:, :G] >= s0["commit"][None, :]).sum(axis=0)
        assert (cnt[changed] >= 3).all()
        assert engs[0].result_counts() == (int(changed.sum()), 0)
        prev_commit = s0["commit"].copy()
        # oracle on the first 50k groups
        m = {"n_groups": n_or, "n_slots": P, "stride": sub["stride"]}
        for name, d in zip(("m_index", "m_commit", "m_hint", "m_rs"), dev):
            a = np.zeros((P, sub["stride"]), dtype=np.uint64)
            a[:, :n_or] = d[:, :n_or].cpu().numpy().view(np.uint64)
            m[name] = a
        m["m_flags"] = np.ascontiguousarray(dflags[:n_or].cpu().numpy())
        cl.tick_soa(m, gout)
        cl.store_soa(sub)
        for k in ("match", "next", "pr_commit"):
            assert (sub[k][:, :n_or] == s0[k][:, :n_or]).all(), k
        assert (sub["commit"] == s0["commit"][:n_or]).all()
        assert (sub["pflags"] == s0["pflags"][:n_or]).all()
        assert (gout == s0["out"][:n_or]).all()
    for e in engs:
        e.close()


@pytest.mark.parametrize("workload,n_slots,T", [(2, 5, 4), (5, 7, 8), (3, 5, 3)])
def test_fused_launch_equals_sequential_ticks(rg, workload, n_slots, T):
    """rg_tick_device_fused(T ticks) == T x rg_tick_device: final state, every tick's result word and commit
    index; and both equal the oracle."""
    import torch
    from raft_rs_amd import engine as E
    G = 30000 + 11
    seq = rg.Engine(G, n_slots)
    fus = rg.Engine(G, n_slots)
    for e in (seq, fus):
        e.workload_init(workload)
    st = seq.read_state()
    cl = oracle_from_state(st)
    gout = np.zeros(G, dtype=np.uint32)
    host = rg.MsgBuffers(G, n_slots, seq.stride)
    for rnd in range(2):
        dev_ticks, want_out, want_commit = [], [], []
        for t in range(T):
            cl.store_soa(st)
            E.workload_gen_host(st, host, workload, rnd * T + t)
            cols = [torch.from_numpy(getattr(host, k).view(np.int64).copy()).cuda()
                    for k in ("m_index", "m_commit", "m_hint", "m_rs")]
            flags = torch.from_numpy(host.m_flags.copy()).cuda()
            dev_ticks.append((cols, flags))
            seq.tick_device(*[c.data_ptr() for c in cols], flags.data_ptr())
            cl.tick_soa(host.as_dict(), gout)
            cl.store_soa(st)
            c, o = seq.results()
            assert (o == gout).all() and (c == st["commit"]).all()
            want_out.append(o)
            want_commit.append(c)
        out_t = torch.zeros((T, G), dtype=torch.int32, device="cuda")
        commit_t = torch.zeros((T, G), dtype=torch.int64, device="cuda")
        fus.tick_device_fused([[c.data_ptr() for c in cols] + [flags.data_ptr()] for cols, flags in dev_ticks],
                              out_t.data_ptr(), commit_t.data_ptr())
        fus.sync()
        ot = out_t.cpu().numpy().view(np.uint32)
        ct = commit_t.cpu().numpy().view(np.uint64)
        for t in range(T):
            assert (ot[t] == want_out[t]).all(), (rnd, t)
            assert (ct[t] == want_commit[t]).all(), (rnd, t)
        a, b = seq.read_state(), fus.read_state()
        for k in fuzz.STATE_KEYS + ("out",):
            assert (a[k] == b[k]).all(), (rnd, k)
    seq.close()
    fus.close()


@pytest.mark.parametrize("variant", [1, 2])
@pytest.mark.parametrize("n_slots", [3, 7])
def test_find_conflict_by_term_on_device(rg, n_slots, variant):
    """Rejects that carry Message.log_term: hint resolved on the device against the term-run table
    (raft_log.rs:209-235 via raft.rs:1562,1657-1660); dense tick and the wire-record path."""
    from raft_rs_amd.engine import WIRE_DTYPE
    rng = np.random.default_rng(600 + n_slots)
    G, TERM = 6000, 9
    st = O.add_term_table(O.alloc_state(G, n_slots))
    st["cfg"][:] = fuzz.random_cfg(rng, G, n_slots)
    fuzz.random_state(rng, st, small_values=True, probe_frac=0.5)
    fuzz.random_term_table(rng, st, TERM)
    eng = rg.Engine(G, n_slots, variant=variant)
    eng.load_state(st)
    cl = O.Cluster(G)
    cl.load_soa(st, term=TERM)
    msgs = O.alloc_msgs(G, n_slots)
    gout = np.zeros(G, dtype=np.uint32)
    mb = rg.MsgBuffers(G, n_slots, eng.stride)
    n_lt = 0
    for t in range(5):
        cl.store_soa(st)
        fuzz.random_msgs(rng, st, msgs, reject_p=0.4, logterm_max=TERM)
        n_lt += int(((msgs["m_flags"] & 0x80) != 0).sum())
        if t % 2 == 0:
            for k in ("m_index", "m_commit", "m_hint", "m_rs", "m_logterm", "m_flags"):
                getattr(mb, k)[...] = msgs[k]
            eng.tick(mb)
        else:  # the same messages as wire-order records through the sparse path
            recs = []
            for g in range(G):
                for p in range(n_slots):
                    f = int(msgs["m_flags"][g, p])
                    if f:
                        recs.append((g, msgs["m_index"][p, g], msgs["m_commit"][p, g], msgs["m_hint"][p, g],
                                     msgs["m_rs"][p, g], msgs["m_logterm"][p, g], p, f, 0))
            arr = np.array(recs, dtype=WIRE_DTYPE)
            rng.shuffle(arr)
            assert eng.ingest(arr) == 0
            eng.tick_ingested()
        cl.tick_soa(msgs, gout)
        assert_same(eng, cl, st, gout, f"log_term rejects P={n_slots} variant={variant} tick {t}")
    assert n_lt > 1000
    eng.close()


@pytest.mark.parametrize("form", ["dense", "sparse", "lds"])
@pytest.mark.parametrize("n_slots", [3, 5, 7])
def test_log_history_deeper_than_the_term_run_table(rg, n_slots, form):
    """RaftLog keeps the whole log (raft_log.rs:122-140), the engine's term-run table the newest RG_TERM_RUNS runs of older terms.
    Groups that start with a FULL table and see up to nine more elections (9 .. 17 older runs in the oracle's log), rejects whose
    reject_hint / log_term land anywhere in that history, through the dense tick, the wire-record path and the LDS variant: the
    engine equals the UNMODIFIED oracle wherever RG_OUT_HOST_HINT is clear, the bit / RG_COL_HOST_HINT / rg_host_hints name exactly
    the rejects whose literal find_conflict_by_term walk (raft_log.rs:209-235) consults a dropped run, and once the host has
    answered (a tick of its own, or rg_resolve_host_hints on every other tick) every column equals the oracle again."""
    from raft_rs_amd.engine import COL, WIRE_DTYPE
    box = {}

    def load(st):
        box["eng"] = rg.Engine(st["n_groups"], n_slots, variant=2 if form == "lds" else 1)
        box["eng"].load_state(st)
        box["mb"] = rg.MsgBuffers(st["n_groups"], n_slots, box["eng"].stride)

    def tick(m):
        eng, mb = box["eng"], box["mb"]
        if form == "sparse":
            G = m["n_groups"]
            gs, ps = np.nonzero(m["m_flags"][:, :n_slots])
            arr = np.zeros(len(gs), dtype=WIRE_DTYPE)
            arr["group"], arr["slot"], arr["flags"] = gs, ps, m["m_flags"][gs, ps]
            for k, f in (("m_index", "index"), ("m_commit", "commit"), ("m_hint", "hint"), ("m_rs", "rs"), ("m_logterm", "log_term")):
                arr[f] = m[k][ps, gs]
            assert eng.ingest(arr) == 0
            eng.tick_ingested()
        else:
            for k in ("m_index", "m_commit", "m_hint", "m_rs", "m_logterm", "m_flags"):
                getattr(mb, k)[...] = m[k]
            eng.tick(mb)
        return eng.read_column(COL.OUT)

    def resolve(recs):
        eng = box["eng"]
        listed = {int(r["group"]): int(r["slot_mask"]) for r in eng.host_hints()}
        assert listed == {g: sum(1 << r[3] for r in recs if r[0] == g) for g in {r[0] for r in recs}}
        assert eng.resolve_host_hints(recs).all()
        return eng.read_column(COL.OUT)

    stats = None
    for step in hosthints.deep_history_run(n_slots, 7100 + n_slots, tick, lambda: box["eng"].read_column(COL.HOST_HINT), load,
                                           G=3000, resolve=resolve):
        if isinstance(step, dict):
            stats = step
            break
        cl, st, t = step
        got = box["eng"].read_state()
        diffs = fuzz.diff_states(st, got, st["n_groups"], n_slots)
        assert not diffs, (form, t, diffs[:6])
        for col in (COL.RUN_FIRST, COL.RUN_TERM, COL.CUR_TERM, COL.DUMMY_INDEX, COL.DUMMY_TERM):
            assert (box["eng"].read_column(col) == st[COL.NAMES[col]]).all(), (t, COL.NAMES[col])
    assert stats["settled"] > 200 and stats["applied_logterm_rejects"] > 5 * stats["settled"], stats
    assert stats["max_runs"] >= 12 and len(stats["depths"]) >= 4, stats
    box["eng"].close()


@pytest.mark.parametrize("n_slots", [3, 7])
def test_fused_call_with_log_term_ticks_and_elections(rg, n_slots):
    """rg_tick_device_fused over 6 ticks of which two carry Message.log_term (the library runs those behind their
    find_conflict_by_term pre-pass, between the fused launches of the others) and all carry elections: every tick's
    result word and commit index and the final state against the oracle."""
    import torch
    rng = np.random.default_rng(660 + n_slots)
    G, TERM, T = 5000, 9, 6
    st = O.add_term_table(O.alloc_state(G, n_slots))
    st["cfg"][:] = fuzz.random_cfg(rng, G, n_slots)
    fuzz.random_state(rng, st, small_values=True, probe_frac=0.4)
    # (at most 2 + 6 older runs: the table never overflows inside the call, so no reject goes back to the host in the
    # middle of a fused launch -- tests/test_parity_gpu.py::test_log_history_deeper_than_the_term_run_table has that corner)
    fuzz.random_term_table(rng, st, TERM, max_runs=2)
    eng = rg.Engine(G, n_slots)
    eng.load_state(st)
    cl = O.Cluster(G)
    cl.load_soa(st, term=TERM)
    gout = np.zeros(G, dtype=np.uint32)
    dev, want_out, want_commit, n_lt = [], [], [], 0
    for t in range(T):
        cl.store_soa(st)
        msgs = O.alloc_msgs(G, n_slots)
        with_lt = t in (1, 4)
        fuzz.random_msgs(rng, st, msgs, reject_p=0.3, logterm_max=(TERM + t) if with_lt else 0, elect_p=0.1, elect_term=TERM + 1 + t)
        n_lt += int(((msgs["m_flags"] & 0x80) != 0).sum())
        cl.tick_soa(msgs, gout)
        cl.store_soa(st)
        want_out.append(gout.copy())
        want_commit.append(st["commit"].copy())
        keys = ("m_index", "m_commit", "m_hint", "m_rs", "m_flags") + (("m_logterm",) if with_lt else ())
        dev.append([torch.from_numpy(np.ascontiguousarray(msgs[k]).view(np.uint8 if k == "m_flags" else np.int64).copy()).cuda()
                    for k in keys])
    out_t = torch.zeros((T, G), dtype=torch.int32, device="cuda")
    commit_t = torch.zeros((T, G), dtype=torch.int64, device="cuda")
    eng.tick_device_fused([[c.data_ptr() for c in tick] for tick in dev], out_t.data_ptr(), commit_t.data_ptr())
    eng.sync()
    ot, ct = out_t.cpu().numpy().view(np.uint32), commit_t.cpu().numpy().view(np.uint64)
    for t in range(T):
        bad = np.nonzero(ot[t] != want_out[t])[0]
        assert bad.size == 0, (t, bad[:5], [hex(x) for x in ot[t][bad[:5]]], [hex(x) for x in want_out[t][bad[:5]]])
        assert (ct[t] == want_commit[t]).all(), t
    assert_same(eng, cl, st, gout, f"fused call with log-term ticks P={n_slots}")
    assert n_lt > 300 and (np.array(want_out) & 0x10).any()
    eng.close()


@pytest.mark.parametrize("n_slots", [3, 7])
def test_fused_call_stops_behind_a_tick_that_leaves_a_reject_to_the_host(rg, n_slots):
    """Exact or loud inside fused launches: histories that grow to 12+ runs (a full term-run table plus elections in every
    tick), eight log-term ticks submitted as ONE rg_tick_device_fused call. The reference applies a reject before anything later
    (raft_log.rs:209-235 -> raft.rs:1657-1660), so the call must END behind the first tick that leaves one to the host
    (RG_ERR_HOST_HINT, rg_fused_ticks_done), with RG_COL_OUT / RG_COL_HOST_HINT that tick's; the host answers
    (rg_resolve_host_hints against the oracle's complete log) and submits the rest. Every tick's result word and commit index
    and the final state must be what the unfused sequence -- the oracle, tick by tick -- produces."""
    import torch
    import hosthints
    from raft_rs_amd.engine import COL
    rng = np.random.default_rng(9900 + n_slots)
    G, TERM, T = 3000, 30, 8
    st = O.add_term_table(O.alloc_state(G, n_slots))
    st["cfg"][:] = fuzz.random_cfg(rng, G, n_slots, missing_progress_frac=0.03)
    fuzz.random_state(rng, st, probe_frac=0.5, base=200)
    fuzz.random_term_table(rng, st, TERM, min_runs=O.TERM_RUNS)
    eng = rg.Engine(G, n_slots)
    eng.load_state(st)
    ahead, host = O.Cluster(G), O.Cluster(G)  # the oracle twice: one runs ahead to write the stream, one follows the engine
    ahead.load_soa(st, term=TERM)
    host.load_soa(st, term=TERM)
    gout = np.zeros(G, dtype=np.uint32)
    ticks, dev, want_out, want_commit = [], [], [], []
    for t in range(T):
        ahead.store_soa(st)
        msgs = O.alloc_msgs(G, n_slots)
        fuzz.random_msgs(rng, st, msgs, valid_p=0.8, reject_p=0.6, rs_p=0.05, sent_p=0.2, heartbeat_p=0.05, logterm_max=TERM + t,
                         elect_p=0.5, elect_term=TERM + 1 + t)
        hosthints.spread_reject_hints(rng, st, msgs, TERM + 1 + t)
        ahead.tick_soa(msgs, gout)
        ahead.store_soa(st)
        ticks.append(msgs)
        want_out.append(gout.copy())
        want_commit.append(st["commit"].copy())
        dev.append([torch.from_numpy(np.ascontiguousarray(msgs[k]).view(np.uint8 if k == "m_flags" else np.int64).copy()).cuda()
                    for k in ("m_index", "m_commit", "m_hint", "m_rs", "m_flags", "m_logterm")])
    assert max(len(hosthints.log_runs(ahead, g)[0]) for g in range(0, G, 37)) >= 12
    out_t = torch.zeros((T, G), dtype=torch.int32, device="cuda")
    commit_t = torch.zeros((T, G), dtype=torch.int64, device="cuda")
    pos, calls, early_stops, settled = 0, 0, 0, 0
    hgout = np.zeros(G, dtype=np.uint32)
    while pos < T:
        n = eng.tick_device_fused([[c.data_ptr() for c in tick] for tick in dev[pos:]], out_t[pos:].data_ptr(), commit_t[pos:].data_ptr())
        calls += 1
        assert 1 <= n <= T - pos and eng.fused_ticks_done() == n
        eng.sync()
        rows = out_t[pos:pos + n].cpu().numpy().view(np.uint32)
        # only the LAST applied tick of a call may carry the bit -- and must, if the call ended early
        assert not (rows[:-1] & hosthints.OUT_HOST_HINT).any()
        if pos + n < T:
            early_stops += 1
            assert (rows[-1] & hosthints.OUT_HOST_HINT).any()
        for k in range(n):
            host.tick_soa(ticks[pos + k], hgout)
        last = pos + n - 1
        out = eng.read_column(COL.OUT)
        assert (out == rows[-1]).all()  # RG_COL_OUT is the hinted (= last applied) tick's, not a later one's
        hh = eng.read_column(COL.HOST_HINT)

        def resolve(recs):
            assert eng.resolve_host_hints(recs).all()
            return eng.read_column(COL.OUT)

        merged, k = hosthints.settle(host, ticks[last], out, hh, resolve=resolve)
        settled += k
        for t in range(pos, last):
            assert (rows[t - pos] == want_out[t]).all(), t
        assert (merged == want_out[last]).all(), (last, np.nonzero(merged != want_out[last])[0][:5])
        ct = commit_t[pos:pos + n].cpu().numpy().view(np.uint64)
        for t in range(pos, pos + n):
            assert (ct[t - pos] == want_commit[t]).all(), t
        pos += n
    assert early_stops >= 2 and settled >= 20, (calls, early_stops, settled)
    assert_same(eng, host, st, want_out[-1], f"fused call stopped by host hints P={n_slots}")
    eng.close()


@pytest.mark.parametrize("n_slots", [3, 8])
def test_garbage_events_on_gpu(rg, n_slots):
    rng = np.random.default_rng(4321 + n_slots)
    G, TERM = 5000, 9
    st = O.add_term_table(O.alloc_state(G, n_slots))
    st["cfg"][:] = fuzz.random_cfg(rng, G, n_slots, missing_progress_frac=0.1)
    fuzz.random_state(rng, st, small_values=True)
    fuzz.random_term_table(rng, st, TERM)
    for variant in (1, 2):
        eng = rg.Engine(G, n_slots, variant=variant)
        eng.load_state(st)
        cl = O.Cluster(G)
        cl.load_soa(st, term=TERM)
        ref = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in st.items()}
        msgs = O.alloc_msgs(G, n_slots)
        gout = np.zeros(G, dtype=np.uint32)
        mb = rg.MsgBuffers(G, n_slots, eng.stride)
        for t in range(4):
            cl.store_soa(ref)
            fuzz.garbage_msgs(rng, ref, msgs)
            for k in ("m_index", "m_commit", "m_hint", "m_rs", "m_logterm", "m_flags"):
                getattr(mb, k)[...] = msgs[k]
            eng.tick(mb)
            cl.tick_soa(msgs, gout)
            # (garbage elects about half the groups per tick: histories outgrow the term-run table and some rejects come back)
            hosthints.settle_engine(eng, cl, msgs)
            assert_same(eng, cl, ref, gout, f"garbage events P={n_slots} variant={variant} tick {t}")
        eng.close()


def test_soak_on_gpu(rg):
    """100 ticks of the mixed workload on the GPU against the oracle (compared every 10 ticks)."""
    from raft_rs_amd import engine as E
    G, P, WL = 6000, 7, 5
    eng = rg.Engine(G, P)
    eng.workload_init(WL)
    st = eng.read_state()
    cl = oracle_from_state(st)
    mb = rg.MsgBuffers(G, P, eng.stride)
    gout = np.zeros(G, dtype=np.uint32)
    for t in range(100):
        E.workload_gen_host(st, mb, WL, t)
        eng.tick(mb)
        cl.tick_soa(mb.as_dict(), gout)
        cl.store_soa(st)
        if t % 10 == 9:
            assert_same(eng, cl, st, gout, f"soak tick {t}")
    eng.close()


def test_cfg_words_are_validated(rg):
    eng = rg.Engine(16, 3)
    bad = np.full(16, rg.cfg_make(0x0f, 0, 0), dtype=np.uint32)  # names slot 3 of a 3-slot engine
    with pytest.raises(rg.EngineError) as e:
        eng.load_column(rg.COL.CFG, bad)
    assert e.value.code == -1
    with pytest.raises(rg.EngineError):
        eng.set_config(0, rg.cfg_make(0x07, 0, 5))  # self slot 5
    eng.close()


def test_bench_two_ranks_sharing_the_gpu(rg, tmp_path):
    """The N>1 path of bench.py (rank offsets, commit publication + verification, max-over-ranks timing) with two
    ranks on this box's single GPU (BENCH_SHARE_GPU=1 switches the collective backend to gloo)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_SHARE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29547", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "12",
           "--warmup", "2", "--groups", "60000", "--publish-every", "4", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["steps"] == 12 and d["value"] > 0
    assert d["config"]["groups_per_gpu"] == 60000 and "published every 4 tick(s)" in d["config"]["sharding"] and d["config"]["pub_publications"] > 0
    assert d["config"]["transport"] == "callback" and d["config"]["rccl_engines"] == 0  # (two ranks on one GPU: gloo moves the slices, and the line says so)
    assert len(line) <= 6000


def test_device_info_is_queried_not_assumed(rg):
    eng = rg.Engine(1000, 5)
    info = eng.device_info()
    assert info["arch"] == "gfx950" and info["wavefront"] == 64
    # the cache the regime policy is sized against is ASKED of the device (HSA's cache enumeration, level 3), not a constant
    assert info["infinity_cache_queried"] and info["infinity_cache_bytes"] == 256 << 20, info
    assert info["compute_units"] >= 64 and info["lds_per_workgroup"] >= 64 * 1024 and info["hbm_bytes"] > (100 << 30)
    cols = sum(eng.L.rg_column_bytes(eng.h, c) for c in range(17))
    assert cols <= info["engine_bytes"] <= cols + (64 << 10) + 4 * 5 * eng.stride * 8
    eng.close()


def test_allocation_failures_are_reported_not_fatal(rg):
    """A shard or an inflight window that does not fit in HBM comes back as RG_ERR_OUT_OF_MEMORY; the process and
    the device stay usable."""
    from raft_rs_amd.engine import EngineError, ERR
    with pytest.raises(EngineError) as e:
        rg.Engine(1 << 36, 8)  # 64 G groups x 8 peers: tens of terabytes of columns
    assert e.value.code == ERR["OUT_OF_MEMORY"]
    with pytest.raises(EngineError) as e:
        rg.Engine(4_000_000, 8, max_inflight=65535)  # 16 TB of rings
    assert e.value.code == ERR["OUT_OF_MEMORY"]
    with pytest.raises(EngineError) as e:
        rg.Engine(1000, 5, max_inflight=70000)
    assert e.value.code == ERR["INVALID_ARG"]
    eng = rg.Engine(1000, 5)  # still works
    eng.workload_init(2)
    eng.recompute()
    eng.results()
    eng.close()


def test_fused_launch_applies_elections(rg):
    """RG_MF_BECOME_LEADER inside a fused launch: the group's registers carry on with the new leader's state, the cold
    cells (RG_COL_CUR_TERM, the term-run table) are written in memory -- same result as a single-tick launch, including
    two elections of one group within the launch."""
    import torch
    G, P = 1000, 3
    engs = [rg.Engine(G, P) for _ in range(2)]
    for e in engs:
        e.workload_init(2)
    before = engs[0].read_state()
    ticks = []
    for t, term in enumerate((77, 78, 90)):
        mb = rg.MsgBuffers(G, P, engs[0].stride)
        if t != 1:
            mb.m_flags[::2, 0] = rg.MF.BECOME_LEADER
            mb.m_hint[0, :G] = term
        else:  # in between: the new leader persists its empty entry, a follower answers its first probe
            mb.m_flags[::2, 0] = rg.MF.VALID
            mb.m_index[0, :G] = before["term_hi"] + 1
            mb.m_flags[::2, 1] = rg.MF.VALID
            mb.m_index[1, :G] = before["term_hi"]
        cols = [torch.from_numpy(getattr(mb, k).view(np.int64).copy()).cuda() for k in ("m_index", "m_commit", "m_hint", "m_rs")]
        ticks.append((cols, torch.from_numpy(mb.m_flags.copy()).cuda()))
    out_t = torch.zeros((3, G), dtype=torch.int32, device="cuda")
    engs[0].tick_device_fused([[c.data_ptr() for c in cols] + [flags.data_ptr()] for cols, flags in ticks], out_t.data_ptr())
    engs[0].sync()
    o = out_t.cpu().numpy().view(np.uint32)
    for t, (cols, flags) in enumerate(ticks):
        engs[1].tick_device(*[c.data_ptr() for c in cols], flags.data_ptr())
        _, o1 = engs[1].results()
        assert np.array_equal(o[t], o1), t
    assert (o[0][::2] == (rg.OUT.BECAME_LEADER | rg.OUT.APPENDED)).all() and (o[0][1::2] == 0).all()
    assert (o[2][::2] == (rg.OUT.BECAME_LEADER | rg.OUT.APPENDED)).all()
    a, b = engs[0].read_state(), engs[1].read_state()
    for k in fuzz.STATE_KEYS:
        assert np.array_equal(a[k], b[k]), k
    for col in (rg.COL.CUR_TERM, rg.COL.RUN_FIRST, rg.COL.RUN_TERM):
        assert np.array_equal(engs[0].read_column(col), engs[1].read_column(col)), col
    assert (engs[0].read_column(rg.COL.CUR_TERM)[::2] == 90).all()
    assert (a["term_lo"][::2] == before["term_hi"][::2] + 2).all()
    for e in engs:
        e.close()


# ---------------------------------------------------------------------------------------------------------------
# the 64-bit-offset instantiations (k_tick_lane / _list / _fused / _compact <..., u64>): engines beyond 4 GiB per column
# run them; rg_config.flags = RG_CFGF_IX64 (rg_common.h: rg_ix32) makes every launch of a small engine take them, so the
# device code objects that ship are the ones that are tested
# ---------------------------------------------------------------------------------------------------------------
@pytest.fixture
def force_ix64():
    from raft_rs_amd import engine as E
    with E.config_defaults(flags=E.CFGF.IX64):
        yield


@pytest.mark.parametrize("variant", [1, 5])
@pytest.mark.parametrize("workload,n_slots", [(2, 3), (2, 5), (5, 7)])
def test_workload_stream_matches_oracle_with_64bit_offsets(rg, force_ix64, variant, workload, n_slots):
    test_workload_stream_matches_oracle(rg, variant, workload, n_slots)


@pytest.mark.parametrize("n_slots", [3, 5, 7, 8])
def test_random_streams_match_oracle_with_64bit_offsets(rg, force_ix64, n_slots):
    test_random_streams_match_oracle(rg, 1, n_slots)
    test_random_streams_match_oracle(rg, 5, n_slots)


# ---------------------------------------------------------------------------------------------------------------
# the third memory regime (k_tick_lane / k_tick_classes <.., NTM = 2>: state columns streamed too, loads and stores) is what
# engines far beyond the Infinity Cache run; rg_config.cache_policy = RG_CACHE_STREAM_ALL makes a small engine take it
# ---------------------------------------------------------------------------------------------------------------
@pytest.fixture
def force_nt_all():
    from raft_rs_amd import engine as E
    with E.config_defaults(cache_policy=E.CACHE.STREAM_ALL):
        yield


@pytest.mark.parametrize("workload,n_slots", [(2, 3), (2, 5), (3, 5), (5, 7)])
def test_workload_stream_matches_oracle_with_everything_streamed(rg, force_nt_all, workload, n_slots):
    test_workload_stream_matches_oracle(rg, 1, workload, n_slots)


def test_random_and_class_placed_streams_with_everything_streamed(rg, force_nt_all):
    for n_slots in (4, 7, 8):
        test_random_streams_match_oracle(rg, 1, n_slots)
    test_class_placed_shard_runs_as_one_launch_and_matches_oracle(rg, 7)
    test_sorted_mixed_workload_matches_oracle(rg, 7)


@pytest.fixture(params=[1, 7])
def force_resident(request):
    """k_tick_split: the first `param` workgroups' groups keep their state in the cache, the rest is streamed (both bodies in one
    launch; engines beyond the Infinity Cache). The engines of these tests have a few hundred to a few thousand groups: with 1 and
    7 resident workgroups both bodies run in every one of them."""
    from raft_rs_amd import engine as E
    with E.config_defaults(cache_policy=E.CACHE.RESIDENT, cache_resident_groups=256 * request.param):
        yield


@pytest.mark.parametrize("workload,n_slots", [(2, 3), (2, 5), (3, 5), (5, 7)])
def test_workload_stream_matches_oracle_partly_resident(rg, force_resident, workload, n_slots):
    test_workload_stream_matches_oracle(rg, 1, workload, n_slots)


def test_random_streams_partly_resident(rg, force_resident):
    for n_slots in (4, 7, 8):
        test_random_streams_match_oracle(rg, 1, n_slots)


def test_fused_and_sparse_kernels_with_64bit_offsets(rg, force_ix64):
    """k_tick_fused<..., u64> and k_tick_list<..., u64> on the GPU."""
    test_fused_launch_equals_sequential_ticks(rg, 2, 5, 4)
    test_fused_launch_equals_sequential_ticks(rg, 5, 7, 8)
    import test_sparse_path_gpu as S
    S.test_sparse_ticks_match_oracle(rg, 5)
    S.test_sparse_ticks_match_oracle(rg, 7)


# ---------------------------------------------------------------------------------------------------------------
# shards placed by replica-set size class: ONE launch whose blocks run the tick instantiated for the slots their groups
# have (k_tick_classes; the engine derives the ranges from RG_COL_CFG by itself, rg_size_classes reports them)
# ---------------------------------------------------------------------------------------------------------------
def _run_against_oracle(rg, eng, st, rng, ticks, label, **msg_kw):
    G, P = st["n_groups"], st["n_slots"]
    cl = oracle_from_state(st)
    msgs = O.alloc_msgs(G, P)
    gout = np.zeros(G, dtype=np.uint32)
    mb = rg.MsgBuffers(G, P, eng.stride)
    for t in range(ticks):
        cl.store_soa(st)
        fuzz.random_msgs(rng, st, msgs, **msg_kw)
        for k in ("m_index", "m_commit", "m_hint", "m_rs", "m_flags"):
            getattr(mb, k)[...] = msgs[k]
        eng.tick(mb)
        cl.tick_soa(msgs, gout)
        assert_same(eng, cl, st, gout, f"{label} tick {t}")
    return cl


@pytest.mark.parametrize("n_slots", [5, 7, 8])
def test_class_placed_shard_runs_as_one_launch_and_matches_oracle(rg, n_slots):
    """Groups of 3, 5 and 7 (8) peers in contiguous ranges of one engine -- the boundaries NOT on multiples of 64, so the blocks
    that straddle one take the larger class. The engine finds the ranges itself; random traffic (rejects, heartbeats, elections,
    malformed acks, events on slots a group does not have) against the oracle; then rg_set_config grows one group of the
    3-peer range to the engine's last slot (its block leaves the class), rg_restore brings the old words back (re-derived),
    and an engine created with RG_CFGF_NO_SIZE_CLASSES -- the plain kernel -- ends in the same state."""
    rng = np.random.default_rng(8800 + n_slots)
    sizes = [q for q in (3, 5, 7) if q < n_slots] + [n_slots]
    ranges = [(64 * 20 + 11, sizes[0])] + [(64 * 13 + 5, q) for q in sizes[1:-1]] + [(64 * 9 + 40, sizes[-1])]
    G = sum(n for n, _ in ranges)
    st = O.alloc_state(G, n_slots)
    st["cfg"][:] = fuzz.class_placed_cfg(rng, ranges, n_slots, missing_progress_frac=0.05)
    fuzz.random_state(rng, st, small_values=True)
    eng = rg.Engine(G, n_slots)
    eng.load_state(st)
    # what the engine derived: one range per size, ends rounded UP to the block that still holds a smaller group
    cls = eng.size_classes()
    assert [q for _, _, q in cls] == sizes, cls
    assert cls[0][0] == 0 and sum(n for _, n, _ in cls) == G
    first = 0
    for (f, n, q), (rn, rq) in zip(cls, ranges):
        assert f == (first // 64) * 64 and q == rq, (cls, ranges)  # a class starts with the block its first group lies in
        first += rn
    plain = rg.Engine(G, n_slots, flags=rg.CFGF.NO_SIZE_CLASSES)
    plain.load_state(st)
    assert plain.size_classes() == []
    st_plain = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in st.items()}
    kw = dict(reject_p=0.2, heartbeat_p=0.1, elect_p=0.05, elect_term=50)
    rng2 = np.random.default_rng(99)
    _run_against_oracle(rg, eng, st, np.random.default_rng(99), 5, f"classes P={n_slots}", **kw)
    _run_against_oracle(rg, plain, st_plain, rng2, 5, f"plain P={n_slots}", **kw)
    a, b = eng.read_state(), plain.read_state()
    assert not fuzz.diff_states(a, b, G, n_slots)
    # a conf change that names a slot beyond the group's class: its block leaves the class (more ranges, same results)
    eng.checkpoint()
    g = 70
    full = (1 << n_slots) - 1
    eng.set_config(g, rg.engine.cfg_make(full, 0, 0, False, 0, full))
    cls2 = eng.size_classes()
    assert len(cls2) == len(cls) + 2 and cls2[1] == (64, 64, n_slots), cls2
    st2 = eng.read_state()
    # (the oracle restarts from the engine's columns at TERM: no elections here, their terms would no longer line up)
    _run_against_oracle(rg, eng, st2, np.random.default_rng(5), 3, f"classes after conf change P={n_slots}", reject_p=0.2, heartbeat_p=0.1)
    eng.restore()
    assert eng.size_classes() == cls
    # a conf change INSIDE the class keeps the table
    eng.set_config(g, int(st["cfg"][g]))
    assert eng.size_classes() == cls
    eng.close()
    plain.close()


def test_interleaved_sizes_are_not_a_class_placed_shard(rg):
    """Sizes alternating group by group (BASELINE config 5's one-engine layout): every block names every slot, the plain kernel
    runs, nothing else changes."""
    G = 20000
    eng = rg.Engine(G, 7)
    eng.workload_init(5)
    assert eng.size_classes() == []
    eng.close()


@pytest.mark.parametrize("n_slots", [7, 8])
def test_sorted_mixed_workload_matches_oracle(rg, n_slots):
    """BASELINE config 5's population placed by size class (RG_WL_PLACE_SORTED): the same groups as the interleaved layout --
    group for group, after undoing the placement -- in three ranges the engine runs as one launch; 6 ticks of the rollover
    stream (elections, rejects, probes) against the oracle, messages from the host twin of the generator."""
    from raft_rs_amd import engine as E
    G = 30000 + 77
    eng = rg.Engine(G, n_slots)
    eng.workload_init(5, sorted_classes=True)
    cls = eng.size_classes()
    assert [q for _, _, q in cls] == [3, 5, 7], cls
    n0, n1 = (G + 2) // 3, (G + 1) // 3
    assert cls[1][0] == (n0 // 64) * 64 and cls[2][0] == ((n0 + n1) // 64) * 64
    st = eng.read_state()
    # the same population as the interleaved layout
    ref = rg.Engine(G, n_slots)
    ref.workload_init(5)
    st_i = ref.read_state()
    ref.close()
    place = np.concatenate([np.arange(0, G, 3), np.arange(1, G, 3), np.arange(2, G, 3)])
    for k in ("commit", "term_lo", "term_hi", "cfg"):
        assert (st[k] == st_i[k][place]).all(), k
    assert (st["match"][:, :G] == st_i["match"][:, :G][:, place]).all()
    cl = oracle_from_state(st)
    msgs = rg.MsgBuffers(G, n_slots, eng.stride)
    gout = np.zeros(G, dtype=np.uint32)
    n_elect = 0
    for t in range(6):
        cl.store_soa(st)
        E.workload_gen_host(st, msgs, 5, t, sorted_classes=True)
        eng.tick(msgs)
        cl.tick_soa(msgs.as_dict(), gout)
        assert_same(eng, cl, st, gout, f"sorted workload 5 P={n_slots} tick {t}")
        n_elect += int(((gout & 0x10) != 0).sum())
    assert n_elect > G // 40
    eng.close()
