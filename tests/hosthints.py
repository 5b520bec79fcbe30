"""RG_OUT_HOST_HINT in the tests -- TEST INFRASTRUCTURE (numpy + the oracle; nothing here is product code).

The engine's term-run table is bounded (RG_TERM_RUNS runs of older terms, the newest ones), RaftLog is not
(src/raft_log.rs:122-140). A reject whose find_conflict_by_term walk (src/raft_log.rs:209-235, applied at
src/raft.rs:1562,1657-1660) needs a term the table no longer holds is not applied by the tick: the group's result word carries
RG_OUT_HOST_HINT, RG_COL_HOST_HINT names the slots, and the host -- here: the oracle, whose log is complete -- resolves the hint
and steps the reject again without its log term. Two things live here:

  settle()    the host's half: resolve and re-step, so that a test can go on comparing whole states with the oracle;
  expected()  an independent restatement of WHICH rejects must come back, from the oracle's complete log: the literal walk of the
              reference, one index at a time.
"""
import ctypes as C

import numpy as np

import fuzz
import oracle_lib as O

OUT_HOST_HINT = 0x20
MF_VALID, MF_REJECT, MF_HAS_RS, MF_HEARTBEAT, MF_HAS_LOGTERM = 0x01, 0x02, 0x04, 0x40, 0x80


def settle(cl, msgs, out, host_hint, tick=None, resolve=None):
    """`out` (u32[G], the engine's result words of the tick that consumed `msgs`) and `host_hint` (u8[G], RG_COL_HOST_HINT after
    it): for every reject the engine left to the host, find_conflict_by_term on the ORACLE's complete log (which has already taken
    the whole tick; entries up to the last_index of message time are the same ones), then either
      tick(msgs2)      one more engine tick carrying exactly those rejects with the resolved hint and no log term (returns its
                       result words, which are merged into the original tick's), or
      resolve(records) rg_resolve_host_hints with [(group, index, hint, slot, 0), ...] (returns RG_COL_OUT afterwards).
    Returns (the completed result words, the number of rejects settled): what the reference reports for the original tick."""
    G, P = msgs["n_groups"], msgs["n_slots"]
    flagged = np.nonzero(out & OUT_HOST_HINT)[0]
    merged = out & ~np.uint32(OUT_HOST_HINT)
    if flagged.size == 0:
        return merged, 0
    L = O.lib()
    m2 = O.alloc_msgs(G, P, stride=msgs["stride"])
    recs = []
    for g in flagged:
        mask = int(host_hint[g])
        assert mask, ("RG_OUT_HOST_HINT without a slot in RG_COL_HOST_HINT", int(g))
        for s in range(P):
            if not (mask >> s) & 1:
                continue
            f = int(msgs["m_flags"][g, s])
            assert f & (MF_VALID | MF_REJECT | MF_HAS_LOGTERM) == (MF_VALID | MF_REJECT | MF_HAS_LOGTERM), (int(g), s, hex(f))
            lt = int(msgs["m_logterm"][s, g])
            assert lt > 0
            hint = L.ro_log_find_conflict_by_term(cl.h, int(g), int(msgs["m_hint"][s, g]), lt)
            m2["m_flags"][g, s] = MF_VALID | MF_REJECT
            m2["m_index"][s, g] = msgs["m_index"][s, g]
            m2["m_commit"][s, g] = msgs["m_commit"][s, g]
            m2["m_hint"][s, g] = hint
            recs.append((int(g), int(msgs["m_index"][s, g]), hint, s, 0))
    if resolve is not None:
        out3 = resolve(recs)
        assert (out3 & OUT_HOST_HINT).sum() == 0
        return out3, len(recs)
    out2 = tick(m2)
    assert (out2 & OUT_HOST_HINT).sum() == 0
    return merged | out2, len(recs)


def settle_engine(eng, cl, msgs):
    """The host's half against a real engine (raft_rs_amd.Engine) after a tick of `msgs` that the oracle `cl` has taken too:
    rg_host_hints names the groups, RG_COL_HOST_HINT the slots (both must agree), rg_resolve_host_hints completes them.
    Returns the number of rejects settled."""
    from raft_rs_amd.engine import COL
    out = eng.read_column(COL.OUT)
    if not (out & OUT_HOST_HINT).any():
        assert len(eng.host_hints()) == 0
        return 0
    hh = eng.read_column(COL.HOST_HINT)
    listed = {int(r["group"]): int(r["slot_mask"]) for r in eng.host_hints()}
    assert listed == {int(g): int(hh[g]) for g in np.nonzero(out & OUT_HOST_HINT)[0]}

    def resolve(recs):
        applied = eng.resolve_host_hints(recs)
        assert applied.all(), "a reject the tick left to the host is never stale: maybe_decr_to applies"
        return eng.read_column(COL.OUT)

    _, n = settle(cl, msgs, out, hh, resolve=resolve)
    assert len(eng.host_hints()) == 0
    return n


def log_runs(cl, g, cap=4096):
    """The oracle's complete log of group g: ([(first, term), ...] oldest first, (dummy_index, dummy_term))."""
    first = (C.c_uint64 * cap)()
    term = (C.c_uint64 * cap)()
    di, dt = C.c_uint64(0), C.c_uint64(0)
    fn = O.lib().ro_group_log_runs
    fn.restype = C.c_size_t
    fn.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_size_t, C.POINTER(C.c_uint64),
                   C.POINTER(C.c_uint64)]
    n = fn(cl.h, g, first, term, cap, C.byref(di), C.byref(dt))
    assert n <= cap
    return [(int(first[k]), int(term[k])) for k in range(n)], (int(di.value), int(dt.value))


def expected(cl, st_before, st_after, msgs, table_runs=O.TERM_RUNS):
    """Which (group, slot) rejects of the tick `msgs` must come back as RG_OUT_HOST_HINT, restated from the reference alone:
    `st_before` / `st_after` are the oracle's SoA projections around the tick (the oracle has already taken it), its complete log
    is read through log_runs(). A reject comes back iff
      * the reference reads its hint at all -- maybe_decr_to's Probe / Snapshot branch, not stale, no snapshot request
        (progress.rs:188-203), in the state the message meets (an election of the same tick resets every follower first), and
      * the literal walk of find_conflict_by_term (one index at a time) evaluates term() at an index the bounded table cannot
        answer -- above the dummy entry and below the first entry of the newest `table_runs` older runs -- and log-term
        monotonicity does not decide that step either (dummy_term <= log_term < the term of that first known entry).
    Returns {group: slot mask}."""
    G, P = msgs["n_groups"], msgs["n_slots"]
    L = O.lib()
    want = {}
    cfg = st_before["cfg"]
    for g in range(G):
        self_slot = (int(cfg[g]) >> 16) & 7
        present = (int(cfg[g]) >> 24) & 0xff
        fl = msgs["m_flags"][g]
        cand = [s for s in range(P) if s != self_slot and (present >> s) & 1 and not fl[s] & MF_HEARTBEAT
                and fl[s] & (MF_VALID | MF_REJECT | MF_HAS_LOGTERM) == (MF_VALID | MF_REJECT | MF_HAS_LOGTERM)
                and int(msgs["m_logterm"][s, g]) > 0]
        if not cand:
            continue
        runs, (dummy, dummy_term) = log_runs(cl, g)
        cur_term = int(st_after["cur_term"][g])
        own = bool(runs) and runs[-1][1] == cur_term  # the leader's own run is the log's last run when it carries the current term
        older = runs[:-1] if own else runs
        kept = older[-table_runs:] if table_runs else []
        if kept:
            known, known_term = kept[0]
        elif own:
            known, known_term = runs[-1]
        else:
            known, known_term = int(st_after["term_hi"][g]) + 1, cur_term
        elected = int(st_after["cur_term"][g]) != int(st_before["cur_term"][g])
        hi_before = int(st_before["term_hi"][g]) + (1 if elected else 0)
        hi_after = int(st_after["term_hi"][g])
        for s in cand:
            idx, rs = int(msgs["m_index"][s, g]), int(msgs["m_rs"][s, g]) if fl[s] & MF_HAS_RS else 0
            state = 0 if elected else int(st_before["pflags"][g, s]) & 3
            nx = int(st_before["term_hi"][g]) + 1 if elected else int(st_before["next"][s, g])
            if state == 1 or rs != 0 or nx == 0 or nx - 1 != idx:
                continue  # the reference never reads this hint
            lt, ci = int(msgs["m_logterm"][s, g]), int(msgs["m_hint"][s, g])
            last = hi_after if s > self_slot else hi_before
            if ci > last:
                continue  # "out of range": returned as is, no walk
            need = False
            while True:
                if dummy < ci < known and dummy_term <= lt < known_term:
                    need = True
                    break
                if L.ro_log_term(cl.h, g, ci) <= lt:
                    break
                ci = (ci - 1) & O.U64_MAX
            if need:
                want[g] = want.get(g, 0) | (1 << s)
    return want


def _copy(st):
    return {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in st.items()}


def spread_reject_hints(rng, st, msgs, max_term):
    """Make the rejects of a tick probe the WHOLE log: reject_hint anywhere in [dummy_index, Message.index], log_term anywhere
    in [1, max_term] -- so that find_conflict_by_term walks start and end in every run of the history."""
    G, P = st["n_groups"], st["n_slots"]
    for p in range(P):
        rej = (msgs["m_flags"][:, p] & 0x83) == 0x83  # VALID | REJECT | HAS_LOGTERM
        lo = st["dummy_index"].astype(np.int64)
        hi = np.maximum(msgs["m_index"][p, :G].astype(np.int64), lo)
        h = lo + (rng.random(G) * (hi - lo + 1)).astype(np.int64)
        msgs["m_hint"][p, :G] = np.where(rej, h, msgs["m_hint"][p, :G].astype(np.int64)).astype(np.uint64)
        msgs["m_logterm"][p, :G] = np.where(rej, rng.integers(1, max_term + 1, size=G), msgs["m_logterm"][p, :G]).astype(np.uint64)


def deep_history_run(n_slots, seed, tick, read_hints, load, G=2500, ticks=9, resolve=None):
    """The scenario of test_log_history_deeper_than_the_term_run_table for any engine form: `load(st)` hands over the initial
    state, `tick(msgs)` runs one tick and returns its result words, `read_hints()` RG_COL_HOST_HINT (u8[G]). Returns the oracle
    cluster, the oracle's final projection and the counters the caller asserts on."""
    rng = np.random.default_rng(seed)
    TERM = 30
    st = O.add_term_table(O.alloc_state(G, n_slots))
    st["cfg"][:] = fuzz.random_cfg(rng, G, n_slots, missing_progress_frac=0.03)
    fuzz.random_state(rng, st, probe_frac=0.5, base=200)
    fuzz.random_term_table(rng, st, TERM, min_runs=O.TERM_RUNS)
    cl = O.Cluster(G)
    cl.load_soa(st, term=TERM)
    load(_copy(st))
    msgs = O.alloc_msgs(G, n_slots)
    gout = np.zeros(G, dtype=np.uint32)
    stats = {"settled": 0, "applied_logterm_rejects": 0, "max_runs": 0, "depths": set()}
    for t in range(ticks):
        cl.store_soa(st)
        st_before = _copy(st)
        # half of the groups elect in every tick (each election files one more run: after a few ticks the histories are 9 to
        # 8 + ticks runs deep); the other half's followers sit in Probe at next = last_index + 1 and get their probes rejected
        fuzz.random_msgs(rng, st, msgs, valid_p=0.8, reject_p=0.6, rs_p=0.05, sent_p=0.2, heartbeat_p=0.05, logterm_max=TERM + t,
                         elect_p=0.5, elect_term=TERM + 1 + t)
        spread_reject_hints(rng, st, msgs, TERM + 1 + t)
        out = tick(msgs)
        cl.tick_soa(msgs, gout)
        cl.store_soa(st)
        want = expected(cl, st_before, st, msgs)
        hh = read_hints()
        flagged = np.nonzero(out & OUT_HOST_HINT)[0]
        got = {int(g): int(hh[g]) for g in flagged}
        assert got == want, (t, sorted(set(got.items()) ^ set(want.items()))[:5])
        # wherever the bit is clear the tick IS the reference's: nothing below has been touched by the host yet
        clear = (out & OUT_HOST_HINT) == 0
        assert (out[clear] == gout[clear]).all(), (t, np.nonzero(clear & (out != gout))[0][:5])
        # the host's answer: stepped again as a tick of their own, or (every other tick) through rg_resolve_host_hints
        merged, n = settle(cl, msgs, out, hh, tick, resolve if t % 2 else None)
        stats["settled"] += n
        assert (merged == gout).all(), (t, np.nonzero(merged != gout)[0][:5])
        stats["applied_logterm_rejects"] += int((((msgs["m_flags"] & 0x83) == 0x83).sum(axis=1)[clear]).sum())
        yield cl, st, t
        for g in want:
            runs, _ = log_runs(cl, g)
            stats["depths"].add(len(runs))
        stats["max_runs"] = max(stats["max_runs"], max(len(log_runs(cl, g)[0]) for g in range(0, G, 97)))
    yield stats
