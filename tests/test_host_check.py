"""CPU-only differential test of the ENGINE'S OWN per-group arithmetic: tests/host_check/ compiles
raft_rs_amd/csrc/rg_group.h (the header the HIP kernels inline) for the host and this file diffs it
against the oracle on seeded random streams. It catches arithmetic regressions without a GPU; the GPU
parity tests (-m gpu) remain the gate for the kernels themselves. Nothing here is product code."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

import fuzz
import hosthints
import oracle_lib as O
import sendstage

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "host_check", "host_tick.hip")
LIB = os.path.join(HERE, "host_check", "libhost_tick.so")
CSRC = os.path.join(os.path.dirname(HERE), "raft_rs_amd", "csrc")


def build_lib():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        return None
    deps = [SRC] + [os.path.join(CSRC, f) for f in ("rg_group.h", "rg_common.h", "rg_tick_kernels.h", "rg_send.h", "rg_publish.h")]
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        # no -march=native: the built .so travels with gpurun snapshots to hosts with other CPUs
        subprocess.check_call([hipcc, "-O3", "-std=c++17", "-shared", "-fPIC", "--offload-arch=gfx950",
                               "-Wno-pass-failed", "-pthread", SRC, "-o", LIB])
    return LIB


STATE_ORDER = ("match", "next", "pr_commit", "pend_snap", "pend_rs", "gid", "pflags", "commit", "term_lo", "term_hi",
               "cfg")
TABLE_ORDER = ("run_first", "run_term", "dummy_index", "dummy_term")
MSG_ORDER = ("m_index", "m_commit", "m_hint", "m_rs", "m_flags", "m_logterm")
_ZERO = {}


def state_ptrs(st, out):
    """19 pointers in the order host_tick.hip expects; a zero table when the state has no term-run table. The
    RG_COL_HOST_HINT byte column is created in the state dict on first use (st["host_hint"])."""
    key = (st["n_groups"], st["stride"])
    if key not in _ZERO:
        _ZERO[key] = (np.zeros((O.TERM_RUNS, st["stride"]), dtype=np.uint64), np.zeros(st["n_groups"], dtype=np.uint64))
    z4, zg = _ZERO[key]
    table = [st.get("run_first", z4), st.get("run_term", z4), st.get("dummy_index", zg), st.get("dummy_term", zg),
             st.get("cur_term", zg)]
    if "host_hint" not in st or "run_count" not in st or st["run_count"].ctypes.data != st["host_hint"].ctypes.data + st["stride"]:
        # RG_COL_HOST_HINT and, `stride` bytes behind it, RG_COL_RUN_COUNT (engine-owned, derived by the twin on its way in): one
        # allocation, as in the engine's arena (rg_run_n)
        both = np.zeros(2 * st["stride"], dtype=np.uint8)
        if "host_hint" in st:
            both[:st["n_groups"]] = st["host_hint"][:st["n_groups"]]
        st["_hint_and_count"] = both
        st["host_hint"] = both[:st["n_groups"]]
        st["run_count"] = both[st["stride"]:st["stride"] + st["n_groups"]]
    cols = [st[k] for k in STATE_ORDER] + [out] + table + [st["host_hint"], st["run_count"]]
    return (C.c_void_p * 19)(*[c.ctypes.data for c in cols])


def msg_ptrs(msgs):
    key = ("m", msgs["m_index"].shape)
    if key not in _ZERO:
        _ZERO[key] = np.zeros(msgs["m_index"].shape, dtype=np.uint64)
    cols = [msgs[k] for k in MSG_ORDER[:5]] + [msgs.get("m_logterm", _ZERO[key]) if isinstance(msgs, dict)
                                                 else msgs["m_logterm"]]
    return (C.c_void_p * 6)(*[c.ctypes.data for c in cols])


@pytest.fixture(scope="module")
def host_tick():
    if build_lib() is None:
        pytest.skip("hipcc not available")
    fn = C.CDLL(LIB).rg_host_check_tick
    fn.restype = C.c_int
    fn.argtypes = [C.c_uint, C.c_ulong, C.c_ulong, C.c_void_p, C.c_void_p, C.c_int, C.c_ulong, C.c_ulong]

    def tick(st, msgs, out, gc):
        rc = fn(st["n_slots"], st["n_groups"], st["stride"], state_ptrs(st, out), msg_ptrs(msgs), int(gc), 0,
                st["n_groups"])
        assert rc == 0
    return tick


@pytest.fixture(scope="module")
def host_fused():
    if build_lib() is None:
        pytest.skip("hipcc not available")
    fn = C.CDLL(LIB).rg_host_check_fused
    fn.restype = C.c_int
    fn.argtypes = [C.c_uint, C.c_ulong, C.c_ulong, C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]

    def fused(st, ticks, out_last, out_t, commit_t, gc):
        per_tick = [msg_ptrs(t) for t in ticks]
        arr = (C.c_void_p * len(ticks))(*[C.addressof(p) for p in per_tick])
        rc = fn(st["n_slots"], st["n_groups"], st["stride"], state_ptrs(st, out_last), len(ticks), arr,
                out_t.ctypes.data, commit_t.ctypes.data, int(gc))
        assert rc == 0
    return fused


def copy_state(st):
    return {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in st.items()}


RESOLVED_DTYPE = np.dtype([("group", "<u8"), ("index", "<u8"), ("hint", "<u8"), ("slot", "<u4"), ("reserved", "<u4")])


def twin_resolve(st, out, recs, meta=None):
    """rg_resolve_host_hints on the host twin: `out` is the twin's RG_COL_OUT (completed in place). Returns applied u8[n]."""
    fn = C.CDLL(LIB).rg_host_check_resolve_hints
    fn.restype = C.c_int
    fn.argtypes = [C.c_uint, C.c_ulong, C.c_ulong, C.c_void_p, C.c_void_p, C.c_void_p, C.c_ulong, C.c_void_p]
    r = np.array(recs, dtype=RESOLVED_DTYPE)
    applied = np.zeros(len(r), dtype=np.uint8)
    assert fn(st["n_slots"], st["n_groups"], st["stride"], state_ptrs(st, out), None if meta is None else meta.ctypes.data,
              r.ctypes.data, len(r), applied.ctypes.data) == 0
    return applied


@pytest.mark.parametrize("gc", [False, True])
@pytest.mark.parametrize("n_slots", [1, 2, 3, 4, 5, 6, 7, 8])
def test_device_arithmetic_on_host_matches_oracle(host_tick, n_slots, gc):
    rng = np.random.default_rng(4000 + n_slots + (100 if gc else 0))
    G = 3000
    st = O.alloc_state(G, n_slots)
    st["cfg"][:] = fuzz.random_cfg(rng, G, n_slots, missing_progress_frac=0.05, group_commit_frac=0.5 if gc else 0.0)
    fuzz.random_state(rng, st, small_values=True, with_gids=gc)
    cl = O.Cluster(G)
    cl.load_soa(st, term=6)
    eng_st = copy_state(st)
    msgs = O.alloc_msgs(G, n_slots)
    gout = np.zeros(G, dtype=np.uint32)
    out = np.zeros(G, dtype=np.uint32)
    for t in range(6):
        cl.store_soa(st)
        fuzz.random_msgs(rng, st, msgs, malformed_p=0.03 if t == 4 else 0.0)
        host_tick(eng_st, msgs, out, gc)
        cl.tick_soa(msgs, gout)
        cl.store_soa(st)
        diffs = fuzz.diff_states(st, eng_st, G, n_slots)
        assert not diffs, (t, diffs[:6])
        assert (out == gout).all(), (t, np.nonzero(out != gout)[0][:5])


@pytest.mark.parametrize("gc", [False, True])
@pytest.mark.parametrize("n_slots", [3, 5, 7, 8])
def test_acks_from_paused_peers_against_the_sequential_reference(host_tick, n_slots, gc):
    """RgTick::paused_acks decides `else if old_paused { send_append }` (raft.rs:1749-1751) from two quorum evaluations
    per paused ack instead of replaying the sequence: streams where HALF the accepted acks come from paused peers (full
    windows, paused probes), joint and majority configurations, small values so that ties and commits at every position
    of the sequence occur, some malformed acks (the replay fallback).
    gc (round 6): the same with group commit on in EVERY group and random commit groups -- the replay of
    RgTick::commit_phase_gc, whose every step is the literal group-commit evaluation through one inlined call site."""
    rng = np.random.default_rng(9100 + n_slots + (50 if gc else 0))
    G = 4000
    st = O.alloc_state(G, n_slots)
    st["cfg"][:] = fuzz.random_cfg(rng, G, n_slots, missing_progress_frac=0.02, group_commit_frac=1.0 if gc else 0.0)
    fuzz.random_state(rng, st, small_values=True, with_gids=gc)
    cl = O.Cluster(G)
    cl.load_soa(st, term=6)
    eng_st = copy_state(st)
    msgs = O.alloc_msgs(G, n_slots)
    gout = np.zeros(G, dtype=np.uint32)
    out = np.zeros(G, dtype=np.uint32)
    n_paused_changed = 0
    for t in range(14):
        cl.store_soa(st)
        fuzz.random_msgs(rng, st, msgs, malformed_p=0.02 if t % 5 == 4 else 0.0)
        f = msgs["m_flags"]
        accept = ((f & 3) == 1)  # VALID, not REJECT
        f[...] = np.where(accept & (rng.random(f.shape) < 0.5), f | fuzz.MF_INS_FULL, f)
        host_tick(eng_st, msgs, out, gc)
        cl.tick_soa(msgs, gout)
        cl.store_soa(st)
        diffs = fuzz.diff_states(st, eng_st, G, n_slots)
        assert not diffs, (t, diffs[:6])
        assert (out == gout).all(), (t, np.nonzero(out != gout)[0][:5], [hex(x) for x in out[out != gout][:5]],
                                     [hex(x) for x in gout[out != gout][:5]])
        n_paused_changed += int((((gout & 1) != 0) & (((gout >> 8) & 0xff) != 0)).sum())
    assert n_paused_changed > (100 if gc else 500), n_paused_changed  # commits AND owed send_appends in the same tick: the decided case


def test_device_arithmetic_near_the_top_of_the_index_range(host_tick):
    """All indices shifted by 2**62 (the reference computes n + 1 on u64: indices < 2**63)."""
    rng = np.random.default_rng(99)
    G, P = 3000, 5
    st = O.alloc_state(G, P)
    st["cfg"][:] = fuzz.random_cfg(rng, G, P)
    fuzz.random_state(rng, st, small_values=True, base=2 ** 62)
    cl = O.Cluster(G)
    cl.load_soa(st, term=6)
    eng_st = copy_state(st)
    msgs = O.alloc_msgs(G, P)
    gout = np.zeros(G, dtype=np.uint32)
    out = np.zeros(G, dtype=np.uint32)
    for t in range(5):
        cl.store_soa(st)
        fuzz.random_msgs(rng, st, msgs)
        host_tick(eng_st, msgs, out, False)
        cl.tick_soa(msgs, gout)
        cl.store_soa(st)
        assert not fuzz.diff_states(st, eng_st, G, P), t
        assert (out == gout).all()
    assert (st["commit"] >= 2 ** 62).mean() > 0.5


@pytest.mark.parametrize("gc", [False, True])
@pytest.mark.parametrize("workload,n_slots", [(2, 5), (3, 5), (5, 7)])
def test_device_arithmetic_on_host_runs_the_workloads(host_tick, workload, n_slots, gc):
    """(gc: the same streams with group commit on in every group and three commit groups over the peers, RG_WL_GROUP_COMMIT --
    what the bench's group-commit figure and tests/test_full_size_gpu.py run at 1 M groups)"""
    from raft_rs_amd import engine as E
    G = 4000
    st = O.add_term_table(O.alloc_state(G, n_slots))  # (its own term columns: config 5's elections write them)
    E.workload_init_host(st, workload, group_commit=gc)
    st["cur_term"][:] = 6
    assert bool((st["cfg"] >> 19 & 1).all()) == gc and bool(st["gid"].any()) == gc
    cl = O.Cluster(G)
    cl.load_soa(st, term=6)
    eng_st = copy_state(st)
    mb = E.MsgBuffers(G, n_slots, st["stride"])
    gout = np.zeros(G, dtype=np.uint32)
    out = np.zeros(G, dtype=np.uint32)
    for t in range(6):
        E.workload_gen_host(st, mb, workload, t, group_commit=gc)
        host_tick(eng_st, mb.as_dict(), out, gc)
        cl.tick_soa(mb.as_dict(), gout)
        cl.store_soa(st)
        assert not fuzz.diff_states(st, eng_st, G, n_slots)
        assert (out == gout).all()


@pytest.mark.parametrize("gc", [False, True])
@pytest.mark.parametrize("n_slots,T", [(1, 2), (3, 8), (5, 4), (7, 3), (8, 5)])
def test_fused_ticks_equal_sequential_ticks(host_fused, n_slots, T, gc):
    """k_tick_fused's per-lane sequence (state in registers across T ticks, one write-back) against the
    oracle stepping the same T ticks one at a time: final state, every tick's result word and commit."""
    rng = np.random.default_rng(7000 + n_slots * 10 + T + (100 if gc else 0))
    G = 2500
    st = O.alloc_state(G, n_slots)
    st["cfg"][:] = fuzz.random_cfg(rng, G, n_slots, missing_progress_frac=0.05, group_commit_frac=0.5 if gc else 0.0)
    fuzz.random_state(rng, st, small_values=True, with_gids=gc)
    cl = O.Cluster(G)
    cl.load_soa(st, term=6)
    eng_st = copy_state(st)
    for rnd in range(2):
        ticks, want_out, want_commit = [], [], []
        gout = np.zeros(G, dtype=np.uint32)
        for t in range(T):
            cl.store_soa(st)
            msgs = O.alloc_msgs(G, n_slots)
            fuzz.random_msgs(rng, st, msgs, malformed_p=0.02 if t == 1 else 0.0)
            cl.tick_soa(msgs, gout)
            cl.store_soa(st)
            ticks.append(msgs)
            want_out.append(gout.copy())
            want_commit.append(st["commit"].copy())
        out_t = np.zeros((T, G), dtype=np.uint32)
        commit_t = np.zeros((T, G), dtype=np.uint64)
        out_last = np.zeros(G, dtype=np.uint32)
        host_fused(eng_st, ticks, out_last, out_t, commit_t, gc)
        for t in range(T):
            bad = np.nonzero(out_t[t] != want_out[t])[0]
            assert bad.size == 0, (rnd, t, bad[:5], [hex(x) for x in out_t[t][bad[:5]]], [hex(x) for x in want_out[t][bad[:5]]])
            assert (commit_t[t] == want_commit[t]).all(), (rnd, t)
        assert (out_last == want_out[-1]).all()
        diffs = fuzz.diff_states(st, eng_st, G, n_slots)
        assert not diffs, (rnd, diffs[:6])


@pytest.mark.parametrize("n_slots,T", [(3, 8), (5, 4), (7, 6)])
def test_fused_ticks_with_elections_equal_sequential_ticks(host_fused, n_slots, T):
    """RG_MF_BECOME_LEADER inside a fused launch (k_tick_fused keeps the group in registers across the ticks and applies
    the election to them; RG_COL_CUR_TERM and the term-run table are written in memory): ~15 % of the groups elect per
    tick, several times per launch, next to ordinary traffic -- every tick's result word and commit index, the final
    state and the term table against the oracle stepping the ticks one at a time."""
    rng = np.random.default_rng(7700 + n_slots)
    G, TERM = 3000, 9
    st = O.add_term_table(O.alloc_state(G, n_slots))
    st["cfg"][:] = fuzz.random_cfg(rng, G, n_slots, missing_progress_frac=0.05, transfer_frac=0.3)
    fuzz.random_state(rng, st, small_values=True, probe_frac=0.3)
    fuzz.random_term_table(rng, st, TERM)
    cl = O.Cluster(G)
    cl.load_soa(st, term=TERM)
    eng_st = copy_state(st)
    term = TERM
    elections = 0
    for rnd in range(2):
        ticks, want_out, want_commit = [], [], []
        gout = np.zeros(G, dtype=np.uint32)
        for t in range(T):
            term += 1
            cl.store_soa(st)
            msgs = O.alloc_msgs(G, n_slots)
            fuzz.random_msgs(rng, st, msgs, reject_p=0.2, elect_p=0.15, elect_term=term)
            cl.tick_soa(msgs, gout)
            cl.store_soa(st)
            ticks.append(msgs)
            want_out.append(gout.copy())
            want_commit.append(st["commit"].copy())
            elections += int(((gout & 0x10) != 0).sum())
        out_t = np.zeros((T, G), dtype=np.uint32)
        commit_t = np.zeros((T, G), dtype=np.uint64)
        out_last = np.zeros(G, dtype=np.uint32)
        host_fused(eng_st, ticks, out_last, out_t, commit_t, False)
        for t in range(T):
            bad = np.nonzero(out_t[t] != want_out[t])[0]
            assert bad.size == 0, (rnd, t, bad[:5], [hex(x) for x in out_t[t][bad[:5]]], [hex(x) for x in want_out[t][bad[:5]]])
            assert (commit_t[t] == want_commit[t]).all(), (rnd, t)
        diffs = fuzz.diff_states(st, eng_st, G, n_slots)
        assert not diffs, (rnd, diffs[:6])
        for k in ("run_first", "run_term", "cur_term"):
            assert (st[k] == eng_st[k]).all(), (rnd, k)
    assert elections > G, elections


@pytest.mark.parametrize("n_slots", [1, 3, 5, 8])
def test_elections_between_and_inside_ticks(host_tick, n_slots):
    """RG_MF_BECOME_LEADER (Raft::reset + become_leader, raft.rs:942-971,1151-1202) on ~15% of the groups per tick,
    several elections per group, together with ordinary traffic of the same tick (acks, rejects with and without
    Message.log_term, heartbeat responses, appends): every column incl. term_lo and the cfg word's transferee, the
    result word, and the term-run table (pushed by every election, its oldest run dropped when full -- the oracle keeps
    its whole log, and the rejects the shortened table cannot answer come back as RG_OUT_HOST_HINT)."""
    rng = np.random.default_rng(4400 + n_slots)
    G, TERM = 4000, 9
    st = O.add_term_table(O.alloc_state(G, n_slots))
    st["cfg"][:] = fuzz.random_cfg(rng, G, n_slots, missing_progress_frac=0.05, transfer_frac=0.3)
    fuzz.random_state(rng, st, small_values=True, probe_frac=0.3)
    fuzz.random_term_table(rng, st, TERM)
    cl = O.Cluster(G)
    cl.load_soa(st, term=TERM)
    eng_st = copy_state(st)
    msgs = O.alloc_msgs(G, n_slots)
    gout = np.zeros(G, dtype=np.uint32)
    out = np.zeros(G, dtype=np.uint32)
    elected = np.zeros(G, dtype=np.int64)
    out2 = np.zeros(G, dtype=np.uint32)
    settled = 0

    def tick_again(m2):
        host_tick(eng_st, m2, out2, False)
        return out2.copy()

    for t in range(10):
        cl.store_soa(st)
        st_before = copy_state(st)
        # stale terms now and then (not above the current one): fault, ignored
        term_t = TERM + 1 + t if t != 6 else TERM
        fuzz.random_msgs(rng, st, msgs, reject_p=0.3, logterm_max=TERM + t, elect_p=0.15, elect_term=term_t)
        host_tick(eng_st, msgs, out, False)
        cl.tick_soa(msgs, gout)
        cl.store_soa(st)
        # rejects whose find_conflict_by_term walk needs terms the bounded table has dropped come back to the host: exactly the
        # ones the reference's literal walk says, and once the host has resolved them everything equals the oracle again
        want = hosthints.expected(cl, st_before, st, msgs)
        flagged = np.nonzero(out & hosthints.OUT_HOST_HINT)[0]
        assert {int(g): int(eng_st["host_hint"][g]) for g in flagged} == want, t
        out[:], n = hosthints.settle(cl, msgs, out, eng_st["host_hint"], tick_again)
        settled += n
        diffs = fuzz.diff_states(st, eng_st, G, n_slots)
        assert not diffs, (t, diffs[:6])
        assert (out == gout).all(), (t, np.nonzero(out != gout)[0][:5])
        for k in ("run_first", "run_term", "cur_term", "dummy_index", "dummy_term"):
            assert (st[k] == eng_st[k]).all(), (t, k)
        elected += (out & 0x10) != 0
        self_slot = (st["cfg"] >> 16) & 7
        flagged = (msgs["m_flags"][np.arange(G), self_slot] & 2) != 0
        has_self = ((st["cfg"] >> 24) >> self_slot) & 1 == 1
        if t == 6:
            assert ((out[flagged & has_self] & 0x12) == 0x2).all(), "a stale term: fault, no election"
        else:
            assert ((out[flagged & has_self] & 0x10) != 0).all() and ((out[~flagged] & 0x10) == 0).all()
            assert ((st["cfg"][flagged & has_self] >> 20) & 0xf == 0).all(), "abort_leader_transfer"
    assert (elected >= 3).sum() > 50, "some groups saw three and more elections (table overflow path)"




@pytest.mark.parametrize("n_slots", [3, 5, 7])
def test_log_history_deeper_than_the_term_run_table(host_tick, n_slots):
    """RaftLog keeps the whole log (raft_log.rs:122-140); the engine's term-run table keeps the newest RG_TERM_RUNS runs of older
    terms. Groups that start with a FULL table and then see up to nine more elections (9 .. 17 older runs in the oracle's log),
    rejects whose reject_hint / log_term land anywhere in that history: the engine equals the UNMODIFIED oracle wherever
    RG_OUT_HOST_HINT is clear, the bit (and RG_COL_HOST_HINT) is set exactly for the rejects whose literal find_conflict_by_term
    walk (raft_log.rs:209-235) consults a dropped run, and after the host has resolved those against the complete log every
    column equals the oracle again."""
    eng = {}
    out = None

    def load(st):
        eng["st"] = st

    def tick(m):
        o = np.zeros(eng["st"]["n_groups"], dtype=np.uint32)
        host_tick(eng["st"], m, o, False)
        eng["out"] = o  # (the twin's RG_COL_OUT)
        return o.copy()

    def resolve(recs):
        applied = twin_resolve(eng["st"], eng["out"], recs)
        assert applied.all(), "a reject the tick left to the host was not stale: maybe_decr_to applies"
        return eng["out"].copy()

    stats = None
    for step in hosthints.deep_history_run(n_slots, 6100 + n_slots, tick, lambda: eng["st"]["host_hint"], load, resolve=resolve):
        if isinstance(step, dict):
            stats = step
            break
        cl, st, t = step
        diffs = fuzz.diff_states(st, eng["st"], st["n_groups"], n_slots)
        assert not diffs, (t, diffs[:6])
        for k in ("run_first", "run_term", "cur_term", "dummy_index", "dummy_term"):
            assert (st[k] == eng["st"][k]).all(), (t, k)
    assert stats["settled"] > 200, stats       # the corner is really exercised ...
    assert stats["applied_logterm_rejects"] > 5 * stats["settled"], stats  # ... and stays a corner
    assert stats["max_runs"] >= 12 and len(stats["depths"]) >= 4, stats


@pytest.mark.parametrize("n_slots", [3, 5, 7])
def test_find_conflict_by_term_on_the_device_table(host_tick, n_slots):
    """Rejects carrying Message.log_term: the engine resolves the hint with find_conflict_by_term over its
    compact term-run table (raft_log.rs:209-235 via raft.rs:1562,1657-1660); the oracle walks its own log."""
    rng = np.random.default_rng(9100 + n_slots)
    G, TERM = 3000, 9
    st = O.add_term_table(O.alloc_state(G, n_slots))
    st["cfg"][:] = fuzz.random_cfg(rng, G, n_slots)
    fuzz.random_state(rng, st, small_values=True, probe_frac=0.5)
    fuzz.random_term_table(rng, st, TERM)
    cl = O.Cluster(G)
    cl.load_soa(st, term=TERM)
    # the table and the oracle's log agree on every term
    L = O.lib()
    for g in range(0, G, 37):
        for idx in range(max(0, int(st["dummy_index"][g]) - 1), int(st["term_hi"][g]) + 2):
            want = L.ro_log_term(cl.h, g, idx)
            runs = [(int(st["run_first"][k, g]), int(st["run_term"][k, g])) for k in range(O.TERM_RUNS) if st["run_first"][k, g]]
            d = int(st["dummy_index"][g])
            if idx < d or idx > int(st["term_hi"][g]):
                got = 0
            elif idx >= int(st["term_lo"][g]):
                got = TERM
            elif idx == d:
                got = int(st["dummy_term"][g])
            else:
                got = ([t for f, t in runs if f <= idx] or [0])[-1]
            assert got == want, (g, idx, got, want)
    eng_st = copy_state(st)
    msgs = O.alloc_msgs(G, n_slots)
    gout = np.zeros(G, dtype=np.uint32)
    out = np.zeros(G, dtype=np.uint32)
    n_lt = 0
    for t in range(5):
        cl.store_soa(st)
        fuzz.random_msgs(rng, st, msgs, reject_p=0.4, logterm_max=TERM)
        n_lt += int(((msgs["m_flags"] & 0x80) != 0).sum())
        host_tick(eng_st, msgs, out, False)
        cl.tick_soa(msgs, gout)
        cl.store_soa(st)
        diffs = fuzz.diff_states(st, eng_st, G, n_slots)
        assert not diffs, (t, diffs[:6])
        assert (out == gout).all()
    assert n_lt > 500


@pytest.mark.parametrize("n_slots", [2, 5, 8])
def test_garbage_events_still_follow_the_reference(host_tick, n_slots):
    rng = np.random.default_rng(1234 + n_slots)
    G, TERM = 3000, 9
    st = O.add_term_table(O.alloc_state(G, n_slots))
    st["cfg"][:] = fuzz.random_cfg(rng, G, n_slots, missing_progress_frac=0.1)
    fuzz.random_state(rng, st, small_values=True)
    fuzz.random_term_table(rng, st, TERM)
    cl = O.Cluster(G)
    cl.load_soa(st, term=TERM)
    eng_st = copy_state(st)
    msgs = O.alloc_msgs(G, n_slots)
    gout = np.zeros(G, dtype=np.uint32)
    out = np.zeros(G, dtype=np.uint32)
    for t in range(6):
        cl.store_soa(st)
        fuzz.garbage_msgs(rng, st, msgs)
        host_tick(eng_st, msgs, out, False)
        cl.tick_soa(msgs, gout)
        cl.store_soa(st)
        diffs = fuzz.diff_states(st, eng_st, G, n_slots)
        assert not diffs, (t, diffs[:6])
        assert (out == gout).all(), (t, np.nonzero(out != gout)[0][:5])


@pytest.mark.parametrize("workload,n_slots", [(5, 7), (2, 5)])
def test_soak_many_ticks(host_tick, workload, n_slots):
    """120 ticks of the synthetic stream: long-horizon paths (Probe -> Replicate cycles, snapshot requests,
    full windows) keep matching the oracle."""
    from raft_rs_amd import engine as E
    G = 1500
    st = O.alloc_state(G, n_slots)
    E.workload_init_host(st, workload)
    cl = O.Cluster(G)
    cl.load_soa(st, term=6)
    eng_st = copy_state(st)
    mb = E.MsgBuffers(G, n_slots, st["stride"])
    gout = np.zeros(G, dtype=np.uint32)
    out = np.zeros(G, dtype=np.uint32)
    for t in range(120):
        E.workload_gen_host(st, mb, workload, t)
        host_tick(eng_st, mb.as_dict(), out, False)
        cl.tick_soa(mb.as_dict(), gout)
        cl.store_soa(st)
        assert (out == gout).all(), t
        if t % 10 == 9:
            assert not fuzz.diff_states(st, eng_st, G, n_slots), t
    assert (st["commit"] > 1000).all()


# ---- the send stage (rg_send.h): device Inflights + maybe_send_append decisions ----------------------------
SEND_ITEM_DTYPE = np.dtype([("group", "<u8"), ("prev_index", "<u8"), ("last_index", "<u8"), ("slot", "<u4"),
                            ("n_msgs", "<u2"), ("kind", "<u2")])


@pytest.fixture(scope="module")
def host_send():
    if build_lib() is None:
        pytest.skip("hipcc not available")
    fn = C.CDLL(LIB).rg_host_check_send
    fn.restype = C.c_long
    fn.argtypes = [C.c_uint, C.c_ulong, C.c_ulong, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint,
                   C.c_ulong, C.c_uint, C.c_void_p, C.c_ulong, C.c_void_p, C.c_uint]

    def send(st, out, meta, head, ring, cap, max_entries, flags=0, esz=None):
        """meta/ring as rg_read_inflights reports them; `head` = (head, tail), the engine-internal columns of the
        oldest / newest entry of every window."""
        head, tail = head
        items = np.zeros(st["n_groups"] * st["n_slots"], dtype=SEND_ITEM_DTYPE)
        n = fn(st["n_slots"], st["n_groups"], st["stride"], state_ptrs(st, out), meta.ctypes.data, head.ctypes.data,
               tail.ctypes.data, ring.ctypes.data, cap, max_entries, flags, items.ctypes.data, len(items),
               None if esz is None else esz.ctypes.data, 0 if esz is None else esz.shape[1])
        assert 0 <= n <= len(items)
        sendstage.patch_ring(meta, head, tail, ring, cap, st["n_groups"], st["n_slots"])  # (what rg_read_inflights does)
        return items[:n]
    return send


@pytest.fixture(scope="module")
def host_tick_send():
    """rg_group_tick_send (k_tick_send's per-lane code: the tick and its send stage on one set of registers) on the host."""
    if build_lib() is None:
        pytest.skip("hipcc not available")
    fn = C.CDLL(LIB).rg_host_check_tick_send
    fn.restype = C.c_long
    fn.argtypes = [C.c_uint, C.c_ulong, C.c_ulong, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                   C.c_void_p, C.c_uint, C.c_ulong, C.c_uint, C.c_void_p, C.c_ulong, C.c_void_p, C.c_uint]

    def tick_send(st, msgs, out, gc, meta, head, ring, cap, max_entries, flags=0, esz=None):
        head, tail = head
        items = np.zeros(st["n_groups"] * st["n_slots"], dtype=SEND_ITEM_DTYPE)
        n = fn(st["n_slots"], st["n_groups"], st["stride"], state_ptrs(st, out), msg_ptrs(msgs), int(gc), meta.ctypes.data,
               head.ctypes.data, tail.ctypes.data, ring.ctypes.data, cap, max_entries, flags, items.ctypes.data, len(items),
               None if esz is None else esz.ctypes.data, 0 if esz is None else esz.shape[1])
        assert 0 <= n <= len(items)
        sendstage.patch_ring(meta, head, tail, ring, cap, st["n_groups"], st["n_slots"])
        return items[:n]
    return tick_send


def apply_snapshots(rng, items, cl, st, meta):
    """The host's half of a snapshot send: Progress::become_snapshot(snapshot index) on both sides."""
    for (g, p), (kind, prev, last, n) in items.items():
        if kind != O.SEND_SNAPSHOT:
            continue
        sidx = int(st["commit"][g])  # what a storage would hand out: a snapshot at the applied index
        pr = cl.pr(g, p + 1)
        cl.L.ro_progress_become_snapshot(C.byref(pr), sidx)
        # (RG_PF_PEND_SNAP 0x40: rg_write_cells re-derives the engine-owned summary bits of the two pending fields)
        st["pflags"][g, p] = (int(st["pflags"][g, p]) & ~(0x3 | 0x4 | 0x10 | 0x40)) | O.SNAPSHOT | (0x40 if sidx else 0)
        st["pend_snap"][p, g] = sidx
        meta[p, g] = 0  # what rg_write_cells does on a state change


@pytest.mark.parametrize("cap,max_entries", [(1, 0), (3, 2), (4, 1), (256, 0), (2, 7), (5, 1)])
@pytest.mark.parametrize("n_slots", [1, 3, 5, 8])
@pytest.mark.parametrize("fused", [False, True])
def test_send_stage_on_host_matches_oracle(host_tick, host_send, host_tick_send, fused, n_slots, cap, max_entries):
    """fused: the tick and its stage as ONE pass over the group's registers (rg_group_tick_send, k_tick_send's lane code)."""
    rng = np.random.default_rng(9100 + 17 * n_slots + cap + max_entries)
    G, ticks = (300, 60) if cap == 5 else (1500, 10)  # cap 5: a long run, the ring positions wrap many times
    seen = send_stage_round(rng, host_tick, host_send, host_tick_send, fused, n_slots, cap, max_entries, G, ticks)
    if n_slots > 1:
        assert seen["items"] > 100, seen
        assert seen["snap"] > 0, seen
        if cap <= 4:
            assert seen["full"] > 0, seen
        if max_entries in (1, 2) and cap > 1:
            assert seen["multi"] > 0, seen


def send_stage_round(rng, host_tick, host_send, host_tick_send, fused, n_slots, cap, max_entries, G, ticks):
    """One seeded run of tick + send stage on the host twins against the oracle with its own Inflights (the body of the test
    above; tools/fuzz_host_check.py --send calls it with fresh seeds). Returns what the run exercised."""
    st = O.add_term_table(O.alloc_state(G, n_slots))
    st["cfg"][:] = fuzz.random_cfg(rng, G, n_slots, missing_progress_frac=0.05)
    fuzz.random_state(rng, st, small_values=True)
    fuzz.random_term_table(rng, st, term=6)
    sendstage.mark_pending_conf(rng, st)
    cl = O.Cluster(G)
    cl.load_soa(st, term=6, max_inflight=cap)
    cl.set_own_inflights(True)
    eng_st = copy_state(st)
    meta = np.zeros((n_slots, st["stride"]), dtype=np.uint32)
    head = (np.zeros((n_slots, st["stride"]), dtype=np.uint64), np.zeros((n_slots, st["stride"]), dtype=np.uint64))
    ring = np.zeros((G, n_slots, cap), dtype=np.uint64)
    msgs = O.alloc_msgs(G, n_slots)
    gout = np.zeros(G, dtype=np.uint32)
    out = np.zeros(G, dtype=np.uint32)
    seen = {"items": 0, "multi": 0, "snap": 0, "full": 0}
    for t in range(ticks):
        cl.store_soa(st)
        if t % 5 == 4:  # arbitrary event bytes and values now and then
            fuzz.garbage_msgs(rng, st, msgs)
        else:
            fuzz.random_msgs(rng, st, msgs, sent_p=0.0, heartbeat_p=0.2)
        sendstage.prepare_msgs(msgs)
        skip = t % 2 == 1  # Config::skip_bcast_commit on every other tick
        if fused:
            items = host_tick_send(eng_st, msgs, out, False, meta, head, ring, cap, max_entries, 1 if skip else 0)
        else:
            host_tick(eng_st, msgs, out, False)
        cl.tick_soa(msgs, gout)
        # (garbage ticks elect: histories outgrow the term-run table) a group with a reject left to the host keeps ALL its sends
        # until rg_resolve_host_hints has completed its result word -- the stage skips it, the resolve call runs it
        hinted = (out & hosthints.OUT_HOST_HINT) != 0
        if hinted.any():
            seen["hinted"] = seen.get("hinted", 0) + int(hinted.sum())
            if fused:
                assert not set(int(g) for g in items["group"]) & set(np.nonzero(hinted)[0].tolist())
            hosthints.settle(cl, msgs, out, eng_st["host_hint"], resolve=lambda recs: (twin_resolve(eng_st, out, recs, meta), out)[1])
        assert (out == gout).all(), (t, np.nonzero(out != gout)[0][:5])
        if not fused:
            items = host_send(eng_st, out, meta, head, ring, cap, max_entries, 1 if skip else 0)
        elif hinted.any():
            late = host_send(eng_st, np.where(hinted, out, 0).astype(np.uint32), meta, head, ring, cap, max_entries, 1 if skip else 0)
            items = np.concatenate([items, late])
        omsgs = cl.send_stage_soa(gout, max_entries, skip_bcast_commit=skip)
        got = sendstage.compare_items(items, omsgs)
        apply_snapshots(rng, got, cl, eng_st, meta)
        cl.store_soa(st)
        diffs = fuzz.diff_states(st, eng_st, G, n_slots)
        assert not diffs, (t, diffs[:6])
        sendstage.compare_rings(cl, meta, ring, st, cap)
        seen["items"] += len(got)
        seen["multi"] += sum(1 for v in got.values() if v[3] > 1)
        seen["snap"] += sum(1 for v in got.values() if v[0] == O.SEND_SNAPSHOT)
        seen["full"] += int(((st["pflags"][:, :n_slots] & 0x10) != 0).sum())
    return seen


@pytest.mark.parametrize("n_slots,cap,window,max_bytes", [(3, 4, 8, 900), (5, 256, 64, 1500), (5, 3, 16, 0), (7, 8, 32, 2**32 + 5),
                                                         (5, 16, 64, O.U64_MAX), (8, 5, 8, 300)])
@pytest.mark.parametrize("fused", [False, True])
def test_send_stage_byte_limit_on_host_matches_oracle(host_tick, host_send, host_tick_send, fused, n_slots, cap, window, max_bytes):
    """Config::max_size_per_msg in BYTES (RG_SEND_BYTES): rg_limit_size over the device's window of cumulative entry sizes
    against util::limit_size restated literally in the oracle, entry sizes random with a tenth of them zero; peers that
    need entries outside the window come back as RG_SEND_HOST and are served the way rg_update_state does."""
    rng = np.random.default_rng(9300 + 31 * n_slots + cap + window)
    G, ticks = 1200, 12
    st = O.add_term_table(O.alloc_state(G, n_slots))
    st["cfg"][:] = fuzz.random_cfg(rng, G, n_slots, missing_progress_frac=0.05)
    fuzz.random_state(rng, st, small_values=True)
    fuzz.random_term_table(rng, st, term=6)
    cl = O.Cluster(G)
    cl.load_soa(st, term=6, max_inflight=cap)
    cl.set_own_inflights(True)
    cl.set_limit_bytes(True)
    n_index = int(st["term_hi"].max()) + 64 * ticks + 64
    sizes, cum = sendstage.entry_sizes(rng, G, n_index)
    for g in range(G):
        cl.append_entry_sizes(g, 1, sizes[g, 1:])
    eng_st = copy_state(st)
    meta = np.zeros((n_slots, st["stride"]), dtype=np.uint32)
    head = (np.zeros((n_slots, st["stride"]), dtype=np.uint64), np.zeros((n_slots, st["stride"]), dtype=np.uint64))
    ring = np.zeros((G, n_slots, cap), dtype=np.uint64)
    esz = np.zeros((G, window), dtype=np.uint32)
    msgs = O.alloc_msgs(G, n_slots)
    gout = np.zeros(G, dtype=np.uint32)
    out = np.zeros(G, dtype=np.uint32)
    seen = {"items": 0, "multi": 0, "host": 0, "limited": 0}
    for t in range(ticks):
        cl.store_soa(st)
        fuzz.random_msgs(rng, st, msgs, sent_p=0.0, heartbeat_p=0.2)
        sendstage.prepare_msgs(msgs)
        if fused:
            # the host writes the size records of what it appends BEFORE the launch that learns of them: the last_index this
            # tick leaves is taken from a dry run of the tick alone on a copy
            dry, dry_out = copy_state(eng_st), out.copy()
            host_tick(dry, msgs, dry_out, False)
            sendstage.fill_size_window(esz, cum, dry["term_hi"])
            items = host_tick_send(eng_st, msgs, out, False, meta, head, ring, cap, max_bytes, 2, esz=esz)
        else:
            host_tick(eng_st, msgs, out, False)
        cl.tick_soa(msgs, gout)
        assert (out == gout).all(), (t, np.nonzero(out != gout)[0][:5])
        assert int(eng_st["term_hi"].max()) < n_index
        if not fused:
            sendstage.fill_size_window(esz, cum, eng_st["term_hi"])
            items = host_send(eng_st, out, meta, head, ring, cap, max_bytes, 2, esz=esz)  # RG_SEND_BYTES
        omsgs = cl.send_stage_soa(gout, max_bytes)
        items, omsgs_dev, served = sendstage.split_host_items(items, omsgs)
        got = sendstage.compare_items(items, omsgs_dev)
        for (g, p), lasts in served.items():
            sendstage.host_update_state(eng_st, meta, head[0], head[1], ring, cap, g, p, lasts)
        apply_snapshots(rng, got, cl, eng_st, meta)
        cl.store_soa(st)
        diffs = fuzz.diff_states(st, eng_st, G, n_slots)
        assert not diffs, (t, diffs[:6])
        sendstage.compare_rings(cl, meta, ring, st, cap)
        seen["items"] += len(got)
        seen["multi"] += sum(1 for v in got.values() if v[3] > 1)
        seen["host"] += len(served)
        seen["limited"] += sum(1 for m in omsgs if 0 < int(m["n_entries"]))
    assert seen["items"] > 100, seen
    if window <= 16:
        assert seen["host"] > 0, seen
    if max_bytes < 2000 and cap > 1:
        assert seen["multi"] > 0, seen


def _reference_limit_size(sizes, max_bytes):
    """util::limit_size (src/util.rs:52-76), literally: `entries.len() <= 1` and NO_LIMIT return early; then take_while with
    the `size == 0` test that keeps the first entry -- and everything behind a prefix of zero-size entries."""
    if len(sizes) <= 1 or max_bytes == O.U64_MAX:
        return len(sizes)
    size, limit = 0, 0
    for e in sizes:
        if size == 0:
            size += e
            limit += 1
            continue
        size += e
        if size <= max_bytes:
            limit += 1
        else:
            break
    return limit


def test_limit_size_matches_the_reference_rule_property():
    """rg_limit_size over the device's ring of cumulative u32 sizes == util::limit_size over the entries themselves, for
    random sizes (many of them zero), ring positions that wrap, cumulative sums that wrap modulo 2^32, and limits from 0 to
    NO_LIMIT."""
    from hypothesis import given, settings, strategies as hst
    if build_lib() is None:
        pytest.skip("hipcc not available")
    fn = C.CDLL(LIB).rg_host_check_limit_size
    fn.restype = C.c_ulong
    fn.argtypes = [C.c_void_p, C.c_uint, C.c_ulong, C.c_ulong, C.c_ulong]
    size = hst.one_of(hst.just(0), hst.integers(0, 3), hst.integers(1, 2000), hst.integers(1, 1 << 20))
    limit = hst.one_of(hst.sampled_from([0, 1, 100, 2048, (1 << 32) - 1, 1 << 32, O.U64_MAX - 1, O.U64_MAX]),
                       hst.integers(0, 5000), hst.integers(0, 1 << 24))

    @settings(max_examples=600, deadline=None)
    @given(hst.sampled_from([8, 16, 64]), hst.data())
    def check(window, data):
        avail = data.draw(hst.integers(0, window - 1))
        sizes = data.draw(hst.lists(size, min_size=avail, max_size=avail))
        next_idx = data.draw(hst.integers(1, 1 << 40))
        base = data.draw(hst.one_of(hst.integers(0, 1 << 20), hst.integers((1 << 32) - 3000, (1 << 32) - 1)))  # wrap mod 2^32
        max_bytes = data.draw(limit)
        row = np.zeros(window, dtype=np.uint32)
        cum = base
        row[(next_idx - 1) & (window - 1)] = cum & 0xffffffff
        for k, e in enumerate(sizes):
            cum += e
            row[(next_idx + k) & (window - 1)] = cum & 0xffffffff
        got = fn(row.ctypes.data, window, next_idx, avail, max_bytes)
        assert got == _reference_limit_size(sizes, max_bytes), (window, next_idx, sizes, max_bytes, got)

    check()


# ---- property tests (hypothesis) of the quorum arithmetic over the FULL u64 range --------------------------
from hypothesis import HealthCheck, given, settings, strategies as hst  # noqa: E402

U64 = hst.one_of(hst.integers(0, (1 << 64) - 1), hst.sampled_from([0, 1, (1 << 63) - 1, 1 << 63, (1 << 64) - 2, (1 << 64) - 1]),
                 hst.integers(0, 6))


def _mci_fn():
    if build_lib() is None:
        pytest.skip("hipcc not available")
    fn = C.CDLL(LIB).rg_host_check_mci
    fn.restype = C.c_int
    fn.argtypes = [C.c_uint, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_uint, C.c_uint, C.c_int, C.c_int, C.c_ulong,
                   C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
    return fn


def _oracle_joint(match, gid, incoming, outgoing, gc):
    """ProgressTracker::maximal_committed_index through the oracle's majority routine (slot order = voter order)."""
    res, used = [], []
    for mask in (incoming, outgoing):
        pairs = [(match[i], gid[i]) for i in range(len(match)) if (mask >> i) & 1]
        if not pairs:
            res.append(O.U64_MAX)  # empty config (majority.rs:71-75)
            used.append(True)
            continue
        v, u = O.committed_index(pairs, gc)
        res.append(v)
        used.append(u)
    return min(res), all(used)


@settings(max_examples=400, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
@given(hst.integers(1, 8).flatmap(lambda p: hst.tuples(
    hst.just(p), hst.lists(U64, min_size=p, max_size=p), hst.integers(0, (1 << p) - 1), hst.integers(0, (1 << p) - 1),
    hst.integers(-1, p - 1), U64)))
def test_quorum_bit_matrix_equals_sort_based_reference(case):
    """RgQuorum (rank select on the >= bit matrix, with and without the incremental row/column update) ==
    the reference's sort-based committed_index for arbitrary u64 matches and voter masks."""
    P, match, incoming, outgoing, raise_slot, old = case
    fn = _mci_fn()
    m = (C.c_uint64 * P)(*match)
    g = (C.c_uint64 * P)(*([0] * P))
    if raise_slot >= 0:
        old = min(old, match[raise_slot])  # matches only grow
    out, used = C.c_uint64(0), C.c_int(0)
    assert fn(P, m, g, incoming, outgoing, 0, raise_slot, old, C.byref(out), C.byref(used)) == 0
    want, _ = _oracle_joint(match, [0] * P, incoming, outgoing, False)
    assert out.value == want, (case, out.value, want)


@settings(max_examples=400, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
@given(hst.integers(1, 8).flatmap(lambda p: hst.tuples(
    hst.just(p), hst.lists(U64, min_size=p, max_size=p), hst.lists(hst.integers(0, 3), min_size=p, max_size=p),
    hst.integers(0, (1 << p) - 1), hst.integers(0, (1 << p) - 1))))
def test_group_commit_routine_equals_reference(case):
    """rg_mci_group == majority.rs:99-123 (group commit) composed by joint.rs:47-51, value and flag."""
    P, match, gid, incoming, outgoing = case
    fn = _mci_fn()
    out, used = C.c_uint64(0), C.c_int(0)
    assert fn(P, (C.c_uint64 * P)(*match), (C.c_uint64 * P)(*gid), incoming, outgoing, 1, -1, 0, C.byref(out),
              C.byref(used)) == 0
    want, want_used = _oracle_joint(match, gid, incoming, outgoing, True)
    assert out.value == want and bool(used.value) == want_used, (case, out.value, used.value, want, want_used)


def _apply_events_to_oracle(cl, events, G, P):
    """The oracle's side of an rg_progress_events batch: MsgUnreachable / MsgSnapStatus stepped one by one (ids = slot + 1)."""
    L = O.lib()
    for g, s, kind in events:
        if g >= G or s >= P:
            continue
        if kind == 1:
            L.ro_handle_unreachable(cl.h, g, s + 1)
        else:
            L.ro_handle_snapshot_status(cl.h, g, s + 1, kind == 3)


@pytest.mark.parametrize("n_slots", [1, 3, 5, 8])
def test_progress_events_on_host_match_the_oracle(n_slots):
    """rg_progress_events (RawNode::report_unreachable / report_snapshot applied to the cells in place): the engine's
    per-cell arithmetic and record loop, compiled for the host, against handle_unreachable / handle_snapshot_status
    restated in the oracle -- random states with a quarter of the peers in Snapshot, runs of several events on one cell."""
    if build_lib() is None:
        pytest.skip("hipcc not available")
    fn = C.CDLL(LIB).rg_host_check_progress_events
    fn.restype = C.c_int
    fn.argtypes = [C.c_uint, C.c_ulong, C.c_ulong, C.c_void_p, C.c_void_p, C.c_void_p, C.c_ulong]
    rng = np.random.default_rng(7100 + n_slots)
    G = 2000
    st = O.alloc_state(G, n_slots)
    st["cfg"][:] = fuzz.random_cfg(rng, G, n_slots, missing_progress_frac=0.05)
    fuzz.random_state(rng, st, small_values=True, snapshot_frac=0.25)
    cl = O.Cluster(G)
    cl.load_soa(st, term=6)
    eng_st = copy_state(st)
    out = np.zeros(G, dtype=np.uint32)
    n_changed = 0
    for rnd in range(4):
        cl.store_soa(st)
        before = copy_state(st)
        events = fuzz.random_progress_events(rng, G, n_slots, 1500)
        ev = np.array(events, dtype=[("group", "<u8"), ("slot", "<u4"), ("kind", "<u4")])
        assert fn(n_slots, G, st["stride"], state_ptrs(eng_st, out), None, ev.ctypes.data, len(ev)) == 0
        _apply_events_to_oracle(cl, events, G, n_slots)
        cl.store_soa(st)
        diffs = fuzz.diff_states(st, eng_st, G, n_slots)
        assert not diffs, (rnd, diffs[:6])
        n_changed += int((before["next"] != st["next"]).sum() + (before["pflags"] != st["pflags"]).sum())
        if rnd == 1:  # put a fresh crop of peers into Snapshot / Replicate for the later rounds
            fuzz.random_state(rng, st, small_values=True, snapshot_frac=0.3)
            cl.load_soa(st, term=6)
            eng_st = copy_state(st)
    assert n_changed > 500, n_changed


@pytest.mark.parametrize("n_slots,q", [(7, 3), (7, 5), (8, 7), (5, 3), (8, 3)])
def test_tick_for_fewer_slots_is_the_full_tick_where_no_higher_slot_is_named(host_tick, n_slots, q):
    """What k_tick_classes rests on (one launch over a shard placed by replica-set size class): for a group whose cfg word
    names only slots < q -- Progress set, both voter sets, the leader's own slot, the transferee -- the tick instantiated for q
    slots, run over the SAME P-slot columns, is the tick instantiated for P slots: every column, every result word. The
    messages are random over all P slots (events on slots without a Progress are ignored by both, raft.rs:1663-1673)."""
    rng = np.random.default_rng(7700 + 10 * n_slots + q)
    G = 3000
    st = O.add_term_table(O.alloc_state(G, n_slots))
    st["cfg"][:] = fuzz.class_placed_cfg(rng, [(G, q)], n_slots, missing_progress_frac=0.05)
    fuzz.random_state(rng, st, small_values=True)
    fuzz.random_term_table(rng, st, 9, max_runs=4)
    full, few = copy_state(st), copy_state(st)
    few["n_slots"] = q  # the q-slot instantiation over the P-slot columns (same stride, same pointers)
    msgs = O.alloc_msgs(G, n_slots)
    out_full = np.zeros(G, dtype=np.uint32)
    out_few = np.zeros(G, dtype=np.uint32)
    for t in range(8):
        fuzz.random_msgs(rng, full, msgs, reject_p=0.2, heartbeat_p=0.1, malformed_p=0.02 if t == 5 else 0.0,
                         elect_p=0.1, elect_term=10 + t)
        host_tick(full, msgs, out_full, False)
        host_tick(few, msgs, out_few, False)
        few["n_slots"] = n_slots
        diffs = fuzz.diff_states(full, few, G, n_slots)
        few["n_slots"] = q
        assert not diffs, (t, diffs[:6])
        assert (out_full == out_few).all(), (t, np.nonzero(out_full != out_few)[0][:5])
        for k in ("run_first", "run_term", "cur_term"):
            assert (full[k] == few[k]).all(), (t, k)


@pytest.mark.parametrize("cap,spread", [(1, 3), (2, 5), (3, 1 << 22), (5, 9), (5, 1 << 23), (8, 40), (256, 7), (256, 1 << 21), (64, 1 << 62)])
def test_window_arithmetic_equals_a_plain_queue(cap, spread):
    """Inflights (src/tracker/inflights.rs:42-125) as the engine holds it since round 6 -- windows of up to four entries whose
    distances fit 21 bits live entirely in two column cells (compact), deeper or wider ones in the ring, and they move between
    the two forms -- against a plain Python list: add / free_to / free_first_one in random order, every intermediate state
    compared. `spread` sets the distance between consecutive entries: small ones stay compact, 2^21 and more force the ring."""
    if build_lib() is None:
        pytest.skip("hipcc not available")
    fn = C.CDLL(LIB).rg_host_check_window_ops
    fn.restype = C.c_int
    fn.argtypes = [C.c_uint, C.c_ulong] + [C.c_void_p] * 6
    rng = np.random.default_rng(cap * 31 + (spread % 1000))
    n = 6000
    ops = np.zeros(n, dtype=np.uint32)
    vals = np.zeros(n, dtype=np.uint64)
    model, want, last = [], [], 1000
    for k in range(n):
        r = rng.random()
        if r < 0.5 and len(model) < cap and last < (1 << 64) - 8:  # (at the top of the index range only frees are left)
            step = (1 + int(rng.random() * spread)) if rng.random() < 0.8 else int(rng.integers(1, 4))
            last = min(last + step, (1 << 64) - 4)
            if model and last <= model[-1]:
                last = model[-1] + 1
            ops[k], vals[k] = 0, last
            model.append(last)
        elif r < 0.85:
            # free_to anywhere: below the window, on an entry, between entries, beyond the newest
            lo = (model[0] - 2) if model else last
            hi = (model[-1] + 2) if model else last + 2
            pick = int(model[int(rng.integers(0, len(model)))]) if (model and rng.random() < 0.5) else (max(lo, 0) + int(rng.random() * (hi - max(lo, 0) + 1)))
            pick = min(pick, (1 << 64) - 1)
            ops[k], vals[k] = 1, pick
            model = [e for e in model if e > pick]
        else:
            ops[k] = 2
            model = model[1:]
        want.append(list(model))
    ring = np.zeros(cap, dtype=np.uint64)
    contents = np.zeros((n, cap), dtype=np.uint64)
    counts = np.zeros(n, dtype=np.uint32)
    modes = np.zeros(n, dtype=np.uint32)
    rc = fn(cap, n, ops.ctypes.data, vals.ctypes.data, ring.ctypes.data, contents.ctypes.data, counts.ctypes.data, modes.ctypes.data)
    assert rc == 0, rc
    for k in range(n):
        assert counts[k] == len(want[k]) and contents[k, :counts[k]].tolist() == want[k], (k, int(ops[k]), int(vals[k]), want[k], contents[k, :counts[k]].tolist())
    deep = max(len(w) for w in want)
    if spread < (1 << 21):
        assert modes[counts <= min(4, cap)].mean() > 0.5  # small windows of near entries live in the columns
    if cap > 4 and spread < (1 << 60):  # (steps of 2^62 run out of index range after four entries)
        assert deep > 4 and (modes == 0).any() and (modes[1:][(modes[:-1] == 0)] == 1).any()  # ... deeper ones in the ring, and back
    if spread >= (1 << 22) and cap >= 3:
        assert (modes[counts >= 2] == 0).any()  # a distance beyond 21 bits forces the ring
