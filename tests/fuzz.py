"""Random engine states and message ticks for differential testing (engine vs oracle).

Test infrastructure: plain numpy, no dependency on the oracle or the engine.
"""
import numpy as np

TERM_RUNS = 8  # RG_TERM_RUNS (include/raftgroups.h)

MF_VALID, MF_REJECT, MF_HAS_RS, MF_INS_FULL, MF_SENT, MF_APPEND, MF_HEARTBEAT = 1, 2, 4, 8, 16, 32, 64


def random_cfg(rng, n_groups, n_slots, joint_frac=0.3, learner_frac=0.2, group_commit_frac=0.0,
               missing_progress_frac=0.0, transfer_frac=0.1):
    full = (1 << n_slots) - 1
    cfg = np.zeros(n_groups, dtype=np.uint32)
    for g in range(n_groups):
        inc = int(rng.integers(1, full + 1))
        out = int(rng.integers(0, full + 1)) if rng.random() < joint_frac else 0
        present = inc | out
        if rng.random() < learner_frac:
            present |= int(rng.integers(0, full + 1))
        voters = [s for s in range(n_slots) if (inc | out) >> s & 1]
        self_slot = int(rng.choice(voters)) if voters and rng.random() < 0.9 else int(rng.integers(0, n_slots))
        present |= 1 << self_slot
        if rng.random() < missing_progress_frac:  # a voter without a Progress entry
            v = int(rng.choice(voters))
            if v != self_slot:
                present &= ~(1 << v)
        gc = rng.random() < group_commit_frac
        xfer = int(rng.integers(1, n_slots + 1)) if rng.random() < transfer_frac else 0
        cfg[g] = (inc | (out << 8) | (self_slot << 16) | (0x80000 if gc else 0) | (xfer << 20) |
                  ((present & 0xff) << 24))
    return cfg


def random_state(rng, st, snapshot_frac=0.05, probe_frac=0.2, small_values=False, with_gids=False, base=0):
    """Fill an oracle_lib.alloc_state() dict in place (cfg must already be set). `base` shifts every
    log index (e.g. 2**62: arithmetic near the top of the u64 range the reference allows)."""
    G, P, stride = st["n_groups"], st["n_slots"], st["stride"]
    hi_max = 40 if small_values else 1 << 20
    last = (rng.integers(5, hi_max, size=G).astype(np.uint64) + np.uint64(base)).astype(np.uint64)
    span = rng.integers(0, 8, size=G).astype(np.uint64)
    lo = np.maximum(last - np.minimum(span, last - 1), 1).astype(np.uint64)
    empty = rng.random(G) < 0.05  # no entry of the current term yet: lo == hi + 1
    lo = np.where(empty, last + 1, lo).astype(np.uint64)
    st["term_lo"][:] = lo
    st["term_hi"][:] = last
    st["commit"][:] = (last - np.minimum(rng.integers(0, 12, size=G).astype(np.uint64), last)).astype(np.uint64)
    for p in range(P):
        lag = rng.integers(0, 10, size=G).astype(np.uint64)
        m = (last - np.minimum(lag, last)).astype(np.uint64)
        m = np.where(rng.random(G) < 0.1, 0, m).astype(np.uint64)  # a peer that never acked
        # ill-formed leftovers (matched beyond last_index, as after a malformed ack): the engine must
        # still follow the reference arithmetic exactly
        m = np.where(rng.random(G) < 0.02, last + rng.integers(1, 4, size=G).astype(np.uint64), m).astype(np.uint64)
        st["match"][p, :G] = m
        nxt = m + 1 + rng.integers(0, 4, size=G).astype(np.uint64)
        nxt = np.where(rng.random(G) < 0.03, m, nxt)  # next <= matched corner (progress.rs:145-147)
        nxt = np.where(rng.random(G) < 0.01, 0, nxt)
        st["next"][p, :G] = nxt.astype(np.uint64)
        st["pr_commit"][p, :G] = np.minimum(st["commit"], m)
        r = rng.random(G)
        state = np.where(r < snapshot_frac, 2, np.where(r < snapshot_frac + probe_frac, 0, 1)).astype(np.uint8)
        paused = (rng.random(G) < 0.3).astype(np.uint8)
        ra = (rng.random(G) < 0.5).astype(np.uint8)
        st["pflags"][:, p] = state | (paused << 2) | (ra << 3)
        ps = np.where(state == 2, m.astype(np.int64) + rng.integers(0, 5, size=G) - 2, 0)
        ps = np.where(rng.random(G) < 0.02, rng.integers(0, 50, size=G), ps)  # stale pending_snapshot on non-Snapshot
        st["pend_snap"][p, :G] = np.maximum(ps, 0).astype(np.uint64)
        st["pend_rs"][p, :G] = np.where(rng.random(G) < 0.05, rng.integers(1, 30, size=G), 0).astype(np.uint64)
        if with_gids:
            st["gid"][p, :G] = rng.integers(0, 4, size=G).astype(np.uint64)
    for p in range(P, 8):
        st["pflags"][:, p] = 0
    return st


def random_term_table(rng, st, term, min_runs=0, max_runs=TERM_RUNS):
    """Fill st's term-run table (oracle_lib.add_term_table): min_runs..TERM_RUNS runs of terms OLDER than the leader's `term`
    right below term_lo (the leader's own entries [term_lo, term_hi] are implicit), the dummy entry below them. The table
    is contiguous: its first run starts right above the dummy entry. (max_runs < TERM_RUNS leaves room for elections: a
    table that never overflows never hands a reject back to the host.)"""
    G = st["n_groups"]
    st["cur_term"][:] = term
    for g in range(G):
        lo, hi = int(st["term_lo"][g]), int(st["term_hi"][g])
        top = lo if lo <= hi else hi + 1  # first index that is NOT an older entry
        runs, first, t = [], top, term
        for _ in range(int(rng.integers(min_runs, max_runs + 1))):
            if first <= 1 or t <= 1:
                break
            first = max(1, first - int(rng.integers(1, 6)))
            t = max(1, t - int(rng.integers(1, 3)))
            runs.insert(0, (first, t))
        d_idx = (runs[0][0] - 1) if runs else top - 1
        # the dummy (snapshot) entry is older than the leader's term: a snapshot index is always committed,
        # so "term(dummy) == current term" can never gate a commit in a real log
        d_term = int(rng.integers(0, min(runs[0][1] if runs else term - 1, term - 1) + 1))
        for k in range(TERM_RUNS):
            st["run_first"][k, g] = runs[k][0] if k < len(runs) else 0
            st["run_term"][k, g] = runs[k][1] if k < len(runs) else 0
        st["dummy_index"][g] = d_idx
        st["dummy_term"][g] = d_term
    return st


def random_msgs(rng, st, msgs, valid_p=0.7, reject_p=0.15, rs_p=0.1, malformed_p=0.0, sent_p=0.5, heartbeat_p=0.1,
                logterm_max=0, elect_p=0.0, elect_term=0):
    """elect_p > 0: that fraction of the groups carries RG_MF_BECOME_LEADER on the leader's slot with the new term
    `elect_term` in m_hint (well-formed when elect_term is above every group's current term)."""
    """logterm_max > 0: 60% of the rejects carry Message.log_term in [1, logterm_max + 1] (needs a term table)."""
    """One tick of random messages against state `st` (fills an alloc_msgs() dict in place)."""
    G, P = st["n_groups"], st["n_slots"]
    self_slot = ((st["cfg"] >> 16) & 7).astype(np.int64)
    msgs["m_flags"][...] = 0
    for p in range(P):
        m, nx, hi = st["match"][p, :G], st["next"][p, :G], st["term_hi"]
        is_self = self_slot == p
        hb = (rng.random(G) < heartbeat_p) & ~is_self  # a MsgHeartbeatResponse instead of an AppendResponse
        valid = (rng.random(G) < valid_p) & ~hb
        reject = valid & (rng.random(G) < reject_p) & ~is_self
        has_rs = reject & (rng.random(G) < rs_p)
        sent = (rng.random(G) < sent_p) & ~is_self
        ins_full = rng.random(G) < 0.1
        # accept index: around match, bounded by last_index unless malformed
        delta = rng.integers(-3, 12, size=G)
        idx = np.clip(m.astype(np.int64) + delta, 0, None).astype(np.uint64)
        idx = np.minimum(idx, hi)
        mal = rng.random(G) < malformed_p
        idx = np.where(mal, hi + rng.integers(1, 5, size=G).astype(np.uint64), idx)
        # reject index: mostly next-1 (Probe-valid) or around match (Replicate)
        ridx = np.where(rng.random(G) < 0.6, np.where(nx > 0, nx - 1, 0),
                        np.clip(m.astype(np.int64) + rng.integers(-2, 4, size=G), 0, None)).astype(np.uint64)
        idx = np.where(reject, ridx, idx).astype(np.uint64)
        hint = np.where(rng.random(G) < 0.8, np.minimum(m + rng.integers(0, 3, size=G).astype(np.uint64), idx),
                        rng.integers(0, 5, size=G)).astype(np.uint64)
        # self slot: append + persist
        append = is_self & (rng.random(G) < 0.7)
        new_last = hi + rng.integers(0, 6, size=G).astype(np.uint64)
        self_idx = np.where(append, new_last - np.minimum(rng.integers(0, 3, size=G).astype(np.uint64), new_last), idx)
        idx = np.where(is_self, np.where(mal, new_last + 2, self_idx), idx).astype(np.uint64)
        mc = np.where(is_self, new_last, np.minimum(st["commit"] + rng.integers(0, 2, size=G).astype(np.uint64), idx))
        msgs["m_index"][p, :G] = idx
        msgs["m_commit"][p, :G] = mc.astype(np.uint64)
        msgs["m_hint"][p, :G] = hint
        msgs["m_rs"][p, :G] = np.where(has_rs, rng.integers(1, 100, size=G), 0).astype(np.uint64)
        # rejects that carry the follower's log term: the hint goes through find_conflict_by_term
        has_lt = reject & (rng.random(G) < 0.6) & bool(logterm_max)
        if "m_logterm" in msgs:
            msgs["m_logterm"][p, :G] = np.where(has_lt, rng.integers(1, max(2, logterm_max + 2), size=G), 0).astype(np.uint64)
            hint_lt = np.clip(hi.astype(np.int64) - rng.integers(-2, 14, size=G), 0, None).astype(np.uint64)
            msgs["m_hint"][p, :G] = np.where(has_lt, hint_lt, msgs["m_hint"][p, :G])
        elect = is_self & (rng.random(G) < elect_p)
        msgs["m_hint"][p, :G] = np.where(elect, elect_term, msgs["m_hint"][p, :G]).astype(np.uint64)
        f = (valid * MF_VALID) | (reject * MF_REJECT) | (has_rs * MF_HAS_RS) | (ins_full * MF_INS_FULL) | \
            (sent * MF_SENT) | (append * MF_APPEND) | (hb * MF_HEARTBEAT) | (has_lt * 0x80) | (elect * MF_REJECT)
        msgs["m_flags"][:, p] = f.astype(np.uint8)
    return msgs


STATE_KEYS = ("match", "next", "pr_commit", "pend_snap", "pend_rs", "gid", "pflags", "commit", "term_lo",
              "term_hi", "cfg")


def diff_states(a, b, n_groups, n_slots, keys=STATE_KEYS, present_only=True):
    """Return a list of human-readable differences between two state dicts (empty = bit-exact)."""
    diffs = []
    present = ((a["cfg"] >> 24) & 0xff).astype(np.uint32)
    for k in keys:
        x, y = a[k], b[k]
        if k == "pflags":
            x, y = x[:n_groups, :n_slots], y[:n_groups, :n_slots]
            mask = np.stack([(present >> p) & 1 for p in range(n_slots)], axis=1).astype(bool)
            bad = (x != y) & (mask if present_only else True)
            for g, p in zip(*np.nonzero(bad)):
                diffs.append(f"pflags[g={g},slot={p}]: {x[g, p]:#x} != {y[g, p]:#x}")
        elif x.ndim == 2:
            x, y = x[:n_slots, :n_groups], y[:n_slots, :n_groups]
            mask = np.stack([(present >> p) & 1 for p in range(n_slots)], axis=0).astype(bool)
            bad = (x != y) & (mask if present_only else True)
            for p, g in zip(*np.nonzero(bad)):
                diffs.append(f"{k}[slot={p},g={g}]: {x[p, g]} != {y[p, g]}")
        else:
            bad = x[:n_groups] != y[:n_groups]
            for g in np.nonzero(bad)[0]:
                diffs.append(f"{k}[g={g}]: {x[g]} != {y[g]}")
        if len(diffs) > 20:
            break
    return diffs


def garbage_msgs(rng, st, msgs):
    """Arbitrary event bytes (all 256 combinations, also meaningless ones such as REJECT without VALID or
    VALID|HEARTBEAT) and arbitrary values: the engine must still follow the reference arithmetic exactly."""
    G, P = st["n_groups"], st["n_slots"]
    hi = st["term_hi"]
    msgs["m_flags"][...] = 0
    msgs["m_flags"][:, :P] = rng.integers(0, 256, size=(G, P), dtype=np.uint8)
    for k in ("m_index", "m_commit", "m_hint", "m_rs"):
        near = np.clip(hi.astype(np.int64)[None, :] + rng.integers(-12, 6, size=(P, G)), 0, None).astype(np.uint64)
        # "wild" stays small: the reference's find_conflict_by_term walks the log index by index (and so does
        # the oracle), so astronomically large garbage indices would only test patience
        wild = rng.integers(0, 3000, size=(P, G), dtype=np.uint64)
        msgs[k][:, :G] = np.where(rng.random((P, G)) < 0.9, near, wild)
    if "m_logterm" in msgs:
        msgs["m_logterm"][:, :G] = rng.integers(0, 12, size=(P, G), dtype=np.uint64)
    return msgs


def random_progress_events(rng, n_groups, n_slots, n, dup_frac=0.2):
    """[(group, slot, kind)] for rg_progress_events: RawNode::report_unreachable / report_snapshot on random cells (kinds 1..3),
    sorted so that the records of one (group, slot) are adjacent -- the entry point's contract -- with runs of several
    events on one cell, and a few slots / groups that do not exist (ignored)."""
    ev = []
    for _ in range(n):
        g, s = int(rng.integers(0, n_groups)), int(rng.integers(0, n_slots))
        ev.append((g, s, int(rng.integers(1, 4))))
        while rng.random() < dup_frac:
            ev.append((g, s, int(rng.integers(1, 4))))
    keyed = {}
    for g, s, k in ev:  # group the runs, keep the order inside a run
        keyed.setdefault((g, s), []).append(k)
    out = [(g, s, k) for (g, s), ks in keyed.items() for k in ks]
    out.insert(len(out) // 2, (n_groups + 5, 0, 1))  # no such group
    out.insert(len(out) // 3, (0, 8, 2))             # no such slot
    return out


def class_placed_cfg(rng, ranges, n_slots, **kw):
    """cfg words of a shard whose groups are placed by replica-set size class: `ranges` = [(n_groups, q), ...] in placement order,
    the groups of a range name only slots < q (random_cfg over q slots). Returns u32[sum n]."""
    parts = [random_cfg(rng, n, min(q, n_slots), **kw) for n, q in ranges]
    return np.concatenate(parts).astype(np.uint32)
