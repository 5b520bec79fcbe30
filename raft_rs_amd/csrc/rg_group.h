// rg_group.h -- per-group arithmetic of the hot path, written for one lane = one raft group.
//
// The functions are __host__ __device__ only so that tests/host_check/ can compile this very header
// for the host and diff it against the CPU restatement of the reference without a GPU; the product library never runs
// them on the host.
//
// All loops are over compile-time slot counts so every array lives in VGPRs (no dynamic register
// indexing, no scratch). Citations are file:line in the pingcap/raft-rs v0.6.0 tree.
#pragma once

#include "rg_common.h"
#include "rg_publish.h"

// compile-time slot sequence (a minimal std::integer_sequence)
template <int... S> struct rg_seq {};
template <int N, int... S> struct rg_make_seq : rg_make_seq<N - 1, N - 1, S...> {};
template <int... S> struct rg_make_seq<0, S...> { typedef rg_seq<S...> type; };

// ---------------------------------------------------------------------------------------------
// Quorum index. MajorityConfig::committed_index (src/quorum/majority.rs:70-124) sorts the voters'
// matched indices descending and takes element n/2 (the q-th largest, q = n/2+1, src/util.rs:118-120).
// Here: the q-th largest of {v_i : i in M} is max{ v_i : #{ j in M : v_j >= v_i } >= q }, evaluated
// from a bit matrix ge[i] (bit j = v_j >= v_i) that is maintained incrementally: an accepted ack
// changes one v_s, i.e. one row and one column (2(P-1) compares), so re-evaluating the commit index
// after EVERY accepted ack -- exactly as Raft::handle_append_response does (src/raft.rs:1745) -- costs
// O(P) per message instead of a sort.
// ---------------------------------------------------------------------------------------------
template <int P> struct RgQuorum {
    // v = the group's matched registers; the caller zeroes slots without a Progress first
    // (a voter without a Progress acks 0: unwrap_or_default, majority.rs:80-82)
    u32 ge[P]; // bit j of ge[i] = (v[j] >= v[i])

    RG_HD void init(const u64 (&v)[P]) {
#pragma unroll
        for (int i = 0; i < P; i++) ge[i] = 1u << i;
#pragma unroll
        for (int i = 0; i < P; i++) {
#pragma unroll
            for (int j = i + 1; j < P; j++) {
                const bool lt = v[i] < v[j];  // v_j >  v_i
                const bool eq = v[i] == v[j];
                ge[i] |= (u32)(lt | eq) << j;  // v_j >= v_i
                ge[j] |= (u32)(!lt) << i;      // v_i >= v_j
            }
        }
    }

    // slot S's value was raised to nv = v[S]: refresh row S and column S.
    template <int S> RG_HD void update(const u64 (&v)[P]) {
        const u64 nv = v[S];
        u32 row = 1u << S;
#pragma unroll
        for (int j = 0; j < P; j++) {
            if (j == S) continue;
            const bool lt = nv < v[j];
            const bool eq = nv == v[j];
            row |= (u32)(lt | eq) << j;                                // v_j >= v_S
            ge[j] = (ge[j] & ~(1u << S)) | ((u32)(!lt) << S);          // v_S >= v_j
        }
        ge[S] = row;
    }

    // q-th largest over the voter mask M (majority.rs:95-101); empty config => u64::MAX (:71-75).
    RG_HD u64 kth(const u64 (&v)[P], u32 M) const {
        const u32 n = (u32)__builtin_popcount(M);
        if (n == 0) return ~0ULL;
        const u32 q = n / 2u + 1u;
        u64 t = 0;
#pragma unroll
        for (int i = 0; i < P; i++) {
            const bool in = (M >> i) & 1u;
            const bool ok = in && ((u32)__builtin_popcount(ge[i] & M) >= q);
            const u64 c = ok ? v[i] : 0ULL;
            t = c > t ? c : t;
        }
        return t;
    }

    // JointConfig::committed_index (src/quorum/joint.rs:47-51): min of both majorities.
    RG_HD u64 mci(const u64 (&v)[P], u32 incoming, u32 outgoing) const {
        const u64 a = kth(v, incoming);
        const u64 b = kth(v, outgoing);
        return a < b ? a : b;
    }
};

// ---------------------------------------------------------------------------------------------
// The quorum index WHILE acks land one at a time (the exact replay of RgTick::commit_phase): JointConfig::committed_index
// (src/quorum/joint.rs:47-51) after every single Progress::maybe_update, without re-evaluating the quorum. Per majority
// config the pair (T, c) is carried along: T = the q-th largest matched index over the voter mask M (what
// MajorityConfig::committed_index returns, majority.rs:70-101), c = #{j in M : v_j > T}. T is the q-th largest iff
// c <= q - 1 and #{v_j >= T} >= q. An accepted ack raises ONE v_S from u to v (u < v); that moves T only if the voter sat at or
// below T and now lies above it (u <= T < v) while q - 1 voters were above T already (c = q - 1): then exactly q voters are
// above T and the q-th largest is the smallest of them; in every other case T stays and at most c grows by one. So a step
// costs a few compares, and the O(P) rescan only where the index really moves -- instead of a row + a column of the bit matrix
// and 2 P popcounts per ack (what made the replay 1 159 VALU per wave, profiles/r03_pmc_sq_c5_kinds.txt). The initial (T, c)
// come from one ordinary evaluation on the matches as they were before the tick.
// ---------------------------------------------------------------------------------------------
template <int P> struct RgRunningQuorum {
    struct One {   // one majority config (no arrays over the two configs: nothing here may ever be indexed dynamically)
        u64 T;     // an empty config is u64::MAX for good (majority.rs:71-75)
        u32 c, q, M;
    };
    One in, out; // incoming, outgoing majority

    RG_HD static void init_one(One &o, const RgQuorum<P> &qm, const u64 (&v)[P], u32 mask) {
        o.M = mask;
        o.q = (u32)__builtin_popcount(mask) / 2u + 1u;
        o.T = qm.kth(v, mask);
        u32 above = 0;
#pragma unroll
        for (int j = 0; j < P; j++) above += (((mask >> j) & 1u) && v[j] > o.T) ? 1u : 0u;
        o.c = above;
    }
    RG_HD void init(const RgQuorum<P> &qm, const u64 (&v)[P], u32 incoming, u32 outgoing) {
        init_one(in, qm, v, incoming);
        init_one(out, qm, v, outgoing);
    }
    // v[S] has just been raised from `u` (v already holds the new value)
    template <int S> RG_HD static void raise_one(One &o, const u64 (&v)[P], u64 u) {
        if (!((o.M >> S) & 1u) || !(u <= o.T && v[S] > o.T)) return;
        if (o.c + 1u < o.q) {
            o.c += 1u;
            return;
        }
        // q voters lie above T now: the q-th largest is the smallest of them
        u64 t = ~0ULL;
#pragma unroll
        for (int j = 0; j < P; j++)
            if (((o.M >> j) & 1u) && v[j] > o.T && v[j] < t) t = v[j];
        u32 above = 0;
#pragma unroll
        for (int j = 0; j < P; j++) above += (((o.M >> j) & 1u) && v[j] > t) ? 1u : 0u;
        o.T = t;
        o.c = above;
    }
    template <int S> RG_HD void raise(const u64 (&v)[P], u64 u) {
        raise_one<S>(in, v, u);
        raise_one<S>(out, v, u);
    }
    RG_HD u64 mci() const { return in.T < out.T ? in.T : out.T; }
};

// ---------------------------------------------------------------------------------------------
// Group commit (src/quorum/majority.rs:99-123), the literal algorithm: stable descending sort by
// index (voter order = slot order), Q = sorted[q-1], then the scan over sorted order. Ranks replace
// the sort so that all indexing stays static. Rare path (ProgressTracker.group_commit, tracker.rs:207).
// ---------------------------------------------------------------------------------------------
template <int P>
RG_HD u64 rg_majority_ci_group(const u64 (&v)[P], const u64 (&gid)[P], u32 M, bool &flag) {
    const u32 n = (u32)__builtin_popcount(M);
    if (n == 0) {
        flag = true;
        return ~0ULL;
    }
    const u32 q = n / 2u + 1u;
    u32 rank[P];
#pragma unroll
    for (int i = 0; i < P; i++) {
        u32 r = 0;
#pragma unroll
        for (int j = 0; j < P; j++) {
            if (j == i) continue;
            const bool before = (v[j] > v[i]) || (v[j] == v[i] && j < i);
            r += (((M >> j) & 1u) && before) ? 1u : 0u;
        }
        rank[i] = r;
    }
    u64 q_index = 0, q_gid = 0, last_index = 0;
#pragma unroll
    for (int i = 0; i < P; i++) {
        const bool in = (M >> i) & 1u;
        if (in && rank[i] == q - 1u) {
            q_index = v[i];
            q_gid = gid[i];
        }
        if (in && rank[i] == n - 1u) last_index = v[i];
    }
    u64 checked = q_gid, res = 0;
    bool single = true, done = false;
#pragma unroll
    for (int pos = 0; pos < P; pos++) {
#pragma unroll
        for (int i = 0; i < P; i++) {
            const bool here = ((M >> i) & 1u) && rank[i] == (u32)pos && !done;
            if (here) {
                const u64 g = gid[i];
                if (g == 0) {
                    single = false;
                } else if (checked == 0) {
                    checked = g;
                } else if (checked != g) {
                    res = v[i] < q_index ? v[i] : q_index;
                    done = true;
                }
            }
        }
    }
    if (done) {
        flag = true;
        return res;
    }
    flag = false;
    return single ? q_index : last_index;
}

// ProgressTracker::maximal_committed_index with group commit on (tracker.rs:294-298, joint.rs:47-51).
// Inlined, and every kernel has ONE call site of it (the tick's commit phase walks its evaluations in a loop around that one
// site: RgTick::commit_phase_gc): through round 5 this was a __noinline__ function taking the match and gid arrays by
// reference, which put both arrays -- and with them the group's whole register image -- into 400-512 B of scratch per lane
// in every group-commit kernel.
template <int P>
RG_HD u64 rg_mci_group(const u64 (&v)[P], const u64 (&gid)[P], u32 incoming, u32 outgoing, bool &used) {
    bool fi, fo;
    const u64 a = rg_majority_ci_group<P>(v, gid, incoming, fi);
    const u64 b = rg_majority_ci_group<P>(v, gid, outgoing, fo);
    used = fi && fo;
    return a < b ? a : b;
}

// ---------------------------------------------------------------------------------------------
// RaftLog::term (src/raft_log.rs:122-140) and RaftLog::find_conflict_by_term (:209-235) over the group's
// compact term-run table: dummy entry + up to RG_TERM_RUNS runs of older terms (ascending) + the implicit
// run [term_lo, last_index] of the leader's own term. Rare path (rejects with log_term > 0), so the
// table is read straight from its cold columns.
// ---------------------------------------------------------------------------------------------
// The leader-side view of the log a message of the tick is evaluated against. Normally [lo, last] is the leader's own
// run at cur_term and everything below comes from the table; when the tick carries RG_MF_BECOME_LEADER the previous
// leader run [xlo, lo) has just become an older run of term xterm and is not in the table yet (the tick kernel
// pushes it), so the pre-pass sees it through these two fields. xlo == lo: no such run.
//
// The table is BOUNDED (RG_TERM_RUNS runs of older terms, the newest ones: rg_push_run drops the oldest when it is full),
// the reference's log is not (RaftLog::term reads the whole log, raft_log.rs:122-140). What the device knows is the dummy
// entry and every index from `known` on; the terms of (dummy, known) are NOT on the device. A walk that would need one of
// them reports `unknown` instead of guessing -- the tick then leaves that one reject to the host (RG_OUT_HOST_HINT), which
// owns the log. With a table that still reaches down to the dummy entry (known == dummy + 1: every log that has seen at
// most RG_TERM_RUNS older terms since its last snapshot) nothing is ever unknown.
struct RgLogView {
    u64 lo, last, cur_term;
    u64 xlo, xterm;
    int skip;       // table run this tick's election is about to drop (table full: run 0, the oldest), -1 = none
    u64 dummy, dummy_term;
    u64 known, known_term; // first index above the dummy entry whose term the device knows, and that term
};

// (dummy, known): first used run of the table that survives this tick's push, else the previous leader's run, else the
// leader's own. Used runs come first, in ascending order.
RG_HD void rg_log_view_known(const RgState &st, u64 g, RgLogView &v) {
    v.dummy = st.dummy_idx[g];
    v.dummy_term = st.dummy_term[g];
    const int k0 = v.skip == 0 ? 1 : 0;
    const u64 f0 = k0 < RG_TERM_RUNS ? st.run_first[(u64)k0 * st.stride + g] : 0;
    if (f0 != 0) {
        v.known = f0;
        v.known_term = st.run_term[(u64)k0 * st.stride + g];
    } else if (v.xlo < v.lo) {
        v.known = v.xlo;
        v.known_term = v.xterm;
    } else {
        v.known = v.lo;
        v.known_term = v.cur_term;
    }
}

// RaftLog::term(idx) (raft_log.rs:122-140); `unk` is raised when idx lies in (dummy, known)
RG_HD u64 rg_log_term(const RgState &st, u64 g, const RgLogView &v, u64 idx, bool &unk) {
    if (idx < v.dummy || idx > v.last) return 0; // outside [dummy, last_index]
    if (idx >= v.lo) return v.cur_term;          // the leader's own entries (lo <= idx <= last_index)
    if (idx >= v.xlo) return v.xterm;            // the previous leader's run (election in this tick)
    if (idx == v.dummy) return v.dummy_term;
    if (idx < v.known) {
        unk = true;
        return 0;
    }
    u64 t = 0;
    for (int k = 0; k < RG_TERM_RUNS; k++) {
        const u64 first = st.run_first[(u64)k * st.stride + g];
        if (k != v.skip && first != 0 && first <= idx) t = st.run_term[(u64)k * st.stride + g];
    }
    return t;
}

// RaftLog::find_conflict_by_term (raft_log.rs:209-235). `unk`: the answer depends on terms the device no longer has.
// Log terms never decrease with the index, so every entry of (dummy, known) has a term in [dummy_term, known_term]: a
// walk that arrives there with term >= known_term stops on the spot (the entry's term is <= term whatever it is), one
// with term < dummy_term passes through the whole gap and the dummy entry; only dummy_term <= term < known_term needs
// the real terms.
RG_HD u64 rg_find_conflict_by_term(const RgState &st, u64 g, const RgLogView &v, u64 index, u64 term, bool &unk) {
    if (index > v.last) return index; // "out of range": returned as is (raft_log.rs:214-223)
    u64 ci = index;
    for (;;) { // every iteration leaves a whole run (or the dummy entry) behind: <= RG_TERM_RUNS + 5 rounds
        bool u = false;
        const u64 t = rg_log_term(st, g, v, ci, u);
        if (u) {
            if (term >= v.known_term) return ci;
            if (term >= v.dummy_term) {
                unk = true;
                return index;
            }
            ci = v.dummy; // every term of the gap is above `term`: the reference steps down to the dummy entry
            continue;
        }
        if (t <= term) return ci;
        // t > term: the reference steps ci -= 1 until the term changes; skip to just below this run
        u64 run_start;
        if (ci >= v.lo) {
            run_start = v.lo; // inside the leader's own run
        } else if (ci >= v.xlo) {
            run_start = v.xlo;
        } else {
            run_start = v.dummy; // ci == dummy: step below it
            for (int k = 0; k < RG_TERM_RUNS; k++) {
                const u64 first = st.run_first[(u64)k * st.stride + g];
                if (k != v.skip && first != 0 && first <= ci) run_start = first;
            }
        }
        ci = run_start - 1;
    }
}

// Raft::become_leader as a tick event (RG_MF_BECOME_LEADER on the leader's own slot, new term in m_hint): is the
// event well-formed? A new leader's term is higher than every term in its log (src/raft.rs:1284-1348: terms only
// grow); anything else is malformed input: RG_OUT_FAULT and the event is ignored.
RG_HD bool rg_election_valid(const RgState &st, const RgMsgs &ms, u64 g, u32 self, u64 &new_term) {
    new_term = ms.mh[(u64)self * st.stride + g];
    return new_term > st.cur_term[g];
}

// Pre-pass of a tick for ONE group: for every slot whose event is a reject with RG_MF_HAS_LOGTERM, store
// find_conflict_by_term(reject_hint, log_term) (or the hint itself when log_term == 0) into `rh`.
// last_index "at message time" is reproduced exactly: an election lands before every message of the tick, the
// leader's APPEND at its own slot, so slots after it see the grown log (RgTick::slot).
// A walk that needs log terms the bounded table no longer has (RgLogView) cannot be answered here. The group's byte of
// RG_COL_HOST_HINT receives the slots where that happens AND the tick is going to read the hint -- a reject reads it only in
// maybe_decr_to's Probe / Snapshot branch, when it is not stale and carries no snapshot request (progress.rs:188-203),
// which this pre-pass decides from the state the tick will see (an election of the same tick puts every peer in Probe
// with next = last_index + 1 first; nothing else before a slot's message changes its state or, outside Replicate, its
// next_idx). The tick leaves exactly those rejects alone and raises RG_OUT_HOST_HINT (RgTick::slot).
RG_HD void rg_resolve_hints(const RgState &st, const RgMsgs &ms, u64 g, u32 n_slots, u64 *rh, u32 *raised = nullptr) {
    const u64 mf = ms.mflags[g];
    const u32 cfg = st.cfg[g];
    const u32 self = RG_CFG_SELF(cfg), present = RG_CFG_PRESENT(cfg);
    bool any = false;
    for (u32 p = 0; p < n_slots; p++) {
        const u32 f = (u32)(mf >> (8 * p)) & 0xffu;
        any |= p != self && (f & (RG_MF_VALID | RG_MF_REJECT | RG_MF_HAS_LOGTERM)) == (RG_MF_VALID | RG_MF_REJECT | RG_MF_HAS_LOGTERM);
    }
    if (!any) return;
    RgLogView v;
    v.lo = st.lo[g];
    v.last = st.hi[g];
    v.cur_term = st.cur_term[g];
    v.xlo = v.lo;
    v.xterm = 0;
    v.skip = -1;
    u64 hi_after = v.last; // last_index once the leader's own slot has been processed
    bool elected = false;
    u64 next_elected = 0;
    if (self < n_slots && ((present >> self) & 1u)) {
        const u32 fs = (u32)(mf >> (8 * self)) & 0xffu;
        u64 new_term;
        if ((fs & RG_MF_BECOME_LEADER) && rg_election_valid(st, ms, g, self, new_term)) {
            const u64 old_lo = v.lo, old_hi = v.last;
            v.last = old_hi + 1; // the new leader's empty entry (raft.rs:1163-1194)
            v.lo = v.last;
            v.xlo = old_lo <= old_hi ? old_lo : v.lo; // the previous leader's entries keep their term
            v.xterm = v.cur_term;
            // pushing that run into a FULL table drops the oldest run (rg_push_run)
            if (old_lo <= old_hi && st.run_first[(u64)(RG_TERM_RUNS - 1) * st.stride + g] != 0) v.skip = 0;
            v.cur_term = new_term;
            hi_after = v.last;
            elected = true;
            next_elected = old_hi + 1; // Progress::reset(last_index + 1) of every follower (progress.rs:82-92)
        }
        if (fs & RG_MF_APPEND) {
            const u64 nl = ms.mc[(u64)self * st.stride + g];
            if (nl > hi_after) hi_after = nl;
        }
    }
    rg_log_view_known(st, g, v);
    const u64 hi_before = v.last;
    const u64 pf = st.pflags[g];
    u32 defer = 0;
    for (u32 p = 0; p < n_slots; p++) {
        const u32 f = (u32)(mf >> (8 * p)) & 0xffu;
        if (p == self) continue;
        if ((f & (RG_MF_VALID | RG_MF_REJECT | RG_MF_HAS_LOGTERM)) != (RG_MF_VALID | RG_MF_REJECT | RG_MF_HAS_LOGTERM)) continue;
        const u64 o = (u64)p * st.stride + g;
        const u64 lt = ms.mlt[o];
        u64 hint = ms.mh[o];
        v.last = p > self ? hi_after : hi_before;
        bool unk = false;
        if (lt > 0) hint = rg_find_conflict_by_term(st, g, v, hint, lt, unk);
        rh[o] = hint;
        if (unk && ((present >> p) & 1u) && !(f & RG_MF_HEARTBEAT)) {
            // will the tick read this hint? (RgTick::slot, the reject branch)
            const u32 state = elected ? RG_STATE_PROBE : (u32)(pf >> (8 * p)) & RG_PF_STATE_MASK;
            const u64 nx = elected ? next_elected : st.next[o];
            const u64 rs = (f & RG_MF_HAS_RS) ? ms.mrs[o] : 0ULL;
            if (state != RG_STATE_REPLICATE && rs == 0 && nx != 0 && nx - 1 == ms.mi[o]) defer |= 1u << p;
        }
    }
    st.hhint[g] = (u8)defer;
    // `raised` (optional): one word per launch that says "some reject of this tick was left to the host" -- what lets the engine
    // skip the check for unanswered hints (and its synchronisation) after every log-term tick that raised none
    if (defer && raised) {
#if defined(__HIP_DEVICE_COMPILE__)
        atomicOr(raised, 1u);
#else
        *raised |= 1u;
#endif
    }
}

// Does this tick carry RG_MF_BECOME_LEADER for the group (the REJECT bit of the leader's OWN slot)?
RG_HD bool rg_has_election(u64 mf, u32 cfg, u32 n_slots) {
    const u32 self = RG_CFG_SELF(cfg);
    return self < n_slots && ((RG_CFG_PRESENT(cfg) >> self) & 1u) && ((mf >> (8 * self)) & RG_MF_BECOME_LEADER);
}

// Commit publication (rg_publish.h): the group's accumulated delta byte is loaded with the group; after the
// tick `adv` = that byte + the tick's advance. Must run BEFORE the commit column is stored: the exact path
// re-reads the old commit index and the old byte from memory.
template <typename IX> RG_HD u32 rg_pub_load(const RgState &st, IX g) {
    return st.pub ? (u32) reinterpret_cast<const u8 *>(st.pub + st.pub_off_delta)[g] : 0u;
}
template <typename IX> RG_HD void rg_pub_store(const RgState &st, IX g, u32 adv, u64 new_commit) {
    u8 *dlt = reinterpret_cast<u8 *>(st.pub + st.pub_off_delta);
    if (adv > 255u) // rare: saturated byte, the excess goes to the exact-value list
        adv = rg_pub_accumulate(dlt[g], rg_at(st.commit, g), new_commit, g, reinterpret_cast<RgPubHdr *>(st.pub),
                                reinterpret_cast<RgPubOvf *>(st.pub + sizeof(RgPubHdr)), st.pub_cap);
    dlt[g] = (u8)adv;
}

// ---------------------------------------------------------------------------------------------
// One group's registers for a tick.
// ---------------------------------------------------------------------------------------------
#ifndef RG_HINT_COMPRESS_FROM /* slot counts from which the reject hints share two registers (99 = never: measured in round 5,
                                profiles/r05_c5_occupancy.txt -- 10 registers fewer at 7 slots, no wave gained, 87.5 -> 90.4 us on config 5) */
#define RG_HINT_COMPRESS_FROM 99
#endif
#define RG_HINT_REGS(P) ((P) >= RG_HINT_COMPRESS_FROM ? 2 : (P))
template <int P> struct RgGroup {
    u64 mt[P], nx[P], pc[P]; // Progress.matched / next_idx / committed_index   (in/out)
    u64 mi[P], mc[P];        // Message.index / Message.commit                  (in)
    u64 pf, mf;              // flag rows: RG_PF_* / RG_MF_* byte per slot
    u64 commit, lo, hi;      // RaftLog.committed, current-term index range [lo, hi], hi = last_index
    u32 cfg, out;
    u32 dirty;               // bit s: mt[s] changed, bit 8+s: nx[s], bit 16+s: pc[s], bit 24: pf, 25: commit, 26: hi,
                             // 27: lo, 28: cfg; bit 29: an election was applied in this tick
    u32 evm;                 // slots (with a Progress) that had any event since the state was loaded
    u32 adv;                 // commit publication: the group's delta byte of this interval + this tick's advance
                             // (values > 255 = "saturated, take the exact path"); 0 when not publishing
    // Operands of the rare paths, requested TOGETHER WITH the bulk loads by rg_prefetch_rare (NX mode RG_NX_PREFETCH):
    // reject hints (after find_conflict_by_term where the pre-pass ran) and what an election reads from the cold
    // columns. A wave then pays one memory round trip however many of its lanes take a rare path.
    // HN == P: one register per slot. HN == 2 (RG_HINT_REGS: the 7- and 8-slot bodies, where P registers cost a wave of
    // occupancy): the hints of the first two slots that need one, the election's new term first (`hsel`: bits 0-3 the slot of
    // hint[0] + 1, bits 4-7 the slot of hint[1] + 1, 0 = none); a third reject of the same group in one tick -- rare even
    // under rollover (0.12 rejects per group-tick in BASELINE config 5) -- fetches its hint on the spot.
    static constexpr int HN = RG_HINT_REGS(P);
    u64 hint[HN];
    u32 hsel;
    u64 pc_self;        // early stores / late loads (RgTick's ES): the leader's own committed_index while pc[] is not in registers
    u64 mc_self_v;      // late loads: m_commit of the leader's own slot (RG_MF_APPEND: the new last_index), fetched with the rare batch
    u32 el_n;           // an election: the term-run table's fill count (RG_COL_RUN_COUNT) -- with it rg_push_run files the previous
                        // leader's run on the spot, two stores and no look at the table (round 3 read the table for the first
                        // unused run, behind the group's stores: two dependent round trips at the tail of every wave that holds
                        // an electing group, 7 of config 5's 100 us)
    u64 el_old;         // an election: the current term (the new one comes in hint[self]) = the term of the previous
                        // leader's entries, the run rg_push_run files (round 2 requested the table's CELLS early, into the
                        // registers the dead message columns leave: with RG_TERM_RUNS = 8 that cost the dense kernel a wave of
                        // occupancy at P = 5; the count is one register that is dead again before the slots are walked)
};
#if defined(__HIP_DEVICE_COMPILE__)
// The value stays what it is, but the compiler may not look through: without this it computes `hint + 1` (and the
// election's term comparison) inside the very branch that issued the prefetch, i.e. waits for the load on the spot.
#define RG_OPAQUE64(x) asm volatile("" : "+v"(x))
#define RG_OPAQUE32(x) asm volatile("" : "+v"(x))
#else
#define RG_OPAQUE64(x) (void)(x)
#define RG_OPAQUE32(x) (void)(x)
#endif
// How a tick gets at Progress.next_idx (and the other rarely needed operands):
#define RG_NX_LOADED 0   /* the caller loaded r.nx[] for every slot (LDS-staged variants) */
#define RG_NX_LAZY 1     /* fetched on demand, cell by cell, where the old value can matter (fused launches) */
#define RG_NX_PREFETCH 2 /* rg_prefetch_rare has requested the needed cells, the reject hints and the election's cold
                            cells in one batch behind the bulk loads (lane / list kernels) */
#define RG_DIRTY_PF (1u << 24)
#define RG_DIRTY_COMMIT (1u << 25)
#define RG_DIRTY_HI (1u << 26)
#define RG_DIRTY_LO (1u << 27)      /* term_lo changed (an election) */
#define RG_DIRTY_CFG (1u << 28)     /* the cfg word changed (an election aborts a leader transfer) */
#define RG_TICK_ELECTED (1u << 29)  /* not a store bit: RG_MF_BECOME_LEADER was applied in THIS tick */

// RawNode::report_unreachable / report_snapshot (src/raw_node.rs:692-709) on ONE Progress cell: handle_unreachable
// (src/raft.rs:1931-1954) and handle_snapshot_status (:1891-1929). `pf` is the cell's flag byte. Returns whether
// Progress::reset_state ran (progress.rs:75-80: the caller resets a device-side window and re-derives the engine-owned bits).
RG_HD bool rg_apply_progress_event(u32 kind, u64 match, u64 &next, u64 &psnap, u64 &prs, u32 &pf) {
    const u32 state = pf & RG_PF_STATE_MASK;
    if (kind == RG_EV_UNREACHABLE) {
        // "During optimistic replication, if the remote becomes unreachable, there is huge probability that a MsgAppend is lost"
        if (state != RG_STATE_REPLICATE) return false;
        pf = (pf & ~(RG_PF_STATE_MASK | RG_PF_PAUSED)) | RG_STATE_PROBE; // become_probe(): reset_state(Probe) ...
        psnap = 0;
        next = match + 1; // ... next_idx = matched + 1
        return true;
    }
    if (kind == RG_EV_SNAPSHOT_FINISH || kind == RG_EV_SNAPSHOT_FAILURE) {
        if (state != RG_STATE_SNAPSHOT) return false;
        const u64 pending = kind == RG_EV_SNAPSHOT_FAILURE ? 0 : psnap; // snapshot_failure() (progress.rs:124-127)
        next = rg_max(match + 1, pending + 1);                           // become_probe() from Snapshot (progress.rs:99-102)
        psnap = 0;
        pf = (pf & ~RG_PF_STATE_MASK) | RG_STATE_PROBE | RG_PF_PAUSED; // wait for the msgAppResp / a heartbeat interval: pause()
        prs = 0;                                                       // pending_request_snapshot = INVALID_INDEX
        return true;
    }
    return false;
}


// Record i of an rg_progress_events batch against the columns: the whole run of records of its (group, slot) is applied, in
// order, by the run's first record (k_progress_events: one lane per record; tests/host_check: a loop).
RG_HD void rg_progress_events_at(const RgState &st, u32 *ins_meta, const rg_progress_event *ev, u64 n, u32 P, u64 i) {
    const u64 g = ev[i].group;
    const u32 s = ev[i].slot;
    if (g >= st.G || s >= P) return;
    if (i > 0 && ev[i - 1].group == g && ev[i - 1].slot == s) return;
    if (!((RG_CFG_PRESENT(st.cfg[g]) >> s) & 1u)) return; // "no progress available for {}": ignored (raft.rs:1893-1901, :1933-1942)
    const u64 o = (u64)s * st.stride + g;
    u8 *pfb = reinterpret_cast<u8 *>(st.pflags) + g * 8 + s;
    u32 pf = *pfb;
    const u64 match = st.match[o];
    u64 next = st.next[o], psnap = st.psnap[o], prs = st.prs[o];
    bool reset = false;
    for (u64 j = i; j < n && ev[j].group == g && ev[j].slot == s; j++)
        reset = rg_apply_progress_event(ev[j].kind, match, next, psnap, prs, pf) || reset;
    if (!reset) return; // (an event that does not apply changes nothing)
    st.next[o] = next;
    st.psnap[o] = psnap;
    st.prs[o] = prs;
    if (ins_meta) { // Progress::reset_state: ins.reset() (progress.rs:75-80); an empty window is not full
        ins_meta[o] = 0;
        pf &= ~RG_PF_INS_FULL;
    }
    pf = (pf & ~RG_PF_PENDING) | (psnap ? RG_PF_PEND_SNAP : 0u) | (prs ? RG_PF_PEND_RS : 0u);
    *pfb = (u8)pf;
}

// rg_resolve_host_hints: a reject the tick left to the host (RG_OUT_HOST_HINT) comes back with its hint resolved -- the
// rest of handle_append_response's reject branch (raft.rs:1679-1721) for ONE cell: Progress::maybe_decr_to(index, hint, 0)
// (progress.rs:168-206), become_probe when that leaves Replicate (:95-107). recent_active and update_committed were applied by
// the tick. Returns maybe_decr_to's result: send_append(from) is due (raft.rs:1719).
RG_HD bool rg_apply_resolved_reject(u64 match, u64 &next, u64 &psnap, u32 &pf, u64 index, u64 hint, bool &left_replicate) {
    const u32 state = pf & RG_PF_STATE_MASK;
    left_replicate = false;
    if (state == RG_STATE_REPLICATE) {
        // (a deferred reject never meets this state -- the tick reads a hint only outside Replicate; literal all the same)
        if (index < match || index == match) return false; // stale (request_snapshot == 0)
        pf = (pf & ~(RG_PF_STATE_MASK | RG_PF_PAUSED)) | RG_STATE_PROBE; // become_probe(): reset_state(Probe) ...
        psnap = 0;
        next = match + 1; // ... next_idx = matched + 1 (what maybe_decr_to had set as well)
        left_replicate = true;
        return true;
    }
    if (next == 0 || next - 1 != index) return false; // stale
    const u64 h = hint + 1;
    u64 n = index < h ? index : h;
    if (n < 1) n = 1;
    next = n;
    pf &= ~RG_PF_PAUSED; // resume()
    return true;
}

// Record i of an rg_resolve_host_hints batch against the columns (k_resolve_apply: one lane per record; tests/host_check: a loop).
// A record answers ONE deferred reject: bit `slot` of the group's RG_COL_HOST_HINT byte. A record for a (group, slot) that is
// not waiting -- never flagged, or answered already -- changes nothing (returns false). The group's RG_OUT_HOST_HINT bit, which
// holds back the group's send requests on an engine with device Inflights, falls only when its LAST waiting slot is answered:
// a host that passes some of a group's rejects now and the rest later releases the stage with the last of them.
// `take(g, s)`: atomically clear bit s of hhint[g] and return the byte as it was (records of one group may sit in different
// lanes); `orw(g, bits, clear)`: out[g] = (out[g] | bits) & ~clear, atomically on the device.
// Returns bit 0: maybe_decr_to applied (send_append is due); bit 1: the record answered a waiting slot; bit 2: it was the group's
// last one -- RG_OUT_HOST_HINT fell, the group's held-back send requests can be served.
#define RG_RESOLVE_APPLIED 1u
#define RG_RESOLVE_TAKEN 2u
#define RG_RESOLVE_RELEASED 4u
template <typename TAKE, typename ORW>
RG_HD u32 rg_resolve_hint_at(const RgState &st, u32 *ins_meta, const rg_resolved_hint *it, u32 P, u64 i, TAKE &&take, ORW &&orw) {
    const u64 g = it[i].group;
    const u32 s = it[i].slot;
    if (g >= st.G || s >= P) return 0u;
    if (!((RG_CFG_PRESENT(st.cfg[g]) >> s) & 1u)) return 0u;
    if (!(st.out[g] & RG_OUT_HOST_HINT)) return 0u; // (the byte is only meaningful under the bit: raftgroups.h)
    const u32 before = take(g, s);
    if (!((before >> s) & 1u)) return 0u;
    const u64 o = (u64)s * st.stride + g;
    u8 *pfb = reinterpret_cast<u8 *>(st.pflags) + g * 8 + s;
    u32 pf = *pfb;
    u64 next = st.next[o], psnap = st.psnap[o];
    bool left = false;
    const bool dec = rg_apply_resolved_reject(st.match[o], next, psnap, pf, it[i].index, it[i].hint, left);
    if (dec) {
        st.next[o] = next;
        if (left) {
            st.psnap[o] = 0;
            pf &= ~RG_PF_PEND_SNAP;
            if (ins_meta) { // Progress::reset_state: ins.reset() (progress.rs:75-80)
                ins_meta[o] = 0;
                pf &= ~RG_PF_INS_FULL;
            }
        }
        *pfb = (u8)pf;
    }
    const bool last = (before & ~(1u << s) & 0xffu) == 0;
    orw(g, dec ? 1u << (8 + s) : 0u, last ? (u32)RG_OUT_HOST_HINT : 0u);
    return (dec ? RG_RESOLVE_APPLIED : 0u) | RG_RESOLVE_TAKEN | (last ? RG_RESOLVE_RELEASED : 0u);
}

// RaftLog::maybe_commit (src/raft_log.rs:487-499) with term(mci)==cur_term restated as lo<=mci<=hi
// (log terms are non-decreasing, so the entries of the leader's term are one contiguous range that
// ends at last_index); commit_to (:286-300) can then never exceed last_index.
RG_HD bool rg_log_maybe_commit(u64 mci, u64 &commit, u64 lo, u64 hi) {
    if (mci > commit && mci >= lo && mci <= hi) {
        commit = mci;
        return true;
    }
    return false;
}

// The previous leader's entries [first, ...] of term `term` become one more run of the group's term-run table
// (RgTick::become_leader). Used runs come first; when all RG_TERM_RUNS are in use the OLDEST run is dropped: the table
// then starts above the dummy entry, and a find_conflict_by_term walk that would need the dropped terms is handed to
// the host instead of being answered (RgLogView, RG_OUT_HOST_HINT; include/raftgroups.h: RG_COL_RUN_FIRST).
// `n` = the group's RG_COL_RUN_COUNT byte (how many runs the table holds): an engine-owned count kept beside the table so that
// filing a run needs no look at the table itself -- round 3 found the first unused run by reading it, two dependent memory
// round trips at the tail of every wave that holds an electing group (87 % of the waves of BASELINE config 5: 7 us of its
// 100 us tick, profiles/r04_c5_one_launch.txt). The caller reads the byte (rg_store_group: together with term_lo, ahead of
// the group's stores); only a FULL table is still read, to move its runs down.
template <typename IX> RG_HD void rg_push_run(const RgState &st, IX g, u64 first, u64 term, u32 n) {
    u32 k = n;
    if (k >= RG_TERM_RUNS) { // table full: the oldest run goes
#pragma unroll 1
        for (int j = 0; j + 1 < RG_TERM_RUNS; j++) {
            rg_at(st.run_first, (IX)j * (IX)st.stride + g) = rg_at(st.run_first, (IX)(j + 1) * (IX)st.stride + g);
            rg_at(st.run_term, (IX)j * (IX)st.stride + g) = rg_at(st.run_term, (IX)(j + 1) * (IX)st.stride + g);
        }
        k = RG_TERM_RUNS - 1;
    } else {
        rg_at(rg_run_n(st), g) = (u8)(k + 1u);
    }
    rg_at(st.run_first, (IX)k * (IX)st.stride + g) = first;
    rg_at(st.run_term, (IX)k * (IX)st.stride + g) = term;
}
// What RG_COL_RUN_COUNT holds for a table as loaded (used runs come first): the engine derives it whenever the host loads
// RG_COL_RUN_FIRST (k_fix_run_count), the host twin of the tests on its way in.
RG_HD u32 rg_count_runs(const RgState &st, u64 g) {
    u32 n = 0;
    for (int k = 0; k < RG_TERM_RUNS; k++) n += st.run_first[(u64)k * st.stride + g] != 0 ? 1u : 0u;
    return n;
}
// RG_NX_PREFETCH: everything a tick may read beyond the bulk columns, decided from the two flag rows and the cfg
// word alone and requested in ONE batch right behind the bulk loads -- the old `next` of the slots where it can
// matter (see RgTick: not where SENT on a Replicate peer overwrites it first), the reject hint of every slot whose
// reject can reach maybe_decr_to's Probe/Snapshot branch (progress.rs:188-203), and the cold cells an election
// reads (new term, current term; the term-run table is only touched after the group's stores, rg_push_run). On demand, each of these was a dependent
// memory round trip in the middle of the tick, taken by the WHOLE wave as soon as one lane needed it: under leader-term
// rollover (BASELINE config 5) practically every wave needed a dozen of them, one after the other.
// Every destination is written once, before its load is issued, and not touched again until the tick reads it: a
// later write (even under a disjoint exec mask) would make the compiler wait for the load first.
struct RgNoEarlyStores; // (the tick's store / load policy, below: ES)
template <int P, typename IX, typename ES = RgNoEarlyStores> RG_HD void rg_prefetch_rare(RgGroup<P> &r, const RgState &st, const RgMsgs &ms, IX g) {
    const u32 cfg = r.cfg;
    const u32 self = RG_CFG_SELF(cfg), present = RG_CFG_PRESENT(cfg);
    const bool elect = rg_has_election(r.mf, cfg, P);
    // (ES::late_pc: the own slot's committed_index and Message.commit travel with this batch; a no-op otherwise)
    ES::template load_self<P, IX>(r, st, ms, g, self, self < (u32)P && ((present >> self) & 1u) && ((r.mf >> (8 * self)) & 0xffULL) != 0);
    r.el_old = 0;
    r.el_n = 0;
    if (elect) {
        r.el_old = rg_at(st.cur_term, g); // (the new term arrives in r.hint[self], below)
        r.el_n = (u32)rg_at(rg_run_n(st), g);
    }
    constexpr bool HC = RgGroup<P>::HN != P; // compressed hints
    u32 hneed = 0, hlt = 0;                  // HC: slots that need a hint / of those, the ones that read the pre-pass's column
#pragma unroll
    for (int i = 0; i < P; i++) {
        const u32 f = (u32)(r.mf >> (8 * i)) & 0xffu, pb = (u32)(r.pf >> (8 * i)) & 0xffu;
        const IX o = (IX)i * (IX)st.stride + g;
        const bool here = ((present >> i) & 1u) != 0;
        const bool repl = (pb & RG_PF_STATE_MASK) == RG_STATE_REPLICATE;
        // (an election rewrites every present cell of `next` before anything reads it -- but whether the event is
        // well-formed is only known once its own cells have arrived, so electing groups fetch theirs like the others)
        const bool overwritten = (u32)i != self && (f & RG_MF_SENT) && repl;
        r.nx[i] = 0;
        if (here && f != 0 && !overwritten) r.nx[i] = rg_at(st.next, o);
        // a reject that is not a heartbeat response reads its hint in maybe_decr_to's Probe / Snapshot branch (unless
        // it carries a snapshot request: known only with m_rs); after an election every follower is in Probe
        const u32 rj = RG_MF_VALID | RG_MF_REJECT;
        // (the leader's own slot has no hint: its register carries the new term of an election, m_hint of that slot)
        const bool need = (u32)i == self ? elect : here && (f & (rj | RG_MF_HEARTBEAT)) == rj && (elect || !repl);
        const bool from_prepass = (f & RG_MF_HAS_LOGTERM) && (u32)i != self;
        if constexpr (!HC) {
            r.hint[i] = 0;
            if (need) r.hint[i] = from_prepass ? rg_at(ms.mhr, o) : rg_at(ms.mh, o);
        } else {
            hneed |= need ? 1u << i : 0u;
            hlt |= from_prepass ? 1u << i : 0u;
        }
    }
    if constexpr (HC) {
        // two loads, each into its own register, whatever slots they serve: the ADDRESS is selected, not the loaded value (a
        // select over loaded values would make the wave wait for the loads right here)
        u32 s0 = 0, s1 = 0; // slot + 1
        if (elect) s0 = self + 1u;
        u32 rest = hneed & ~(elect ? 1u << self : 0u);
        if (!s0 && rest) {
            s0 = (u32)__builtin_ctz(rest) + 1u;
            rest &= rest - 1u;
        }
        if (rest) s1 = (u32)__builtin_ctz(rest) + 1u;
        r.hsel = s0 | (s1 << 4);
        r.hint[0] = 0;
        r.hint[1] = 0;
        if (s0) r.hint[0] = ((hlt >> (s0 - 1u)) & 1u) ? rg_at(ms.mhr, (IX)(s0 - 1u) * (IX)st.stride + g) : rg_at(ms.mh, (IX)(s0 - 1u) * (IX)st.stride + g);
        if (s1) r.hint[1] = ((hlt >> (s1 - 1u)) & 1u) ? rg_at(ms.mhr, (IX)(s1 - 1u) * (IX)st.stride + g) : rg_at(ms.mh, (IX)(s1 - 1u) * (IX)st.stride + g);
    } else {
        r.hsel = 0;
    }
}

// The prefetched hint of slot s (RG_NX_PREFETCH). Compressed form: one of the two registers, or -- a third reject of the group
// in this tick -- fetched now.
template <int P, typename IX> RG_HD u64 rg_hint_of(const RgGroup<P> &r, const RgState &st, const RgMsgs &ms, IX g, u32 s, bool from_prepass) {
    if constexpr (RgGroup<P>::HN == P) {
        u64 h = 0;
#pragma unroll
        for (int i = 0; i < P; i++)
            if ((u32)i == s) h = r.hint[i];
        return h;
    } else {
        if ((r.hsel & 0xfu) == s + 1u) return r.hint[0];
        if (((r.hsel >> 4) & 0xfu) == s + 1u) return r.hint[1];
        const IX o = (IX)s * (IX)st.stride + g;
        return from_prepass ? rg_at(ms.mhr, o) : rg_at(ms.mh, o);
    }
}

// Raft::handle_append_response for every slot of one group, in slot order (src/raft.rs:1559-1775),
// on_persist_entries for the leader's own slot (src/raft.rs:994-1016).
// Cold columns (pending_snapshot, pending_request_snapshot, commit_group_id, reject_hint,
// request_snapshot) are touched through `st`/`ms` only on the rare paths that need them.
// GC = false compiles the group-commit path out (the engine launches the GC = true kernel only
// when some group has ProgressTracker.group_commit set).
// FUSED: the group's state stays in registers across several ticks (k_tick_fused): `dirty`/`evm`
// accumulate, and `next` cells already fetched or written are not fetched again.
// ES -- early stores (the 7- and 8-slot lane bodies, rg_tick_kernels.h: RgEarlyStores). A tick's registers peak in the commit
// phase, where the matches, the parked old matches, the quorum state AND the two columns that are merely waiting for the
// group's stores -- `next` and `committed_index`, 4 P registers -- are all live. Neither is read again once its slot is done:
// update_committed is applied to every slot up front (peer_committed), `next` is final when slot<S>() returns. With ES::on
// they are stored on the spot -- the followers' committed_index right behind peer_committed (uniform control flow: whole-line
// ballots as in rg_store_group), `next` behind each slot -- and only the leader's own committed_index (raft.rs:896-900 raises
// it to the new commit index at the very end) travels on, in ONE register pair (RgGroup::pc_self), to be stored with the group. Same cells, same values, same
// number of stores; what changes is when they are issued and that 4 P - 2 registers are free when the quorum is evaluated.
// ES::late_pc -- late loads (round 5, second attempt at the 7-slot body's registers; rg_tick_kernels.h: RgLatePc). The peak of
// the allocation is the load phase: six u64 operands per slot in flight at once. Two of them, `committed_index` and
// Message.commit, are needed by nothing but each other (update_committed) -- except on the leader's own slot (the new last_index
// of RG_MF_APPEND; the election's and the commit phase's `prs[self].update_committed`). With ES::late_pc those two cells of the
// own slot ride with the rare-path prefetch batch (pc_self, mc_self_v), the 2 P column loads are issued only when the slots
// have been walked -- `next`, the hints and the election operands are dead by then -- and land while the quorum is evaluated;
// update_committed runs behind the commit phase. Same loads, same stores, same values; 4 P registers less at the peak.
struct RgNoEarlyStores {
    static constexpr bool on = false;
    static constexpr bool late_pc = false;
    template <int P, typename IX> RG_HD static void load_self(RgGroup<P> &, const RgState &, const RgMsgs &, IX, u32, bool) {}
    template <int P, typename IX> RG_HD static void load_pc_mc(RgGroup<P> &, const RgState &, const RgMsgs &, IX) {}
    template <int P, typename IX> RG_HD static void store_pc(RgGroup<P> &, const RgState &, IX, u32) {}
    template <int S, int P, typename IX> RG_HD static void store_next(RgGroup<P> &, const RgState &, IX) {}
};
template <int P, bool GC, int NXM, bool FUSED, typename IX, typename ES = RgNoEarlyStores> struct RgTick {
    static_assert(!(ES::on || ES::late_pc) || !FUSED, "early stores / late loads: single-tick kernels only");
    static_assert(!(ES::on && ES::late_pc), "early stores and late loads exclude each other");
    static constexpr bool LAZY_NX = NXM == RG_NX_LAZY;
    static constexpr bool PREF = NXM == RG_NX_PREFETCH;
    RgGroup<P> &r;
    const RgState &st;
    const RgMsgs &ms;
    const IX g;
    u64 last0; // last_index the send path saw before this tick
    u32 self, present, out; // (the voter masks and the transferee are extracted from r.cfg where they are used)
    u64 mc_self;  // m_commit of the leader's own slot
    u32 acc;      // slots whose maybe_update returned true this tick (each is followed by a maybe_commit)
    u32 acc_oldp; // ... of those, the ones that were paused before the ack (raft.rs:1724,1749-1751)

    RG_HD RgTick(RgGroup<P> &r_, const RgState &st_, const RgMsgs &ms_, IX g_)
        : r(r_), st(st_), ms(ms_), g(g_) {
        const u32 cfg = r.cfg;
        self = RG_CFG_SELF(cfg);
        present = RG_CFG_PRESENT(cfg);
        last0 = r.hi;
        out = 0;
        acc = 0;
        acc_oldp = 0;
        if (!FUSED) {
            r.dirty = 0;
            r.evm = 0;
        }
#pragma unroll
        for (int i = 0; i < P; i++) // slots without a Progress ack 0 and are never written back
            if (!((present >> i) & 1u)) r.mt[i] = 0;
        // An election lands before every message of the tick (they answer the NEW leader). In a fused launch the group's
        // registers simply carry on with the new leader's state; the cold cells (RG_COL_CUR_TERM, the term-run table) are
        // read and written in memory on the spot, so a second election of the same group later in the launch sees them.
#ifndef RG_NO_ELECT /* (measurement builds only: python -m raft_rs_amd.build --exp noelect -DRG_NO_ELECT) */
        if (rg_has_election(r.mf, r.cfg, P)) become_leader();
#endif
#pragma unroll
        for (int i = 0; i < P; i++)
            if (((r.mf >> (8 * i)) & 0xffULL) && ((present >> i) & 1u)) r.evm |= 1u << i;
        if (LAZY_NX) {
            // next_idx is read only where its old value can matter: a slot with an event, unless the event
            // set starts with SENT on a Replicate peer, which overwrites it (optimistic_update,
            // progress.rs:161) before anything reads it. In steady state that is every follower, so the
            // `next` column is written but (except for the leader's own slot) never read.
            u32 need_mask = 0;
#pragma unroll
            for (int i = 0; i < P; i++) {
                const u32 f = (u32)(r.mf >> (8 * i)) & 0xffu, pb = (u32)(r.pf >> (8 * i)) & 0xffu;
                const bool overwritten = (u32)i != self && (f & RG_MF_SENT) && (pb & RG_PF_STATE_MASK) == RG_STATE_REPLICATE;
                // FUSED: a cell fetched or written by an earlier tick of this launch is already current
                // (bit 8+i of dirty doubles as "r.nx[i] is valid": fetched cells are marked too -- rewriting
                // an unchanged value is harmless)
                // (an election of this tick has just written every present cell: same bit)
                const bool have = (r.dirty >> (8 + i)) & 1u;
                const bool need = ((present >> i) & 1u) && f != 0 && !overwritten && !have;
                // (a register is written BEFORE its load is issued, never after: a write behind a pending load -- even
                // on the lanes that do not load -- makes the compiler wait for the load, slot by slot)
                if (!have) r.nx[i] = 0ULL;
                need_mask |= need ? 1u << i : 0u;
            }
#pragma unroll
            for (int i = 0; i < P; i++) {
                if ((need_mask >> i) & 1u) {
                    r.nx[i] = rg_at(st.next, (IX)i * (IX)st.stride + g);
                    if (FUSED) r.dirty |= 1u << (8 + i);
                }
            }
        }
    }

    // RG_MF_BECOME_LEADER: Raft::reset(term) (src/raft.rs:942-971) + Raft::become_leader (:1151-1202) for this
    // group -- what the reference runs when this node wins the group's election -- over the group's registers:
    // every Progress is reset to Progress::reset(last_index + 1) (progress.rs:82-92: matched 0, Probe, not paused,
    // no pending snapshot / snapshot request, not recently active, empty Inflights; committed_index and
    // commit_group_id survive); the leader's own keeps matched = persisted, takes committed_index = committed and
    // becomes Replicate (:1176-1181); a leader transfer is aborted (:953); the new leader's empty entry is appended
    // at last_index + 1 (:1191-1194), which starts the index range of the new term; the previous leader's range
    // becomes one more run of the term table. Cold columns are written straight to memory (rare path).
    // Malformed (term not above RG_COL_CUR_TERM): RG_OUT_FAULT, ignored.
    RG_HD void become_leader() {
        // everything this rare path reads from the cold columns is requested at once: ONE memory round trip
        // (a wave of config 5 almost always has an electing lane, and it waits for that lane)
        // (RG_NX_PREFETCH: they were requested with the group's bulk loads, rg_prefetch_rare)
        u64 new_term, old_term;
        if (PREF) {
            new_term = rg_hint_of<P, IX>(r, st, ms, g, self, false); // (compressed hints: the election's term is always hint[0])
            old_term = r.el_old;
            RG_OPAQUE64(new_term);
            RG_OPAQUE64(old_term);
        } else {
            new_term = rg_at(ms.mh, (IX)self * (IX)st.stride + g);
            old_term = rg_at(st.cur_term, g);
        }
        if (!(new_term > old_term)) { // rg_election_valid
            out |= RG_OUT_FAULT;
            return;
        }
        const u64 old_lo = r.lo, old_hi = r.hi;
        rg_at(st.cur_term, g) = new_term;
        if (old_lo <= old_hi) { // the previous leader's entries become one more run of an older term
            // (PREF: the fill count came with the prefetch batch -- the stores go out here and now, nothing waits for them)
            u32 n = r.el_n;
            if (PREF) RG_OPAQUE32(n);
            else n = (u32)rg_at(rg_run_n(st), g);
            rg_push_run<IX>(st, g, old_lo, old_term, n);
        }
#pragma unroll
        for (int i = 0; i < P; i++) {
            if (!((present >> i) & 1u)) continue;
            const IX o = (IX)i * (IX)st.stride + g;
            const u32 pb = (u32)(r.pf >> (8 * i)) & 0xffu;
            u32 nb;
            if ((u32)i == self) {
                // assert_eq!(last_index, self.raft_log.persisted) (raft.rs:1170): matched IS the persisted index
                if (r.mt[i] != old_hi) out |= RG_OUT_FAULT;
                if constexpr (ES::late_pc) { // (pc[] is not here yet: the own cell came with the rare batch)
                    if (r.pc_self != r.commit) {
                        r.pc_self = r.commit;
                        r.dirty |= 1u << (16 + i);
                    }
                } else if (r.pc[i] != r.commit) {
                    r.pc[i] = r.commit;
                    r.dirty |= 1u << (16 + i);
                }
                r.nx[i] = r.mt[i] + 1; // become_replicate (progress.rs:111-114)
                nb = (pb & RG_PF_PENDING_CONF) | RG_STATE_REPLICATE;
            } else {
                r.mt[i] = 0;
                r.nx[i] = old_hi + 1;
                nb = RG_STATE_PROBE; // ins.reset(): RG_OUT_BECAME_LEADER tells the send stage to empty the device window
            }
            r.dirty |= (1u << i) | (1u << (8 + i));
            // (RG_PF_PEND_*: almost never -- these were the 2 P cold stores an election used to cost)
            if (pb & RG_PF_PEND_SNAP) rg_at(st.psnap, o) = 0;
            if (pb & RG_PF_PEND_RS) rg_at(st.prs, o) = 0;
            if (nb != pb) {
                r.pf = (r.pf & ~(0xffULL << (8 * i))) | ((u64)nb << (8 * i));
                r.dirty |= RG_DIRTY_PF;
            }
        }
        if (RG_CFG_TRANSFEREE(r.cfg)) { // abort_leader_transfer (raft.rs:953)
            r.cfg &= ~(0xfu << 20);
            r.dirty |= RG_DIRTY_CFG;
        }
        r.hi = old_hi + 1; // append_entry(&mut [Entry::default()]) (raft.rs:1191-1194)
        r.lo = r.hi;
        last0 = r.hi;      // what the send path sees once the election is over
        r.dirty |= RG_DIRTY_HI | RG_DIRTY_LO | RG_TICK_ELECTED;
        out |= RG_OUT_BECAME_LEADER | RG_OUT_APPENDED; // the caller follows with bcast_append (raft.rs:2190-2191)
    }

    // ProgressTracker::maximal_committed_index (tracker.rs:294-298) over the matches `v` (groups with group commit on go
    // through commit_phase_gc instead).
    RG_HD u64 mci_of(const RgQuorum<P> &qm, const u64 (&v)[P]) {
        return qm.mci(v, RG_CFG_INCOMING(r.cfg), RG_CFG_OUTGOING(r.cfg));
    }

    // Progress::reset_state (progress.rs:75-80): paused=false, pending_snapshot=0, state; the
    // Inflights reset is the host's (it sees the transition through the state column).
    RG_HD void reset_state(u32 &pb, u32 new_state, IX o) {
        pb = (pb & ~(RG_PF_PAUSED | RG_PF_STATE_MASK)) | new_state;
        // (stored unconditionally: making this one store conditional on RG_PF_PEND_SNAP costs the kernel 4 VGPRs = a wave of
        // occupancy at P = 5; the election's 2 P stores are the ones worth skipping)
        rg_at(st.psnap, o) = 0;
        pb &= ~RG_PF_PEND_SNAP;
    }

    template <int S> RG_HD void set_next(u64 n) {
        r.nx[S] = n;
        r.dirty |= 1u << (8 + S);
    }

    // Progress::maybe_update (progress.rs:138-150); returns need_update
    template <int S> RG_HD bool maybe_update(u64 idx, u32 &pb) {
        const bool upd = r.mt[S] < idx;
        if (upd) {
            // the matched index this ack replaces is parked where the message's index was (r.mi[S] == idx is not read
            // again): the replay of commit_phase() needs the matches as they were BEFORE each ack, and gets them
            // without re-reading memory or spending registers on a snapshot
            r.mi[S] = r.mt[S];
            r.mt[S] = idx;
            r.dirty |= 1u << S;
            pb &= ~RG_PF_PAUSED; // resume()
            acc |= 1u << S;      // the reference now calls maybe_commit(): evaluated in commit_phase()
        }
        if (r.nx[S] < idx + 1) set_next<S>(idx + 1);
        return upd;
    }

    // Progress::update_committed(m.commit) (progress.rs:153-157) of a follower's AppendResponse (raft.rs:1677) or
    // HeartbeatResponse (:1781): unconditional once the peer has a Progress, and nothing between the start of the tick and
    // that line reads pr.committed_index, so it is applied to every slot up front -- Message.commit is then dead before the
    // slots are walked (registers).
    template <int S> RG_HD void peer_committed() {
        const u32 f = (u32)(r.mf >> (8 * S)) & 0xffu;
        if ((u32)S == self) mc_self = r.mc[S]; // (RG_MF_APPEND: the leader's new last_index)
        if ((u32)S == self || !((present >> S) & 1u) || !(f & (RG_MF_VALID | RG_MF_HEARTBEAT))) return;
        if (r.mc[S] > r.pc[S]) {
            r.pc[S] = r.mc[S];
            r.dirty |= 1u << (16 + S);
        }
    }

    template <int S> RG_HD void slot() {
        const u32 f = (u32)(r.mf >> (8 * S)) & 0xffu;
        if (f == 0 || !((present >> S) & 1u)) return; // no event / no Progress (raft.rs:1663-1673)
        const u32 pb0 = (u32)(r.pf >> (8 * S)) & 0xffu;
        u32 pb = pb0;
        const IX o = (IX)S * (IX)st.stride + g;

        if ((u32)S == self) {
            if (f & RG_MF_APPEND) { // Raft::append_entry (raft.rs:976-991): last_index grows, same term
                const u64 nl = mc_self;
                if (nl > r.hi) {
                    r.hi = nl;
                    r.dirty |= RG_DIRTY_HI;
                    out |= RG_OUT_APPENDED; // MsgPropose: append_entry, then bcast_append (raft.rs:2049-2053)
                }
            }
            if (f & RG_MF_VALID) { // on_persist_entries (raft.rs:994-1016)
                const u64 idx = r.mi[S];
                if ((idx >> 63) || idx > r.hi) out |= RG_OUT_FAULT;
                maybe_update<S>(idx, pb); // && self.maybe_commit() -> commit_phase()
            }
        } else {
            const u32 state = pb & RG_PF_STATE_MASK;
            if (f & RG_MF_SENT) { // Progress::update_state(last) (progress.rs:231-243) via raft.rs:726-729
                if (state == RG_STATE_REPLICATE) set_next<S>(last0 + 1); // optimistic_update
                else if (state == RG_STATE_PROBE) pb |= RG_PF_PAUSED;
                else out |= RG_OUT_FAULT; // the reference panics
            }
            if (f & RG_MF_HEARTBEAT) { // Raft::handle_heartbeat_response (raft.rs:1777-1803)
                // (update_committed(m.commit): peer_committed<S>())
                pb = (pb | RG_PF_RECENT_ACTIVE) & ~RG_PF_PAUSED; // recent_active = true; resume()
                if (state == RG_STATE_REPLICATE && ((f & RG_MF_INS_FULL) || (pb0 & RG_PF_INS_FULL)))
                    out |= 1u << (24 + S); // ins.free_first_one()
                if (r.mt[S] < r.hi || (pb & RG_PF_PEND_RS) /* pending_request_snapshot != 0 */) out |= 1u << (8 + S); // send_append(m.from)
            } else if (f & RG_MF_VALID) {
                const u64 idx = r.mi[S];
                const bool reject = (f & RG_MF_REJECT) != 0;
                if ((idx >> 63) || (!reject && idx > r.hi)) out |= RG_OUT_FAULT;
                pb |= RG_PF_RECENT_ACTIVE;  // raft.rs:1674
                // (update_committed, raft.rs:1677: peer_committed<S>())
                if (reject) {
                    // Progress::maybe_decr_to(m.index, hint, m.request_snapshot) (progress.rs:168-206)
                    const u64 rs = (f & RG_MF_HAS_RS) ? rg_at(ms.mrs, o) : 0ULL;
                    bool dec = false;
                    if (state == RG_STATE_REPLICATE) {
                        const bool stale = idx < r.mt[S] || (idx == r.mt[S] && rs == 0);
                        if (!stale) {
                            if (rs == 0) {
                                set_next<S>(r.mt[S] + 1);
                            } else {
                                rg_at(st.prs, o) = rs;
                                pb |= RG_PF_PEND_RS;
                            }
                            dec = true;
                        }
                    } else {
                        const bool stale = (r.nx[S] == 0 || r.nx[S] - 1 != idx) && rs == 0;
                        // find_conflict_by_term needed log terms the bounded term-run table no longer holds (the
                        // pre-pass, rg_resolve_hints, marks exactly the rejects that get here): nothing that depends on
                        // the hint is applied -- recent_active and update_committed already are, like for any message
                        // of the peer --, the host resolves the hint against its own log and steps the reject again
                        const bool host_hint = !stale && rs == 0 && (f & RG_MF_HAS_LOGTERM) && ms.mhr != ms.mh /* the pre-pass ran */ &&
                                               ((rg_at(st.hhint, g) >> S) & 1u);
                        if (host_hint) out |= RG_OUT_HOST_HINT;
                        if (!stale && !host_hint) {
                            if (rs == 0) {
                                // rejects that carry log_term > 0 read their hint AFTER find_conflict_by_term
                                // (raft.rs:1562,1657-1660): resolved by rg_resolve_hints in a pre-pass so that
                                // the walk over the term table costs this kernel no registers
                                u64 hint;
                                if (PREF) {
                                    hint = rg_hint_of<P, IX>(r, st, ms, g, (u32)S, (f & RG_MF_HAS_LOGTERM) != 0);
                                    RG_OPAQUE64(hint);
                                } else {
                                    hint = (f & RG_MF_HAS_LOGTERM) ? rg_at(ms.mhr, o) : rg_at(ms.mh, o);
                                }
                                const u64 h = hint + 1;
                                u64 n = idx < h ? idx : h;
                                if (n < 1) n = 1;
                                set_next<S>(n);
                            } else if (!(pb & RG_PF_PEND_RS)) { // pending_request_snapshot == 0
                                rg_at(st.prs, o) = rs;
                                pb |= RG_PF_PEND_RS;
                            }
                            pb &= ~RG_PF_PAUSED; // resume()
                            dec = true;
                        }
                    }
                    if (dec) {
                        if (state == RG_STATE_REPLICATE) { // become_probe (progress.rs:95-107)
                            reset_state(pb, RG_STATE_PROBE, o);
                            set_next<S>(r.mt[S] + 1);
                        }
                        out |= 1u << (8 + S); // send_append(m.from), raft.rs:1719
                    }
                } else {
                    // Progress::is_paused (progress.rs:210-216); Inflights::full() comes from the host with the
                    // message or, when the ring lives on the device, from the send stage through the flag byte
                    const bool old_paused = state == RG_STATE_PROBE ? (pb & RG_PF_PAUSED) != 0
                                            : state == RG_STATE_REPLICATE
                                                ? ((f & RG_MF_INS_FULL) | (pb0 & RG_PF_INS_FULL)) != 0
                                                : true;
                    if (maybe_update<S>(idx, pb)) {
                        if (state == RG_STATE_PROBE) { // become_replicate (progress.rs:111-114)
                            reset_state(pb, RG_STATE_REPLICATE, o);
                            set_next<S>(r.mt[S] + 1);
                        } else if (state == RG_STATE_SNAPSHOT) { // maybe_snapshot_abort (progress.rs:132-134)
                            const u64 ps = rg_at(st.psnap, o);
                            if (r.mt[S] >= ps) { // become_probe from Snapshot (progress.rs:99-102)
                                reset_state(pb, RG_STATE_PROBE, o);
                                const u64 a = r.mt[S] + 1, b = ps + 1;
                                set_next<S>(a > b ? a : b);
                            }
                        } else {
                            out |= 1u << (24 + S); // ins.free_to(m.index), raft.rs:1742
                        }
                        // `if self.maybe_commit() {bcast} else if old_paused {send_append}` (raft.rs:1745-1751)
                        // is resolved in commit_phase(), which knows whether THIS ack moved the commit index
                        if (old_paused) acc_oldp |= 1u << S;
                        out |= 1u << (16 + S);                      // raft.rs:1761
                        if (RG_CFG_TRANSFEREE(r.cfg) == (u32)S + 1u && r.mt[S] == r.hi) // raft.rs:1764-1774
                            out |= RG_OUT_TIMEOUT_NOW;
                    }
                }
            }
        }
        if (pb != pb0) {
            r.pf = (r.pf & ~(0xffULL << (8 * S))) | ((u64)pb << (8 * S));
            r.dirty |= RG_DIRTY_PF;
        }
    }

    template <int S> RG_HD void self_committed() { // prs[self].update_committed(committed), raft.rs:896-900
        if constexpr (ES::on || ES::late_pc) { // (pc[] is in memory already / not loaded yet: the own cell travels in pc_self)
            if ((u32)S == self && ((present >> S) & 1u) && r.pc_self < r.commit) {
                r.pc_self = r.commit;
                r.dirty |= 1u << (16 + S);
            }
        } else if ((u32)S == self && ((present >> S) & 1u) && r.pc[S] < r.commit) {
            r.pc[S] = r.commit;
            r.dirty |= 1u << (16 + S);
        }
    }

    // One step of the sequential replay: slot S's accepted ack lands, Raft::maybe_commit runs. The quorum index is carried
    // from step to step (RgRunningQuorum); groups with group commit on take the literal evaluation (GC kernels only).
    // The replay walks the matches IN PLACE: before it r.mt / r.mi of every accepted slot are swapped (r.mt = the matches as
    // they were before the tick, r.mi = the acked values), each step swaps its slot back, so at the end r.mt holds the new
    // matches and r.mi the parked old ones again -- no second copy of the matches in registers (P u64 less in the region
    // that carries the running quorum state as well).
    template <int S> RG_HD void replay_slot(RgQuorum<P> &qm, RgRunningQuorum<P> &run, u64 &commit) {
        if (!((acc >> S) & 1u)) return;
        {
            const u64 old = r.mt[S];
            r.mt[S] = r.mi[S];
            r.mi[S] = old;
        }
        run.template raise<S>(r.mt, r.mi[S]); // (the matched index this ack replaced)
        const u64 mci = run.mci();
        // last_index as it was when this message was processed: the leader's APPEND lands at its own slot
        const u64 hi_then = (u32)S < self ? last0 : r.hi;
        if (rg_log_maybe_commit(mci, commit, r.lo, hi_then)) out |= RG_OUT_CHANGED;
        else if ((acc_oldp >> S) & 1u) out |= 1u << (8 + S); // raft.rs:1749-1751
    }

    // Raft::maybe_commit (src/raft.rs:893-904) after every accepted ack of the tick, in slot order.
    //
    // Matches only grow within a tick, so the quorum index mci_k is non-decreasing over the message
    // sequence. Let k* be the last accepted ack and hi* the last_index at that moment. If
    // mci_final <= hi*, the final commit index is decided by the evaluation at k* alone (an earlier
    // evaluation can only have committed some mci_k <= mci_final, and if the last one fails the gate
    // because mci_final < term_lo or <= commit, every earlier one failed too), and "some maybe_commit
    // returned true" == "the commit index moved". The per-message results are otherwise observable only
    // through `else if old_paused { send_append }` (raft.rs:1749-1751). So: ONE evaluation on the final
    // matches unless an accepted ack came from a paused peer or mci_final > hi* (possible only with
    // acks beyond last_index), else an exact replay of the sequence (from the old matches
    // maybe_update parked in r.mi). Both paths are bit-identical to the message-at-a-time
    // reference for ANY state and input.
    // The same decisions for a group with ProgressTracker.group_commit on (GC kernels only; tracker.rs:294-298 ->
    // majority.rs:99-123): the evaluation on the final matches, then -- under the same conditions as below -- the exact replay.
    // Every evaluation is the LITERAL group-commit algorithm (rg_mci_group) on the matches of that moment; there is no running
    // form of it. All of them go through ONE inlined call site: step -1 is the evaluation on the final matches, steps 0 .. P-1
    // the replay, in a loop that is not unrolled -- the slot a step swaps back is picked by compile-time-indexed predicated
    // swaps, so nothing is ever indexed dynamically and nothing leaves the registers (no scratch).
    template <int... S> RG_HD void commit_phase_gc(rg_seq<S...>, u64 commit0) {
        u64 gidv[P];
#pragma unroll
        for (int i = 0; i < P; i++)
            gidv[i] = ((present >> i) & 1u) ? rg_at(st.gid, (IX)i * (IX)st.stride + g) : 0ULL;
        const u32 inc = RG_CFG_INCOMING(r.cfg), outg = RG_CFG_OUTGOING(r.cfg);
        const u32 last_acc = 31u - (u32)__builtin_clz(acc);
        const u64 hi_last = last_acc < self ? last0 : r.hi;
        bool replay = false;
        u64 commit = commit0;
#pragma unroll 1
        for (int step = -1; step < P; step++) {
            if (step >= 0) {
                if (!((acc >> step) & 1u)) continue;
                (((S == step) ? (void)rg_swap64(r.mt[S], r.mi[S]) : (void)0), ...); // this ack lands: its new match back in
            }
            bool used;
            const u64 mci = rg_mci_group<P>(r.mt, gidv, inc, outg, used);
            if (step < 0) {
                if (mci <= hi_last) { // (as in commit_phase: the evaluation at the last accepted ack decides)
                    u64 c = r.commit;
                    const bool changed = rg_log_maybe_commit(mci, c, r.lo, hi_last);
                    replay = changed && acc_oldp != 0;
                    if (!replay) {
                        r.commit = c;
                        if (changed) out |= RG_OUT_CHANGED;
                        else out |= acc_oldp << 8;
                        break;
                    }
                } else {
                    replay = true;
                }
                ((((acc >> S) & 1u) ? (void)rg_swap64(r.mt[S], r.mi[S]) : (void)0), ...); // the old matches in, the acked ones parked
            } else {
                const u64 hi_then = (u32)step < self ? last0 : r.hi;
                if (rg_log_maybe_commit(mci, commit, r.lo, hi_then)) out |= RG_OUT_CHANGED;
                else if ((acc_oldp >> step) & 1u) out |= 1u << (8 + step); // raft.rs:1749-1751
            }
        }
        if (replay) r.commit = commit;
    }

    template <int... S> RG_HD void commit_phase(rg_seq<S...> seq) {
        if (acc == 0) return;
        const u64 commit0 = r.commit;
        if (GC && (r.cfg & RG_CFG_GROUP_COMMIT)) {
            commit_phase_gc(seq, commit0);
            commit_tail(seq, commit0);
            return;
        }
        RgQuorum<P> qm;
        bool replay;
        {
            const u32 last_acc = 31u - (u32)__builtin_clz(acc);
            const u64 hi_last = last_acc < self ? last0 : r.hi;
            qm.init(r.mt);
            const u64 mci = mci_of(qm, r.mt);
            if (mci <= hi_last) {
                // the evaluation at the last accepted ack decides the final commit index (above). If it does not
                // commit, NO earlier one did either (mci_k <= mci_final fails the same `> commit` / `>= term_lo`
                // test), so every ack from a paused peer takes the `else if old_paused` branch: no replay. That is
                // the steady state of a group between an election and its first commit in the new term.
                u64 c = r.commit;
                const bool changed = rg_log_maybe_commit(mci, c, r.lo, hi_last);
                replay = changed && acc_oldp != 0; // which of the acks moved it? only the sequence tells
                if (!replay) {
                    r.commit = c;
                    if (changed) out |= RG_OUT_CHANGED;
                    else out |= acc_oldp << 8; // send_append(from) for each of them (raft.rs:1749-1751)
                }
            } else {
                replay = true;
            }
        }
        if (replay) {
            // (maybe_update parked the old matches in r.mi: swap them in, the replay swaps them back slot by slot)
            ((((acc >> S) & 1u) ? (void)rg_swap64(r.mt[S], r.mi[S]) : (void)0), ...);
            qm.init(r.mt);
            RgRunningQuorum<P> run;
            run.init(qm, r.mt, RG_CFG_INCOMING(r.cfg), RG_CFG_OUTGOING(r.cfg));
            u64 commit = commit0;
            (replay_slot<S>(qm, run, commit), ...);
            r.commit = commit;
        }
        commit_tail(seq, commit0);
    }
    template <int... S> RG_HD void commit_tail(rg_seq<S...>, u64 commit0) {
        if (r.commit != commit0) {
            r.dirty |= RG_DIRTY_COMMIT;
            (self_committed<S>(), ...);
            r.adv += (u32)rg_min(r.commit - commit0, 0x10000ULL); // (rg_pub_store: what the other ranks learn)
        }
    }

    // ES::late_pc: update_committed behind the commit phase, on the columns that have just arrived
    template <int S> RG_HD void peer_committed_late() {
        if ((u32)S == self) { // (the own cell: what the election / the commit phase made of it)
            if ((present >> S) & 1u) r.pc[S] = r.pc_self;
            return;
        }
        const u32 f = (u32)(r.mf >> (8 * S)) & 0xffu;
        if (!((present >> S) & 1u) || !(f & (RG_MF_VALID | RG_MF_HEARTBEAT))) return;
        if (r.mc[S] > r.pc[S]) {
            r.pc[S] = r.mc[S];
            r.dirty |= 1u << (16 + S);
        }
    }

    template <int... S> RG_HD void run(rg_seq<S...> seq) {
        if constexpr (ES::late_pc) {
            mc_self = r.mc_self_v;
            (slot<S>(), ...);
            // (the group index the late loads use is made to DEPEND on what the slots produced: left alone, the scheduler hoists
            // the 2 P loads to the top of the kernel -- where they would be in flight with everything else again: 174 VGPRs)
#ifndef RG_LATE_PC_AFTER_COMMIT /* 1 = the two columns are requested behind the commit phase (lowest register pressure, one exposed
                                   round trip: 112 VGPRs at 7 slots); 0 = behind the slots, in flight across the quorum evaluation
                                   (138 VGPRs: no fourth wave, measured no gain) */
#define RG_LATE_PC_AFTER_COMMIT 1
#endif
            IX gl = g;
#if defined(__HIP_DEVICE_COMPILE__)
            if (!RG_LATE_PC_AFTER_COMMIT) asm volatile("" : "+v"(gl) : "v"(acc), "v"(out), "v"(r.dirty));
#endif
            if (!RG_LATE_PC_AFTER_COMMIT) ES::template load_pc_mc<P, IX>(r, st, ms, gl); // (2 P loads in flight across the quorum evaluation)
            commit_phase(seq);
#if defined(__HIP_DEVICE_COMPILE__)
            if (RG_LATE_PC_AFTER_COMMIT) asm volatile("" : "+v"(gl) : "v"(out), "v"(r.dirty), "v"((u32)r.commit));
#endif
            if (RG_LATE_PC_AFTER_COMMIT) ES::template load_pc_mc<P, IX>(r, st, ms, gl);
            (peer_committed_late<S>(), ...);
            r.out = out;
            return;
        }
        mc_self = 0;
        (peer_committed<S>(), ...);
        if constexpr (ES::on) {
            u64 ps = 0;
            (((u32)S == self ? (void)(ps = r.pc[S]) : (void)0), ...);
            r.pc_self = ps;
            ES::template store_pc<P, IX>(r, st, g, self);
        }
        ((slot<S>(), ES::template store_next<S, P, IX>(r, st, g)), ...);
        commit_phase(seq);
        r.out = out;
    }
};

// NXM: RG_NX_* -- who provides r.nx (and, for RG_NX_PREFETCH, r.hint / r.el_*).
template <int P, bool GC, int NXM, bool FUSED = false, typename IX = u64, typename ES = RgNoEarlyStores>
RG_HD void rg_group_tick(RgGroup<P> &r, const RgState &st, const RgMsgs &ms, IX g) {
    RgTick<P, GC, NXM, FUSED, IX, ES> t(r, st, ms, g);
    t.run(typename rg_make_seq<P>::type{});
}
