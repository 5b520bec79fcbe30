// host_tick.hip -- TEST INFRASTRUCTURE. Compiles the engine's per-group arithmetic header
// (raft_rs_amd/csrc/rg_group.h, the exact code the HIP kernels inline) for the HOST, so CPU-only
// tests can diff it against the oracle. This file is never part of libraftgroups.so: the product has
// no CPU path. Build: hipcc -O3 -shared -fPIC --offload-arch=gfx950 host_tick.hip -o libhost_tick.so
#include "../../raft_rs_amd/csrc/rg_tick_kernels.h"
#include "../../raft_rs_amd/csrc/rg_send.h"

// state[]: match next pr_commit pend_snap pend_rs gid pflags commit term_lo term_hi cfg out
//          run_first run_term dummy_index dummy_term cur_term host_hint run_count   (19 pointers)
// msg[]:   m_index m_commit m_hint m_rs m_flags m_logterm       (6 pointers)
static RgState make_state(void *const *p, u64 G, u64 stride) {
    RgState st;
    st.match = (u64 *)p[0]; st.next = (u64 *)p[1]; st.prc = (u64 *)p[2]; st.psnap = (u64 *)p[3];
    st.prs = (u64 *)p[4]; st.gid = (u64 *)p[5]; st.pflags = (u64 *)p[6]; st.commit = (u64 *)p[7];
    st.lo = (u64 *)p[8]; st.hi = (u64 *)p[9]; st.cfg = (u32 *)p[10]; st.out = (u32 *)p[11];
    st.run_first = (u64 *)p[12]; st.run_term = (u64 *)p[13]; st.dummy_idx = (u64 *)p[14]; st.dummy_term = (u64 *)p[15];
    st.cur_term = (u64 *)p[16];
    st.hhint = (u8 *)p[17];
    // (p[18] = RG_COL_RUN_COUNT: it must sit `stride` bytes behind p[17], as in the engine's arena -- rg_run_n)
    if ((u8 *)p[18] != (u8 *)p[17] + stride) __builtin_trap();
    st.G = G; st.stride = stride;
    st.pub = nullptr; st.pub_off_delta = 0; st.pub_cap = 0; st.ix64 = 0;
    return st;
}
// RG_PF_PEND_SNAP / RG_PF_PEND_RS are engine-owned: the library derives it when the columns are loaded (k_fix_pending); the arrays a test hands
// over may come straight from a generator, so the same derivation runs here on the way in.
static void derive_pending(const RgState &st, unsigned P) {
    for (u64 g = 0; g < st.G; g++) {
        u64 row = st.pflags[g];
        for (unsigned p = 0; p < P; p++) {
            const u64 o = (u64)p * st.stride + g;
            row &= ~((u64)RG_PF_PENDING << (8 * p));
            row |= (u64)((st.psnap[o] ? RG_PF_PEND_SNAP : 0u) | (st.prs[o] ? RG_PF_PEND_RS : 0u)) << (8 * p);
        }
        st.pflags[g] = row;
        rg_run_n(st)[g] = (u8)rg_count_runs(st, g); // RG_COL_RUN_COUNT is engine-owned too (k_fix_run_count)
    }
}

static RgMsgs make_msgs(const void *const *p) {
    RgMsgs ms;
    ms.mi = (const u64 *)p[0]; ms.mc = (const u64 *)p[1]; ms.mh = (const u64 *)p[2]; ms.mrs = (const u64 *)p[3];
    ms.mflags = (const u64 *)p[4]; ms.mlt = (const u64 *)p[5];
    ms.mhr = ms.mh;
    return ms;
}

#include <vector>
template <int P, typename IX, bool CLS = false> static void host_one(const RgState &st, const RgMsgs &ms, bool gc, IX g) {
    RgGroup<P> r;
    // (the kernels' own choice of store / load policy for this slot count: CLS = as a body of k_tick_classes -- late loads of
    // committed_index / Message.commit from 7 slots on)
    typedef typename RgLaneStores<P, false, false, CLS>::type ES;
    rg_load_group<P, RG_LANE_NX, IX, false, false, ES>(r, st, ms, g);
    if (gc) rg_group_tick<P, true, RG_LANE_NX, false, IX, ES>(r, st, ms, g);
    else rg_group_tick<P, false, RG_LANE_NX, false, IX, ES>(r, st, ms, g);
    rg_store_group<P, IX, 3, false, false, ES::on>(r, st, g);
}

template <int P> static void host_tick(const RgState &st, const RgMsgs &ms_in, bool gc, u64 g0, u64 g1) {
    // the engine's pre-pass (k_resolve_hints) followed by the tick, group by group
    std::vector<u64> rh;
    RgMsgs ms = ms_in;
    bool any = false;
    for (u64 g = g0; g < g1 && !any; g++) any = (ms_in.mflags[g] & 0x8080808080808080ULL) != 0;
    if (any) { // like the engine: the pre-pass only runs when the tick carries log terms
        rh.assign((size_t)P * st.stride, 0);
        for (u64 g = g0; g < g1; g++) rg_resolve_hints(st, ms_in, g, P, rh.data());
        ms.mhr = rh.data();
    }
    // exactly what k_tick_lane does, with the index type rg_launch_tick_t would pick (odd groups take the other one,
    // so both instantiations are diffed against the oracle)
    const bool fits32 = rg_fits_u32_offsets(P, st.stride);
    for (u64 g = g0; g < g1; g++) {
        // ... and every other pair of groups as a body of k_tick_classes would run them (no group commit there)
        const bool cls = !gc && (g & 2);
        if (fits32 && !(g & 1)) cls ? host_one<P, u32, true>(st, ms, gc, (u32)g) : host_one<P, u32>(st, ms, gc, (u32)g);
        else cls ? host_one<P, u64, true>(st, ms, gc, g) : host_one<P, u64>(st, ms, gc, g);
    }
}

// Fused replay on the host: the same sequence k_tick_fused runs per lane (state in "registers" across ticks).
template <int P> static void host_fused(const RgState &st, const RgMsgs *ms, u32 T, u32 *out_t, u64 *commit_t, bool gc) {
    for (u64 g = 0; g < st.G; g++) {
        RgGroup<P> r;
        r.pf = st.pflags[g]; r.cfg = st.cfg[g]; r.commit = st.commit[g]; r.lo = st.lo[g]; r.hi = st.hi[g];
        for (int p = 0; p < P; p++) {
            const u64 o = (u64)p * st.stride + g;
            r.mt[p] = st.match[o]; r.pc[p] = st.prc[o]; r.nx[p] = 0;
        }
        r.dirty = 0; r.evm = 0; r.adv = 0;
        for (u32 t = 0; t < T; t++) {
            r.mf = ms[t].mflags[g];
            for (int p = 0; p < P; p++) {
                const u64 o = (u64)p * st.stride + g;
                r.mi[p] = ms[t].mi[o]; r.mc[p] = ms[t].mc[o];
            }
            if (gc) rg_group_tick<P, true, RG_NX_LAZY, true>(r, st, ms[t], g);
            else rg_group_tick<P, false, RG_NX_LAZY, true>(r, st, ms[t], g);
            out_t[(u64)t * st.G + g] = r.out;
            if (commit_t) commit_t[(u64)t * st.G + g] = r.commit;
        }
        rg_store_group<P>(r, st, g);
    }
}

#define RG_DISPATCH_P(P, CALL)                                                                    \
    switch (P) {                                                                                  \
    case 1: { constexpr int N = 1; CALL; } break;                                                 \
    case 2: { constexpr int N = 2; CALL; } break;                                                 \
    case 3: { constexpr int N = 3; CALL; } break;                                                 \
    case 4: { constexpr int N = 4; CALL; } break;                                                 \
    case 5: { constexpr int N = 5; CALL; } break;                                                 \
    case 6: { constexpr int N = 6; CALL; } break;                                                 \
    case 7: { constexpr int N = 7; CALL; } break;                                                 \
    case 8: { constexpr int N = 8; CALL; } break;                                                 \
    default: return -1;                                                                           \
    }

extern "C" int rg_host_check_tick(unsigned P, unsigned long G, unsigned long stride, void *const *state,
                                  const void *const *msg, int group_commit_kernel, unsigned long g_begin,
                                  unsigned long g_end) {
    const RgState st = make_state(state, G, stride);
    derive_pending(st, P);
    const RgMsgs ms = make_msgs(msg);
    const bool gc = group_commit_kernel != 0;
    const u64 g0 = g_begin, g1 = g_end < G ? g_end : G;
    RG_DISPATCH_P(P, host_tick<N>(st, ms, gc, g0, g1));
    return 0;
}

// The same tick over contiguous group ranges on n_threads host threads (std::thread inside this library, like the
// oracle's ro_tick_soa_mt) -- bench.py's "engine arithmetic on the host cores" line. Every thread derives the pending
// bits of its own range first, then ticks it; groups are independent.
#include <thread>
extern "C" int rg_host_check_tick_mt(unsigned P, unsigned long G, unsigned long stride, void *const *state,
                                     const void *const *msg, int group_commit_kernel, unsigned n_threads) {
    if (P == 0 || P > 8 || n_threads == 0) return -1;
    const RgState st = make_state(state, G, stride);
    const RgMsgs ms = make_msgs(msg);
    const bool gc = group_commit_kernel != 0;
    // (the pre-pass of host_tick<> decides per RANGE whether a tick carries log terms; results are the same either way)
    auto work = [&](u64 g0, u64 g1) {
        RgState sub = st;
        for (u64 g = g0; g < g1; g++) {
            u64 row = st.pflags[g];
            for (unsigned p = 0; p < P; p++) {
                const u64 o = (u64)p * st.stride + g;
                row &= ~((u64)RG_PF_PENDING << (8 * p));
                row |= (u64)((st.psnap[o] ? RG_PF_PEND_SNAP : 0u) | (st.prs[o] ? RG_PF_PEND_RS : 0u)) << (8 * p);
            }
            st.pflags[g] = row;
        }
        switch (P) {
        case 1: host_tick<1>(sub, ms, gc, g0, g1); break;
        case 2: host_tick<2>(sub, ms, gc, g0, g1); break;
        case 3: host_tick<3>(sub, ms, gc, g0, g1); break;
        case 4: host_tick<4>(sub, ms, gc, g0, g1); break;
        case 5: host_tick<5>(sub, ms, gc, g0, g1); break;
        case 6: host_tick<6>(sub, ms, gc, g0, g1); break;
        case 7: host_tick<7>(sub, ms, gc, g0, g1); break;
        default: host_tick<8>(sub, ms, gc, g0, g1); break;
        }
    };
    if (n_threads == 1) {
        work(0, G);
        return 0;
    }
    std::vector<std::thread> th;
    th.reserve(n_threads);
    for (unsigned i = 0; i < n_threads; i++) th.emplace_back(work, (u64)i * G / n_threads, (u64)(i + 1) * G / n_threads);
    for (auto &t : th) t.join();
    return 0;
}

extern "C" int rg_host_check_fused(unsigned P, unsigned long G, unsigned long stride, void *const *state, unsigned T,
                                   const void *const *const *msgs, u32 *out_t, u64 *commit_t, int group_commit_kernel) {
    const RgState st = make_state(state, G, stride);
    derive_pending(st, P);
    RgMsgs ms[RG_MAX_FUSE];
    if (T == 0 || T > RG_MAX_FUSE) return -1;
    for (unsigned t = 0; t < T; t++) ms[t] = make_msgs(msgs[t]);
    const bool gc = group_commit_kernel != 0;
    RG_DISPATCH_P(P, host_fused<N>(st, ms, T, out_t, commit_t, gc));
    return 0;
}

// The send stage (rg_send.h) group by group, exactly what k_send_appends runs per lane; items are appended in
// group order. Returns the number of items (may exceed cap), -1 on a bad slot count.
template <int P>
static long host_send(const RgState &st, const RgIns &ins, u64 max_entries, u32 flags, rg_send_item *items, u64 cap) {
    u64 k = 0;
    for (u64 g = 0; g < st.G; g++) {
        const u32 out = st.out[g];
        if (!out) continue;
        RgSendRegs<P> it;
        rg_group_send<P>(st, ins, g, out, max_entries, flags, it);
        for (int s = 0; s < P; s++) {
            const u32 nk = rg_send_nk<P>(it, s);
            if (!nk) continue;
            if (k < cap) {
                rg_send_item r;
                r.group = g; r.prev_index = it.prev[s]; r.last_index = it.last[s]; r.slot = (u32)s;
                r.n_msgs = (uint16_t)(nk & 0xffffu);
                r.kind = (uint16_t)(nk >> 16);
                items[k] = r;
            }
            // the premise of RG_SEND_NK_LAST_IS_TAIL (the item columns leave `last` unwritten): where the stage says so, the
            // window's tail column -- just stored -- holds the item's last_index
            if (((it.tailm >> s) & 1u) && ins.tail[(u64)s * st.stride + g] != it.last[s]) return -2;
            k++;
        }
    }
    return (long)k;
}

extern "C" long rg_host_check_send(unsigned P, unsigned long G, unsigned long stride, void *const *state, u32 *meta,
                                   u64 *head, u64 *tail, u64 *ring, unsigned cap, unsigned long max_entries, unsigned flags,
                                   rg_send_item *items, unsigned long items_cap, const u32 *esz, unsigned esz_w) {
    const RgState st = make_state(state, G, stride);
    derive_pending(st, P);
    RgIns ins;
    ins.meta = meta; ins.head = head; ins.tail = tail; ins.ring = ring; ins.cap = cap;
    ins.esz = esz; ins.esz_w = esz_w; // entry sizes for RG_SEND_BYTES (NULL / 0 = off)
    long n = -1;
    RG_DISPATCH_P(P, n = host_send<N>(st, ins, max_entries, flags, items, items_cap));
    return n;
}

// The tick and its send stage in ONE pass over the group's registers (rg_group_tick_send, what k_tick_send runs per lane;
// rg_tick_send / rg_tick_device_send): the group is loaded once, ticked, the stage runs on what the tick left and the group
// is stored once. Items are appended in group order, like host_send.
template <int P, typename IX>
static void host_tick_send_one(const RgState &st, const RgMsgs &ms, const RgIns &ins, bool gc, IX g, u64 max_entries, u32 flags,
                               RgSendRegs<P> &it) {
    RgGroup<P> r;
    rg_load_group<P, RG_LANE_NX, IX>(r, st, ms, g);
    if (gc) rg_group_tick_send<P, true, IX>(r, st, ms, ins, g, max_entries, flags, it);
    else rg_group_tick_send<P, false, IX>(r, st, ms, ins, g, max_entries, flags, it);
}
template <int P>
static long host_tick_send(const RgState &st, const RgMsgs &ms_in, const RgIns &ins, bool gc, u64 max_entries, u32 flags,
                           rg_send_item *items, u64 cap) {
    std::vector<u64> rh;
    RgMsgs ms = ms_in;
    bool any = false;
    for (u64 g = 0; g < st.G && !any; g++) any = (ms_in.mflags[g] & 0x8080808080808080ULL) != 0;
    if (any) {
        rh.assign((size_t)P * st.stride, 0);
        for (u64 g = 0; g < st.G; g++) rg_resolve_hints(st, ms_in, g, P, rh.data());
        ms.mhr = rh.data();
    }
    const bool fits32 = rg_fits_u32_offsets(P, st.stride);
    u64 k = 0;
    for (u64 g = 0; g < st.G; g++) {
        RgSendRegs<P> it;
        if (fits32 && !(g & 1)) host_tick_send_one<P, u32>(st, ms, ins, gc, (u32)g, max_entries, flags, it);
        else host_tick_send_one<P, u64>(st, ms, ins, gc, g, max_entries, flags, it);
        for (int s = 0; s < P; s++) {
            const u32 nk = rg_send_nk<P>(it, s);
            if (!nk) continue;
            if (k < cap) {
                rg_send_item r;
                r.group = g; r.prev_index = it.prev[s]; r.last_index = it.last[s]; r.slot = (u32)s;
                r.n_msgs = (uint16_t)(nk & 0xffffu);
                r.kind = (uint16_t)(nk >> 16);
                items[k] = r;
            }
            if (((it.tailm >> s) & 1u) && ins.tail[(u64)s * st.stride + g] != it.last[s]) return -2; // (as in host_send)
            k++;
        }
    }
    return (long)k;
}

extern "C" long rg_host_check_tick_send(unsigned P, unsigned long G, unsigned long stride, void *const *state,
                                        const void *const *msg, int group_commit_kernel, u32 *meta, u64 *head, u64 *tail,
                                        u64 *ring, unsigned cap, unsigned long max_entries, unsigned flags, rg_send_item *items,
                                        unsigned long items_cap, const u32 *esz, unsigned esz_w) {
    const RgState st = make_state(state, G, stride);
    derive_pending(st, P);
    const RgMsgs ms = make_msgs(msg);
    RgIns ins;
    ins.meta = meta; ins.head = head; ins.tail = tail; ins.ring = ring; ins.cap = cap;
    ins.esz = esz; ins.esz_w = esz_w;
    long n = -1;
    RG_DISPATCH_P(P, n = host_tick_send<N>(st, ms, ins, group_commit_kernel != 0, max_entries, flags, items, items_cap));
    return n;
}

// rg_limit_size (rg_send.h) on a caller-provided window of cumulative entry sizes: how many of the entries
// [next, next + avail) one MsgAppend of at most `max` bytes carries.
extern "C" unsigned long rg_host_check_limit_size(const u32 *row, unsigned window, unsigned long next, unsigned long avail,
                                                  unsigned long max) {
    return rg_limit_size(row, window - 1u, next, avail, max);
}

// RgQuorum (the P x P ">=" bit matrix rank select) and the literal group-commit routine on caller-provided
// matches / group ids: mci = ProgressTracker::maximal_committed_index over (incoming, outgoing) slot masks.
// `raise_slot` >= 0: build the matrix on the OLD value of that slot, then apply the incremental update.
template <int P>
static u64 host_mci(const u64 *match, const u64 *gid, u32 incoming, u32 outgoing, int use_group_commit, int raise_slot,
                    u64 old_value, int *used) {
    u64 v[P], g[P];
    for (int i = 0; i < P; i++) { v[i] = match[i]; g[i] = gid[i]; }
    *used = 0;
    if (use_group_commit) {
        bool u = false;
        const u64 r = rg_mci_group<P>(v, g, incoming, outgoing, u);
        *used = u;
        return r;
    }
    RgQuorum<P> q;
    if (raise_slot >= 0 && raise_slot < P) {
        u64 w[P];
        for (int i = 0; i < P; i++) w[i] = v[i];
        w[raise_slot] = old_value;
        q.init(w);
        switch (raise_slot) { // update<S> is a compile-time slot
        case 0: if (P > 0) q.template update<0>(v); break;
        case 1: if (P > 1) q.template update<(P > 1 ? 1 : 0)>(v); break;
        case 2: if (P > 2) q.template update<(P > 2 ? 2 : 0)>(v); break;
        case 3: if (P > 3) q.template update<(P > 3 ? 3 : 0)>(v); break;
        case 4: if (P > 4) q.template update<(P > 4 ? 4 : 0)>(v); break;
        case 5: if (P > 5) q.template update<(P > 5 ? 5 : 0)>(v); break;
        case 6: if (P > 6) q.template update<(P > 6 ? 6 : 0)>(v); break;
        default: if (P > 7) q.template update<(P > 7 ? 7 : 0)>(v); break;
        }
    } else {
        q.init(v);
    }
    return q.mci(v, incoming, outgoing);
}

extern "C" int rg_host_check_mci(unsigned P, const u64 *match, const u64 *gid, unsigned incoming, unsigned outgoing,
                                 int use_group_commit, int raise_slot, unsigned long old_value, u64 *mci, int *used) {
    RG_DISPATCH_P(P, *mci = host_mci<N>(match, gid, incoming, outgoing, use_group_commit, raise_slot, old_value, used));
    return 0;
}


// rg_progress_events (RawNode::report_unreachable / report_snapshot in place) for the host: the record loop the kernel
// runs one lane per record. ins_meta may be NULL (Inflights with the host).
extern "C" int rg_host_check_progress_events(unsigned P, unsigned long G, unsigned long stride, void *const *state, u32 *ins_meta,
                                             const rg_progress_event *ev, unsigned long n) {
    if (P < 1 || P > 8) return -1;
    const RgState st = make_state(state, G, stride);
    derive_pending(st, P);
    for (u64 i = 0; i < n; i++) rg_progress_events_at(st, ins_meta, ev, n, P, i);
    return 0;
}


// rg_resolve_host_hints for the host: the record loop k_resolve_apply runs one lane per record. applied[i] = maybe_decr_to's result.
extern "C" int rg_host_check_resolve_hints(unsigned P, unsigned long G, unsigned long stride, void *const *state, u32 *ins_meta,
                                           const rg_resolved_hint *it, unsigned long n, u8 *applied) {
    if (P < 1 || P > 8) return -1;
    const RgState st = make_state(state, G, stride);
    derive_pending(st, P);
    for (u64 i = 0; i < n; i++)
        applied[i] = rg_resolve_hint_at(st, ins_meta, it, P, i,
                                        [&](u64 g, u32 s) { const u32 b = st.hhint[g]; st.hhint[g] = (u8)(b & ~(1u << s)); return b; },
                                        [&](u64 g, u32 bits, u32 clear) { st.out[g] = (st.out[g] | bits) & ~clear; }) & RG_RESOLVE_APPLIED;
    return 0;
}

// One Inflights window driven op by op (tests/test_host_check.py::test_window_arithmetic_equals_a_plain_queue): op 0 = add(v),
// 1 = free_to(v), 2 = free_first_one. After every op the window's logical contents (oldest first) are written to
// contents[k][0..count) and counts[k]; modes[k] = 1 while the window is compact. The ring is the caller's (cap words).
extern "C" int rg_host_check_window_ops(unsigned cap, unsigned long n_ops, const unsigned *ops, const u64 *vals, u64 *ring,
                                        u64 *contents, unsigned *counts, unsigned *modes) {
    RgIns ins;
    ins.meta = nullptr; ins.head = nullptr; ins.tail = nullptr; ins.ring = ring; ins.cap = cap;
    ins.esz = nullptr; ins.esz_w = 0;
    u32 start = 0, count = 0;
    u64 hd = 0, tail = 0;
    for (unsigned long k = 0; k < n_ops; k++) {
        if (ops[k] == 0) {
            if (count == cap) return -1; // (the caller never adds to a full window: Inflights::add panics)
            rg_ins_add(ins, 0, start, count, hd, tail, vals[k]);
        } else if (ops[k] == 1) {
            rg_ins_free_to(ins, 0, start, count, hd, tail, vals[k]);
        } else if (count) {
            rg_ins_free_first(ins, 0, start, count, hd, tail);
        }
        counts[k] = count;
        modes[k] = start == RG_INS_COMPACT ? 1u : 0u;
        u64 *row = contents + k * cap;
        if (start == RG_INS_COMPACT) {
            if (count > RG_INS_COMPACT_MAX) return -2;
            u64 e[4];
            rg_ins_compact_entries(hd, tail, e);
            for (u32 i = 0; i < count; i++) row[i] = e[count - 1 - i];
            if (count && rg_ins_oldest(start, count, hd, tail) != row[0]) return -3;
        } else {
            for (u32 i = 0; i < count; i++) row[i] = ring[(start + i) % cap];
            if (count) {
                row[0] = hd;
                row[count - 1] = tail;
            }
        }
    }
    return 0;
}
