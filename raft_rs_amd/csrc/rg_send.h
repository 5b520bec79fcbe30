// rg_send.h -- the send stage: Inflights in HBM and the maybe_send_append
// decision, one group per lane. Host+device code so tests/host_check can run it on the CPU.
//
// Restated from (pingcap/raft-rs v0.6.0): src/tracker/inflights.rs:42-125 (ring), src/raft.rs:773-819
// (maybe_send_append), :722-731 (prepare_send_entries), :664-712 (prepare_send_snapshot, decision only),
// :857-864 (bcast_append), :1745-1761 (who gets what after an ack), src/raft_log.rs:382-389,463-484 (entries /
// Compacted), src/tracker/progress.rs:210-216,231-243 (is_paused, update_state).
#pragma once

#include "rg_group.h"

// RG_SEND_EXP: measurement-only knobs (never set in the product build): bit0 = no work-item list,
// bit1 = no ring accesses (results are wrong for windows deeper than one message), bit2 = list positions are
// computed (scan + atomic) but the items are not stored
#ifndef RG_SEND_EXP
#define RG_SEND_EXP 0
#endif
#ifndef RG_SEND_WHOLE_LINES
#define RG_SEND_WHOLE_LINES 1
#endif
#define RG_SEND_EFFECTS_ONLY 0x80000000u  /* engine-internal stage flags (never part of the C ABI's RG_SEND_*) */
#define RG_SEND_APPEND_LIST 0x40000000u   /* the stage's work items are appended to the compact list (the counter is not reset) */
#define RG_SEND_REQUESTS_ONLY 0x20000000u
#ifndef RG_SEND_WAVE_LINES /* the dense kernels: a cell one lane of a wave needs is loaded and rewritten by all of them (rg_wave_any) */
#define RG_SEND_WAVE_LINES 1
#endif

struct RgIns {
    u32 *meta; // [P][stride]: Inflights.start (bits 0-15) | Inflights.count (bits 16-31); start == RG_INS_COMPACT: a COMPACT window
    u64 *head; // [P][stride]: ring mode: the OLDEST inflight, Inflights.buffer[start]; compact mode: the window's deltas (below)
    u64 *tail; // [P][stride]: the NEWEST inflight, Inflights.buffer[start + count - 1]
               // These peer-major columns (coalesced across a wave) are the authoritative copies of those entries; the ring's
               // own words for them may be stale. Entries are the last indices of consecutive MsgAppends, strictly increasing.
               // COMPACT windows (round 6): up to RG_INS_COMPACT_MAX = 4 entries whose distances fit 21 bits live ENTIRELY in
               // the columns -- newest first t0 = tail, t1 = t0 - d1, t2 = t1 - d2, t3 = t2 - d3, with d1 | d2 << 21 | d3 << 42
               // in the `head` cell -- and never touch the ring. Every window starts compact (the first add of an empty one)
               // and becomes a ring window when a fifth message or a distance of 2 M entries arrives; a ring window that a
               // free leaves with at most two entries becomes compact again. Through round 5 only windows of <= 2 messages
               // stayed out of the ring: the third message cost a scattered 8-byte store (a 64-byte line of HBM traffic each,
               // ~70 MB per step at 1 M x 5) and a partial free of a deeper window walked the ring with dependent loads, lane
               // by lane -- with the ring accesses knocked out the one-launch step ran 28 % faster (profiles/r06_send_windows.txt).
    u64 *ring; // [(g * P + slot) * cap + i]: Inflights.buffer of that Progress, contiguous per cell; ring windows only, and of
               // those only the MIDDLE entries (positions start + 1 .. start + count - 2)
    u32 cap;   // Inflights::cap()
    // Byte-accurate Config::max_size_per_msg (rg_log_sizes_enable; nullptr = off): per group a ring of the cumulative
    // Entry::compute_size() of its last `esz_w` log entries, esz[g * esz_w + (index & (esz_w - 1))] = bytes of all
    // entries up to and including `index`, modulo 2^32 (differences inside a window are exact: its total is < 4 GiB).
    const u32 *esz;
    u32 esz_w; // power of two
};

// util::limit_size (src/util.rs:52-76) over the entries [next, next + avail) of one group, from the cumulative sizes:
// how many of them one MsgAppend of at most `max` bytes carries. C(m) = bytes of the first m entries. The reference
// keeps entry k if the running size BEFORE it is 0 (so the first entry always, and any entry behind a prefix of
// zero-size ones) or if the running size including it stays <= max; a prefix rule, so with m* = the largest m with
// C(m*) <= max the count is m* -- plus one when C(m*) == 0. RaftLog::slice applies it to the stable part and to the
// whole again (raft_log.rs:583-608): same count. Requires avail < esz_w (caller).
RG_HD u64 rg_limit_size(const u32 *row, u32 mask, u64 next, u64 avail, u64 max) {
    if (avail <= 1 || max == ~0ULL) return avail; // `entries.len() <= 1` / NO_LIMIT
    if (max >= 0xffffffffULL) return avail;       // a window's total is below 4 GiB
    // the base, the total and the first four candidates are independent cells of one 4 * window-byte row: six loads in
    // flight together decide most messages (a limit that admits a handful of entries) in ONE round trip; only a longer
    // message goes on with the binary search, whose loads depend on each other
    const u32 mx = (u32)max;
    const u32 base = row[(u32)(next - 1) & mask];
    const u32 tot = row[(u32)(next - 1 + avail) & mask] - base;
    u32 c[5];
    c[0] = 0;
#pragma unroll
    for (u32 m = 1; m <= 4; m++) c[m] = row[(u32)(next - 1 + (m <= avail ? m : avail)) & mask] - base;
    if (tot <= mx) return avail; // everything fits
    u64 lo = 0;                  // the largest m with C(m) <= max, C = cumulative bytes from `next`
    u32 c_lo = 0;
#pragma unroll
    for (u32 m = 1; m <= 4; m++)
        if (m <= avail && c[m] <= mx) {
            lo = m;
            c_lo = c[m];
        }
    if (lo == 4 && avail > 5) { // (C(avail) > max is known: the answer is below avail)
        u64 hi = avail - 1;
        while (lo < hi) {
            const u64 mid = (lo + hi + 1) >> 1;
            const u32 cm = row[(u32)(next - 1 + mid) & mask] - base;
            if (cm <= mx) {
                lo = mid;
                c_lo = cm;
            } else {
                hi = mid - 1;
            }
        }
    }
    const u64 n = c_lo == 0 ? lo + 1 : lo; // the `size == 0` rule (at least one entry: C(0) = 0)
    return n < avail ? n : avail;
}

// Work items of a DENSE stage (every group walked): peer-major columns, one cell per (slot, group), so the stage
// stores them like every other column -- coalesced, no compaction, no atomics, no workgroup barrier. A device-side
// consumer (a message builder) reads them in place; the compact rg_send_item list is materialised on request.
struct RgSendCols {
    u64 *prev, *last; // [P][stride] Message.index of the first message / index of the last entry sent (valid where n != 0)
    u32 *n;           // [P][stride] n_msgs (bits 0-15) | kind RG_SEND_* (bits 16-29); 0 = nothing to send to this peer;
                      // bit 31 (RG_SEND_LAST_IS_TAIL): the `last` cell was NOT written -- the index of the last entry sent is the
                      // peer's newest inflight, the window's tail column (RgIns::tail), which the stage has just stored anyway;
                      // bit 30 (RG_SEND_LAST_IS_PREV): not written either -- an empty MsgAppend: last_index == prev_index
};
#define RG_SEND_NK_LAST_IS_TAIL 0x80000000u
#define RG_SEND_NK_LAST_IS_PREV 0x40000000u

template <int P> struct RgSendRegs {
    u64 prev[P], last[P];
    u32 n[P];   // messages per slot, 0 = nothing to send
    u32 snap;   // bit s: slot s needs a snapshot instead (RG_SEND_SNAPSHOT)
    u32 hostm;  // bit s: slot s is the host's to serve (RG_SEND_HOST: entry sizes outside the device's window)
    u32 tailm;  // bit s: the item's last_index IS the window's new newest inflight (a Replicate peer that was sent entries:
                // Progress::update_state -> ins.add(last)) -- the item COLUMNS then leave the `last` cell alone (RG_SEND_LAST_IS_TAIL)
    u32 count;  // items of this group
};
template <int P> RG_HD void rg_send_regs_clear(RgSendRegs<P> &it) {
#pragma unroll
    for (int s = 0; s < P; s++) {
        it.prev[s] = 0;
        it.last[s] = 0;
        it.n[s] = 0;
    }
    it.snap = 0;
    it.hostm = 0;
    it.tailm = 0;
    it.count = 0;
}
// The work item of slot s as the `n_msgs | kind << 16` word of the item columns (0 = nothing for this peer)
template <int P> RG_HD u32 rg_send_nk(const RgSendRegs<P> &it, int s) {
    if ((it.snap >> s) & 1u) return 1u | (RG_SEND_SNAPSHOT << 16);
    if ((it.hostm >> s) & 1u) return 1u | (RG_SEND_HOST << 16);
    return it.n[s] ? (it.n[s] | (RG_SEND_APPEND << 16)) : 0u;
}

#define RG_INS_COMPACT 0xffffu      /* Inflights.start of a compact window (cap <= 65535: never a ring position) */
#ifndef RG_INS_COMPACT_MAX
#define RG_INS_COMPACT_MAX 4u      /* entries a compact window holds (2 = what rounds 1-5 kept out of the ring: the A/B build) */
#endif
#define RG_INS_DBITS 21
#define RG_INS_DMASK ((1ULL << RG_INS_DBITS) - 1ULL)

// The entries of a compact window, newest first (e[0] = tail); entries beyond `count` are meaningless
RG_HD void rg_ins_compact_entries(u64 hd, u64 tail, u64 (&e)[4]) {
    e[0] = tail;
    e[1] = e[0] - (hd & RG_INS_DMASK);
    e[2] = e[1] - ((hd >> RG_INS_DBITS) & RG_INS_DMASK);
    e[3] = e[2] - ((hd >> (2 * RG_INS_DBITS)) & RG_INS_DMASK);
}
// ... with only the newest `keep` of them left: the distances behind the survivors are dropped (canonical form)
RG_HD u64 rg_ins_compact_keep(u64 hd, u32 keep) {
    return keep >= 4u ? hd : keep == 3u ? (hd & ((1ULL << (2 * RG_INS_DBITS)) - 1ULL)) : keep == 2u ? (hd & RG_INS_DMASK) : 0ULL;
}
// The OLDEST inflight of a window (Inflights.buffer[start])
RG_HD u64 rg_ins_oldest(u32 start, u32 count, u64 hd, u64 tail) {
    if (start != RG_INS_COMPACT) return hd;
    u64 e[4];
    rg_ins_compact_entries(hd, tail, e);
    return count >= 4u ? e[3] : count == 3u ? e[2] : count == 2u ? e[1] : e[0];
}
// A ring window that holds at most two entries goes back to the columns (no ring word is read for it)
RG_HD void rg_ins_ring_to_compact(u32 &start, u32 count, u64 &hd, u64 tail) {
    if (count <= 1u) {
        start = RG_INS_COMPACT;
        hd = 0;
    } else if (count == 2u && tail - hd <= RG_INS_DMASK) {
        start = RG_INS_COMPACT;
        hd = tail - hd;
    }
}

// Inflights::free_to (inflights.rs:84-110). `hd`: the window's `head` cell (ring mode: the oldest entry; compact: the deltas).
// Written as arithmetic on predicates (the stage is instruction-bound: every `if` of lane-varying code is a saveexec /
// branch / restore sequence): a compact window is decided from its registers alone; of the ring windows only a partial free
// of MORE than two messages walks the ring.
RG_HD void rg_ins_free_to(const RgIns &ins, u64 base, u32 &start, u32 &count, u64 &hd, u64 tail, u64 to) {
    if (start == RG_INS_COMPACT) {
        u64 e[4];
        rg_ins_compact_entries(hd, tail, e);
        // entries ascend from e[count - 1] to e[0]: everything from the NEWEST entry <= `to` down is freed -- what stays is
        // the entries newer than it, i.e. its index
        u32 keep = count;
        keep = (count > 3u && to >= e[3]) ? 3u : keep;
        keep = (count > 2u && to >= e[2]) ? 2u : keep;
        keep = (count > 1u && to >= e[1]) ? 1u : keep;
        keep = (count > 0u && to >= e[0]) ? 0u : keep;
        hd = rg_ins_compact_keep(hd, keep);
        count = keep;
        return;
    }
    u64 head = hd;
    const bool hit = count != 0 && to >= head; // (else: out of the left side of the window)
    const bool all = hit && to >= tail;        // everything in the window is <= tail <= to
    if (hit && !all && count > 2) {
        // head <= to < tail: the new oldest is the first entry > to -- a middle one, or the newest
        u32 i = 1, idx = start + 1;
        if (idx >= ins.cap) idx -= ins.cap;
        u64 nh = tail;
        while (i + 1 < count) { // middle entries only
            const u64 v = (RG_SEND_EXP & 2) ? ~0ULL : ins.ring[base + idx];
            if (to < v) {
                nh = v;
                break;
            }
            idx++;
            if (idx >= ins.cap) idx -= ins.cap;
            i++;
        }
        head = nh;
        count -= i;
        start = idx;
    } else {
        // nothing freed, everything freed, or the older of exactly two: no ring word is involved
        const bool one = hit && !all;
        const u32 adv = all ? count : (one ? 1u : 0u);
        start += adv;
        if (start >= ins.cap) start -= ins.cap;
        count -= adv;
        head = one ? tail : head;
    }
    hd = head;
    if (count <= 2u) rg_ins_ring_to_compact(start, count, hd, tail);
}

// Inflights::free_first_one (inflights.rs:114-117): free_to(the oldest entry); the caller has checked count != 0
RG_HD void rg_ins_free_first(const RgIns &ins, u64 base, u32 &start, u32 &count, u64 &hd, u64 tail) {
    if (start == RG_INS_COMPACT) {
        count -= 1u;
        hd = rg_ins_compact_keep(hd, count);
        return;
    }
    rg_ins_free_to(ins, base, start, count, hd, tail, hd);
}

// Inflights::add (inflights.rs:65-81). An empty window starts compact; a compact window that cannot take the entry -- it
// holds RG_INS_COMPACT_MAX already, or the entry lies 2 M indices or more beyond the newest -- moves to the ring first: its
// middle entries are written to positions 1 .. count - 2 (start = 0; the oldest and the newest stay in the columns, as for
// every ring window). Ring mode: the previous newest becomes a middle entry (only then does it need a ring word).
RG_HD void rg_ins_add(const RgIns &ins, u64 base, u32 &start, u32 &count, u64 &hd, u64 &tail, u64 v) {
    if (count == 0) {
        start = RG_INS_COMPACT;
        hd = 0;
        tail = v;
        count = 1;
        return;
    }
    if (start == RG_INS_COMPACT) {
        const u64 d = v - tail;
        if (count < RG_INS_COMPACT_MAX && d <= RG_INS_DMASK) {
            hd = ((hd << RG_INS_DBITS) & ((1ULL << (3 * RG_INS_DBITS)) - 1ULL)) | d;
            tail = v;
            count++;
            return;
        }
        u64 e[4];
        rg_ins_compact_entries(hd, tail, e);
        if (!(RG_SEND_EXP & 2)) { // (entries newest first; ring position i holds the (i + 1)-th oldest)
            if (count == 3u) ins.ring[base + 1] = e[1];
            if (count == 4u) {
                ins.ring[base + 1] = e[2];
                ins.ring[base + 2] = e[1];
            }
        }
        hd = count >= 4u ? e[3] : count == 3u ? e[2] : count == 2u ? e[1] : e[0];
        start = 0;
    }
    if (count >= 2) {
        u32 pos = start + count - 1;
        if (pos >= ins.cap) pos -= ins.cap;
        if (!(RG_SEND_EXP & 2)) ins.ring[base + pos] = tail;
    }
    tail = v;
    count++;
}

// One group of the send stage. `out` is the group's RG_OUT_* word of the tick that just ran.
// IX: index type of the column accesses (rg_common.h: rg_at); the ring is always addressed with 64 bits.
// SPEC (the dense stage): the per-peer cells of ALL P slots are requested together with the group-level words, before
// the result word says which peers are in the work set -- one memory round trip like the tick's instead of two, at the
// price of the cells that turn out not to be needed (the leader's own slot; groups with nothing to do). After a dense
// tick of a busy shard nearly every follower is in the work set (every commit advance broadcasts).
// FUSED (k_tick_send: the stage runs in the SAME launch as the tick, on the registers the tick leaves): the result
// word, cfg, the flag row, last_index, `matched` and every `next` cell the tick holds (bit s of `nxv`) come from the
// group's registers `r` instead of memory, and what the stage changes of them (`next`, the flag row) goes back there --
// the caller stores the group once, behind the stage. The stage is split in two so that such a caller can put the
// tick's own stores between the stage's loads and their first use (rg_send_request / rg_send_serve).

// Does ANY lane of the wave want it? (the dense kernels, WAVE: a column cell is loaded and rewritten by every lane of a wave
// as soon as one lane needs it, so that the wave's store covers whole 128-B lines -- a lane-masked partial line costs the
// memory side a read-merge-write: tools/microbench/send_shape.hip measures one skipped lane in 20 at -30 % throughput for the
// stage's access shape. On the host, and where lanes hold unrelated groups (k_send_appends), a lane decides for itself.)
template <bool WAVE> RG_HD bool rg_wave_any(bool x) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (WAVE) return __builtin_amdgcn_ballot_w64(x) != 0;
#endif
    return x;
}

// The operands of one group's stage, as requested from memory (or taken over from the tick)
template <int P> struct RgSendOps {
    u32 cfg, work;
    u32 wv, sv; // WAVE: bit s = some lane of the wave has slot s in its work set / sends to it (whole-line accesses)
    u64 row0, hi, first_index;
    u32 meta_v[P];
    u64 head_v[P], tail_v[P], next_v[P], prs_v[P], match_v[P];
    bool bcast, serve, elected;
};

// The window columns (meta, head, tail: 20 B per Progress, read and rewritten by every stage) go past the Infinity Cache, loads
// and stores (RG_SEND_NT_WIN = 1; 2 = head and tail only; 0 = through the cache, as up to round 3): what the cache is for in
// a send engine is the tick's state, which the NEXT launch re-reads, and 20 P B per group of window columns competed with it.
// profiles/r04_send_window.txt: faster or equal at every engine size tried -- 1 M x 5 one launch -3..-5 %, 1.25 M x 5 -12 %,
// 2 M x 5 as two launches -15 %, 1 M x 7 -14 %, 1 M x 3 -10 %, 500 k x 5 -1..-4 %.
// (The choice is a template argument: see rg_st in rg_tick_kernels.h for what a run-time one costs.)
#ifndef RG_SEND_NT_WIN
#define RG_SEND_NT_WIN 1
#endif
template <int K, typename T> RG_HD T rg_wld(const T &c) {
    if constexpr (RG_SEND_NT_WIN == 1 || (RG_SEND_NT_WIN == 2 && K == 1)) return __builtin_nontemporal_load(&c);
    else return c;
}
template <int K, typename T, typename V> RG_HD void rg_wst(T &c, V v) {
    if constexpr (RG_SEND_NT_WIN == 1 || (RG_SEND_NT_WIN == 2 && K == 1)) __builtin_nontemporal_store((T)v, &c);
    else c = (T)v;
}

// Part 0 (k_tick_send with RG_TS_SPEC): the window columns of EVERY slot and first_index requested together with the
// group's own loads, before the tick has run -- the stage then needs no second memory round trip (after a dense tick nearly
// every follower is in the work set anyway); rg_send_request<.., PRE = true> skips what is already on its way.
template <int P, typename IX> RG_HD void rg_send_prefetch(const RgState &st, const RgIns &ins, IX g, RgSendOps<P> &q) {
    q.first_index = rg_at(st.dummy_idx, g) + 1;
#pragma unroll
    for (int s = 0; s < P; s++) {
        const IX o = (IX)s * (IX)st.stride + g;
        q.meta_v[s] = rg_wld<0>(rg_at(ins.meta, o));
        q.head_v[s] = rg_wld<1>(rg_at(ins.head, o));
        q.tail_v[s] = rg_wld<1>(rg_at(ins.tail, o));
    }
}

// Part 1: decide the work set and request every cell the stage reads (no loaded value is touched here, except -- unless
// SPEC or FUSED -- the group-level words the work set is decided from, which the caller requested with `out`).
template <int P, typename IX, bool SPEC, bool FUSED, bool PRE = false, bool WAVE = false>
RG_HD void rg_send_request(const RgState &st, const RgIns &ins, IX g, u32 out, u32 flags, RgSendOps<P> &q,
                           RgGroup<P> *r, u32 nxv, bool hold = false) {
    // everything indexed by the group alone is requested at once, before anything is decided: with the result word the
    // caller loaded that is ONE memory round trip ahead of the per-peer cells (it used to be three: out, cfg, the rest)
    if (FUSED) {
        q.cfg = r->cfg;
        q.row0 = r->pf;
        q.hi = r->hi;
    } else {
        q.cfg = rg_at(st.cfg, g);
        q.row0 = rg_at(st.pflags, g);
        q.hi = rg_at(st.hi, g);                     // last_index
    }
    if (!PRE) q.first_index = rg_at(st.dummy_idx, g) + 1; // RaftLog::first_index (dummy entry = first_index - 1)
    if (SPEC) {
#pragma unroll
        for (int s = 0; s < P; s++) {
            const IX o = (IX)s * (IX)st.stride + g;
            q.meta_v[s] = rg_wld<0>(rg_at(ins.meta, o));
            q.head_v[s] = rg_wld<1>(rg_at(ins.head, o));
            q.tail_v[s] = rg_wld<1>(rg_at(ins.tail, o));
            q.next_v[s] = rg_at(st.next, o);
            q.prs_v[s] = rg_at(st.prs, o); // (the flag row is not known yet)
            q.match_v[s] = rg_at(st.match, o);
        }
    }
    const u32 present = RG_CFG_PRESENT(q.cfg), self = RG_CFG_SELF(q.cfg);
    // bcast_append: the leader appended entries (a proposal, raft.rs:2049-2053), or the commit index moved and
    // should_bcast_commit() (raft.rs:1745-1748, :2684-2686: !skip_bcast_commit || has_pending_conf())
    // RG_SEND_EFFECTS_ONLY (engine-internal: a skipped stage being settled, a group that waits for its host hint): only the
    // Inflights effects below. (RG_SEND_REQUESTS_ONLY, rg_resolve_host_hints completing such a group: rg_group_send masks the
    // effect bits out of the result word it passes.)
    // `hold` (per group, where `flags` is the launch's): this group's requests wait for rg_resolve_host_hints
    q.serve = !(flags & RG_SEND_EFFECTS_ONLY) && !hold;
    q.bcast = q.serve && (out & RG_OUT_APPENDED) != 0;
    if (q.serve && (out & RG_OUT_CHANGED))
        q.bcast = q.bcast || !(flags & RG_SEND_SKIP_BCAST_COMMIT) || ((q.row0 >> (8 * self)) & RG_PF_PENDING_CONF);
    const u32 sa_bits = RG_OUT_SEND_APPEND(out), sm_bits = RG_OUT_SEND_MORE(out), fr_bits = RG_OUT_FREE_TO(out);
    q.elected = (out & RG_OUT_BECAME_LEADER) != 0; // Raft::reset: every Progress's ins.reset() (progress.rs:82-92)
    u32 work = sa_bits | sm_bits | fr_bits;
    if (q.bcast || q.elected) work |= present;
    work &= present & ~(1u << self);
    q.work = work;
    q.wv = q.sv = 0;
    if ((work == 0 && !WAVE) || SPEC) return; // (WAVE: a lane with nothing to do still helps fill its wave's lines)
    // all column loads of the group are issued before any of the (dependent, scattered) ring accesses
#pragma unroll
    for (int s = 0; s < P; s++) {
        const bool w = (work >> s) & 1u;
        const IX o = (IX)s * (IX)st.stride + g;
        const bool sends = w && (q.bcast || (((sa_bits | sm_bits) >> s) & 1u));
        const bool wv = rg_wave_any<WAVE>(w), sv = rg_wave_any<WAVE>(sends);
        q.wv |= (wv ? 1u : 0u) << s;
        q.sv |= (sv ? 1u : 0u) << s;
        // (each destination is written once BEFORE its load is issued and not again: `x = w ? load : 0` made the
        // compiler wait for slot s's loads -- a pending write to the same registers -- before issuing slot s+1's)
        if (!PRE) {
            q.meta_v[s] = 0u;
            q.head_v[s] = q.tail_v[s] = 0ULL;
            if (wv) {
                q.meta_v[s] = rg_wld<0>(rg_at(ins.meta, o));
                q.head_v[s] = rg_wld<1>(rg_at(ins.head, o));
                q.tail_v[s] = rg_wld<1>(rg_at(ins.tail, o));
            }
        }
        if (FUSED) {
            // the cells of `next` the tick fetched or wrote are in its registers (bit s of nxv); a broadcast also reaches
            // peers that had no event in this tick -- only theirs are read here, into the register the tick left unused
            // (it holds 0, written before the group's bulk loads were issued: no write behind a pending load)
            if (sv && !((nxv >> s) & 1u)) r->nx[s] = rg_at(st.next, o);
            if (WAVE && sv) r->dirty |= 1u << (8 + s); // (the caller stores the group's `next` cells: every lane's, for this slot)
            // (a pending snapshot request -- RG_PF_PEND_RS, almost never -- is read where it is needed, rg_send_serve)
            continue;
        }
        q.next_v[s] = q.prs_v[s] = q.match_v[s] = 0ULL;
        if (sv) q.next_v[s] = rg_at(st.next, o);
        // pending_request_snapshot: zero unless the flag byte says otherwise (RG_PF_PEND_RS) -- a column the stage used
        // to read for every peer it sends to
        if (sends && ((q.row0 >> (8 * s)) & RG_PF_PEND_RS)) q.prs_v[s] = rg_at(st.prs, o);
        if (w && (((fr_bits & sm_bits) >> s) & 1u)) q.match_v[s] = rg_at(st.match, o);
    }
}

// Part 2: the Inflights effects of the tick and the send decisions, peer by peer; stores what it changes.
template <int P, typename IX, bool FUSED, bool WAVE = false>
RG_HD void rg_send_serve(const RgState &st, const RgIns &ins, IX g, u32 out, u64 max_entries, u32 flags, const RgSendOps<P> &q,
                         RgSendRegs<P> &it, RgGroup<P> *r, u32 nxv) {
    it.snap = 0;
    it.hostm = 0;
    it.tailm = 0;
    it.count = 0;
#pragma unroll
    for (int s = 0; s < P; s++) it.n[s] = 0;
    const u32 work = q.work;
    if (work == 0 && !(WAVE && q.wv)) return;
    const u32 sa_bits = RG_OUT_SEND_APPEND(out), sm_bits = RG_OUT_SEND_MORE(out), fr_bits = RG_OUT_FREE_TO(out);
    const bool bcast = q.bcast, serve = q.serve, elected = q.elected;
    const u64 hi = q.hi, first_index = q.first_index, row0 = q.row0;
    u64 row = row0;
#pragma unroll
    for (int s = 0; s < P; s++) {
        const IX o = (IX)s * (IX)st.stride + g;
        if (!((work >> s) & 1u)) {
            // WAVE: another lane of the wave works on this slot -- this one rewrites what it loaded, so that the wave's
            // stores cover whole lines
            if (WAVE && ((q.wv >> s) & 1u)) {
                rg_wst<0>(rg_at(ins.meta, o), q.meta_v[s]);
                rg_wst<1>(rg_at(ins.head, o), q.head_v[s]);
                rg_wst<1>(rg_at(ins.tail, o), q.tail_v[s]);
                if (!FUSED && ((q.sv >> s) & 1u)) rg_at(st.next, o) = q.next_v[s];
            }
            continue;
        }
        const u64 base = ((u64)g * (u64)P + (u64)s) * ins.cap;
        u32 pb = (u32)(row >> (8 * s)) & 0xffu;
        const u32 state = pb & RG_PF_STATE_MASK;
        const u32 meta0 = q.meta_v[s];
        u32 start = meta0 & 0xffffu, count = meta0 >> 16;
        const u64 head0 = q.head_v[s];
        u64 head = head0;
        const u64 tail0 = q.tail_v[s];
        u64 tail = tail0;

        // ---- what the tick did to this peer's Inflights ----
        if (state != RG_STATE_REPLICATE || elected) {
            // Progress::reset_state (progress.rs:75-80) ran when it left Replicate; nothing is added outside it.
            // After an election the peer may already be back in Replicate (its first ack): the window is new.
            start = 0;
            count = 0;
        } else if ((fr_bits >> s) & 1u) {
            const u64 matched = FUSED ? r->mt[s] : q.match_v[s];
            if ((sm_bits >> s) & 1u) rg_ins_free_to(ins, base, start, count, head, tail, matched); // accepted ack: m.index == matched
            else if (count) rg_ins_free_first(ins, base, start, count, head, tail);                // free_first_one (:114-117)
        }

        // ---- send_append(to) then `while maybe_send_append(to, false)` ----
        const bool sa = bcast || (serve && ((sa_bits >> s) & 1u));
        const bool sm = serve && ((sm_bits >> s) & 1u);
        if (sa || sm) {
            u64 next = FUSED ? r->nx[s] : q.next_v[s];
            const u64 next0 = next;
            (void)next0;
            const u64 prs = !FUSED ? q.prs_v[s] : ((row0 >> (8 * s)) & RG_PF_PEND_RS) ? rg_at(st.prs, o) : 0ULL;
            u32 n = 0;
            bool snap = false, host = false;
            // The common peer needs at most ONE message and no decision beyond "is there anything to send": paused (nothing
            // happens: send_append and every maybe_send_append return at is_paused), or no pending snapshot request, nothing
            // compacted away and everything that is left fits one message. For it the loop below collapses to
            //   n = !paused && (entries left || send_append was requested)   [send_append sends an empty message, the
            //   `while maybe_send_append(to, false)` loop does not]; update_state(last) if entries went out
            // -- the second pass of the loop finds nothing left (or a Probe peer paused) whatever the first one did. Written
            // as predicate arithmetic; every other peer takes the literal loop.
            const bool repl = state == RG_STATE_REPLICATE, probe = state == RG_STATE_PROBE;
            const bool paused0 = probe ? (pb & RG_PF_PAUSED) != 0 : repl ? count == ins.cap : true; // Progress::is_paused (progress.rs:210-216)
            const u64 avail0 = next > hi ? 0 : hi - next + 1;
            const bool limited = (flags & RG_SEND_BYTES) != 0 || (max_entries != 0 && avail0 > max_entries);
            const bool simple = paused0 || (prs == 0 && !(next <= hi && next < first_index) && !limited);
            if (simple) {
                const bool snd = !paused0 && (avail0 != 0 || sa);
                const bool took = snd && avail0 != 0;
                const u64 last = next - 1 + avail0;
                if (snd) {
                    it.prev[s] = next - 1;
                    it.last[s] = last;
                }
                n = snd ? 1u : 0u;
                const bool addw = took && repl; // Progress::update_state(last) (progress.rs:231-243): optimistic_update + ins.add(last)
                it.tailm |= addw ? 1u << s : 0u;
                // Inflights::add. The common window is compact with room to spare: one shift and one or, no memory access;
                // everything else (a fifth message, a distance beyond 21 bits, a ring window) takes rg_ins_add's general road
                const bool fast = addw && ((count == 0) | ((start == RG_INS_COMPACT) & (count < RG_INS_COMPACT_MAX) & (last - tail <= RG_INS_DMASK)));
                if (addw && !fast) {
                    rg_ins_add(ins, base, start, count, head, tail, last);
                } else {
                    const u64 shifted = ((head << RG_INS_DBITS) & ((1ULL << (3 * RG_INS_DBITS)) - 1ULL)) | (last - tail);
                    head = fast ? (count == 0 ? 0ULL : shifted) : head;
                    start = fast ? RG_INS_COMPACT : start;
                    tail = fast ? last : tail;
                    count += fast ? 1u : 0u;
                }
                next = addw ? last + 1 : next;
                pb |= (took && probe) ? RG_PF_PAUSED : 0u;
            } else {
                bool first = sa; // the first call is send_append (allow_empty) only if one was requested
                for (;;) {
                    const bool allow_empty = first;
                    // Progress::is_paused (progress.rs:210-216)
                    const bool paused = state == RG_STATE_PROBE       ? (pb & RG_PF_PAUSED) != 0
                                        : state == RG_STATE_REPLICATE ? count == ins.cap
                                                                      : true;
                    bool sent = false;
                    if (!paused) {
                        if (prs != 0) { // pending_request_snapshot: prepare_send_snapshot
                            snap = (pb & RG_PF_RECENT_ACTIVE) != 0;
                        } else {
                            const u64 avail = next > hi ? 0 : hi - next + 1;
                            const bool compacted = next <= hi && next < first_index; // entries() = Err(Compacted)
                            if (compacted) {
                                if (allow_empty) snap = (pb & RG_PF_RECENT_ACTIVE) != 0; // else: `return false`
                            } else if (avail != 0 || allow_empty) {
                                u64 take;
                                if (flags & RG_SEND_BYTES) {
                                    // Config::max_size_per_msg in bytes: util::limit_size over the group's entry sizes. A peer
                                    // so far behind that the entries it needs have left the window is the host's to serve
                                    // (it owns the log): RG_SEND_HOST, Progress untouched, like a snapshot.
                                    if (avail > 1 && max_entries != ~0ULL && avail >= ins.esz_w) {
                                        host = true;
                                        break;
                                    }
                                    take = rg_limit_size(ins.esz + (u64)g * ins.esz_w, ins.esz_w - 1u, next, avail, max_entries);
                                } else {
                                    take = (max_entries && avail > max_entries) ? max_entries : avail;
                                }
                                if (n == 0) it.prev[s] = next - 1;
                                it.last[s] = next - 1 + take;
                                n++;
                                if (take) { // Progress::update_state(last) (progress.rs:231-243)
                                    if (state == RG_STATE_REPLICATE) {
                                        next += take; // optimistic_update
                                        rg_ins_add(ins, base, start, count, head, tail, next - 1);
                                        it.tailm |= 1u << s; // (it.last[s] == tail from here on: every later message adds its own last)
                                    } else {
                                        pb |= RG_PF_PAUSED;
                                    }
                                }
                                sent = true;
                            }
                        }
                    }
                    // send_append runs once; the loop goes on while something was sent. A snapshot pauses the
                    // Progress (become_snapshot, applied by the host), which ends the loop as well.
                    if (snap) break;
                    if (first) {
                        first = false;
                        if (!sm) break;
                    } else if (!sent) {
                        break;
                    }
                }
            }
            if (snap || host) it.tailm &= ~(1u << s); // (the item carries the snapshot index / the leader's last_index instead)
            if (snap) {
                it.snap |= 1u << s;
                it.prev[s] = next - 1;
                it.last[s] = prs; // the requested snapshot index (0 = any)
            }
            if (host) { // (only ever the first send of a peer: the entries left shrink with every message)
                it.hostm |= 1u << s;
                it.prev[s] = next - 1;
                it.last[s] = hi;
            }
            it.n[s] = n;
            if (n || snap || host) it.count++;
            if (FUSED) { // the group's `next` cells are stored once, by the caller (whole lines: every cell the stage looked at)
                r->nx[s] = next;
                r->dirty |= 1u << (8 + s);
            } else if (RG_SEND_WHOLE_LINES || next != next0) {
                rg_at(st.next, o) = next;
            }
        } else if (WAVE && !FUSED && ((q.sv >> s) & 1u)) {
            rg_at(st.next, o) = q.next_v[s]; // (another lane of the wave sends to this slot: whole lines)
        }
        pb = (pb & ~RG_PF_INS_FULL) | ((state == RG_STATE_REPLICATE && count == ins.cap) ? RG_PF_INS_FULL : 0u);
        row = (row & ~(0xffULL << (8 * s))) | ((u64)pb << (8 * s));
        const u32 meta = start | (count << 16);
        // RG_SEND_WHOLE_LINES: every cell of the work set is rewritten, changed or not -- whole 128-B lines instead of
        // lane-masked partial ones, which the memory side has to read before it can merge them (the tick kernel's
        // RG_OPT bit 1, same reason)
        if (RG_SEND_WHOLE_LINES || meta != meta0) rg_wst<0>(rg_at(ins.meta, o), meta);
        // (an empty window's two cells keep whatever they held: rewritten with the value just read)
        if (RG_SEND_WHOLE_LINES || (head != head0 && count)) rg_wst<1>(rg_at(ins.head, o), count ? head : head0);
        if (RG_SEND_WHOLE_LINES || (tail != tail0 && count)) rg_wst<1>(rg_at(ins.tail, o), count ? tail : tail0);
    }
    if (row != row0) {
        if (FUSED) {
            r->pf = row;
            r->dirty |= RG_DIRTY_PF;
        } else {
            rg_at(st.pflags, g) = row;
        }
    }
}

// What a stage of its own does with a group that waits -- or waited -- for its host hint: returns `hold` (serve no request now) and
// adjusts the result word the stage sees.
RG_HD bool rg_send_hold(u32 &out, u32 flags) {
    if ((out & RG_OUT_HOST_HINT) && (flags & RG_SEND_REQUESTS_ONLY)) out = 0; // (still waiting for more answers; its effects are done)
    else if (out & RG_OUT_HOST_HINT) return true;
    else if (flags & RG_SEND_REQUESTS_ONLY) out &= ~(0xff000000u | (u32)RG_OUT_BECAME_LEADER); // (free_to / free_first_one / reset: done)
    return false;
}

// The stage of one group in its own launch (k_send_dense, k_send_appends, the host twin of the tests).
// WAVE: the lanes of a wave hold consecutive groups (k_send_dense): whole-line accesses, see rg_wave_any.
template <int P, typename IX = u64, bool SPEC = false, bool WAVE = false>
RG_HD void rg_group_send(const RgState &st, const RgIns &ins, IX g, u32 out, u64 max_entries, u32 flags,
                         RgSendRegs<P> &it) {
    // a group that waits for the host's answer to RG_OUT_HOST_HINT keeps its send REQUESTS for the stage that
    // rg_resolve_host_hints runs (the deferred reject's send_append comes before the group's other sends of the tick); the
    // tick's Inflights EFFECTS -- free_to, free_first_one, the window resets -- are applied here and now, once, whatever the host
    // does next (round 4 dropped them with the requests: a host that moved on without resolving left the windows stale)
    const bool hold = rg_send_hold(out, flags);
    RgSendOps<P> q;
    rg_send_request<P, IX, SPEC, false, false, WAVE && !SPEC>(st, ins, g, out, flags, q, nullptr, 0u, hold);
    rg_send_serve<P, IX, false, WAVE && !SPEC>(st, ins, g, out, max_entries, flags, q, it, nullptr, 0u);
}
