// rg_workload.h -- the synthetic AppendResponse stream of BASELINE.md section 4 / SURVEY.md 8(d).
//
// One definition compiled for host and device, so the GPU bench, the CPU baseline and the parity
// tests consume bit-identical streams. Counter-based: every value is a pure function of
// (seed, tick, global group, slot) and of the CURRENT engine state of that group, which is what a
// real cluster's followers would be answering to.
//
// What it models (reference behaviour cited per case):
//  * steady state: followers in Replicate acknowledge the entries the leader sent after the last
//    tick (accept acks, some stale acks, some silence); the leader appends 0..7 entries per tick and
//    persists them (Raft::append_entry + on_persist_entries, src/raft.rs:976-1016);
//  * the send path's writes to Progress (Progress::update_state, src/tracker/progress.rs:231-243)
//    as the RG_MF_SENT event: Replicate => optimistic next = last+1, Probe => paused;
//  * RG_WL_MIXED (BASELINE config 5, "leader-term rollover"): 10% of the groups START right after an election
//    (Raft::reset + become_leader, src/raft.rs:942-971,1151-1202) and, on EVERY tick, a hash-selected 1/32 of all
//    groups elects a new leader (the RG_MF_BECOME_LEADER event; the old term's in-flight responses are dropped by
//    the host's term gate, so that group carries no follower messages in its election tick): followers
//    Probe / match 0 / next = last + 1; the first probe is rejected with the follower's real last index as hint
//    (maybe_decr_to, progress.rs:186-205), the second accepted (Probe -> Replicate), then the followers catch up
//    until a quorum holds the new term's first entry and the commit index moves again. In steady state ~10% of
//    the groups are in that probe / reject / catch-up phase and ~0.1 rejects per group arrive per tick. Plus rare
//    Replicate-state rejects, request_snapshot rejects and full-inflight acks so every branch of
//    handle_append_response runs.
#pragma once

#include "rg_common.h"

#define RG_WL_TERM0 5u          /* leader term of every group when the run starts */
#define RG_WL_ELECT_ONE_IN 32u  /* RG_WL_MIXED: per tick, one group in 32 elects a new leader */

struct RgWlGroup { // static (tick-independent) facts of a group
    u32 n_peers;   // slots in use (P_g)
    u32 incoming, outgoing, present;
    u64 last0;     // last_index before the run starts
    u64 lo0;
    bool post_election;
};

RG_HD RgWlGroup rg_wl_group(u64 seed, u32 workload_word, u32 n_slots, u64 gg) {
    // workload_word: low 8 bits = RG_WL_*, bits 8-11 = fixed replica-set size for RG_WL_MIXED (0 = by gg % 3):
    // a size-class shard holds only groups of one size
    const u32 workload = workload_word & 0xffu, fixed_peers = (workload_word >> 8) & 0xfu;
    RgWlGroup w;
    const u64 h = rg_hash(seed, 0, gg, 0);
    w.n_peers = n_slots;
    if (workload == RG_WL_MIXED) {
        const u32 k = (u32)(gg % 3);
        w.n_peers = fixed_peers ? fixed_peers : (k == 0 ? 3u : (k == 1 ? 5u : 7u));
        if (w.n_peers > n_slots) w.n_peers = n_slots;
    }
    const u32 all = (1u << w.n_peers) - 1u;
    w.present = all;
    w.incoming = all;
    w.outgoing = 0;
    if (workload == RG_WL_JOINT) { // joint {0,1,2} && {1,2,3}; slots >= 4 are learners
        w.incoming = 0x07u & all;
        w.outgoing = 0x0eu & all;
    }
    w.last0 = 1000 + (h & 0xFFFFF);
    w.lo0 = w.last0 - ((h >> 20) & 63);
    w.post_election = (workload == RG_WL_MIXED) && (((h >> 32) % 10) == 0);
    return w;
}

// RG_WL_SORTED (bit 12 of the workload word; rg_workload.reserved bit 4): the SAME population of groups, placed by
// replica-set size class -- the local groups of a shard of G are first all those whose global id is 0 mod 3 (3 peers under
// RG_WL_MIXED), then 1 mod 3 (5), then 2 mod 3 (7), each class in ascending id order. Group placement inside a shard is the
// host's choice; contiguous classes are what lets ONE launch skip the peer slots a class does not have
// (k_tick_classes). Returns the offset of local group g's global id from the shard's first id.
#define RG_WL_SORTED_BIT 0x1000u
// RG_WL_GC_BIT (bit 13 of the workload word; rg_workload.reserved bit 5, RG_WL_GROUP_COMMIT): the same stream with
// ProgressTracker.group_commit on in every group (Raft::enable_group_commit, src/raft.rs:513-518) and every peer assigned to
// one of three commit groups (assign_commit_groups, :531-544: ids 1..3 by hash -- availability zones; some replica sets end
// up in two, a few in one), so that every commit evaluation is the group-commit form of majority.rs:99-123.
#define RG_WL_GC_BIT 0x2000u
RG_HD u64 rg_wl_place(u32 workload_word, u64 g, u64 G) {
    if (!(workload_word & RG_WL_SORTED_BIT)) return g;
    const u64 n0 = (G + 2) / 3, n1 = (G + 1) / 3;
    if (g < n0) return 3 * g;
    if (g < n0 + n1) return 3 * (g - n0) + 1;
    return 3 * (g - n0 - n1) + 2;
}

// follower p's real log end when a leader whose log ends at `last0` is elected (what a rejected probe reports
// as hint): up to 31 entries behind
RG_HD u64 rg_wl_follower_last(u64 seed, u64 gg, u32 p, u64 last0) {
    return last0 - rg_min(last0, rg_hash(seed, 0, gg, p) & 31);
}

// RG_WL_MIXED: does group gg elect a new leader in tick `tick`?
RG_HD bool rg_wl_elects(u64 seed, u32 workload_word, u64 gg, u64 tick) {
    return (workload_word & 0xffu) == RG_WL_MIXED && (rg_hash(seed, tick + 1, gg, 15) % RG_WL_ELECT_ONE_IN) == 0;
}

// q-th largest of up to 8 values under a mask (generator-side, for the initial commit only)
RG_HD u64 rg_wl_kth(const u64 *v, u32 mask) {
    u32 n = 0;
    for (int i = 0; i < 8; i++) n += (mask >> i) & 1u;
    if (n == 0) return ~0ULL;
    const u32 q = n / 2 + 1;
    u64 best = 0;
    for (int i = 0; i < 8; i++) {
        if (!((mask >> i) & 1u)) continue;
        u32 c = 0;
        for (int j = 0; j < 8; j++)
            if (((mask >> j) & 1u) && v[j] >= v[i]) c++;
        if (c >= q && v[i] > best) best = v[i];
    }
    return best;
}

// Initial state of group g (local index) / gg (global index). Writes every column of the group.
RG_HD void rg_wl_init_group(u64 seed, u32 workload, u32 n_slots, u64 stride, u64 g, u64 gg, u64 *match,
                            u64 *next, u64 *prc, u64 *psnap, u64 *prs, u64 *gid, u8 *pflags,
                            u64 *commit, u64 *lo, u64 *hi, u32 *cfg) {
    const RgWlGroup w = rg_wl_group(seed, workload, n_slots, gg);
    u64 v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    u64 last = w.last0, tlo = w.lo0;
    if (w.post_election) { // become_leader: noop appended at last0+1 (raft.rs:1163-1194)
        last = w.last0 + 1;
        tlo = last;
    }
    for (u32 p = 0; p < n_slots; p++) {
        const u64 o = (u64)p * stride + g;
        u64 m = 0, n = 0;
        u8 f = 0;
        if (p < w.n_peers) {
            if (p == 0) { // the leader: Replicate, matched = persisted = last_index (its own empty entry included)
                m = last;
                n = m + 1;
                f = RG_STATE_REPLICATE;
            } else if (w.post_election) { // Progress::reset(last_index + 1) (progress.rs:82-92)
                m = 0;
                n = w.last0 + 1;
                f = RG_STATE_PROBE;
            } else {
                m = rg_wl_follower_last(seed, gg, p, w.last0);
                n = m + 1;
                f = RG_STATE_REPLICATE;
            }
        }
        v[p] = m;
        match[o] = m;
        next[o] = n;
        psnap[o] = 0;
        prs[o] = 0;
        gid[o] = ((workload & RG_WL_GC_BIT) && p < w.n_peers) ? 1 + rg_hash(seed, 0, gg, 8 + p) % 3 : 0;
        pflags[g * 8 + p] = f;
    }
    for (u32 p = n_slots; p < 8; p++) pflags[g * 8 + p] = 0;
    u64 c;
    if (w.post_election) {
        c = w.last0 - ((rg_hash(seed, 0, gg, 0) >> 26) & 31);
    } else {
        const u64 a = rg_wl_kth(v, w.incoming), b = rg_wl_kth(v, w.outgoing);
        c = rg_min(rg_min(a, b), last);
    }
    for (u32 p = 0; p < n_slots; p++) prc[(u64)p * stride + g] = p < w.n_peers ? c : 0;
    commit[g] = c;
    lo[g] = tlo;
    hi[g] = last;
    cfg[g] = RG_CFG_MAKE(w.incoming, w.outgoing, 0, (workload & RG_WL_GC_BIT) != 0, 0, w.present);
}

// Messages of tick `tick` for group g, generated from the group's CURRENT state.
RG_HD void rg_wl_gen_group(u64 seed, u32 workload, u32 n_slots, u64 stride, u64 g, u64 gg, u64 tick,
                           const u64 *match, const u64 *next, const u8 *pflags, const u64 *commit,
                           const u64 *lo, const u64 *hi, u64 *m_index, u64 *m_commit, u64 *m_hint, u64 *m_rs,
                           u8 *m_flags) {
    const RgWlGroup w = rg_wl_group(seed, workload, n_slots, gg);
    const u64 last = hi[g], cm = commit[g], tlo = lo[g];
    const bool mixed = (workload & 0xffu) == RG_WL_MIXED; // rejects / snapshot requests / full windows: config 5 only
    const bool elects = rg_wl_elects(seed, workload, gg, tick);
    for (u32 p = 0; p < 8; p++) {
        u8 f = 0;
        u64 idx = 0, mcm = 0, hint = 0, rs = 0;
        if (p < w.n_peers) {
            const u64 o = (u64)p * stride + g;
            const u64 r = rg_hash(seed, tick + 1, gg, p);
            if (p == 0) { // leader: append d entries and persist them
                const u64 d = r & 7;
                f = RG_MF_APPEND | RG_MF_VALID;
                idx = last + d;
                mcm = last + d;
                if (elects) { // a new leader of this group: its empty entry lands at last + 1, then the d entries
                    f |= RG_MF_BECOME_LEADER;
                    hint = 100 + tick; // the new term (strictly increasing per group)
                    idx += 1;
                    mcm += 1;
                }
            } else if (elects) {
                // responses to the OLD leader carry the old term: dropped by the term gate (raft.rs:1349-1411)
            } else {
                const u32 pb = pflags[g * 8 + p];
                const u32 state = pb & RG_PF_STATE_MASK;
                const u64 mt = match[o], nx = next[o];
                const u32 u = (u32)(r % 100);
                const u32 rare = (u32)(r >> 16) & 63;
                if (state == RG_STATE_REPLICATE) {
                    f = RG_MF_SENT; // the leader streamed entries up to `last` after the previous tick
                    if (mixed && ((r >> 24) & 255) == 0) f |= RG_MF_INS_FULL;
                    if (u < 90) { // accept
                        f |= RG_MF_VALID;
                        idx = rg_min(last, mt + ((r >> 8) & 15));
                        mcm = rg_min(cm, idx);
                    } else if (u < 95) { // stale (duplicate / reordered) ack
                        f |= RG_MF_VALID;
                        const u64 back = (r >> 8) & 3;
                        idx = mt - rg_min(mt, back);
                        mcm = rg_min(cm, idx);
                    } else if (mixed && u == 99 && rare == 0 && last > mt) { // follower lost its tail: real reject
                        f |= RG_MF_VALID | RG_MF_REJECT;
                        idx = last;
                        hint = mt;
                        mcm = rg_min(cm, mt);
                    } else if (mixed && u == 98 && rare == 1) { // follower asks for a snapshot
                        f |= RG_MF_VALID | RG_MF_REJECT | RG_MF_HAS_RS;
                        idx = mt;
                        hint = mt;
                        rs = cm ? cm : 1;
                        mcm = rg_min(cm, mt);
                    }
                } else if (state == RG_STATE_PROBE) {
                    const bool paused = (pb & RG_PF_PAUSED) != 0;
                    if (!paused) f = RG_MF_SENT; // one probe, then paused (progress.rs:238)
                    if (u < 90 && nx > 0) {
                        // the first probe after an election asks for last_index-at-election = term_lo - 1; the
                        // follower's real log ends up to 31 entries earlier
                        const u64 flast = rg_wl_follower_last(seed, gg, p, nx - 1);
                        f |= RG_MF_VALID;
                        if (mt == 0 && nx == tlo && nx - 1 > flast) { // probe beyond the follower's log: reject
                            f |= RG_MF_REJECT;
                            idx = nx - 1;
                            hint = flast;
                            mcm = rg_min(cm, flast);
                        } else { // probe accepted, follower appends what was sent
                            const u64 base = rg_max(nx - 1, mt);
                            idx = rg_min(last, base + ((r >> 8) & 15));
                            mcm = rg_min(cm, idx);
                        }
                    }
                }
                // Snapshot state: the follower is silent until the snapshot lands (host-side path)
            }
            m_index[o] = idx;
            m_commit[o] = mcm;
            m_hint[o] = hint;
            m_rs[o] = rs;
        }
        m_flags[g * 8 + p] = f;
    }
}
