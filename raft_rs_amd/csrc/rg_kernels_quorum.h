// rg_kernels_quorum.h -- kernels of abi_tick.hip: maybe_commit without messages, votes, liveness, heartbeat commits, the find_conflict_by_term pre-pass, host-hint answers, result / message reductions, size classes
// Included by exactly one abi_*.hip unit (the kernels are not templates: one definition per library).
#pragma once
#include "rg_engine.h"

// ------------------------------------------------------------------------------------------------
// kernels: Raft::maybe_commit for all groups without messages, and maximal_committed_index
// ------------------------------------------------------------------------------------------------
// GC = false compiles the group-commit routine (and the scratch its out-of-line call needs) out: launched unless some
// group has ProgressTracker.group_commit set, like the tick kernels.
template <int P, bool COMMIT, bool GC>
__global__ __launch_bounds__(RG_BLOCK) void k_recompute(RgState st, u64 *mci_out, u8 *gc_out) {
    const u64 g = (u64)blockIdx.x * RG_BLOCK + threadIdx.x;
    if (g >= st.G) return;
    const u32 cfg = st.cfg[g];
    const u32 present = RG_CFG_PRESENT(cfg), incoming = RG_CFG_INCOMING(cfg), outgoing = RG_CFG_OUTGOING(cfg);
    u64 mt[P];
#pragma unroll
    for (int p = 0; p < P; p++) mt[p] = st.match[(u64)p * st.stride + g];
#pragma unroll
    for (int p = 0; p < P; p++)
        if (!((present >> p) & 1u)) mt[p] = 0; // a voter without a Progress acks 0 (majority.rs:80-82)
    u64 mci;
    bool used = false;
    if (GC && (cfg & RG_CFG_GROUP_COMMIT)) {
        u64 gidv[P];
#pragma unroll
        for (int p = 0; p < P; p++) gidv[p] = ((present >> p) & 1u) ? st.gid[(u64)p * st.stride + g] : 0ULL;
        mci = rg_mci_group<P>(mt, gidv, incoming, outgoing, used);
    } else {
        RgQuorum<P> qm;
        qm.init(mt);
        mci = qm.mci(mt, incoming, outgoing);
        // joint.rs:47-51 flag: only an empty majority reports true without group commit (majority.rs:71-75,:99-101)
        used = incoming == 0 && outgoing == 0;
    }
    if (COMMIT) {
        u64 commit = st.commit[g];
        const u64 lo = st.lo[g], hi = st.hi[g], commit0 = commit;
        u32 out = 0;
        if (rg_log_maybe_commit(mci, commit, lo, hi)) { // src/raft.rs:893-904
            if (st.pub) rg_pub_store(st, g, rg_pub_load(st, g) + (u32)rg_min(commit - commit0, 0x10000ULL), commit);
            st.commit[g] = commit;
            const u32 self = RG_CFG_SELF(cfg);
            if ((present >> self) & 1u) {
                const u64 o = (u64)self * st.stride + g;
                if (st.prc[o] < commit) st.prc[o] = commit;
            }
            out = RG_OUT_CHANGED;
        }
        st.out[g] = out;
    } else {
        mci_out[g] = mci;
        if (gc_out) gc_out[g] = used ? 1 : 0;
    }
}

// Two groups per lane: every column access of a lane is one 16-B load (a wave moves 1 KiB per instruction, half as many
// waves for the same bytes). Measured (profiles/r02_recompute.txt): 12.0 us against 11.2 us for one group per lane once
// the group-commit routine -- and the scratch its out-of-line call forced on EVERY wave -- was compiled out of the
// common kernel (that, not the access width, was what held the sweep at 16.7 us). Kept as a build-time variant.
#ifndef RG_RECOMPUTE_X2
#define RG_RECOMPUTE_X2 0
#endif
typedef u64 rg_u64x2 __attribute__((ext_vector_type(2)));
typedef u32 rg_u32x2 __attribute__((ext_vector_type(2)));

// one group of k_recompute2 (everything it needs already in registers); returns the group's result word
template <int P, bool COMMIT, bool GC>
RG_D u32 rg_recompute_one(const RgState &st, u64 g, u32 cfg, u64 (&mt)[P], u64 commit, u64 lo, u64 hi, u64 *mci_out, u8 *gc_out) {
    const u32 present = RG_CFG_PRESENT(cfg), incoming = RG_CFG_INCOMING(cfg), outgoing = RG_CFG_OUTGOING(cfg);
#pragma unroll
    for (int p = 0; p < P; p++)
        if (!((present >> p) & 1u)) mt[p] = 0; // a voter without a Progress acks 0 (majority.rs:80-82)
    u64 mci;
    bool used = false;
    if (GC && (cfg & RG_CFG_GROUP_COMMIT)) {
        u64 gidv[P];
#pragma unroll
        for (int p = 0; p < P; p++) gidv[p] = ((present >> p) & 1u) ? st.gid[(u64)p * st.stride + g] : 0ULL;
        mci = rg_mci_group<P>(mt, gidv, incoming, outgoing, used);
    } else {
        RgQuorum<P> qm;
        qm.init(mt);
        mci = qm.mci(mt, incoming, outgoing);
        used = incoming == 0 && outgoing == 0;
    }
    if (!COMMIT) {
        mci_out[g] = mci;
        if (gc_out) gc_out[g] = used ? 1 : 0;
        return 0;
    }
    const u64 commit0 = commit;
    if (!rg_log_maybe_commit(mci, commit, lo, hi)) return 0; // src/raft.rs:893-904
    if (st.pub) rg_pub_store(st, g, rg_pub_load(st, g) + (u32)rg_min(commit - commit0, 0x10000ULL), commit);
    st.commit[g] = commit;
    const u32 self = RG_CFG_SELF(cfg);
    if ((present >> self) & 1u) {
        const u64 o = (u64)self * st.stride + g;
        if (st.prc[o] < commit) st.prc[o] = commit;
    }
    return RG_OUT_CHANGED;
}

template <int P, bool COMMIT, bool GC>
__global__ __launch_bounds__(RG_BLOCK) void k_recompute2(RgState st, u64 *mci_out, u8 *gc_out) {
    const u64 g0 = ((u64)blockIdx.x * RG_BLOCK + threadIdx.x) * 2;
    if (g0 >= st.G) return;
    const bool two = g0 + 1 < st.G; // (the columns are padded to a multiple of 256 groups: the 16-B loads stay in bounds)
    const rg_u32x2 cfg2 = *reinterpret_cast<const rg_u32x2 *>(st.cfg + g0);
    rg_u64x2 mt2[P];
#pragma unroll
    for (int p = 0; p < P; p++) mt2[p] = *reinterpret_cast<const rg_u64x2 *>(st.match + (u64)p * st.stride + g0);
    rg_u64x2 commit2 = {0, 0}, lo2 = {0, 0}, hi2 = {0, 0};
    if (COMMIT) {
        commit2 = *reinterpret_cast<const rg_u64x2 *>(st.commit + g0);
        lo2 = *reinterpret_cast<const rg_u64x2 *>(st.lo + g0);
        hi2 = *reinterpret_cast<const rg_u64x2 *>(st.hi + g0);
    }
    u64 ma[P], mb[P];
#pragma unroll
    for (int p = 0; p < P; p++) {
        ma[p] = mt2[p].x;
        mb[p] = mt2[p].y;
    }
    const u32 oa = rg_recompute_one<P, COMMIT, GC>(st, g0, cfg2.x, ma, commit2.x, lo2.x, hi2.x, mci_out, gc_out);
    u32 ob = 0;
    if (two) ob = rg_recompute_one<P, COMMIT, GC>(st, g0 + 1, cfg2.y, mb, commit2.y, lo2.y, hi2.y, mci_out, gc_out);
    if (COMMIT) {
        if (two) {
            rg_u32x2 o;
            o.x = oa;
            o.y = ob;
            *reinterpret_cast<rg_u32x2 *>(st.out + g0) = o;
        } else {
            st.out[g0] = oa;
        }
    }
}

// Wave-cooperative recompute (RG_VARIANT_COOP): 8 lanes per group, lane s holds slot s's matched index;
// the q-th largest is found by a cross-lane rank select: every lane counts, with 7 xor-shuffles inside
// its 8-lane group, how many voters are >= its own value, and a 3-step butterfly max picks the largest
// value whose count reaches the quorum. No group commit here (the caller falls back to the lane kernel).
template <bool COMMIT>
__global__ __launch_bounds__(256) void k_recompute_coop(RgState st, u32 P, u64 *mci_out, u8 *gc_out) {
    const u32 s = threadIdx.x & 7u;
    const u64 g = (u64)blockIdx.x * 32 + (threadIdx.x >> 3);
    const bool live = g < st.G;
    const u64 gc = live ? g : st.G - 1; // keep every lane in the shuffles
    const u32 cfg = st.cfg[gc];
    const u32 present = RG_CFG_PRESENT(cfg), incoming = RG_CFG_INCOMING(cfg), outgoing = RG_CFG_OUTGOING(cfg);
    const u64 v = (s < P && ((present >> s) & 1u)) ? st.match[(u64)s * st.stride + gc] : 0ULL;
    u64 result[2];
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const u32 M = c == 0 ? incoming : outgoing;
        const u32 n = (u32)__builtin_popcount(M);
        const bool mine = (M >> s) & 1u;
        u32 cnt = mine ? 1u : 0u;
#pragma unroll
        for (int k = 1; k < 8; k++) {
            const u64 pv = __shfl_xor(v, k, 8);
            const bool theirs = (M >> (s ^ (u32)k)) & 1u;
            cnt += (theirs && pv >= v) ? 1u : 0u;
        }
        u64 cand = (mine && cnt >= n / 2u + 1u) ? v : 0ULL;
#pragma unroll
        for (int k = 1; k < 8; k <<= 1) {
            const u64 o = __shfl_xor(cand, k, 8);
            cand = o > cand ? o : cand;
        }
        result[c] = n == 0 ? ~0ULL : cand;
    }
    if (!live || s != 0) return;
    const u64 mci = result[0] < result[1] ? result[0] : result[1];
    if (COMMIT) {
        u64 commit = st.commit[g];
        const u64 commit0 = commit;
        u32 out = 0;
        if (rg_log_maybe_commit(mci, commit, st.lo[g], st.hi[g])) {
            if (st.pub) rg_pub_store(st, g, rg_pub_load(st, g) + (u32)rg_min(commit - commit0, 0x10000ULL), commit);
            st.commit[g] = commit;
            const u32 self = RG_CFG_SELF(cfg);
            if ((present >> self) & 1u) {
                const u64 o = (u64)self * st.stride + g;
                if (st.prc[o] < commit) st.prc[o] = commit;
            }
            out = RG_OUT_CHANGED;
        }
        st.out[g] = out;
    } else {
        mci_out[g] = mci;
        if (gc_out) gc_out[g] = (incoming == 0 && outgoing == 0) ? 1 : 0;
    }
}

// ------------------------------------------------------------------------------------------------
// kernels: votes and quorum liveness (src/quorum/majority.rs:130-154, joint.rs:56-67, tracker.rs:346-372)
// ------------------------------------------------------------------------------------------------
RG_D u32 rg_vote_majority(u32 M, u32 yes, u32 no) {
    const u32 n = (u32)__popc(M);
    if (n == 0) return 2u; // empty config wins
    const u32 q = n / 2u + 1u;
    const u32 y = (u32)__popc(M & yes), missing = (u32)__popc(M & ~(yes | no));
    if (y >= q) return 2u;            // Won
    if (y + missing >= q) return 0u;  // Pending
    return 1u;                        // Lost
}
RG_D u32 rg_vote_joint(u32 i, u32 o) {
    if (i == 2u && o == 2u) return 2u;
    if (i == 1u || o == 1u) return 1u;
    return 0u;
}

__global__ __launch_bounds__(RG_BLOCK) void k_vote(RgState st, const u8 *yes, const u8 *no, u8 *res, u8 *granted,
                                                   u8 *rejected) {
    const u64 g = (u64)blockIdx.x * RG_BLOCK + threadIdx.x;
    if (g >= st.G) return;
    const u32 cfg = st.cfg[g];
    const u32 y = yes[g], n = no[g] & ~y; // record_vote keeps the first vote (tracker.rs:307-309): yes wins a clash
    res[g] = (u8)rg_vote_joint(rg_vote_majority(RG_CFG_INCOMING(cfg), y, n),
                               rg_vote_majority(RG_CFG_OUTGOING(cfg), y, n));
    if (granted) { // tally_votes: votes of current voters only (tracker.rs:319-330)
        const u32 voters = RG_CFG_INCOMING(cfg) | RG_CFG_OUTGOING(cfg);
        granted[g] = (u8)__builtin_popcount(y & voters);
        rejected[g] = (u8)__builtin_popcount(n & voters);
    }
}

__global__ __launch_bounds__(RG_BLOCK) void k_quorum_active(RgState st, u8 *res) {
    const u64 g = (u64)blockIdx.x * RG_BLOCK + threadIdx.x;
    if (g >= st.G) return;
    const u32 cfg = st.cfg[g];
    const u32 self = RG_CFG_SELF(cfg), present = RG_CFG_PRESENT(cfg);
    u64 pf = st.pflags[g];
    u32 active = 0;
#pragma unroll
    for (int p = 0; p < 8; p++) {
        if (!((present >> p) & 1u)) continue;
        const u64 bit = (u64)RG_PF_RECENT_ACTIVE << (8 * p);
        if ((u32)p == self) {
            pf |= bit;
            active |= 1u << p;
        } else if (pf & bit) {
            active |= 1u << p;
            pf &= ~bit;
        }
    }
    st.pflags[g] = pf;
    // has_quorum: vote_result(|id| set.get(id).map(|_| true)) == Won (tracker.rs:367-372)
    res[g] = rg_vote_joint(rg_vote_majority(RG_CFG_INCOMING(cfg), active, 0),
                           rg_vote_majority(RG_CFG_OUTGOING(cfg), active, 0)) == 2u;
}

// send_heartbeat's commit = min(pr.matched, raft_log.committed) (src/raft.rs:830-838) for every slot
__global__ __launch_bounds__(RG_BLOCK) void k_heartbeat_commits(RgState st, u32 P, u64 *hb) {
    const u64 g = (u64)blockIdx.x * RG_BLOCK + threadIdx.x;
    if (g >= st.G) return;
    const u64 commit = st.commit[g];
    const u32 present = RG_CFG_PRESENT(st.cfg[g]);
    for (u32 p = 0; p < P; p++) {
        const u64 o = (u64)p * st.stride + g;
        const u64 m = st.match[o];
        hb[o] = ((present >> p) & 1u) ? (m < commit ? m : commit) : 0ULL;
    }
}

// ------------------------------------------------------------------------------------------------
// kernels: find_conflict_by_term pre-pass (only launched when a tick carries Message.log_term values)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(RG_BLOCK) void k_resolve_hints(RgState st, RgMsgs ms, u32 P, u64 *rh, u32 *raised) {
    const u64 g = (u64)blockIdx.x * RG_BLOCK + threadIdx.x;
    if (g >= st.G) return;
    rg_resolve_hints(st, ms, g, P, rh, raised);
}

// rg_resolve_host_hints: one lane per record
__global__ __launch_bounds__(256) void k_resolve_apply(RgState st, u32 *ins_meta, const rg_resolved_hint *it, u64 n, u32 P, u8 *applied) {
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const u32 res = rg_resolve_hint_at(
        st, ins_meta, it, P, i,
        [&](u64 g, u32 s) { // (byte g of the column lives in the aligned word g / 4; the column is padded to a multiple of 256)
            u32 *w = reinterpret_cast<u32 *>(st.hhint) + (g >> 2);
            const u32 sh = 8u * (u32)(g & 3u);
            return (atomicAnd(w, ~(1u << (sh + s))) >> sh) & 0xffu;
        },
        [&](u64 g, u32 bits, u32 clear) {
            if (bits) atomicOr(&st.out[g], bits);
            if (clear) atomicAnd(&st.out[g], ~clear);
        });
    applied[i] = (u8)res; // RG_RESOLVE_*
}


__global__ __launch_bounds__(RG_BLOCK) void k_count_out(const u32 *out, u64 G, u64 *counts) {
    u64 ch = 0, fl = 0, hh = 0;
    for (u64 g = (u64)blockIdx.x * RG_BLOCK + threadIdx.x; g < G; g += (u64)gridDim.x * RG_BLOCK) {
        const u32 o = out[g];
        ch += o & RG_OUT_CHANGED ? 1 : 0;
        fl += o & RG_OUT_FAULT ? 1 : 0;
        hh += o & RG_OUT_HOST_HINT ? 1 : 0;
    }
    for (int off = 32; off > 0; off >>= 1) {
        ch += __shfl_down(ch, off, 64);
        fl += __shfl_down(fl, off, 64);
        hh += __shfl_down(hh, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd((unsigned long long *)&counts[0], (unsigned long long)ch);
        atomicAdd((unsigned long long *)&counts[1], (unsigned long long)fl);
        if (hh) atomicAdd((unsigned long long *)&counts[2], (unsigned long long)hh);
    }
}

// rg_host_hints: the groups whose result word carries RG_OUT_HOST_HINT, packed group | slot mask << 56
__global__ __launch_bounds__(RG_BLOCK) void k_host_hints(const u32 *out, const u8 *hhint, u64 G, u64 *items, u64 *counter) {
    for (u64 g = (u64)blockIdx.x * RG_BLOCK + threadIdx.x; g < G; g += (u64)gridDim.x * RG_BLOCK) {
        if (!(out[g] & RG_OUT_HOST_HINT)) continue; // (rare: one atomic per flagged group)
        const u64 k = atomicAdd((unsigned long long *)counter, 1ULL);
        items[k] = g | ((u64)hhint[g] << 56);
    }
}

// message census of a tick: [0] VALID messages, [1] rejects, [2] slots in use (present), [3] groups with >=1 event,
// [4] elections (RG_MF_BECOME_LEADER on the leader's own slot, where the same bit does not mean "reject")
__global__ __launch_bounds__(RG_BLOCK) void k_msg_stats(const u64 *mflags, const u32 *cfg, u64 G, u64 *counts) {
    u64 a = 0, r = 0, s = 0, e = 0, el = 0;
    for (u64 g = (u64)blockIdx.x * RG_BLOCK + threadIdx.x; g < G; g += (u64)gridDim.x * RG_BLOCK) {
        const u64 mf = mflags[g];
        const u32 c = cfg[g];
        const u64 own = 0xffULL << (8 * RG_CFG_SELF(c));
        a += __popcll(mf & 0x0101010101010101ULL);
        r += __popcll((mf >> 1) & mf & 0x0101010101010101ULL & ~own);
        el += ((RG_CFG_PRESENT(c) >> RG_CFG_SELF(c)) & 1u) ? __popcll((mf >> 1) & 0x0101010101010101ULL & own) : 0;
        s += __popc(RG_CFG_PRESENT(c));
        e += mf != 0;
    }
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_down(a, off, 64);
        r += __shfl_down(r, off, 64);
        s += __shfl_down(s, off, 64);
        e += __shfl_down(e, off, 64);
        el += __shfl_down(el, off, 64);
    }
    __shared__ u64 part[5][RG_BLOCK / 64];
    if ((threadIdx.x & 63) == 0) {
        part[0][threadIdx.x >> 6] = a;
        part[1][threadIdx.x >> 6] = r;
        part[2][threadIdx.x >> 6] = s;
        part[3][threadIdx.x >> 6] = e;
        part[4][threadIdx.x >> 6] = el;
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        u64 t = 0;
        for (int w = 0; w < RG_BLOCK / 64; w++) t += part[threadIdx.x][w];
        atomicAdd((unsigned long long *)&counts[threadIdx.x], (unsigned long long)t);
    }
}


__global__ __launch_bounds__(256) void k_block_slots(const u32 *cfg, u64 G, u64 n_blocks, u32 P, u8 *need) {
    const u64 b = (u64)blockIdx.x * 256 + threadIdx.x;
    if (b >= n_blocks) return;
    u32 m = 1;
    for (u64 g = b * RG_BLOCK; g < (b + 1) * RG_BLOCK && g < G; g++) {
        const u32 k = rg_cfg_slots_named(cfg[g]);
        m = k > m ? k : m;
    }
    need[b] = (u8)rg_class_body(m, P);
}


