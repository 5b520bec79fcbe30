// engine.hip -- HIP kernels (gfx950 / CDNA4) and the C ABI of the multi-raft progress/commit engine.
//
// The hot path is HBM-bound integer work: per group ~9P+58A+37 algorithmic bytes and
// a few hundred integer ops. Layout and launch choices:
//   * per-slot u64 columns are peer-major [P][stride]: lane i of a wave reads group g0+i, so every
//     column access of a wave is one contiguous 512-B (dwordx2) or 1-KiB (dwordx4) segment;
//   * the per-slot flag bytes of a group are packed into ONE u64 row, so flags cost one 8-B load;
//   * one lane owns one group (RG_VARIANT_LANE) and keeps the whole group in VGPRs: all column loads
//     of a group are issued back to back (>= 16 KB in flight per wave) before any arithmetic;
//   * RG_VARIANT_LDS stages the peer columns of a 64-group batch through LDS with 16-B/lane global
//     loads (two groups per lane on the global side, one group per lane on the compute side);
//   * no atomics, no inter-workgroup communication: groups are independent;
//   * grid = one group per thread; consecutive workgroups walk consecutive 256-group tiles, so the 8
//     XCDs (block b -> XCD b%8) stream disjoint, interleaved 2-KiB column segments.
// There is NO CPU fallback anywhere in this file: without a HIP device every entry point fails.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <unordered_map>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <vector>

#include "rg_group.h"
#include "rg_send.h"
#include "rg_wire.h"
#include "rg_workload.h"

#include "rg_tick_kernels.h"

// live engines of this process PER DEVICE: the Infinity Cache is one per device, and a range of ONE engine only stays
// resident there (k_tick_split) while no other engine's traffic goes through it. Looked at ONCE, by rg_create, when the cache
// policy of the new engine is decided (RG_CACHE_AUTO grants a resident range only to an engine that is alone on its device);
// a live engine's kernel never changes because another engine comes or goes.
#define RG_MAX_DEVICES 64
static std::mutex g_live_mu;
static int g_live_on_device[RG_MAX_DEVICES];

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

static int rg_fail(int code, const char *fmt, ...) {
    // HIP keeps the last error until somebody reads it: a failed hipMalloc must not resurface later as the
    // "launch error" of an unrelated kernel (every launch site checks hipGetLastError)
    (void)hipGetLastError();
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define RG_HIP(expr)                                                                               \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return rg_fail(e__ == hipErrorOutOfMemory ? RG_ERR_OUT_OF_MEMORY : RG_ERR_NO_DEVICE,   \
                           "%s failed: %s", #expr, hipGetErrorString(e__));                        \
    } while (0)

struct rg_engine;
static int rg_mailbox_quiesce(rg_engine *h);
static int rg_require_hints_resolved(rg_engine *h, const char *who);
// Every entry point that puts work on the engine's stream starts here: select the device and, if the resident mailbox
// kernel is on the stream (rg_mailbox_start), tell it to leave -- stream order would make the call wait for it anyway
// (until its idle timeout), this makes the wait a few microseconds.
#define RG_ENTER(h)                                                                                \
    do {                                                                                           \
        RG_HIP(hipSetDevice((h)->cfg.device));                                                     \
        int rc__ = rg_mailbox_quiesce(h);                                                          \
        if (rc__) return rc__;                                                                     \
    } while (0)

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
__global__ __launch_bounds__(RG_BLOCK) void k_resolve_hints(RgState st, RgMsgs ms, u32 P, u64 *rh) {
    const u64 g = (u64)blockIdx.x * RG_BLOCK + threadIdx.x;
    if (g >= st.G) return;
    rg_resolve_hints(st, ms, g, P, rh);
}
__global__ __launch_bounds__(RG_BLOCK) void k_resolve_hints_list(RgState st, RgMsgs ms, u32 P, u64 *rh, const u64 *list,
                                                                 const u32 *n_ptr) {
    const u64 i = (u64)blockIdx.x * RG_BLOCK + threadIdx.x;
    if (i >= *n_ptr) return;
    rg_resolve_hints(st, ms, list[i], P, rh);
}

// ------------------------------------------------------------------------------------------------
// kernels: ingest (wire-order AoS records -> the slot matrix) and helpers of the sparse path
// ------------------------------------------------------------------------------------------------
// (the ingest arithmetic itself, rg_ingest_block, lives in rg_tick_kernels.h: the one-launch small-batch flush uses it too)
__global__ __launch_bounds__(RG_INGEST_BLOCK) void k_ingest(RgIngest a) {
    __shared__ uint4 stage[RG_INGEST_BLOCK * 4];
    rg_ingest_housekeeping(a.clr);
    rg_ingest_block(a, stage);
}

// 24-byte result record of the single-copy flush path (header: u32 n_groups, u32 n_duplicates, 8 B pad)
struct rg_res_rec {
    u64 group, commit;
    u32 out, pad;
};
#define RG_PACKED_HDR 16
#define RG_ROUNDTRIP_MAX 16384 /* records: above this the three-call sequence wins (measured crossover ~20 k) */
#define RG_ZEROCOPY_MAX 1024   /* groups: up to here the kernels read / write pinned host memory directly */

__global__ void k_gather_results(const u64 *list, const u32 *n_ptr, const u64 *commit, const u32 *out, u64 *rl, u64 *rc,
                                 u32 *ro, char *packed) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (packed && i == 0) {
        reinterpret_cast<u32 *>(packed)[0] = n_ptr[0];
        reinterpret_cast<u32 *>(packed)[1] = n_ptr[1];
    }
    if (i >= *n_ptr) return;
    const u64 g = list[i];
    const u64 c = commit[g];
    const u32 o = out[g];
    rl[i] = g;
    rc[i] = c;
    ro[i] = o;
    if (packed) {
        rg_res_rec r;
        r.group = g;
        r.commit = c;
        r.out = o;
        r.pad = 0;
        reinterpret_cast<rg_res_rec *>(packed + RG_PACKED_HDR)[i] = r;
    }
}

__global__ void k_clear_out(const u64 *list, u64 n, u32 *out) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[list[i]] = 0;
}

// ------------------------------------------------------------------------------------------------
// kernels: sparse cell writes, counters, workload
// ------------------------------------------------------------------------------------------------
// The send stage (rg_send.h): one lane per group. The work items of a whole 1024-thread workgroup are appended
// to the compact list with ONE atomic (wave prefix sums by shuffles, the 16 wave totals through LDS): at one
// atomic per wave the 15.6 K same-address atomics of a 1 M-group launch cost more than everything else together.
#ifndef RG_SEND_SPEC_LOADS
#define RG_SEND_SPEC_LOADS 0 /* 1: the dense stage requests the per-peer cells before the work set is known (rg_send.h: SPEC); measured slower (125 vs 112 us: the extra cells cost more than the round trip saves) */
#endif
#ifndef RG_SEND_WAVES
#define RG_SEND_WAVES 4 /* minimum waves per SIMD the dense stage is compiled for */
#endif
#define RG_SEND_BLOCK 1024
#define RG_SEND_SPEC 16384 /* work items copied speculatively with their count (512 KB of pinned memory) */
template <int P>
__global__ __launch_bounds__(RG_SEND_BLOCK) void k_send_appends(RgState st, RgIns ins, u64 max_entries, u32 flags,
                                                               const u64 *list, u64 n, const u32 *n_ptr,
                                                               rg_send_item *items, u32 *counter) {
    __shared__ u32 wave_tot[RG_SEND_BLOCK / 64];
    __shared__ u32 block_base;
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (n_ptr) n = *n_ptr; // list length still on the device (the single-round-trip flush); the grid covers a bound
    const bool active = i < n;
    const u64 g = active ? (list ? list[i] : i) : 0;
    RgSendRegs<P> it;
    it.count = 0;
    it.snap = 0;
    it.hostm = 0;
    if (active) {
        const u32 out = st.out[g];
        if (out) rg_group_send<P>(st, ins, g, out, max_entries, flags, it);
    }
    if (RG_SEND_EXP & 1) return;
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    u32 incl = it.count;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u32 v = __shfl_up(incl, d, 64);
        if (lane >= (u32)d) incl += v;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 total = 0;
#pragma unroll
        for (int w = 0; w < RG_SEND_BLOCK / 64; w++) {
            const u32 t = wave_tot[w];
            wave_tot[w] = total; // exclusive prefix
            total += t;
        }
        block_base = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    if (it.count == 0) return;
    u32 k = block_base + wave_tot[wave] + incl - it.count;
#pragma unroll
    for (int s = 0; s < P; s++) {
        const u32 nk = rg_send_nk<P>(it, s);
        if (!nk) continue;
        rg_send_item r;
        r.group = g;
        r.prev_index = it.prev[s];
        r.last_index = it.last[s];
        r.slot = (u32)s;
        r.n_msgs = (uint16_t)(nk & 0xffffu);
        r.kind = (uint16_t)(nk >> 16);
        if (!(RG_SEND_EXP & 4)) items[k] = r;
        k++;
    }
}

// ---- entry sizes for RG_SEND_BYTES (include/raftgroups.h: rg_log_sizes_*) ----
__global__ __launch_bounds__(256) void k_log_sizes_write(const rg_log_size *recs, u64 n, u64 G, u32 *esz, u32 w) {
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const rg_log_size r = recs[i];
    if (r.group < G) esz[r.group * w + ((u32)r.index & (w - 1u))] = (u32)r.cum_bytes;
}
// synthetic sizes: the window (last_index - w, last_index] of every group, cumulative from its oldest entry
__global__ __launch_bounds__(RG_BLOCK) void k_wl_sizes(const u64 *hi, u64 G, u32 *esz, u32 w, u64 seed, u32 min_bytes, u32 spread) {
    const u64 g = (u64)blockIdx.x * RG_BLOCK + threadIdx.x;
    if (g >= G) return;
    const u64 last = hi[g];
    const u64 first = last >= w ? last - w + 1 : 1;
    u32 acc = 0;
    for (u64 idx = first; idx <= last; idx++) {
        acc += min_bytes + (u32)(rg_hash(seed, 0x517eULL, g, idx) % ((u64)spread + 1));
        esz[g * w + ((u32)idx & (w - 1u))] = acc;
    }
}
// Progress::update_state(last) (src/tracker/progress.rs:231-243) for messages the HOST sent (rg_update_state). Lane i
// applies the whole run of records of its (group, slot) if it holds the run's first record.
__global__ __launch_bounds__(256) void k_update_state(RgState st, RgIns ins, const rg_sent_msg *m, u64 n, u32 P) {
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const u64 g = m[i].group;
    const u32 s = m[i].slot;
    if (g >= st.G || s >= P) return;
    if (i > 0 && m[i - 1].group == g && m[i - 1].slot == s) return;
    const u64 o = (u64)s * st.stride + g, base = (g * (u64)P + s) * ins.cap;
    u8 *pfb = reinterpret_cast<u8 *>(st.pflags) + g * 8 + s;
    u32 pb = *pfb;
    const u32 state = pb & RG_PF_STATE_MASK;
    if (state == RG_STATE_SNAPSHOT) return; // (the reference panics: nothing is sent to a peer in Snapshot)
    if (state == RG_STATE_PROBE) {
        *pfb = (u8)(pb | RG_PF_PAUSED);
        return;
    }
    const u32 meta0 = ins.meta[o];
    u32 start = meta0 & 0xffffu, count = meta0 >> 16;
    u64 head = ins.head[o], tail = ins.tail[o], next = st.next[o];
    for (u64 j = i; j < n && m[j].group == g && m[j].slot == s; j++) {
        if (count == ins.cap) break; // Inflights::add on a full window panics in the reference (inflights.rs:66-68)
        const u64 last = m[j].last;
        next = last + 1; // optimistic_update
        rg_ins_add(ins, base, start, count, head, tail, last);
    }
    st.next[o] = next;
    ins.meta[o] = start | (count << 16);
    ins.head[o] = head;
    ins.tail[o] = tail;
    *pfb = (u8)((pb & ~RG_PF_INS_FULL) | (count == ins.cap ? RG_PF_INS_FULL : 0u));
}

// The dense stage: every group of the shard, one lane each, work items into the peer-major columns (RgSendCols).
// A pure streaming kernel like the tick: 64-thread workgroups, no LDS, no atomics, no barrier.
// IX = u32 when every cell lies within 4 GiB of its column's start (32-bit cell offsets, rg_common.h: rg_at).
// The arguments are ONE struct and every phase -- the result word, the requests, the serve loop, the item stores -- reads the
// column pointers it needs from the kernarg segment ITSELF (as k_tick_send's phases do, rg_tick_kernels.h: RgTsKernarg): taken
// from the parameters all ~30 pointers are live from the first load to the last store, twice what the scalar registers hold
// -- round 4's build moved them in and out of VGPR lanes with 138 spill slots.
struct RgSendDenseArgs {
    RgState st;
    RgIns ins;
    u64 max_entries;
    u32 flags;
    RgSendCols oc;
};
#if defined(__HIP_DEVICE_COMPILE__)
struct RgSdKernarg {
    typedef const __attribute__((address_space(4))) RgSendDenseArgs *KA;
    RG_D static KA ptr() {
        KA ka = (KA)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka)); // (opaque per call: a phase's scalar loads cannot be hoisted into an earlier phase)
        return ka;
    }
};
#endif
template <int P, typename IX>
__global__ __launch_bounds__(RG_BLOCK, RG_SEND_WAVES) void k_send_dense(RgSendDenseArgs a_) {
#if defined(__HIP_DEVICE_COMPILE__)
    const u64 g64 = (u64)blockIdx.x * RG_BLOCK + threadIdx.x;
    if (g64 >= a_.st.G) return;
    const IX g = (IX)g64;
    const u32 flags = a_.flags;
    RgSendRegs<P> it;
    it.count = 0;
    it.snap = 0;
    it.hostm = 0;
#pragma unroll
    for (int s = 0; s < P; s++) it.n[s] = 0;
    u32 out;
    {
        const RgState st = RgSdKernarg::ptr()->st;
        out = rg_at(st.out, g);
    }
    const bool hold = rg_send_hold(out, flags);
    RgSendOps<P> q;
    constexpr bool SPEC = RG_SEND_SPEC_LOADS != 0, WAVE = RG_SEND_WAVE_LINES != 0;
    {   // (unconditional: its loads ride with `out`)
        const RgState st = RgSdKernarg::ptr()->st;
        const RgIns ins = RgSdKernarg::ptr()->ins;
        rg_send_request<P, IX, SPEC, false, false, WAVE && !SPEC>(st, ins, g, out, flags, q, nullptr, 0u, hold);
    }
    {
        const RgState st = RgSdKernarg::ptr()->st;
        const RgIns ins = RgSdKernarg::ptr()->ins;
        rg_send_serve<P, IX, false, WAVE && !SPEC>(st, ins, g, out, RgSdKernarg::ptr()->max_entries, flags, q, it, nullptr, 0u);
    }
    const RgSendCols oc = RgSdKernarg::ptr()->oc;
    const u64 stride = RgSdKernarg::ptr()->st.stride;
    rg_store_send_items<P, IX>(it, oc, stride, g);
#endif
}
#ifndef RG_SEND_IX32 /* the dense send stage's 32-bit cell index (rg_u32o measured: 125 -> 123 VGPRs, nothing else: profiles/r04_addressing.txt) */
#define RG_SEND_IX32 u32
#endif
template <int P>
static void rg_launch_send_dense(hipStream_t stream, dim3 grid, dim3 block, const RgState &st, const RgIns &ins, u64 max_entries,
                                 u32 flags, const RgSendCols &oc) {
    RgSendDenseArgs a;
    a.st = st;
    a.ins = ins;
    a.max_entries = max_entries;
    a.flags = flags;
    a.oc = oc;
    if (rg_ix32(st, P))
        hipLaunchKernelGGL((k_send_dense<P, RG_SEND_IX32>), grid, block, 0, stream, a);
    else
        hipLaunchKernelGGL((k_send_dense<P, u64>), grid, block, 0, stream, a);
}

// Compact list out of the columns, on request (rg_send_items / rg_send_items_ptr after a dense stage).
__global__ __launch_bounds__(256) void k_send_compact(RgSendCols oc, const u64 *tail, u64 G, u64 stride, u32 P, rg_send_item *items, u32 *counter) {
    __shared__ u32 wave_tot[4];
    __shared__ u32 block_base;
    const u64 g = (u64)blockIdx.x * 256 + threadIdx.x;
    u32 cnt = 0;
    if (g < G)
        for (u32 s = 0; s < P; s++) cnt += oc.n[(u64)s * stride + g] != 0;
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    u32 incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u32 v = __shfl_up(incl, d, 64);
        if (lane >= (u32)d) incl += v;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 total = 0;
        for (int w = 0; w < 4; w++) {
            const u32 t = wave_tot[w];
            wave_tot[w] = total;
            total += t;
        }
        block_base = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    if (cnt == 0) return;
    u32 k = block_base + wave_tot[wave] + incl - cnt;
    for (u32 s = 0; s < P; s++) {
        const u64 o = (u64)s * stride + g;
        const u32 nk = oc.n[o];
        if (!nk) continue;
        rg_send_item r;
        r.group = g;
        r.prev_index = oc.prev[o];
        // (the window's newest inflight / the item's own prev_index: rg_store_send_items)
        r.last_index = (nk & RG_SEND_NK_LAST_IS_TAIL) ? tail[o] : (nk & RG_SEND_NK_LAST_IS_PREV) ? r.prev_index : oc.last[o];
        r.slot = s;
        r.n_msgs = (uint16_t)(nk & 0xffffu);
        r.kind = (uint16_t)((nk >> 16) & 0x3fffu);
        items[k++] = r;
    }
}

// Status read-back: one thread per requested group gathers its cells into one record.
__global__ void k_read_groups(RgState st, const u64 *groups, u64 n, u32 P, const u32 *ins_meta, rg_group_status *out) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 g = groups[i];
    rg_group_status r;
    memset(&r, 0, sizeof(r));
    r.group = g;
    if (g < st.G) {
        r.commit = st.commit[g];
        r.term_lo = st.lo[g];
        r.last_index = st.hi[g];
        r.cfg = st.cfg[g];
        r.out = st.out[g];
        const u64 row = st.pflags[g];
        for (u32 p = 0; p < P; p++) {
            const u64 o = (u64)p * st.stride + g;
            r.match[p] = st.match[o];
            r.next[p] = st.next[o];
            r.pr_commit[p] = st.prc[o];
            r.pend_snap[p] = st.psnap[o];
            r.pend_rs[p] = st.prs[o];
            r.pflags[p] = (u8)(row >> (8 * p));
            if (ins_meta) {
                const u32 c = ins_meta[o] >> 16;
                r.inflights[p] = (u8)(c > 255u ? 255u : c);
            }
        }
    } else {
        r.group = ~0ULL; // no such group
    }
    out[i] = r;
}

__global__ void k_write_cells(RgState st, const rg_cell_write *cells, u64 n, u32 P, u32 *ins_meta) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    rg_cell_write c = cells[i];
    if (c.group >= st.G || c.slot >= P) return;
    const u64 o = (u64)c.slot * st.stride + c.group;
    if (ins_meta && (c.field_mask & (1u << RG_COL_PFLAGS))) {
        // device Inflights: a state change is Progress::reset_state (ins.reset(), progress.rs:75-80); the FULL
        // bit belongs to the engine and survives every other flag write
        const u8 old = reinterpret_cast<u8 *>(st.pflags)[c.group * 8 + c.slot];
        c.pflags &= (u8)~RG_PF_INS_FULL;
        if ((old ^ c.pflags) & RG_PF_STATE_MASK) ins_meta[o] = 0;
        else c.pflags |= old & RG_PF_INS_FULL;
    }
    if (c.field_mask & (1u << RG_COL_MATCH)) st.match[o] = c.match;
    if (c.field_mask & (1u << RG_COL_NEXT)) st.next[o] = c.next;
    if (c.field_mask & (1u << RG_COL_PR_COMMIT)) st.prc[o] = c.pr_commit;
    if (c.field_mask & (1u << RG_COL_PEND_SNAP)) st.psnap[o] = c.pend_snap;
    if (c.field_mask & (1u << RG_COL_PEND_RS)) st.prs[o] = c.pend_rs;
    if (c.field_mask & (1u << RG_COL_GID)) st.gid[o] = c.gid;
    // RG_PF_PEND_SNAP / _RS are the engine's as well: exact for the cell as it now stands
    u8 *pfb = reinterpret_cast<u8 *>(st.pflags) + c.group * 8 + c.slot;
    u8 nf = (c.field_mask & (1u << RG_COL_PFLAGS)) ? c.pflags : *pfb;
    nf = (u8)((nf & ~RG_PF_PENDING) | (st.psnap[o] ? RG_PF_PEND_SNAP : 0u) | (st.prs[o] ? RG_PF_PEND_RS : 0u));
    *pfb = nf;
}

// RawNode::report_unreachable / report_snapshot applied to the cells in place (rg_progress_events). Lane i applies the whole
// run of records of its (group, slot), in order, if it holds the run's first record.
__global__ __launch_bounds__(256) void k_progress_events(RgState st, u32 *ins_meta, const rg_progress_event *ev, u64 n, u32 P) {
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i < n) rg_progress_events_at(st, ins_meta, ev, n, P, i);
}

// ... and one kind of event for every group that names a slot (rg_progress_event_dense): lane = group
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

__global__ __launch_bounds__(256) void k_progress_event_dense(RgState st, u32 *ins_meta, const u8 *slot_plus1, u32 kind, u32 P) {
    const u64 g = (u64)blockIdx.x * 256 + threadIdx.x;
    if (g >= st.G) return;
    const u32 s1 = slot_plus1[g];
    if (!s1) return;
    const rg_progress_event ev = {g, s1 - 1u, kind};
    rg_progress_events_at(st, ins_meta, &ev, 1, P, 0);
}

// RG_PF_PEND_SNAP / RG_PF_PEND_RS (pending_snapshot / pending_request_snapshot != 0) re-derived for every cell: after the
// flag column or one of the two columns was loaded wholesale.
// RG_COL_RUN_COUNT from a freshly loaded RG_COL_RUN_FIRST
__global__ __launch_bounds__(RG_BLOCK) void k_fix_run_count(RgState st) {
    const u64 g = (u64)blockIdx.x * RG_BLOCK + threadIdx.x;
    if (g < st.G) rg_run_n(st)[g] = (u8)rg_count_runs(st, g);
}
__global__ __launch_bounds__(RG_BLOCK) void k_fix_pending(RgState st, u32 P) {
    const u64 g = (u64)blockIdx.x * RG_BLOCK + threadIdx.x;
    if (g >= st.G) return;
    const u64 row0 = st.pflags[g];
    u64 row = row0;
    for (u32 p = 0; p < P; p++) {
        const u64 o = (u64)p * st.stride + g;
        row &= ~((u64)RG_PF_PENDING << (8 * p));
        row |= (u64)((st.psnap[o] ? RG_PF_PEND_SNAP : 0u) | (st.prs[o] ? RG_PF_PEND_RS : 0u)) << (8 * p);
    }
    if (row != row0) st.pflags[g] = row;
}

// RG_PF_INS_FULL is engine-owned: Inflights::full() of a Replicate peer's device-side ring. Re-derived from the
// window counts whenever the host loads the windows or the flag column wholesale (rg_load_inflights,
// rg_load_column(RG_COL_PFLAGS)), so the next tick's is_paused() (progress.rs:210-216) sees the loaded window.
__global__ __launch_bounds__(RG_BLOCK) void k_fix_ins_full(RgState st, RgIns ins, u32 P) {
    const u64 g = (u64)blockIdx.x * RG_BLOCK + threadIdx.x;
    if (g >= st.G) return;
    const u64 row0 = st.pflags[g];
    u64 row = row0;
    for (u32 p = 0; p < P; p++) {
        const u32 pb = (u32)(row >> (8 * p)) & 0xffu;
        const bool full = (pb & RG_PF_STATE_MASK) == RG_STATE_REPLICATE && (ins.meta[(u64)p * st.stride + g] >> 16) == ins.cap;
        const u32 nb = (pb & ~RG_PF_INS_FULL) | (full ? RG_PF_INS_FULL : 0u);
        row = (row & ~(0xffULL << (8 * p))) | ((u64)nb << (8 * p));
    }
    if (row != row0) st.pflags[g] = row;
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

// rg_refresh_classes: per block of RG_BLOCK groups (= one workgroup of the lane kernels), the number of slots the block's cfg
// words name: 1 + the highest slot that is present, a voter of either majority, the leader's own, or the transferee.
RG_HD u32 rg_cfg_slots_named(u32 cfg) {
    const u32 tr = RG_CFG_TRANSFEREE(cfg);
    const u32 m = RG_CFG_PRESENT(cfg) | RG_CFG_INCOMING(cfg) | RG_CFG_OUTGOING(cfg) | (1u << RG_CFG_SELF(cfg)) | (tr ? 1u << (tr - 1u) : 0u);
    return 32u - (u32)__builtin_clz(m | 1u);
}
// the smallest body k_tick_classes<P> has for k slots (3, 5, 7 below P; P)
RG_HD u32 rg_class_body(u32 k, u32 P) { return (P > 3 && k <= 3) ? 3u : (P > 5 && k <= 5) ? 5u : (P > 7 && k <= 7) ? 7u : P; }
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

__global__ __launch_bounds__(RG_BLOCK) void k_wl_init(RgState st, u64 seed, u32 workload, u32 P, u64 first) {
    const u64 g = (u64)blockIdx.x * RG_BLOCK + threadIdx.x;
    if (g >= st.G) return;
    rg_wl_init_group(seed, workload, P, st.stride, g, first + rg_wl_place(workload, g, st.G), st.match, st.next, st.prc, st.psnap,
                     st.prs, st.gid, reinterpret_cast<u8 *>(st.pflags), st.commit, st.lo, st.hi, st.cfg);
    st.out[g] = 0;
    // the cold log-model columns: nothing compacted, no older runs known, leader term RG_WL_TERM0
    for (int k = 0; k < RG_TERM_RUNS; k++) {
        st.run_first[(u64)k * st.stride + g] = 0;
        st.run_term[(u64)k * st.stride + g] = 0;
    }
    rg_run_n(st)[g] = 0;
    st.dummy_idx[g] = 0;
    st.dummy_term[g] = 0;
    st.cur_term[g] = RG_WL_TERM0;
}

__global__ __launch_bounds__(RG_BLOCK) void k_wl_gen(RgState st, u64 seed, u32 workload, u32 P, u64 first,
                                                     u64 tick, u64 *mi, u64 *mc, u64 *mh, u64 *mrs, u8 *mf) {
    const u64 g = (u64)blockIdx.x * RG_BLOCK + threadIdx.x;
    if (g >= st.G) return;
    rg_wl_gen_group(seed, workload, P, st.stride, g, first + rg_wl_place(workload, g, st.G), tick, st.match, st.next,
                    reinterpret_cast<const u8 *>(st.pflags), st.commit, st.lo, st.hi, mi, mc, mh, mrs, mf);
}

// ------------------------------------------------------------------------------------------------
// engine object
// ------------------------------------------------------------------------------------------------
struct rg_engine {
    rg_config cfg;
    rg_device_info dev;
    u64 G, stride;
    u32 P;
    hipStream_t stream;
    char *arena;      // state columns
    size_t state_bytes;
    char *ckpt;       // checkpoint copy of the state columns (lazy)
    char *msg_arena;  // device staging for rg_tick(host msgs) / rg_flush (lazy)
    u64 *zero_col;    // [P][stride] zeros, substituted for NULL m_hint / m_rs
    u64 *rhint;       // [P][stride] reject hints after find_conflict_by_term (pre-pass output)
    u64 *d_counts;    // 4 x u64 scratch for reductions
    void *d_scratch;  // G x 8 B scratch for host<->device result shuttles
    size_t col_off[RG_COL_COUNT];
    RgState st;
    RgMsgs staged;    // views into msg_arena
    bool ticked;
    u64 tick_launches; // ticks enqueued so far (rg_flush: did a failed flush already change device state?)
    // sparse path (rg_ingest / rg_tick_ingested)
    char *sparse_arena;       // gmark | list | res_list | res_commit | res_out | counters
    u32 *gmark, *counters, *counters_base, *res_out;
    u64 *list, *res_list, *res_commit;
    // single-sync flush of the host mirror: pinned staging for the records, one packed D2H copy of the results
    rg_wire_msg *pin_records; // hipHostMalloc
    u64 pin_records_cap;
    char *d_packed, *pin_packed; // device / pinned host: header + rg_res_rec[]
    u64 packed_cap;           // records
    std::vector<u64> host_res_groups, host_res_commit;
    std::vector<u32> host_res_out;
    bool host_res_valid;      // the vectors hold the results of the last tick (served by rg_ingested_results)
    rg_cell_write *d_cells;   // device staging for rg_write_cells
    u64 d_cells_cap;
    rg_wire_msg *d_records;   // device staging for records
    u64 d_records_cap;
    u32 epoch;
    u64 ingested_upper;       // records accepted for upload since the last sparse tick (>= touched groups)
    u64 last_sparse_n;        // groups of the last rg_tick_ingested (result arrays are valid for them)
    bool out_is_dense;        // RG_COL_OUT was last written by a dense tick
    // send stage (rg_config.max_inflight > 0): Inflights rings, work items
    char *ins_arena;   // meta | head | tail | ring | items | counter
    char *ins_ckpt;    // checkpoint copy of meta | ring (lazy)
    u32 *esz, *esz_ckpt; // entry sizes for RG_SEND_BYTES (rg_log_sizes_enable), u32 [G][esz_w]; checkpoint copy (lazy)
    void *d_recs;      // staging for rg_log_sizes_write / rg_update_state records
    size_t d_recs_cap;
    // resident small-batch path (rg_mailbox_start): request / answer block in pinned host memory, whether the feature is
    // on, whether the host has launched an instance it has not seen leave, the last request number
    RgMbox *mbox;
    bool mbox_on, mbox_running;
    u32 mbox_seq;
    u64 mbox_idle_ticks;
    u64 mbox_served, mbox_launches; // flushes the resident workgroup answered / times it was (re)launched
    size_t ins_state_bytes;
    RgIns ins;
    rg_send_item *send_items;
    u32 *send_counter;
    RgSendCols send_cols;  // work items of a dense stage (peer-major columns)
    bool send_cols_fresh;  // ... hold the last stage's items and the compact list has not been materialised from them
    bool send_last_dense;  // the last stage was a dense one (the columns are its output)
    u64 send_bound;    // upper bound of the last stage's work items (groups it walked x peers)
    std::vector<rg_send_item> host_items; // items of the last stage when rg_flush_send fetched them
    bool host_items_valid;
    char *pin_send;    // pinned host: u32 count | pad | rg_send_item[RG_SEND_SPEC] (small stages: one round trip)
    // size classes (k_tick_classes): derived from RG_COL_CFG, lazily, by the first dense tick after anything wrote the column
    u8 *cls_need;      // device: one byte per block of RG_BLOCK groups (k_block_slots), padded to whole words
    std::vector<u8> cls_host; // its host copy
    bool cls_on;       // some block names fewer slots than the engine has: the dense lane tick runs k_tick_classes
    u32 *cls_order;    // device: one word per workgroup of that kernel, in launch order: block | slots << 28 (RgClasses::order)
    bool cls_stale;    // RG_COL_CFG may have changed since the bytes were derived
    bool cls_off;      // never use them: RG_CFGF_NO_SIZE_CLASSES at rg_create, or the cfg column's device pointer was handed out
    bool nt_msgs;      // dense ticks stream their message columns (non-temporal loads): state + one tick's messages > Infinity Cache
    bool nt_all;       // ... and the state columns, loads and stores: the state ALONE is far beyond the cache
    u64 nt_resident;   // ... except those of the first nt_resident workgroups' groups, which stay in the cache (k_tick_split); 0 = off
    bool counted_live;   // this engine is in g_live_on_device
    bool cls_block_order; // RG_CFGF_CLASS_BLOCK_ORDER
    bool hint_check_due; // device Inflights: a tick that may have raised RG_OUT_HOST_HINT (it carried log terms) ran and nobody has
                         // verified since that every hint was resolved (rg_require_hints_resolved)
    bool send_ready;   // a tick ran since the last rg_send_appends
    u64 stage_max_entries; // limit and flags of the last send stage (any form): rg_resolve_host_hints runs the stage of the
    u32 stage_flags;       // groups that stage skipped (RG_OUT_HOST_HINT) with the same ones
    bool ckpt_send_ready;
    bool ckpt_any_group_commit;
    bool any_group_commit; // some group's cfg word has RG_CFG_GROUP_COMMIT (tracked on cfg loads)
    // host mirror of RawNode::step (rg_set_peers / rg_step / rg_flush)
    std::vector<u64> peer_ids; // [G][8], 0 = unused
    std::vector<u64> terms;    // [G]
    std::vector<u64> q_mi, q_mc, q_mh, q_mrs, q_mlt; // [P][stride] host queues
    std::vector<u8> q_mf;                      // [G][8]
    std::vector<u64> q_dirty;                  // groups touched since the last flush
    std::vector<rg_wire_msg> q_records;        // flush staging (wire-order records of the dirty groups)
    bool q_any_logterm;                        // some queued message carries Message.log_term
    struct RgQueuedElection { u64 group, old_term; };
    std::vector<RgQueuedElection> q_elections; // rg_local_become_leader calls of the pending flush: group, the term its gate had
    std::vector<u32> host_cfg;                 // host copy of RG_COL_CFG for the mirror (self slots)
    bool host_cfg_valid;
    bool host_mirror;
    struct RgPub *pub; // commit publication across ranks (rg_comm_init), nullptr = single engine
};

// RCCL is bound lazily (the library is ~0.5 GB; single-GPU users never load it). The types come from its header.
#include <dlfcn.h>
#include <rccl/rccl.h>

struct RgRccl {
    void *lib;
    decltype(&ncclGetUniqueId) GetUniqueId;
    decltype(&ncclCommInitRank) CommInitRank;
    decltype(&ncclCommDestroy) CommDestroy;
    decltype(&ncclAllGather) AllGather;
    decltype(&ncclGetErrorString) GetErrorString;
    decltype(&ncclGroupStart) GroupStart;
    decltype(&ncclGroupEnd) GroupEnd;
};
static RgRccl g_rccl = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
static std::mutex g_rccl_mu; // (engines of one process may be driven by one thread each: the first loads, the others wait)

static int rg_rccl_load() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.lib) return RG_OK;
    static const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *lib = nullptr;
    for (const char *n : names)
        if ((lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!lib) return rg_fail(RG_ERR_NO_DEVICE, "rg_comm: cannot load RCCL (librccl.so.1): %s", dlerror());
    g_rccl.GetUniqueId = reinterpret_cast<decltype(&ncclGetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(&ncclCommInitRank)>(dlsym(lib, "ncclCommInitRank"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(&ncclCommDestroy)>(dlsym(lib, "ncclCommDestroy"));
    g_rccl.AllGather = reinterpret_cast<decltype(&ncclAllGather)>(dlsym(lib, "ncclAllGather"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(&ncclGetErrorString)>(dlsym(lib, "ncclGetErrorString"));
    g_rccl.GroupStart = reinterpret_cast<decltype(&ncclGroupStart)>(dlsym(lib, "ncclGroupStart"));
    g_rccl.GroupEnd = reinterpret_cast<decltype(&ncclGroupEnd)>(dlsym(lib, "ncclGroupEnd"));
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllGather || !g_rccl.GetErrorString ||
        !g_rccl.GroupStart || !g_rccl.GroupEnd) {
        dlclose(lib);
        return rg_fail(RG_ERR_NO_DEVICE, "rg_comm: the RCCL library lacks an expected symbol");
    }
    g_rccl.lib = lib;
    return RG_OK;
}

// Send slices in rotation: the ticks of interval i accumulate into slice i % RG_PUB_SEND while the exchanges of the
// previous intervals still read theirs. Re-use is gated on the HOST (hipEventSynchronize on the exchange that last
// read the slice, three publications back: normally long finished), so the engine's stream carries no cross-stream
// wait -- a barrier packet per tick costs ~5 us of a ~58 us tick (profiles/r02_publish_overhead.txt).
#define RG_PUB_SEND 4
struct RgPub {
    u32 rank, world;
    ncclComm_t comm;          // RCCL transport (nullptr with a custom transport)
    rg_allgather_fn transport;
    void *transport_user;
    RgPubLayout lay;
    u32 ring;                 // publications buffered before the replica is brought up to date
    hipStream_t side;         // the exchange runs here; the engine's stream only records / waits events
    char *send[RG_PUB_SEND];  // this rank's slice under construction (a small ring), bytes_per_rank each
    char *ring_buf;           // [ring][world][bytes_per_rank]
    u64 *replica;             // [world][Gpad]
    u64 *full_send;           // [Gpad] snapshot of the commit column for a full publication
    u32 *d_lost;              // device: some gathered slice asked for a resynchronisation (set by the replica update)
    u32 *pin_lost;            // pinned host: [2] copies of d_lost taken at the last two check points
    hipEvent_t ev_tick[RG_PUB_SEND], ev_done[RG_PUB_SEND], ev_chk[2];
    bool done_pending[RG_PUB_SEND], chk_pending[2];
    u64 n_pub;                // publications so far
    u32 pending;              // ring slots gathered and not yet folded into the replica
    bool in_process;          // one of several ranks of ONE process driven by one thread (rg_comm_init_all / rg_publish_commit_all)
    hipEvent_t ev_read;       // ... in-process transport: this rank's side stream has read every rank's slice of the current publication
    bool local_lost;          // this rank's deltas no longer describe its commit column (restore / column load)
    bool lost_announced;      // ... and a slice carrying RG_PUB_LOST has gone out (the full snapshot follows)
    rg_publish_stats stats;
};

static size_t rg_align(size_t x) { return (x + 255) & ~(size_t)255; }

static size_t rg_col_elem(int c) {
    if (c == RG_COL_PFLAGS) return 8; // one u64 row per group
    if (c == RG_COL_CFG || c == RG_COL_OUT) return 4;
    if (c == RG_COL_HOST_HINT || c == RG_COL_RUN_COUNT) return 1;
    return 8;
}
static bool rg_col_per_slot(int c) { return c <= RG_COL_GID; }
static bool rg_col_per_run(int c) { return c == RG_COL_RUN_FIRST || c == RG_COL_RUN_TERM; }

#define RG_STR2(x) #x
#define RG_STR(x) RG_STR2(x)
extern "C" const char *rg_version(void) { return "raftgroups 0.1 (gfx950, opt " RG_STR(RG_OPT) ")"; }
extern "C" const char *rg_last_error(void) { return g_last_error.c_str(); }

extern "C" int rg_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" uint64_t rg_column_bytes(const rg_engine *h, int c) {
    if (!h || c < 0 || c >= RG_COL_COUNT) return 0;
    if (rg_col_per_slot(c)) return (uint64_t)h->P * h->stride * 8;
    if (rg_col_per_run(c)) return (uint64_t)RG_TERM_RUNS * h->stride * 8;
    return (uint64_t)h->G * rg_col_elem(c);
}

static void *rg_col(rg_engine *h, int c) { return h->arena + h->col_off[c]; }

// (rg_create's failure paths and rg_destroy)
static void rg_drop(rg_engine *h) {
    if (h->counted_live) {
        std::lock_guard<std::mutex> lk(g_live_mu);
        g_live_on_device[h->cfg.device]--;
    }
    delete h;
}

extern "C" int rg_create(const rg_config *cfg, rg_engine **out) {
    if (!cfg || !out) return rg_fail(RG_ERR_INVALID_ARG, "rg_create: null argument");
    if (cfg->n_groups == 0 || cfg->n_slots == 0 || cfg->n_slots > RG_MAX_SLOTS)
        return rg_fail(RG_ERR_INVALID_ARG, "rg_create: n_groups=%llu n_slots=%u out of range",
                       (unsigned long long)cfg->n_groups, cfg->n_slots);
    if (cfg->variant > RG_VARIANT_COMPACT) return rg_fail(RG_ERR_INVALID_ARG, "rg_create: unknown variant %u", cfg->variant);
    if (cfg->cache_policy > RG_CACHE_RESIDENT) return rg_fail(RG_ERR_INVALID_ARG, "rg_create: unknown cache_policy %u", cfg->cache_policy);
    if (cfg->flags & ~(RG_CFGF_NO_SIZE_CLASSES | RG_CFGF_CLASS_BLOCK_ORDER | RG_CFGF_IX64))
        return rg_fail(RG_ERR_INVALID_ARG, "rg_create: unknown flags %#x", cfg->flags);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return rg_fail(RG_ERR_NO_DEVICE, "rg_create: no HIP device visible (this engine has no CPU fallback)");
    if (cfg->device < 0 || cfg->device >= ndev || cfg->device >= RG_MAX_DEVICES)
        return rg_fail(RG_ERR_INVALID_ARG, "rg_create: device %d of %d", cfg->device, ndev);
    RG_HIP(hipSetDevice(cfg->device));
    hipDeviceProp_t prop;
    RG_HIP(hipGetDeviceProperties(&prop, cfg->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return rg_fail(RG_ERR_NO_DEVICE, "rg_create: device %d is %s; this library carries gfx950 (CDNA4) kernels only",
                       cfg->device, prop.gcnArchName);
    if (prop.warpSize != 64)
        return rg_fail(RG_ERR_NO_DEVICE, "rg_create: wavefront size %d, the kernels are written for 64", prop.warpSize);
    rg_engine *h = new (std::nothrow) rg_engine();
    if (!h) return rg_fail(RG_ERR_OUT_OF_MEMORY, "rg_create: host allocation failed");
    memset(&h->dev, 0, sizeof(h->dev));
    for (int i = 0; i < 31 && prop.gcnArchName[i] && prop.gcnArchName[i] != ':'; i++) h->dev.arch[i] = prop.gcnArchName[i];
    h->dev.compute_units = (uint32_t)prop.multiProcessorCount;
    h->dev.wavefront = (uint32_t)prop.warpSize;
    h->dev.lds_per_workgroup = prop.sharedMemPerBlock;
    h->dev.hbm_bytes = prop.totalGlobalMem;
    h->dev.l2_bytes = (uint64_t)prop.l2CacheSize;
    h->cfg = *cfg;
    h->G = cfg->n_groups;
    h->P = cfg->n_slots;
    h->stride = (h->G + 255) / 256 * 256;
    h->stream = nullptr;
    h->ckpt = nullptr;
    h->msg_arena = nullptr;
    h->ticked = false;
    h->tick_launches = 0;
    h->sparse_arena = nullptr;
    h->d_records = nullptr;
    h->d_records_cap = 0;
    h->d_cells = nullptr;
    h->d_cells_cap = 0;
    h->pin_records = nullptr;
    h->pin_records_cap = 0;
    h->d_packed = nullptr;
    h->pin_packed = nullptr;
    h->packed_cap = 0;
    h->host_res_valid = false;
    h->epoch = 1;
    h->ingested_upper = 0;
    h->last_sparse_n = 0;
    h->out_is_dense = true;
    h->any_group_commit = false;
    h->host_mirror = false;
    h->host_cfg_valid = false;
    h->q_any_logterm = false;
    h->ins_arena = nullptr;
    h->ins_ckpt = nullptr;
    h->esz = h->esz_ckpt = nullptr;
    h->d_recs = nullptr;
    h->d_recs_cap = 0;
    h->mbox = nullptr;
    h->mbox_on = h->mbox_running = false;
    h->mbox_seq = 0;
    h->mbox_idle_ticks = 0;
    h->mbox_served = h->mbox_launches = 0;
    h->ins.esz = nullptr;
    h->ins.esz_w = 0;
    h->ins_state_bytes = 0;
    h->ins.meta = nullptr;
    h->ins.head = nullptr;
    h->ins.tail = nullptr;
    h->ins.ring = nullptr;
    h->ins.cap = 0;
    h->send_items = nullptr;
    h->send_counter = nullptr;
    h->send_cols.prev = h->send_cols.last = nullptr;
    h->send_cols.n = nullptr;
    h->send_cols_fresh = false;
    h->send_last_dense = false;
    h->send_ready = false;
    h->hint_check_due = false;
    h->cls_need = nullptr;
    h->cls_on = false;
    h->cls_order = nullptr;
    h->cls_stale = true;
    h->cls_off = (cfg->flags & RG_CFGF_NO_SIZE_CLASSES) != 0;
    h->cls_block_order = (cfg->flags & RG_CFGF_CLASS_BLOCK_ORDER) != 0;
    h->stage_max_entries = 0;
    h->stage_flags = 0;
    // ---- cache policy (rg_config.cache_policy; include/raftgroups.h: RG_CACHE_*), decided here and nowhere else ----
    // Infinity Cache (256 MB on MI355X). AUTO, by footprint:
    //  * STREAM_MSGS when the state a dense tick re-reads (24 P + 40 B per group) and the message columns of ONE tick
    //    (16 P + 8 B per group, read once) do not fit together (with the Inflights on the device a step also touches the
    //    window and work-item columns: 40 P B per group more);
    //  * STREAM_ALL when the state alone is more than 1.5 x the cache -- by the time a launch comes back to a line the cache
    //    has turned over, so allocating there only costs -- and the shard holds at most 13 M groups. Both ends are measured, at
    //    3, 5 and 7 slots (profiles/r04_nt_state.txt; profiles/r05_cache_policy_sweep.txt): the lower one follows the BYTES of
    //    state (1.3 x: streamed loses 1-5 %; 1.5 x: wins 6-9 % at every slot count; 8 M x 5: 528 -> 483 us, fraction 0.68 ->
    //    0.75), the upper one the NUMBER of groups -- at 12 M groups everything streamed wins at 3, 5 and 7 slots alike (529 /
    //    786 / 1046 us against 559 / 812 / 1068), at 14 M it loses or ties (705 / 1006 / 1275 against 664 / 975 / 1270), at 16 M
    //    it loses 6-8 % -- although the state of those engines spans 1.3 to 3.2 GB (round 4 had put that end at 7.5 x the cache
    //    in bytes, from 5 slots alone: right there, 12 % wrong at 3 slots);
    //  * RESIDENT (k_tick_split): a leading range of the groups keeps its state in the cache, the rest is streamed. Measured
    //    (profiles/r04_resident.txt, r05_cache_policy_sweep.txt): with 176 MB of state resident 2.4 M x 5 runs in 134 us instead
    //    of 150 (all streamed; 155 plain), 4 M x 5 in 227-233 instead of 246; at 3 / 7 slots it is the fastest policy from 1.3 x
    //    (114 / 126 us against 125 / 128) through 2.5 x the cache (237 / 248 against 243 / 259) and loses beyond (3.5 x at 3
    //    slots: 393 against 339; 8 M x 5: 491 -> 556) -- over a launch that long the resident lines are gone before the next one
    //    comes back to them; below 1.25 x (1.1 x: 95 / 109 against 94 / 102) the plain accesses with streamed messages win.
    //    So: 1.25 x cache < state <= 2.5 x cache. The cache is ONE per device: three size-class engines of config 5 at
    //    8 M groups, two of them with a resident range, took 893 us instead of 724 -- so AUTO grants the range only to an
    //    engine that is ALONE on its device when it is created (engines_on_device == 1 in rg_device_info); a later engine on the
    //    same device gets the streaming policy of its size and the first one keeps what it was given. A host that knows better
    //    says so: an explicit policy is honoured as given.
    {
        std::lock_guard<std::mutex> lk(g_live_mu);
        h->dev.engines_on_device = (uint32_t)++g_live_on_device[cfg->device];
        h->counted_live = true;
    }
    {
        const double mall = 256.0 * 1024.0 * 1024.0;
        const double per_group = (double)(24u * h->P + 40u), state = (double)h->G * per_group;
        const double with_msgs = (double)h->G * (double)(40u * h->P + 48u + (cfg->max_inflight ? 40u * h->P : 0u));
        const bool lane = cfg->variant == RG_VARIANT_DEFAULT || cfg->variant == RG_VARIANT_LANE || cfg->variant == RG_VARIANT_COOP;
        u32 pol = cfg->cache_policy;
        if (pol == RG_CACHE_AUTO) {
            pol = with_msgs > mall ? RG_CACHE_STREAM_MSGS : RG_CACHE_PLAIN;
            if (!cfg->max_inflight && state > 1.5 * mall && h->G <= 13000000ull) pol = RG_CACHE_STREAM_ALL;
            if (!cfg->max_inflight && lane && state > 1.25 * mall && state <= 2.5 * mall && h->dev.engines_on_device == 1)
                pol = RG_CACHE_RESIDENT;
        }
        if (cfg->max_inflight && pol == RG_CACHE_RESIDENT) { // (k_tick_split has no send stage)
            rg_drop(h);
            return rg_fail(RG_ERR_INVALID_ARG, "rg_create: RG_CACHE_RESIDENT needs max_inflight = 0 (engines with device Inflights: "
                                               "RG_CACHE_PLAIN, RG_CACHE_STREAM_MSGS or RG_CACHE_STREAM_ALL)");
        }
        h->nt_msgs = pol >= RG_CACHE_STREAM_MSGS;
        h->nt_all = pol >= RG_CACHE_STREAM_ALL;
        h->nt_resident = 0;
        if (pol == RG_CACHE_RESIDENT) {
            const u64 groups = cfg->cache_resident_groups ? cfg->cache_resident_groups : (u64)(176.0 * 1024.0 * 1024.0 / per_group);
            h->nt_resident = rg_min(groups, h->G) / RG_BLOCK;
            if (h->nt_resident == 0) pol = RG_CACHE_STREAM_ALL; // (less than one workgroup: nothing to keep)
        }
        h->dev.cache_policy = pol;
        h->dev.resident_groups = h->nt_resident * RG_BLOCK;
    }
    h->send_bound = 0;
    h->pin_send = nullptr;
    h->host_items_valid = false;
    h->ckpt_send_ready = false;
    h->ckpt_any_group_commit = false;
    if (cfg->max_inflight > 65535u) {
        rg_drop(h);
        return rg_fail(RG_ERR_INVALID_ARG, "rg_create: max_inflight=%u, at most 65535", cfg->max_inflight);
    }
    size_t off = 0;
    for (int c = 0; c < RG_COL_COUNT; c++) {
        h->col_off[c] = off;
        const size_t bytes = rg_col_per_slot(c)  ? (size_t)h->P * h->stride * 8
                             : rg_col_per_run(c) ? (size_t)RG_TERM_RUNS * h->stride * 8
                                                 : (size_t)h->stride * rg_col_elem(c);
        off += rg_align(bytes);
    }
    h->state_bytes = off;
    const size_t zero_bytes = rg_align((size_t)h->P * h->stride * 8);
    hipError_t e = hipMalloc(&h->arena, off + 2 * zero_bytes + 256 + rg_align(h->stride * 8));
    if (e != hipSuccess) {
        rg_drop(h);
        return rg_fail(RG_ERR_OUT_OF_MEMORY, "rg_create: hipMalloc(%zu) failed: %s", off, hipGetErrorString(e));
    }
    e = hipMemset(h->arena, 0, off + 2 * zero_bytes + 256 + rg_align(h->stride * 8));
    // hipMemset of device memory returns before the fill has run (it is queued on the NULL stream), and a caller's
    // stream created non-blocking (every torch.cuda.Stream) is not ordered behind the NULL stream: without this wait a
    // first kernel on such a stream races the fill (seen at 8 M groups: 0.1 % of the groups zeroed again after
    // rg_workload_init had written them)
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) {
        (void)hipFree(h->arena);
        rg_drop(h);
        return rg_fail(RG_ERR_NO_DEVICE, "rg_create: hipMemset failed: %s", hipGetErrorString(e));
    }
    h->dev.engine_bytes = off + 2 * zero_bytes + 256 + rg_align(h->stride * 8);
    h->zero_col = reinterpret_cast<u64 *>(h->arena + off);
    h->d_counts = reinterpret_cast<u64 *>(h->arena + off + zero_bytes);
    h->d_scratch = h->arena + off + zero_bytes + 256;
    h->rhint = reinterpret_cast<u64 *>(h->arena + off + zero_bytes + 256 + rg_align(h->stride * 8));
    RgState &s = h->st;
    s.match = (u64 *)rg_col(h, RG_COL_MATCH);
    s.next = (u64 *)rg_col(h, RG_COL_NEXT);
    s.prc = (u64 *)rg_col(h, RG_COL_PR_COMMIT);
    s.psnap = (u64 *)rg_col(h, RG_COL_PEND_SNAP);
    s.prs = (u64 *)rg_col(h, RG_COL_PEND_RS);
    s.gid = (u64 *)rg_col(h, RG_COL_GID);
    s.pflags = (u64 *)rg_col(h, RG_COL_PFLAGS);
    s.commit = (u64 *)rg_col(h, RG_COL_COMMIT);
    s.lo = (u64 *)rg_col(h, RG_COL_TERM_LO);
    s.hi = (u64 *)rg_col(h, RG_COL_TERM_HI);
    s.cfg = (u32 *)rg_col(h, RG_COL_CFG);
    s.out = (u32 *)rg_col(h, RG_COL_OUT);
    s.run_first = (u64 *)rg_col(h, RG_COL_RUN_FIRST);
    s.run_term = (u64 *)rg_col(h, RG_COL_RUN_TERM);
    s.dummy_idx = (u64 *)rg_col(h, RG_COL_DUMMY_INDEX);
    s.dummy_term = (u64 *)rg_col(h, RG_COL_DUMMY_TERM);
    s.cur_term = (u64 *)rg_col(h, RG_COL_CUR_TERM);
    s.hhint = (u8 *)rg_col(h, RG_COL_HOST_HINT);
    s.G = h->G;
    s.stride = h->stride;
    // (consecutive byte columns, strides of 256: RG_COL_RUN_COUNT sits `stride` bytes behind RG_COL_HOST_HINT by construction)
    if ((u8 *)rg_col(h, RG_COL_RUN_COUNT) != rg_run_n(s)) {
        (void)hipFree(h->arena);
        rg_drop(h);
        return rg_fail(RG_ERR_INVALID_ARG, "rg_create: column layout");
    }
    s.ix64 = (cfg->flags & RG_CFGF_IX64) ? 1u : 0u; // (rg_common.h: rg_ix32)
    s.pub = nullptr;
    s.pub_off_delta = 0;
    s.pub_cap = 0;
    h->pub = nullptr;
    if (cfg->max_inflight) { // Inflights rings + the send stage's work-item list
        const size_t meta_b = rg_align((size_t)h->P * h->stride * 4) + 2 * rg_align((size_t)h->P * h->stride * 8); // meta | head | tail
        const size_t ring_b = rg_align((size_t)h->G * h->P * cfg->max_inflight * 8);
        const size_t items_b = rg_align((size_t)h->G * h->P * sizeof(rg_send_item));
        const size_t col8_b = rg_align((size_t)h->P * h->stride * 8), col4_b = rg_align((size_t)h->P * h->stride * 4);
        const size_t cols_b = 2 * col8_b + col4_b; // RgSendCols: prev | last | n
        e = hipMalloc(&h->ins_arena, meta_b + ring_b + items_b + 256 + cols_b);
        if (e == hipSuccess) e = hipMemset(h->ins_arena, 0, meta_b + ring_b + items_b + 256 + cols_b);
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr); // (as above: the fill must have run before rg_create returns)
        if (e != hipSuccess) {
            if (h->ins_arena) (void)hipFree(h->ins_arena);
            (void)hipFree(h->arena);
            rg_drop(h);
            return rg_fail(RG_ERR_OUT_OF_MEMORY, "rg_create: %zu bytes of Inflights (cap %u): %s", meta_b + ring_b + items_b + cols_b,
                           cfg->max_inflight, hipGetErrorString(e));
        }
        h->ins.meta = reinterpret_cast<u32 *>(h->ins_arena);
        h->ins.head = reinterpret_cast<u64 *>(h->ins_arena + rg_align((size_t)h->P * h->stride * 4));
        h->ins.tail = reinterpret_cast<u64 *>(h->ins_arena + rg_align((size_t)h->P * h->stride * 4) +
                                              rg_align((size_t)h->P * h->stride * 8));
        h->ins.ring = reinterpret_cast<u64 *>(h->ins_arena + meta_b);
        h->ins.cap = cfg->max_inflight;
        h->ins_state_bytes = meta_b + ring_b;
        h->send_items = reinterpret_cast<rg_send_item *>(h->ins_arena + meta_b + ring_b);
        h->send_counter = reinterpret_cast<u32 *>(h->ins_arena + meta_b + ring_b + items_b);
        char *cols = h->ins_arena + meta_b + ring_b + items_b + 256;
        h->send_cols.prev = reinterpret_cast<u64 *>(cols);
        h->send_cols.last = reinterpret_cast<u64 *>(cols + col8_b);
        h->send_cols.n = reinterpret_cast<u32 *>(cols + 2 * col8_b);
        h->dev.engine_bytes += meta_b + ring_b + items_b + 256 + cols_b;
    }
    *out = h;
    return RG_OK;
}

extern "C" int rg_comm_destroy(rg_engine *h);

extern "C" void rg_destroy(rg_engine *h) {
    if (!h) return;
    (void)hipSetDevice(h->cfg.device);
    (void)rg_mailbox_quiesce(h);
    (void)hipStreamSynchronize(h->stream);
    if (h->mbox) (void)hipHostFree(h->mbox);
    if (h->pub) (void)rg_comm_destroy(h);
    if (h->arena) (void)hipFree(h->arena);
    if (h->ckpt) (void)hipFree(h->ckpt);
    if (h->cls_need) (void)hipFree(h->cls_need);
    if (h->cls_order) (void)hipFree(h->cls_order);
    if (h->ins_arena) (void)hipFree(h->ins_arena);
    if (h->ins_ckpt) (void)hipFree(h->ins_ckpt);
    if (h->esz) (void)hipFree(h->esz);
    if (h->esz_ckpt) (void)hipFree(h->esz_ckpt);
    if (h->d_recs) (void)hipFree(h->d_recs);
    if (h->msg_arena) (void)hipFree(h->msg_arena);
    if (h->sparse_arena) (void)hipFree(h->sparse_arena);
    if (h->d_records) (void)hipFree(h->d_records);
    if (h->d_cells) (void)hipFree(h->d_cells);
    if (h->pin_records) (void)hipHostFree(h->pin_records);
    if (h->pin_send) (void)hipHostFree(h->pin_send);
    if (h->d_packed) (void)hipFree(h->d_packed);
    if (h->pin_packed) (void)hipHostFree(h->pin_packed);
    rg_drop(h);
}

extern "C" uint64_t rg_stride(const rg_engine *h) { return h ? h->stride : 0; }

extern "C" int rg_get_device_info(const rg_engine *h, rg_device_info *info) {
    if (!h || !info) return rg_fail(RG_ERR_INVALID_ARG, "rg_get_device_info: bad argument");
    *info = h->dev;
    return RG_OK;
}

extern "C" int rg_set_stream(rg_engine *h, void *hip_stream) {
    if (!h) return rg_fail(RG_ERR_INVALID_ARG, "rg_set_stream: null engine");
    RG_ENTER(h); // (a resident mailbox workgroup sits on the OLD stream: it has to leave before the engine moves)
    h->stream = reinterpret_cast<hipStream_t>(hip_stream);
    return RG_OK;
}

extern "C" int rg_sync(rg_engine *h) {
    if (!h) return rg_fail(RG_ERR_INVALID_ARG, "rg_sync: null engine");
    RG_ENTER(h);
    RG_HIP(hipStreamSynchronize(h->stream));
    return RG_OK;
}

extern "C" int rg_load_column(rg_engine *h, int c, const void *src, uint64_t bytes) {
    if (!h || !src || c < 0 || c >= RG_COL_COUNT) return rg_fail(RG_ERR_INVALID_ARG, "rg_load_column: bad argument");
    if (c == RG_COL_HOST_HINT) return rg_fail(RG_ERR_INVALID_ARG, "rg_load_column: RG_COL_HOST_HINT is written by the ticks only");
    if (c == RG_COL_RUN_COUNT) return rg_fail(RG_ERR_INVALID_ARG, "rg_load_column: RG_COL_RUN_COUNT is derived from RG_COL_RUN_FIRST by the engine");
    if (bytes != rg_column_bytes(h, c))
        return rg_fail(RG_ERR_INVALID_ARG, "rg_load_column(%d): %llu bytes given, %llu expected", c,
                       (unsigned long long)bytes, (unsigned long long)rg_column_bytes(h, c));
    if (c == RG_COL_CFG) { // a configuration word may only name slots the engine has
        const u32 *w = static_cast<const u32 *>(src);
        for (u64 g = 0; g < h->G; g++) {
            const u32 x = w[g];
            if (RG_CFG_SELF(x) >= h->P || (RG_CFG_PRESENT(x) >> h->P) || (RG_CFG_INCOMING(x) >> h->P) ||
                (RG_CFG_OUTGOING(x) >> h->P) || RG_CFG_TRANSFEREE(x) > h->P)
                return rg_fail(RG_ERR_INVALID_ARG, "rg_load_column(CFG): group %llu: word %#x names a slot >= %u",
                               (unsigned long long)g, x, h->P);
        }
    }
    RG_ENTER(h);
    RG_HIP(hipMemcpyAsync(rg_col(h, c), src, bytes, hipMemcpyHostToDevice, h->stream));
    if (c == RG_COL_PFLAGS && h->ins_arena) // the FULL bit is the engine's: re-derive it from the windows
        hipLaunchKernelGGL(k_fix_ins_full, dim3((unsigned)((h->G + RG_BLOCK - 1) / RG_BLOCK)), dim3(RG_BLOCK), 0, h->stream,
                           h->st, h->ins, h->P);
    if (c == RG_COL_RUN_FIRST) // ... and the table's fill count
        hipLaunchKernelGGL(k_fix_run_count, dim3((unsigned)((h->G + RG_BLOCK - 1) / RG_BLOCK)), dim3(RG_BLOCK), 0, h->stream, h->st);
    if (c == RG_COL_PFLAGS || c == RG_COL_PEND_SNAP || c == RG_COL_PEND_RS) // ... and so is RG_PF_PENDING
        hipLaunchKernelGGL(k_fix_pending, dim3((unsigned)((h->G + RG_BLOCK - 1) / RG_BLOCK)), dim3(RG_BLOCK), 0, h->stream,
                           h->st, h->P);
    RG_HIP(hipStreamSynchronize(h->stream));
    if (c == RG_COL_COMMIT && h->pub) h->pub->local_lost = true;
    if (c == RG_COL_CFG) {
        h->host_cfg_valid = false;
        h->cls_stale = true;
        const u32 *w = static_cast<const u32 *>(src);
        bool any = false;
        for (u64 g = 0; g < h->G && !any; g++) any = (w[g] & RG_CFG_GROUP_COMMIT) != 0;
        h->any_group_commit = any;
    }
    return RG_OK;
}

extern "C" int rg_read_column(rg_engine *h, int c, void *dst, uint64_t bytes) {
    if (!h || !dst || c < 0 || c >= RG_COL_COUNT) return rg_fail(RG_ERR_INVALID_ARG, "rg_read_column: bad argument");
    if (bytes != rg_column_bytes(h, c))
        return rg_fail(RG_ERR_INVALID_ARG, "rg_read_column(%d): %llu bytes given, %llu expected", c,
                       (unsigned long long)bytes, (unsigned long long)rg_column_bytes(h, c));
    RG_ENTER(h);
    RG_HIP(hipMemcpyAsync(dst, rg_col(h, c), bytes, hipMemcpyDeviceToHost, h->stream));
    RG_HIP(hipStreamSynchronize(h->stream));
    return RG_OK;
}

extern "C" void *rg_column_ptr(rg_engine *h, int c) {
    if (!h || c < 0 || c >= RG_COL_COUNT) return nullptr;
    if (c == RG_COL_CFG) h->cls_off = true; // whoever holds this pointer can rewrite cfg words behind the engine's back
    return rg_col(h, c);
}

extern "C" int rg_checkpoint(rg_engine *h) {
    if (!h) return rg_fail(RG_ERR_INVALID_ARG, "rg_checkpoint: null engine");
    RG_ENTER(h);
    if (!h->ckpt) RG_HIP(hipMalloc(&h->ckpt, h->state_bytes));
    RG_HIP(hipMemcpyAsync(h->ckpt, h->arena, h->state_bytes, hipMemcpyDeviceToDevice, h->stream));
    h->ckpt_any_group_commit = h->any_group_commit;
    if (h->ins_arena) {
        if (!h->ins_ckpt) RG_HIP(hipMalloc(&h->ins_ckpt, h->ins_state_bytes));
        RG_HIP(hipMemcpyAsync(h->ins_ckpt, h->ins_arena, h->ins_state_bytes, hipMemcpyDeviceToDevice, h->stream));
        h->ckpt_send_ready = h->send_ready;
    }
    if (h->esz) {
        const size_t b = (size_t)h->G * h->ins.esz_w * 4;
        if (!h->esz_ckpt) RG_HIP(hipMalloc(&h->esz_ckpt, b));
        RG_HIP(hipMemcpyAsync(h->esz_ckpt, h->esz, b, hipMemcpyDeviceToDevice, h->stream));
    }
    return RG_OK;
}

extern "C" int rg_restore(rg_engine *h) {
    if (!h) return rg_fail(RG_ERR_INVALID_ARG, "rg_restore: null engine");
    if (!h->ckpt) return rg_fail(RG_ERR_STATE, "rg_restore: no checkpoint taken");
    RG_ENTER(h);
    RG_HIP(hipMemcpyAsync(h->arena, h->ckpt, h->state_bytes, hipMemcpyDeviceToDevice, h->stream));
    h->host_res_valid = false;
    if (h->pub) h->pub->local_lost = true; // the published advances no longer describe this commit column
    h->out_is_dense = true; // RG_COL_OUT is whatever it was at the checkpoint: the next sparse tick clears all of it
    h->host_cfg_valid = false; // RG_COL_CFG came back too: the mirror re-reads its copy
    h->cls_stale = true;
    if (h->ckpt_any_group_commit) h->any_group_commit = true; // ... and so may group-commit configurations
    if (h->ins_arena && h->ins_ckpt) {
        RG_HIP(hipMemcpyAsync(h->ins_arena, h->ins_ckpt, h->ins_state_bytes, hipMemcpyDeviceToDevice, h->stream));
        h->send_ready = h->ckpt_send_ready; // RG_COL_OUT is part of the state: the tick's requests are back too
    }
    if (h->esz && h->esz_ckpt)
        RG_HIP(hipMemcpyAsync(h->esz, h->esz_ckpt, (size_t)h->G * h->ins.esz_w * 4, hipMemcpyDeviceToDevice, h->stream));
    return RG_OK;
}

static unsigned rg_grid(u64 n, unsigned per_block) { return (unsigned)((n + per_block - 1) / per_block); }

extern "C" int rg_write_cells(rg_engine *h, const rg_cell_write *cells, uint64_t n) {
    if (!h || (!cells && n)) return rg_fail(RG_ERR_INVALID_ARG, "rg_write_cells: bad argument");
    if (n == 0) return RG_OK;
    RG_ENTER(h);
    if (n > h->d_cells_cap) { // engine-owned staging, grown geometrically (this call sits between ticks)
        if (h->d_cells) {
            RG_HIP(hipStreamSynchronize(h->stream));
            (void)hipFree(h->d_cells);
            h->d_cells = nullptr;
            h->d_cells_cap = 0;
        }
        u64 cap = 1024;
        while (cap < n) cap *= 2;
        RG_HIP(hipMalloc(&h->d_cells, cap * sizeof(rg_cell_write)));
        h->d_cells_cap = cap;
    }
    RG_HIP(hipMemcpyAsync(h->d_cells, cells, n * sizeof(rg_cell_write), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_write_cells, dim3(rg_grid(n, 256)), dim3(256), 0, h->stream, h->st, h->d_cells, (u64)n, h->P,
                       h->ins.meta);
    RG_HIP(hipStreamSynchronize(h->stream)); // the caller's array may be reused after return
    return RG_OK;
}

extern "C" int rg_read_groups(rg_engine *h, const uint64_t *groups, uint64_t n, rg_group_status *host_out) {
    if (!h || (n && (!groups || !host_out))) return rg_fail(RG_ERR_INVALID_ARG, "rg_read_groups: bad argument");
    if (n == 0) return RG_OK;
    RG_ENTER(h);
    char *d = nullptr; // [n x u64 group ids | n x rg_group_status]
    const size_t ids_b = rg_align(n * 8);
    RG_HIP(hipMalloc(&d, ids_b + n * sizeof(rg_group_status)));
    hipError_t e = hipMemcpyAsync(d, groups, n * 8, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_read_groups, dim3(rg_grid(n, 128)), dim3(128), 0, h->stream, h->st, (const u64 *)d, (u64)n, h->P,
                           (const u32 *)h->ins.meta, reinterpret_cast<rg_group_status *>(d + ids_b));
        e = hipMemcpyAsync(host_out, d + ids_b, n * sizeof(rg_group_status), hipMemcpyDeviceToHost, h->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return rg_fail(RG_ERR_NO_DEVICE, "rg_read_groups: %s", hipGetErrorString(e));
    for (u64 i = 0; i < n; i++)
        if (host_out[i].group == ~0ULL)
            return rg_fail(RG_ERR_INVALID_ARG, "rg_read_groups: group %llu does not exist (engine holds %llu)",
                           (unsigned long long)groups[i], (unsigned long long)h->G);
    return RG_OK;
}

extern "C" int rg_set_config(rg_engine *h, uint64_t group, uint32_t cfg_word) {
    if (!h || group >= h->G) return rg_fail(RG_ERR_INVALID_ARG, "rg_set_config: bad argument");
    if (RG_CFG_SELF(cfg_word) >= h->P || (RG_CFG_PRESENT(cfg_word) >> h->P) || (RG_CFG_INCOMING(cfg_word) >> h->P) ||
        (RG_CFG_OUTGOING(cfg_word) >> h->P) || RG_CFG_TRANSFEREE(cfg_word) > h->P)
        return rg_fail(RG_ERR_INVALID_ARG, "rg_set_config: cfg word %#x names a slot >= %u", cfg_word, h->P);
    RG_ENTER(h);
    RG_HIP(hipMemcpyAsync(h->st.cfg + group, &cfg_word, 4, hipMemcpyHostToDevice, h->stream));
    RG_HIP(hipStreamSynchronize(h->stream));
    if (cfg_word & RG_CFG_GROUP_COMMIT) h->any_group_commit = true; // (stays set: the GC kernel is a superset)
    if (h->host_cfg_valid) h->host_cfg[group] = cfg_word;
    // (a word that stays inside its block's class changes nothing: the class is an upper bound)
    if (!h->cls_stale && h->cls_on && rg_cfg_slots_named(cfg_word) > h->cls_host[group / RG_BLOCK]) h->cls_stale = true;
    return RG_OK;
}

// ------------------------------------------------------------------------------------------------
// the hot path
// ------------------------------------------------------------------------------------------------
static int rg_send_enqueue(rg_engine *h, uint64_t max_entries_per_msg, uint32_t flags, const u64 *list, u64 n,
                           const u32 *n_ptr);
// (RG_SEND_EFFECTS_ONLY / _APPEND_LIST / _REQUESTS_ONLY: rg_send.h)

// Device Inflights: a tick's result word carries free_to / free_first_one / left-Replicate effects for the rings. If
// the host skipped rg_send_appends, apply those effects (and nothing else: the send requests are dropped, which
// is what skipping the stage means) before the next tick overwrites RG_COL_OUT, so no window is left stale.
// Device Inflights and RG_OUT_HOST_HINT: the reference runs a deferred reject's send_append BEFORE the group's other sends of
// the step, so the group's send requests wait for rg_resolve_host_hints, which serves them (its Inflights effects -- free_to,
// free_first_one, the window resets -- are applied by the stage either way). A host that moved on without resolving would drop
// those requests for good and leave `next` / the windows behind the reference's: exact or loud -- every entry point that
// starts the next step refuses while such a group exists. Checked only after a tick that carried log terms (nothing else can
// raise the bit): one reduction over RG_COL_OUT and one synchronisation on that rare path, nothing on the others.
static int rg_require_hints_resolved(rg_engine *h, const char *who) {
    if (!h->ins_arena || !h->hint_check_due) return RG_OK;
    RG_HIP(hipMemsetAsync(h->d_counts, 0, 32, h->stream));
    const unsigned grid = rg_grid(h->G, RG_BLOCK) < 2048 ? rg_grid(h->G, RG_BLOCK) : 2048;
    hipLaunchKernelGGL(k_count_out, dim3(grid), dim3(RG_BLOCK), 0, h->stream, (const u32 *)h->st.out, h->G, h->d_counts);
    u64 c[3] = {0, 0, 0};
    RG_HIP(hipMemcpyAsync(c, h->d_counts, 24, hipMemcpyDeviceToHost, h->stream));
    RG_HIP(hipStreamSynchronize(h->stream));
    if (c[2])
        return rg_fail(RG_ERR_STATE, "%s: %llu group(s) still carry RG_OUT_HOST_HINT; with device Inflights (max_inflight > 0) "
                                     "rg_resolve_host_hints must answer every hint before the next step (rg_host_hints lists them)",
                       who, (unsigned long long)c[2]);
    h->hint_check_due = false;
    return RG_OK;
}

static int rg_settle_send(rg_engine *h) {
    int hrc = rg_require_hints_resolved(h, "next step");
    if (hrc) return hrc;
    if (!h->ins_arena || !h->send_ready) return RG_OK;
    const u64 *list = h->out_is_dense ? nullptr : h->res_list;
    const u64 n = h->out_is_dense ? h->G : h->last_sparse_n;
    int rc = rg_send_enqueue(h, 0, RG_SEND_EFFECTS_ONLY, list, n, nullptr);
    h->send_ready = false;
    h->send_bound = 0;
    return rc;
}

// Size classes of the shard, from RG_COL_CFG as it stands: per block of RG_BLOCK groups the number of slots its cfg words
// name, rounded up to the slot counts k_tick_classes has a body for (k_block_slots) -- one byte per block, kept in device
// memory for the kernel and copied to the host, where the engine decides whether the layout pays (some block below P) and
// rg_size_classes reports it as ranges. A control-path step (one small kernel, one copy of G / 64 bytes, one
// synchronisation) taken by the first dense tick after something wrote the column.
static int rg_refresh_classes(rg_engine *h) {
    h->cls_on = false;
    h->cls_stale = false;
    if (h->cls_off || h->P < 4) return RG_OK;
    const u64 nb = (h->G + RG_BLOCK - 1) / RG_BLOCK;
    if (!h->cls_need) {
        RG_HIP(hipMalloc(&h->cls_need, (nb + 3) & ~(u64)3));
        RG_HIP(hipMemsetAsync(h->cls_need, 0, (nb + 3) & ~(u64)3, h->stream));
    }
    hipLaunchKernelGGL(k_block_slots, dim3(rg_grid(nb, 256)), dim3(256), 0, h->stream, (const u32 *)h->st.cfg, h->G, nb, h->P, h->cls_need);
    h->cls_host.resize(nb);
    RG_HIP(hipMemcpyAsync(h->cls_host.data(), h->cls_need, nb, hipMemcpyDeviceToHost, h->stream));
    RG_HIP(hipStreamSynchronize(h->stream));
    for (u64 b = 0; b < nb && !h->cls_on; b++) h->cls_on = h->cls_host[b] < h->P;
    if (!h->cls_on) return RG_OK;
    // launch order (RgClasses::order): the ranges of equal blocks dealt out proportionally -- block i of a range of n blocks sorts
    // by (i + 1/2) / n, ties by block index. RG_CFGF_CLASS_BLOCK_ORDER keeps block order (measurement).
    if (nb >= (1ull << 28)) { // (the word holds 28 bits of block index)
        h->cls_on = false;
        return RG_OK;
    }
    std::vector<std::pair<double, u32>> key(nb);
    const bool deal = !h->cls_block_order;
    for (u64 b = 0; b < nb;) {
        u64 e = b + 1;
        while (e < nb && h->cls_host[e] == h->cls_host[b]) e++;
        for (u64 i = b; i < e; i++) key[i] = {deal ? ((double)(i - b) + 0.5) / (double)(e - b) : 0.0, (u32)i};
        b = e;
    }
    std::stable_sort(key.begin(), key.end(), [](const std::pair<double, u32> &x, const std::pair<double, u32> &y) { return x.first < y.first; });
    std::vector<u32> order(nb);
    for (u64 w = 0; w < nb; w++) order[w] = key[w].second | ((u32)h->cls_host[key[w].second] << 28);
    if (!h->cls_order) RG_HIP(hipMalloc(&h->cls_order, nb * 4));
    RG_HIP(hipMemcpyAsync(h->cls_order, order.data(), nb * 4, hipMemcpyHostToDevice, h->stream));
    RG_HIP(hipStreamSynchronize(h->stream)); // (`order` is a local)
    return RG_OK;
}

// `send` != NULL: the tick and its send stage as ONE launch (k_tick_send; rg_tick_send / rg_tick_device_send)
struct RgSendReq {
    u64 max_entries;
    u32 flags;
};
static int rg_tick_impl(rg_engine *h, const RgMsgs &ms, const RgSendReq *send = nullptr) {
    int src = rg_settle_send(h);
    if (src) return src;
    if (send) {
        const bool nts = h->nt_all && !h->any_group_commit && rg_ix32(h->st, h->P);
        switch (h->P) {
        case 1: rg_launch_tick_send_t<1>(h->stream, h->st, ms, h->any_group_commit, h->ins, send->max_entries, send->flags, h->send_cols, nts); break;
        case 2: rg_launch_tick_send_t<2>(h->stream, h->st, ms, h->any_group_commit, h->ins, send->max_entries, send->flags, h->send_cols, nts); break;
        case 3: rg_launch_tick_send_t<3>(h->stream, h->st, ms, h->any_group_commit, h->ins, send->max_entries, send->flags, h->send_cols, nts); break;
        case 4: rg_launch_tick_send_t<4>(h->stream, h->st, ms, h->any_group_commit, h->ins, send->max_entries, send->flags, h->send_cols, nts); break;
        case 5: rg_launch_tick_send_t<5>(h->stream, h->st, ms, h->any_group_commit, h->ins, send->max_entries, send->flags, h->send_cols, nts); break;
        case 6: rg_launch_tick_send_t<6>(h->stream, h->st, ms, h->any_group_commit, h->ins, send->max_entries, send->flags, h->send_cols, nts); break;
        case 7: rg_launch_tick_send_t<7>(h->stream, h->st, ms, h->any_group_commit, h->ins, send->max_entries, send->flags, h->send_cols, nts); break;
        default: rg_launch_tick_send_t<8>(h->stream, h->st, ms, h->any_group_commit, h->ins, send->max_entries, send->flags, h->send_cols, nts); break;
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return rg_fail(RG_ERR_NO_DEVICE, "tick + send stage launch failed: %s", hipGetErrorString(e));
        h->dev.last_tick_kernel = RG_KERNEL_TICK_SEND;
        h->dev.last_tick_streaming = nts ? 2u : 1u; // (k_tick_send streams its message columns at any size)
        h->tick_launches++;
        h->ticked = true;
        h->out_is_dense = true;
        h->host_res_valid = false;
        // what rg_send_appends leaves behind a dense stage
        h->stage_max_entries = send->max_entries;
        h->stage_flags = send->flags;
        h->send_ready = false;
        h->send_bound = h->G * h->P;
        h->send_cols_fresh = true;
        h->send_last_dense = true;
        h->host_items_valid = false;
        return RG_OK;
    }
    // one translation unit per slot count (tick_inst.hip, -DRG_P=n); the group-commit kernel is only
    // needed when some group has ProgressTracker.group_commit set
    const u32 variant = ((h->cfg.variant == RG_VARIANT_LDS || h->cfg.variant == RG_VARIANT_LDS_DMA || h->cfg.variant == RG_VARIANT_COMPACT)
                             ? h->cfg.variant : RG_VARIANT_LANE) | (h->nt_msgs ? RG_VARIANT_NT_MSGS : 0u) | (h->nt_all ? RG_VARIANT_NT_ALL : 0u);
    // a class-placed shard (replica sets of different sizes in contiguous ranges): ONE launch whose blocks run the tick
    // instantiated for the slots their groups have (k_tick_classes). Lane variant, no group commit, 32-bit cell offsets.
    if ((variant & ~(RG_VARIANT_NT_MSGS | RG_VARIANT_NT_ALL)) == RG_VARIANT_LANE && !h->any_group_commit && h->P >= 4 && !h->cls_off && rg_ix32(h->st, h->P)) {
        if (h->cls_stale) {
            // (the refresh synchronises: not inside a stream capture -- a captured tick of a stale engine takes the plain kernel)
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(h->stream, &cs) != hipSuccess) cs = hipStreamCaptureStatusNone;
            if (cs == hipStreamCaptureStatusNone) {
                const int crc = rg_refresh_classes(h);
                if (crc) return crc;
            }
        }
        if (h->cls_on && !h->cls_stale) {
            RgClasses cls;
            cls.order = h->cls_order;
            switch (h->P) {
            case 4: rg_launch_tick_classes_t<4>(h->stream, h->st, ms, h->nt_all ? 2 : h->nt_msgs ? 1 : 0, cls); break;
            case 5: rg_launch_tick_classes_t<5>(h->stream, h->st, ms, h->nt_all ? 2 : h->nt_msgs ? 1 : 0, cls); break;
            case 6: rg_launch_tick_classes_t<6>(h->stream, h->st, ms, h->nt_all ? 2 : h->nt_msgs ? 1 : 0, cls); break;
            case 7: rg_launch_tick_classes_t<7>(h->stream, h->st, ms, h->nt_all ? 2 : h->nt_msgs ? 1 : 0, cls); break;
            default: rg_launch_tick_classes_t<8>(h->stream, h->st, ms, h->nt_all ? 2 : h->nt_msgs ? 1 : 0, cls); break;
            }
            hipError_t ce = hipGetLastError();
            if (ce != hipSuccess) return rg_fail(RG_ERR_NO_DEVICE, "tick launch failed: %s", hipGetErrorString(ce));
            h->dev.last_tick_kernel = RG_KERNEL_CLASSES;
            h->dev.last_tick_streaming = h->nt_all ? 2u : h->nt_msgs ? 1u : 0u;
            h->tick_launches++;
            h->ticked = true;
            h->out_is_dense = true;
            h->send_ready = true;
            h->host_res_valid = false;
            return RG_OK;
        }
    }
    u32 kernel = (variant & ~(RG_VARIANT_NT_MSGS | RG_VARIANT_NT_ALL)) == RG_VARIANT_LANE ? RG_KERNEL_LANE
                 : (variant & 0xffu) == RG_VARIANT_COMPACT                                 ? RG_KERNEL_COMPACT
                                                                                           : RG_KERNEL_LDS;
    if (h->nt_resident && kernel == RG_KERNEL_LANE && !h->any_group_commit && rg_ix32(h->st, h->P)) {
        kernel = RG_KERNEL_SPLIT;
        switch (h->P) {
        case 1: rg_launch_tick_split_t<1>(h->stream, h->st, ms, h->nt_resident); break;
        case 2: rg_launch_tick_split_t<2>(h->stream, h->st, ms, h->nt_resident); break;
        case 3: rg_launch_tick_split_t<3>(h->stream, h->st, ms, h->nt_resident); break;
        case 4: rg_launch_tick_split_t<4>(h->stream, h->st, ms, h->nt_resident); break;
        case 5: rg_launch_tick_split_t<5>(h->stream, h->st, ms, h->nt_resident); break;
        case 6: rg_launch_tick_split_t<6>(h->stream, h->st, ms, h->nt_resident); break;
        case 7: rg_launch_tick_split_t<7>(h->stream, h->st, ms, h->nt_resident); break;
        default: rg_launch_tick_split_t<8>(h->stream, h->st, ms, h->nt_resident); break;
        }
    } else
    switch (h->P) {
    case 1: rg_launch_tick_t<1>(h->stream, h->st, ms, variant, h->any_group_commit); break;
    case 2: rg_launch_tick_t<2>(h->stream, h->st, ms, variant, h->any_group_commit); break;
    case 3: rg_launch_tick_t<3>(h->stream, h->st, ms, variant, h->any_group_commit); break;
    case 4: rg_launch_tick_t<4>(h->stream, h->st, ms, variant, h->any_group_commit); break;
    case 5: rg_launch_tick_t<5>(h->stream, h->st, ms, variant, h->any_group_commit); break;
    case 6: rg_launch_tick_t<6>(h->stream, h->st, ms, variant, h->any_group_commit); break;
    case 7: rg_launch_tick_t<7>(h->stream, h->st, ms, variant, h->any_group_commit); break;
    default: rg_launch_tick_t<8>(h->stream, h->st, ms, variant, h->any_group_commit); break;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return rg_fail(RG_ERR_NO_DEVICE, "tick launch failed: %s", hipGetErrorString(e));
    h->dev.last_tick_kernel = kernel;
    // (the group-commit instantiation and the LDS / compact variants have no streaming twins: rg_launch_tick_gc)
    h->dev.last_tick_streaming = kernel == RG_KERNEL_SPLIT ? 2u : (kernel == RG_KERNEL_LANE && !h->any_group_commit) ? (h->nt_all ? 2u : h->nt_msgs ? 1u : 0u) : 0u;
    h->tick_launches++;
    h->ticked = true;
    h->out_is_dense = true;
    h->send_ready = true;
    h->host_res_valid = false;
    return RG_OK;
}

extern "C" int rg_size_classes(rg_engine *h, rg_size_class *out, uint32_t cap, uint32_t *n) {
    if (!h || !n || (cap && !out)) return rg_fail(RG_ERR_INVALID_ARG, "rg_size_classes: bad argument");
    *n = 0;
    RG_ENTER(h);
    if (h->cls_stale) {
        const int rc = rg_refresh_classes(h);
        if (rc) return rc;
    }
    const bool usable = (h->cfg.variant != RG_VARIANT_LDS && h->cfg.variant != RG_VARIANT_LDS_DMA && h->cfg.variant != RG_VARIANT_COMPACT) &&
                        !h->any_group_commit && !h->cls_off && rg_ix32(h->st, h->P);
    if (!usable) return RG_OK;
    if (!h->cls_on) return RG_OK;
    u32 k = 0; // run-length encode the per-block bytes
    const u64 nb = h->cls_host.size();
    for (u64 b = 0; b < nb;) {
        u64 e = b + 1;
        while (e < nb && h->cls_host[e] == h->cls_host[b]) e++;
        if (k < cap) {
            out[k].first_group = b * RG_BLOCK;
            out[k].n_groups = rg_min(e * RG_BLOCK, h->G) - b * RG_BLOCK;
            out[k].n_slots = h->cls_host[b];
            out[k].reserved = 0;
        }
        k++;
        b = e;
    }
    *n = k;
    return RG_OK;
}

static int rg_send_check(rg_engine *h, uint32_t flags, const char *who) {
    if (!h->ins_arena) return rg_fail(RG_ERR_STATE, "%s: engine created with max_inflight = 0 (Inflights are the host's)", who);
    if (flags & ~(RG_SEND_SKIP_BCAST_COMMIT | RG_SEND_BYTES)) return rg_fail(RG_ERR_INVALID_ARG, "%s: unknown flags %#x", who, flags);
    if ((flags & RG_SEND_BYTES) && !h->esz) return rg_fail(RG_ERR_STATE, "%s: RG_SEND_BYTES needs the entry sizes (rg_log_sizes_enable)", who);
    return RG_OK;
}

static int rg_tick_device_impl(rg_engine *h, const rg_msgs *m, const RgSendReq *send) {
    RG_ENTER(h);
    {   // (device Inflights: nothing of the next step is enqueued while a host hint of the last one is unanswered)
        const int hrc__ = rg_require_hints_resolved(h, "rg_tick_device");
        if (hrc__) return hrc__;
    }
    RgMsgs ms;
    ms.mi = (const u64 *)m->m_index;
    ms.mc = (const u64 *)m->m_commit;
    ms.mh = m->m_hint ? (const u64 *)m->m_hint : h->zero_col;
    ms.mrs = m->m_rs ? (const u64 *)m->m_rs : h->zero_col;
    ms.mlt = m->m_logterm ? (const u64 *)m->m_logterm : h->zero_col;
    ms.mflags = (const u64 *)m->m_flags;
    ms.mhr = ms.mh;
    if (m->m_logterm) { // this tick may carry log terms: resolve the flagged hints first
        hipLaunchKernelGGL(k_resolve_hints, dim3(rg_grid(h->G, RG_BLOCK)), dim3(RG_BLOCK), 0, h->stream, h->st, ms, h->P,
                           h->rhint);
        ms.mhr = h->rhint;
    }
    const int trc = rg_tick_impl(h, ms, send);
    if (trc == RG_OK && m->m_logterm) h->hint_check_due = true; // (rg_require_hints_resolved: this tick can have raised RG_OUT_HOST_HINT)
    return trc;
}

extern "C" int rg_tick_device(rg_engine *h, const rg_msgs *m) {
    if (!h || !m || !m->m_index || !m->m_commit || !m->m_flags)
        return rg_fail(RG_ERR_INVALID_ARG, "rg_tick_device: m_index, m_commit and m_flags are required");
    return rg_tick_device_impl(h, m, nullptr);
}

extern "C" int rg_tick_device_send(rg_engine *h, const rg_msgs *m, uint64_t max_entries_per_msg, uint32_t flags) {
    if (!h || !m || !m->m_index || !m->m_commit || !m->m_flags)
        return rg_fail(RG_ERR_INVALID_ARG, "rg_tick_device_send: m_index, m_commit and m_flags are required");
    int rc = rg_send_check(h, flags, "rg_tick_device_send");
    if (rc) return rc;
    const RgSendReq send = {(u64)max_entries_per_msg, (u32)flags};
    return rg_tick_device_impl(h, m, &send);
}

// One fused launch over ticks [t0, t0 + n) of the caller's array (none of them carries Message.log_term).
static int rg_fused_run(rg_engine *h, const rg_msgs *m, u32 t0, u32 n, uint32_t *dev_out_t, uint64_t *dev_commit_t) {
    RgFused fm;
    memset(&fm, 0, sizeof(fm));
    for (u32 i = 0; i < n; i++) {
        const rg_msgs &x = m[t0 + i];
        fm.m[i].mi = (const u64 *)x.m_index;
        fm.m[i].mc = (const u64 *)x.m_commit;
        fm.m[i].mh = x.m_hint ? (const u64 *)x.m_hint : h->zero_col;
        fm.m[i].mrs = x.m_rs ? (const u64 *)x.m_rs : h->zero_col;
        fm.m[i].mlt = h->zero_col;
        fm.m[i].mhr = fm.m[i].mh;
        fm.m[i].mflags = (const u64 *)x.m_flags;
    }
    fm.out_t = dev_out_t + (size_t)t0 * h->G;
    fm.commit_t = dev_commit_t ? (u64 *)dev_commit_t + (size_t)t0 * h->G : nullptr;
    fm.n_ticks = n;
    switch (h->P) {
    case 1: rg_launch_tick_fused_t<1>(h->stream, h->st, fm, h->any_group_commit); break;
    case 2: rg_launch_tick_fused_t<2>(h->stream, h->st, fm, h->any_group_commit); break;
    case 3: rg_launch_tick_fused_t<3>(h->stream, h->st, fm, h->any_group_commit); break;
    case 4: rg_launch_tick_fused_t<4>(h->stream, h->st, fm, h->any_group_commit); break;
    case 5: rg_launch_tick_fused_t<5>(h->stream, h->st, fm, h->any_group_commit); break;
    case 6: rg_launch_tick_fused_t<6>(h->stream, h->st, fm, h->any_group_commit); break;
    case 7: rg_launch_tick_fused_t<7>(h->stream, h->st, fm, h->any_group_commit); break;
    default: rg_launch_tick_fused_t<8>(h->stream, h->st, fm, h->any_group_commit); break;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return rg_fail(RG_ERR_NO_DEVICE, "fused tick launch failed: %s", hipGetErrorString(e));
    return RG_OK;
}

extern "C" int rg_tick_device_fused(rg_engine *h, const rg_msgs *m, uint32_t n_ticks, uint32_t *dev_out_t,
                                    uint64_t *dev_commit_t) {
    if (!h || !m || !dev_out_t || n_ticks == 0 || n_ticks > RG_MAX_FUSE)
        return rg_fail(RG_ERR_INVALID_ARG, "rg_tick_device_fused: need 1..%d ticks and an out buffer", RG_MAX_FUSE);
    if (h->ins_arena)
        return rg_fail(RG_ERR_STATE, "rg_tick_device_fused: engines with device Inflights (max_inflight > 0) need "
                                     "rg_send_appends after every tick; fused launches are not available");
    for (u32 t = 0; t < n_ticks; t++)
        if (!m[t].m_index || !m[t].m_commit || !m[t].m_flags)
            return rg_fail(RG_ERR_INVALID_ARG, "rg_tick_device_fused: tick %u lacks m_index/m_commit/m_flags", t);
    RG_ENTER(h);
    // A tick that carries Message.log_term needs find_conflict_by_term against the log as it stands BEFORE that tick
    // (last_index, the leader's range and the term table change from tick to tick): such a tick runs as a single-tick
    // launch behind its pre-pass, between the fused launches of the ticks around it -- same results, in the caller's arrays.
    u32 t = 0;
    while (t < n_ticks) {
        u32 e = t;
        while (e < n_ticks && !m[e].m_logterm) e++;
        if (e > t) {
            int rc = rg_fused_run(h, m, t, e - t, dev_out_t, dev_commit_t);
            if (rc) return rc;
        }
        if (e < n_ticks) {
            RgMsgs ms;
            ms.mi = (const u64 *)m[e].m_index;
            ms.mc = (const u64 *)m[e].m_commit;
            ms.mh = m[e].m_hint ? (const u64 *)m[e].m_hint : h->zero_col;
            ms.mrs = m[e].m_rs ? (const u64 *)m[e].m_rs : h->zero_col;
            ms.mlt = (const u64 *)m[e].m_logterm;
            ms.mflags = (const u64 *)m[e].m_flags;
            hipLaunchKernelGGL(k_resolve_hints, dim3(rg_grid(h->G, RG_BLOCK)), dim3(RG_BLOCK), 0, h->stream, h->st, ms, h->P,
                               h->rhint);
            ms.mhr = h->rhint;
            int rc = rg_tick_impl(h, ms);
            if (rc) return rc;
            RG_HIP(hipMemcpyAsync(dev_out_t + (size_t)e * h->G, h->st.out, h->G * 4, hipMemcpyDeviceToDevice, h->stream));
            if (dev_commit_t)
                RG_HIP(hipMemcpyAsync(dev_commit_t + (size_t)e * h->G, h->st.commit, h->G * 8, hipMemcpyDeviceToDevice, h->stream));
            e++;
        }
        t = e;
    }
    h->ticked = true;
    h->host_res_valid = false;
    h->out_is_dense = true;
    return RG_OK;
}

static int rg_ensure_msg_arena(rg_engine *h) {
    if (h->msg_arena) return RG_OK;
    const size_t col = rg_align((size_t)h->P * h->stride * 8);
    RG_HIP(hipMalloc(&h->msg_arena, 5 * col + rg_align(h->stride * 8)));
    RG_HIP(hipMemsetAsync(h->msg_arena, 0, 5 * col + rg_align(h->stride * 8), h->stream));
    h->staged.mi = (u64 *)(h->msg_arena);
    h->staged.mc = (u64 *)(h->msg_arena + col);
    h->staged.mh = (u64 *)(h->msg_arena + 2 * col);
    h->staged.mrs = (u64 *)(h->msg_arena + 3 * col);
    h->staged.mlt = (u64 *)(h->msg_arena + 4 * col);
    h->staged.mflags = (u64 *)(h->msg_arena + 5 * col);
    return RG_OK;
}

static int rg_tick_host_impl(rg_engine *h, const rg_msgs *m, const RgSendReq *send) {
    RG_ENTER(h);
    {   // (device Inflights: nothing of the next step is enqueued while a host hint of the last one is unanswered)
        const int hrc__ = rg_require_hints_resolved(h, "rg_tick");
        if (hrc__) return hrc__;
    }
    int rc = rg_ensure_msg_arena(h);
    if (rc) return rc;
    const size_t colb = (size_t)h->P * h->stride * 8;
    RG_HIP(hipMemcpyAsync((void *)h->staged.mi, m->m_index, colb, hipMemcpyHostToDevice, h->stream));
    RG_HIP(hipMemcpyAsync((void *)h->staged.mc, m->m_commit, colb, hipMemcpyHostToDevice, h->stream));
    RgMsgs ms = h->staged;
    if (m->m_hint) RG_HIP(hipMemcpyAsync((void *)h->staged.mh, m->m_hint, colb, hipMemcpyHostToDevice, h->stream));
    else ms.mh = h->zero_col;
    if (m->m_rs) RG_HIP(hipMemcpyAsync((void *)h->staged.mrs, m->m_rs, colb, hipMemcpyHostToDevice, h->stream));
    else ms.mrs = h->zero_col;
    ms.mhr = ms.mh;
    if (m->m_logterm) RG_HIP(hipMemcpyAsync((void *)h->staged.mlt, m->m_logterm, colb, hipMemcpyHostToDevice, h->stream));
    else ms.mlt = h->zero_col;
    RG_HIP(hipMemcpyAsync((void *)h->staged.mflags, m->m_flags, h->G * 8, hipMemcpyHostToDevice, h->stream));
    if (m->m_logterm) { // pre-pass (after ALL message columns are on the device): find_conflict_by_term
        hipLaunchKernelGGL(k_resolve_hints, dim3(rg_grid(h->G, RG_BLOCK)), dim3(RG_BLOCK), 0, h->stream, h->st, ms, h->P,
                           h->rhint);
        ms.mhr = h->rhint;
    }
    rc = rg_tick_impl(h, ms, send);
    if (rc) return rc;
    if (m->m_logterm) h->hint_check_due = true;
    // the engine-owned message columns must read "no events" outside a tick (sparse-path invariant)
    RG_HIP(hipMemsetAsync((void *)h->staged.mflags, 0, h->stride * 8, h->stream));
    RG_HIP(hipStreamSynchronize(h->stream)); // caller-owned host buffers may be reused after return
    return RG_OK;
}

extern "C" int rg_tick(rg_engine *h, const rg_msgs *m) {
    if (!h || !m || !m->m_index || !m->m_commit || !m->m_flags)
        return rg_fail(RG_ERR_INVALID_ARG, "rg_tick: m_index, m_commit and m_flags are required");
    return rg_tick_host_impl(h, m, nullptr);
}

extern "C" int rg_tick_send(rg_engine *h, const rg_msgs *m, uint64_t max_entries_per_msg, uint32_t flags) {
    if (!h || !m || !m->m_index || !m->m_commit || !m->m_flags)
        return rg_fail(RG_ERR_INVALID_ARG, "rg_tick_send: m_index, m_commit and m_flags are required");
    int rc = rg_send_check(h, flags, "rg_tick_send");
    if (rc) return rc;
    const RgSendReq send = {(u64)max_entries_per_msg, (u32)flags};
    return rg_tick_host_impl(h, m, &send);
}

static RgIngest rg_ingest_args(rg_engine *h, const rg_wire_msg *rec, u64 n, const RgClear &clr) {
    RgIngest a;
    a.rec = rec;
    a.n = n;
    a.G = h->G;
    a.stride = h->stride;
    a.P = h->P;
    a.mi = (u64 *)h->staged.mi;
    a.mc = (u64 *)h->staged.mc;
    a.mh = (u64 *)h->staged.mh;
    a.mrs = (u64 *)h->staged.mrs;
    a.mlt = (u64 *)h->staged.mlt;
    a.mflags32 = (u32 *)h->staged.mflags;
    a.gmark = h->gmark;
    a.epoch = h->epoch;
    a.list = h->list;
    a.counters = h->counters;
    a.clr = clr;
    return a;
}

// Two {touched groups, dropped records} counter pairs take turns: a sparse tick uses one, the ingest kernel of the same
// window resets the other for the tick after it, rg_ctr_flip switches -- no memset command per tick.
static u32 *rg_ctr_other(rg_engine *h) { return h->counters == h->counters_base ? h->counters_base + 2 : h->counters_base; }
static void rg_ctr_flip(rg_engine *h) { h->counters = rg_ctr_other(h); }

static int rg_ensure_sparse(rg_engine *h) {
    if (h->sparse_arena) return RG_OK;
    int rc = rg_ensure_msg_arena(h);
    if (rc) return rc;
    const size_t G = h->stride;
    const size_t o_gmark = 0, o_list = rg_align(G * 4), o_rl = o_list + rg_align(G * 8), o_rc = o_rl + rg_align(G * 8);
    const size_t o_ro = o_rc + rg_align(G * 8), o_cnt = o_ro + rg_align(G * 4), total = o_cnt + 256;
    RG_HIP(hipMalloc(&h->sparse_arena, total));
    RG_HIP(hipMemsetAsync(h->sparse_arena, 0, total, h->stream));
    h->gmark = (u32 *)(h->sparse_arena + o_gmark);
    h->list = (u64 *)(h->sparse_arena + o_list);
    h->res_list = (u64 *)(h->sparse_arena + o_rl);
    h->res_commit = (u64 *)(h->sparse_arena + o_rc);
    h->res_out = (u32 *)(h->sparse_arena + o_ro);
    h->counters_base = (u32 *)(h->sparse_arena + o_cnt);
    h->counters = h->counters_base;
    return RG_OK;
}

extern "C" int rg_ingest(rg_engine *h, const rg_wire_msg *records, uint64_t n, uint64_t *n_duplicates) {
    if (!h || (!records && n)) return rg_fail(RG_ERR_INVALID_ARG, "rg_ingest: bad argument");
    if (n_duplicates) *n_duplicates = 0;
    if (n == 0) return RG_OK;
    RG_ENTER(h);
    {   // (device Inflights: nothing of the next step is enqueued while a host hint of the last one is unanswered)
        const int hrc__ = rg_require_hints_resolved(h, "rg_ingest");
        if (hrc__) return hrc__;
    }
    int rc = rg_ensure_sparse(h);
    if (rc) return rc;
    if (n > h->d_records_cap) {
        if (h->d_records) {
            RG_HIP(hipStreamSynchronize(h->stream));
            (void)hipFree(h->d_records);
            h->d_records = nullptr;
        }
        u64 cap = h->d_records_cap ? h->d_records_cap : 4096;
        while (cap < n) cap *= 2;
        RG_HIP(hipMalloc(&h->d_records, (cap + RG_INGEST_BLOCK) * sizeof(rg_wire_msg)));
        h->d_records_cap = cap;
    }
    u32 dup0 = 0; // duplicates so far in this tick window (device ingests included)
    RG_HIP(hipMemcpyAsync(&dup0, h->counters + 1, 4, hipMemcpyDeviceToHost, h->stream));
    RG_HIP(hipMemcpyAsync(h->d_records, records, n * sizeof(rg_wire_msg), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_ingest, dim3(rg_grid(n, RG_INGEST_BLOCK)), dim3(RG_INGEST_BLOCK), 0, h->stream,
                       rg_ingest_args(h, h->d_records, n, RgClear{nullptr, nullptr, 0u, rg_ctr_other(h)}));
    u32 dup = 0;
    RG_HIP(hipMemcpyAsync(&dup, h->counters + 1, 4, hipMemcpyDeviceToHost, h->stream));
    RG_HIP(hipStreamSynchronize(h->stream)); // the caller's record array may be reused after return
    if (n_duplicates) *n_duplicates = dup - dup0; // dup0 was read before the kernel ran (stream order)
    h->ingested_upper += n;
    return RG_OK;
}

extern "C" int rg_ingest_device(rg_engine *h, const rg_wire_msg *dev_records, uint64_t n) {
    if (!h || (!dev_records && n)) return rg_fail(RG_ERR_INVALID_ARG, "rg_ingest_device: bad argument");
    if (n == 0) return RG_OK;
    RG_ENTER(h);
    {   // (device Inflights: nothing of the next step is enqueued while a host hint of the last one is unanswered)
        const int hrc__ = rg_require_hints_resolved(h, "rg_ingest_device");
        if (hrc__) return hrc__;
    }
    int rc = rg_ensure_sparse(h);
    if (rc) return rc;
    hipLaunchKernelGGL(k_ingest, dim3(rg_grid(n, RG_INGEST_BLOCK)), dim3(RG_INGEST_BLOCK), 0, h->stream,
                       rg_ingest_args(h, dev_records, n, RgClear{nullptr, nullptr, 0u, rg_ctr_other(h)}));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return rg_fail(RG_ERR_NO_DEVICE, "rg_ingest_device: %s", hipGetErrorString(e));
    h->ingested_upper += n;
    return RG_OK;
}

extern "C" int rg_ingested_duplicates(rg_engine *h, uint64_t *n_duplicates) {
    if (!h || !n_duplicates) return rg_fail(RG_ERR_INVALID_ARG, "rg_ingested_duplicates: bad argument");
    *n_duplicates = 0;
    if (!h->sparse_arena) return RG_OK;
    RG_ENTER(h);
    u32 dup = 0;
    RG_HIP(hipMemcpyAsync(&dup, h->counters + 1, 4, hipMemcpyDeviceToHost, h->stream));
    RG_HIP(hipStreamSynchronize(h->stream));
    *n_duplicates = dup;
    return RG_OK;
}

// Everything of a sparse tick that needs no host round trip: clear the previous results, resolve hints, tick the
// listed groups, gather their results (also into `packed` when given). `upper` bounds the list length.
static int rg_sparse_enqueue(rg_engine *h, u64 upper, char *packed, bool any_logterm, bool out_cleared = false,
                             const RgIngest *one_launch = nullptr, const RgSmallSend *small_send = nullptr) {
    int src = rg_settle_send(h); // (walks the PREVIOUS tick's result list, before it is cleared below)
    if (src) return src;
    // RG_COL_OUT must hold zeros for every group this tick does not touch
    if (out_cleared) {
        // (the ingest kernel of this flush has done it)
    } else if (h->out_is_dense) {
        RG_HIP(hipMemsetAsync(h->st.out, 0, h->stride * 4, h->stream));
    } else if (h->last_sparse_n) {
        hipLaunchKernelGGL(k_clear_out, dim3(rg_grid(h->last_sparse_n, 256)), dim3(256), 0, h->stream, h->res_list,
                           h->last_sparse_n, h->st.out);
    }
    h->out_is_dense = false;
    h->last_sparse_n = 0;
    h->host_res_valid = false;
    if (!upper) return RG_OK;
    if (any_logterm) h->hint_check_due = true; // (rg_require_hints_resolved)
    RgMsgs ms = h->staged;
    ms.mhr = ms.mh;
    u64 *mf = (u64 *)h->staged.mflags;
    RgListOut lo; // the tick gathers its own results (one launch less than a separate gather kernel)
    lo.rl = h->res_list;
    lo.rc = h->res_commit;
    lo.ro = h->res_out;
    lo.packed = packed;
    if (one_launch && small_send) { // ... and the touched groups' send stage as well (rg_flush_send)
        if (any_logterm) ms.mhr = h->rhint;
        switch (h->P) {
        case 1: rg_launch_flush_small_send_t<1>(h->stream, h->st, ms, h->any_group_commit, *one_launch, h->rhint, mf, lo, *small_send); break;
        case 2: rg_launch_flush_small_send_t<2>(h->stream, h->st, ms, h->any_group_commit, *one_launch, h->rhint, mf, lo, *small_send); break;
        case 3: rg_launch_flush_small_send_t<3>(h->stream, h->st, ms, h->any_group_commit, *one_launch, h->rhint, mf, lo, *small_send); break;
        case 4: rg_launch_flush_small_send_t<4>(h->stream, h->st, ms, h->any_group_commit, *one_launch, h->rhint, mf, lo, *small_send); break;
        case 5: rg_launch_flush_small_send_t<5>(h->stream, h->st, ms, h->any_group_commit, *one_launch, h->rhint, mf, lo, *small_send); break;
        case 6: rg_launch_flush_small_send_t<6>(h->stream, h->st, ms, h->any_group_commit, *one_launch, h->rhint, mf, lo, *small_send); break;
        case 7: rg_launch_flush_small_send_t<7>(h->stream, h->st, ms, h->any_group_commit, *one_launch, h->rhint, mf, lo, *small_send); break;
        default: rg_launch_flush_small_send_t<8>(h->stream, h->st, ms, h->any_group_commit, *one_launch, h->rhint, mf, lo, *small_send); break;
        }
        hipError_t e1 = hipGetLastError();
        if (e1 != hipSuccess) return rg_fail(RG_ERR_NO_DEVICE, "sparse tick + send stage: launch failed: %s", hipGetErrorString(e1));
        h->tick_launches++;
        return RG_OK;
    }
    if (one_launch) { // <= 256 records: ingest, hint resolution, tick and results in ONE single-workgroup launch
        if (any_logterm) ms.mhr = h->rhint;
        switch (h->P) {
        case 1: rg_launch_flush_small_t<1>(h->stream, h->st, ms, h->any_group_commit, *one_launch, h->rhint, mf, lo); break;
        case 2: rg_launch_flush_small_t<2>(h->stream, h->st, ms, h->any_group_commit, *one_launch, h->rhint, mf, lo); break;
        case 3: rg_launch_flush_small_t<3>(h->stream, h->st, ms, h->any_group_commit, *one_launch, h->rhint, mf, lo); break;
        case 4: rg_launch_flush_small_t<4>(h->stream, h->st, ms, h->any_group_commit, *one_launch, h->rhint, mf, lo); break;
        case 5: rg_launch_flush_small_t<5>(h->stream, h->st, ms, h->any_group_commit, *one_launch, h->rhint, mf, lo); break;
        case 6: rg_launch_flush_small_t<6>(h->stream, h->st, ms, h->any_group_commit, *one_launch, h->rhint, mf, lo); break;
        case 7: rg_launch_flush_small_t<7>(h->stream, h->st, ms, h->any_group_commit, *one_launch, h->rhint, mf, lo); break;
        default: rg_launch_flush_small_t<8>(h->stream, h->st, ms, h->any_group_commit, *one_launch, h->rhint, mf, lo); break;
        }
        hipError_t e1 = hipGetLastError();
        if (e1 != hipSuccess) return rg_fail(RG_ERR_NO_DEVICE, "sparse tick: launch failed: %s", hipGetErrorString(e1));
        h->tick_launches++;
        return RG_OK;
    }
    if (any_logterm) { // records may carry log terms: resolve the touched groups' flagged hints first
        ms.mhr = h->rhint;
        hipLaunchKernelGGL(k_resolve_hints_list, dim3(rg_grid(upper, RG_BLOCK)), dim3(RG_BLOCK), 0, h->stream, h->st, ms,
                           h->P, h->rhint, (const u64 *)h->list, (const u32 *)h->counters);
    }
    switch (h->P) {
    case 1: rg_launch_tick_list_t<1>(h->stream, h->st, ms, h->any_group_commit, h->list, h->counters, upper, mf, lo); break;
    case 2: rg_launch_tick_list_t<2>(h->stream, h->st, ms, h->any_group_commit, h->list, h->counters, upper, mf, lo); break;
    case 3: rg_launch_tick_list_t<3>(h->stream, h->st, ms, h->any_group_commit, h->list, h->counters, upper, mf, lo); break;
    case 4: rg_launch_tick_list_t<4>(h->stream, h->st, ms, h->any_group_commit, h->list, h->counters, upper, mf, lo); break;
    case 5: rg_launch_tick_list_t<5>(h->stream, h->st, ms, h->any_group_commit, h->list, h->counters, upper, mf, lo); break;
    case 6: rg_launch_tick_list_t<6>(h->stream, h->st, ms, h->any_group_commit, h->list, h->counters, upper, mf, lo); break;
    case 7: rg_launch_tick_list_t<7>(h->stream, h->st, ms, h->any_group_commit, h->list, h->counters, upper, mf, lo); break;
    default: rg_launch_tick_list_t<8>(h->stream, h->st, ms, h->any_group_commit, h->list, h->counters, upper, mf, lo); break;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return rg_fail(RG_ERR_NO_DEVICE, "sparse tick: launch failed: %s", hipGetErrorString(e));
    h->tick_launches++;
    return RG_OK;
}

// Bookkeeping once the sparse tick's group count is known on the host.
static int rg_sparse_finish(rg_engine *h, u64 n_groups) {
    h->last_sparse_n = n_groups;
    h->ingested_upper = 0;
    h->epoch++;
    if (h->epoch == 0) { // epoch wrapped: the marks are ambiguous, reset them
        RG_HIP(hipMemsetAsync(h->gmark, 0, h->stride * 4, h->stream));
        h->epoch = 1;
    }
    h->ticked = true;
    h->send_ready = true;
    return RG_OK;
}

extern "C" int rg_tick_ingested(rg_engine *h, uint64_t *n_groups) {
    if (!h) return rg_fail(RG_ERR_INVALID_ARG, "rg_tick_ingested: null engine");
    if (n_groups) *n_groups = 0;
    RG_ENTER(h);
    {   // (device Inflights: nothing of the next step is enqueued while a host hint of the last one is unanswered)
        const int hrc__ = rg_require_hints_resolved(h, "rg_tick_ingested");
        if (hrc__) return hrc__;
    }
    int rc = rg_ensure_sparse(h);
    if (rc) return rc;
    const u64 upper = h->ingested_upper < h->G ? h->ingested_upper : h->G;
    rc = rg_sparse_enqueue(h, upper, nullptr, true); // device-side ingests may carry log terms
    if (rc) return rc;
    u32 n = 0;
    if (upper) {
        RG_HIP(hipMemcpyAsync(&n, h->counters, 4, hipMemcpyDeviceToHost, h->stream));
        RG_HIP(hipStreamSynchronize(h->stream));
        rg_ctr_flip(h); // (the next window's pair was reset by this window's ingest kernels)
    }
    rc = rg_sparse_finish(h, n);
    if (rc) return rc;
    if (n_groups) *n_groups = h->last_sparse_n;
    return RG_OK;
}

extern "C" int rg_ingested_results(rg_engine *h, uint64_t *groups, uint64_t *commit, uint32_t *out, uint64_t cap,
                                   uint64_t *n) {
    if (!h || !n) return rg_fail(RG_ERR_INVALID_ARG, "rg_ingested_results: bad argument");
    *n = h->last_sparse_n;
    const u64 k = h->last_sparse_n < cap ? h->last_sparse_n : cap;
    if (k == 0) return RG_OK;
    if (h->host_res_valid) { // the single-copy flush already brought them over
        if (groups) memcpy(groups, h->host_res_groups.data(), k * 8);
        if (commit) memcpy(commit, h->host_res_commit.data(), k * 8);
        if (out) memcpy(out, h->host_res_out.data(), k * 4);
        return RG_OK;
    }
    RG_ENTER(h);
    if (groups) RG_HIP(hipMemcpyAsync(groups, h->res_list, k * 8, hipMemcpyDeviceToHost, h->stream));
    if (commit) RG_HIP(hipMemcpyAsync(commit, h->res_commit, k * 8, hipMemcpyDeviceToHost, h->stream));
    if (out) RG_HIP(hipMemcpyAsync(out, h->res_out, k * 4, hipMemcpyDeviceToHost, h->stream));
    RG_HIP(hipStreamSynchronize(h->stream));
    return RG_OK;
}

template <int P, bool COMMIT> static void rg_launch_recompute_p(rg_engine *h, u64 *mci, u8 *gc, bool x2) {
    const bool group_commit = h->any_group_commit;
    if (x2) {
        const dim3 grid(rg_grid((h->G + 1) / 2, RG_BLOCK)), block(RG_BLOCK);
        if (group_commit) hipLaunchKernelGGL((k_recompute2<P, COMMIT, true>), grid, block, 0, h->stream, h->st, mci, gc);
        else hipLaunchKernelGGL((k_recompute2<P, COMMIT, false>), grid, block, 0, h->stream, h->st, mci, gc);
    } else {
        const dim3 grid(rg_grid(h->G, RG_BLOCK)), block(RG_BLOCK);
        if (group_commit) hipLaunchKernelGGL((k_recompute<P, COMMIT, true>), grid, block, 0, h->stream, h->st, mci, gc);
        else hipLaunchKernelGGL((k_recompute<P, COMMIT, false>), grid, block, 0, h->stream, h->st, mci, gc);
    }
}

template <bool COMMIT> static int rg_recompute_impl(rg_engine *h, u64 *mci, u8 *gc) {
    if (h->cfg.variant == RG_VARIANT_COOP && !h->any_group_commit) {
        hipLaunchKernelGGL((k_recompute_coop<COMMIT>), dim3(rg_grid(h->G, 32)), dim3(256), 0, h->stream, h->st, h->P, mci, gc);
        hipError_t ce = hipGetLastError();
        if (ce != hipSuccess) return rg_fail(RG_ERR_NO_DEVICE, "recompute launch failed: %s", hipGetErrorString(ce));
        return RG_OK;
    }
    const bool x2 = RG_RECOMPUTE_X2 && h->cfg.variant != RG_VARIANT_LANE; // (variant LANE pins one group per lane)
    switch (h->P) {
    case 1: rg_launch_recompute_p<1, COMMIT>(h, mci, gc, x2); break;
    case 2: rg_launch_recompute_p<2, COMMIT>(h, mci, gc, x2); break;
    case 3: rg_launch_recompute_p<3, COMMIT>(h, mci, gc, x2); break;
    case 4: rg_launch_recompute_p<4, COMMIT>(h, mci, gc, x2); break;
    case 5: rg_launch_recompute_p<5, COMMIT>(h, mci, gc, x2); break;
    case 6: rg_launch_recompute_p<6, COMMIT>(h, mci, gc, x2); break;
    case 7: rg_launch_recompute_p<7, COMMIT>(h, mci, gc, x2); break;
    default: rg_launch_recompute_p<8, COMMIT>(h, mci, gc, x2); break;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return rg_fail(RG_ERR_NO_DEVICE, "recompute launch failed: %s", hipGetErrorString(e));
    return RG_OK;
}

extern "C" int rg_recompute(rg_engine *h) {
    if (!h) return rg_fail(RG_ERR_INVALID_ARG, "rg_recompute: null engine");
    RG_ENTER(h);
    {   // (device Inflights: nothing of the next step is enqueued while a host hint of the last one is unanswered)
        const int hrc__ = rg_require_hints_resolved(h, "rg_recompute");
        if (hrc__) return hrc__;
    }
    int rc = rg_settle_send(h);
    if (rc) return rc;
    rc = rg_recompute_impl<true>(h, nullptr, nullptr);
    if (rc == RG_OK) {
        h->ticked = true;
        h->host_res_valid = false;
        h->out_is_dense = true; // every group's result word was rewritten
        h->send_ready = true;   // post_conf_change: `if self.maybe_commit() { self.bcast_append() }` (raft.rs:2630-2633)
    }
    return rc;
}

extern "C" int rg_maximal_committed_index(rg_engine *h, uint64_t *host_mci, uint8_t *host_gc) {
    if (!h || !host_mci) return rg_fail(RG_ERR_INVALID_ARG, "rg_maximal_committed_index: bad argument");
    RG_ENTER(h);
    u64 *d_mci = nullptr;
    u8 *d_gc = nullptr;
    RG_HIP(hipMalloc(&d_mci, h->G * 8));
    if (host_gc && hipMalloc(&d_gc, h->G) != hipSuccess) {
        (void)hipFree(d_mci);
        return rg_fail(RG_ERR_OUT_OF_MEMORY, "rg_maximal_committed_index: hipMalloc failed");
    }
    int rc = rg_recompute_impl<false>(h, d_mci, d_gc);
    hipError_t e = hipSuccess;
    if (rc == RG_OK) e = hipMemcpyAsync(host_mci, d_mci, h->G * 8, hipMemcpyDeviceToHost, h->stream);
    if (rc == RG_OK && e == hipSuccess && host_gc) e = hipMemcpyAsync(host_gc, d_gc, h->G, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(d_mci);
    if (d_gc) (void)hipFree(d_gc);
    if (rc) return rc;
    if (e != hipSuccess) return rg_fail(RG_ERR_NO_DEVICE, "rg_maximal_committed_index: %s", hipGetErrorString(e));
    return RG_OK;
}

// ------------------------------------------------------------------------------------------------
// send stage
// ------------------------------------------------------------------------------------------------
// Enqueue the send stage over `list[0..n)` (NULL = all groups); n_ptr != NULL: the length is read on the device and
// `n` only sizes the grid.
static int rg_send_enqueue(rg_engine *h, uint64_t max_entries_per_msg, uint32_t flags, const u64 *list, u64 n,
                           const u32 *n_ptr) {
    const bool append = (flags & RG_SEND_APPEND_LIST) != 0; // (rg_resolve_host_hints: the list keeps what it holds)
    flags &= ~RG_SEND_APPEND_LIST;
    h->stage_max_entries = max_entries_per_msg;
    h->stage_flags = flags & ~RG_SEND_REQUESTS_ONLY;
    h->send_cols_fresh = false;
    h->send_last_dense = false;
    if (!list && !n_ptr && n == h->G) { // every group: work items into the peer-major columns, no list
        const dim3 grid(rg_grid(h->G, RG_BLOCK)), block(RG_BLOCK);
        switch (h->P) {
        case 1: rg_launch_send_dense<1>(h->stream, grid, block, h->st, h->ins, (u64)max_entries_per_msg, (u32)flags, h->send_cols); break;
        case 2: rg_launch_send_dense<2>(h->stream, grid, block, h->st, h->ins, (u64)max_entries_per_msg, (u32)flags, h->send_cols); break;
        case 3: rg_launch_send_dense<3>(h->stream, grid, block, h->st, h->ins, (u64)max_entries_per_msg, (u32)flags, h->send_cols); break;
        case 4: rg_launch_send_dense<4>(h->stream, grid, block, h->st, h->ins, (u64)max_entries_per_msg, (u32)flags, h->send_cols); break;
        case 5: rg_launch_send_dense<5>(h->stream, grid, block, h->st, h->ins, (u64)max_entries_per_msg, (u32)flags, h->send_cols); break;
        case 6: rg_launch_send_dense<6>(h->stream, grid, block, h->st, h->ins, (u64)max_entries_per_msg, (u32)flags, h->send_cols); break;
        case 7: rg_launch_send_dense<7>(h->stream, grid, block, h->st, h->ins, (u64)max_entries_per_msg, (u32)flags, h->send_cols); break;
        default: rg_launch_send_dense<8>(h->stream, grid, block, h->st, h->ins, (u64)max_entries_per_msg, (u32)flags, h->send_cols); break;
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return rg_fail(RG_ERR_NO_DEVICE, "send stage: launch failed: %s", hipGetErrorString(e));
        h->send_cols_fresh = true;
        h->send_last_dense = true;
        h->host_items_valid = false;
        return RG_OK;
    }
    if (!append) RG_HIP(hipMemsetAsync(h->send_counter, 0, 4, h->stream));
    if (n) {
        const dim3 grid(rg_grid(n, RG_SEND_BLOCK)), block(RG_SEND_BLOCK);
        switch (h->P) {
        case 1: hipLaunchKernelGGL(k_send_appends<1>, grid, block, 0, h->stream, h->st, h->ins, (u64)max_entries_per_msg, (u32)flags, list, n, n_ptr, h->send_items, h->send_counter); break;
        case 2: hipLaunchKernelGGL(k_send_appends<2>, grid, block, 0, h->stream, h->st, h->ins, (u64)max_entries_per_msg, (u32)flags, list, n, n_ptr, h->send_items, h->send_counter); break;
        case 3: hipLaunchKernelGGL(k_send_appends<3>, grid, block, 0, h->stream, h->st, h->ins, (u64)max_entries_per_msg, (u32)flags, list, n, n_ptr, h->send_items, h->send_counter); break;
        case 4: hipLaunchKernelGGL(k_send_appends<4>, grid, block, 0, h->stream, h->st, h->ins, (u64)max_entries_per_msg, (u32)flags, list, n, n_ptr, h->send_items, h->send_counter); break;
        case 5: hipLaunchKernelGGL(k_send_appends<5>, grid, block, 0, h->stream, h->st, h->ins, (u64)max_entries_per_msg, (u32)flags, list, n, n_ptr, h->send_items, h->send_counter); break;
        case 6: hipLaunchKernelGGL(k_send_appends<6>, grid, block, 0, h->stream, h->st, h->ins, (u64)max_entries_per_msg, (u32)flags, list, n, n_ptr, h->send_items, h->send_counter); break;
        case 7: hipLaunchKernelGGL(k_send_appends<7>, grid, block, 0, h->stream, h->st, h->ins, (u64)max_entries_per_msg, (u32)flags, list, n, n_ptr, h->send_items, h->send_counter); break;
        default: hipLaunchKernelGGL(k_send_appends<8>, grid, block, 0, h->stream, h->st, h->ins, (u64)max_entries_per_msg, (u32)flags, list, n, n_ptr, h->send_items, h->send_counter); break;
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return rg_fail(RG_ERR_NO_DEVICE, "send stage: launch failed: %s", hipGetErrorString(e));
    }
    h->host_items_valid = false;
    return RG_OK;
}

extern "C" int rg_send_appends(rg_engine *h, uint64_t max_entries_per_msg, uint32_t flags) {
    if (!h) return rg_fail(RG_ERR_INVALID_ARG, "rg_send_appends: null engine");
    int src = rg_send_check(h, flags, "rg_send_appends");
    if (src) return src;
    if (!h->send_ready) return rg_fail(RG_ERR_STATE, "rg_send_appends: no tick since the last send stage");
    RG_ENTER(h);
    const u64 *list = h->out_is_dense ? nullptr : h->res_list; // sparse tick: only the touched groups have an out word
    const u64 n = h->out_is_dense ? h->G : h->last_sparse_n;
    int rc = rg_send_enqueue(h, max_entries_per_msg, flags, list, n, nullptr);
    if (rc) return rc;
    h->send_ready = false;
    h->send_bound = n * h->P;
    return RG_OK;
}

extern "C" int rg_log_sizes_enable(rg_engine *h, uint32_t window) {
    if (!h) return rg_fail(RG_ERR_INVALID_ARG, "rg_log_sizes_enable: null engine");
    if (!h->ins_arena) return rg_fail(RG_ERR_STATE, "rg_log_sizes_enable: engine created with max_inflight = 0 (no send stage)");
    if (window < 8 || window > 4096 || (window & (window - 1)))
        return rg_fail(RG_ERR_INVALID_ARG, "rg_log_sizes_enable: window %u, a power of two in 8..4096", window);
    if (h->esz) return rg_fail(RG_ERR_STATE, "rg_log_sizes_enable: already enabled (window %u)", h->ins.esz_w);
    RG_ENTER(h);
    const size_t b = (size_t)h->G * window * 4;
    hipError_t e = hipMalloc(&h->esz, b);
    if (e != hipSuccess) {
        h->esz = nullptr;
        (void)hipGetLastError();
        return rg_fail(RG_ERR_OUT_OF_MEMORY, "rg_log_sizes_enable: hipMalloc(%zu) failed: %s", b, hipGetErrorString(e));
    }
    RG_HIP(hipMemsetAsync(h->esz, 0, b, h->stream));
    h->ins.esz = h->esz;
    h->ins.esz_w = window;
    return RG_OK;
}

// host records -> device staging buffer on the engine's stream (grown on demand; the copy is stream-ordered, the host
// array may be reused once the call returns: pageable copies are staged by the runtime)
static int rg_stage_records(rg_engine *h, const void *recs, size_t bytes) {
    if (bytes > h->d_recs_cap) {
        RG_HIP(hipStreamSynchronize(h->stream)); // (kernels reading the old buffer)
        if (h->d_recs) (void)hipFree(h->d_recs);
        h->d_recs = nullptr;
        h->d_recs_cap = 0;
        const size_t cap = bytes < 65536 ? 65536 : bytes + bytes / 2;
        hipError_t e = hipMalloc(&h->d_recs, cap);
        if (e != hipSuccess) {
            h->d_recs = nullptr;
            (void)hipGetLastError();
            return rg_fail(RG_ERR_OUT_OF_MEMORY, "record staging: hipMalloc(%zu) failed: %s", cap, hipGetErrorString(e));
        }
        h->d_recs_cap = cap;
    }
    RG_HIP(hipMemcpyAsync(h->d_recs, recs, bytes, hipMemcpyHostToDevice, h->stream));
    return RG_OK;
}

extern "C" int rg_log_sizes_write(rg_engine *h, const rg_log_size *recs, uint64_t n) {
    if (!h || (!recs && n)) return rg_fail(RG_ERR_INVALID_ARG, "rg_log_sizes_write: bad argument");
    if (!h->esz) return rg_fail(RG_ERR_STATE, "rg_log_sizes_write: rg_log_sizes_enable first");
    if (n == 0) return RG_OK;
    RG_ENTER(h);
    int rc = rg_stage_records(h, recs, (size_t)n * sizeof(rg_log_size));
    if (rc) return rc;
    hipLaunchKernelGGL(k_log_sizes_write, dim3(rg_grid(n, 256)), dim3(256), 0, h->stream, (const rg_log_size *)h->d_recs, (u64)n,
                       h->G, h->esz, h->ins.esz_w);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return rg_fail(RG_ERR_NO_DEVICE, "rg_log_sizes_write: launch failed: %s", hipGetErrorString(e));
    return RG_OK;
}

extern "C" int rg_workload_sizes(rg_engine *h, uint64_t seed, uint32_t min_bytes, uint32_t spread) {
    if (!h) return rg_fail(RG_ERR_INVALID_ARG, "rg_workload_sizes: null engine");
    if (!h->esz) return rg_fail(RG_ERR_STATE, "rg_workload_sizes: rg_log_sizes_enable first");
    RG_ENTER(h);
    hipLaunchKernelGGL(k_wl_sizes, dim3(rg_grid(h->G, RG_BLOCK)), dim3(RG_BLOCK), 0, h->stream, (const u64 *)h->st.hi, h->G, h->esz,
                       h->ins.esz_w, (u64)seed, min_bytes, spread);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return rg_fail(RG_ERR_NO_DEVICE, "rg_workload_sizes: launch failed: %s", hipGetErrorString(e));
    return RG_OK;
}

extern "C" int rg_update_state(rg_engine *h, const rg_sent_msg *msgs, uint64_t n) {
    if (!h || (!msgs && n)) return rg_fail(RG_ERR_INVALID_ARG, "rg_update_state: bad argument");
    if (!h->ins_arena) return rg_fail(RG_ERR_STATE, "rg_update_state: engine created with max_inflight = 0 (use RG_MF_SENT events)");
    if (n == 0) return RG_OK;
    RG_ENTER(h);
    int rc = rg_stage_records(h, msgs, (size_t)n * sizeof(rg_sent_msg));
    if (rc) return rc;
    hipLaunchKernelGGL(k_update_state, dim3(rg_grid(n, 256)), dim3(256), 0, h->stream, h->st, h->ins, (const rg_sent_msg *)h->d_recs,
                       (u64)n, h->P);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return rg_fail(RG_ERR_NO_DEVICE, "rg_update_state: launch failed: %s", hipGetErrorString(e));
    return RG_OK;
}

extern "C" int rg_progress_events(rg_engine *h, const rg_progress_event *events, uint64_t n) {
    if (!h || (!events && n)) return rg_fail(RG_ERR_INVALID_ARG, "rg_progress_events: bad argument");
    for (u64 i = 0; i < n; i++)
        if (events[i].kind < RG_EV_UNREACHABLE || events[i].kind > RG_EV_SNAPSHOT_FAILURE)
            return rg_fail(RG_ERR_INVALID_ARG, "rg_progress_events: record %llu has kind %u", (unsigned long long)i, events[i].kind);
    if (n == 0) return RG_OK;
    RG_ENTER(h);
    // Call order is event order: a tick whose send stage has not run yet still owes the windows its free_to / free_first_one /
    // left-Replicate effects, and they belong BEFORE this event's become_probe (which empties the window) -- the reference
    // applies everything handle_append_response does before the next local message is stepped. Settled exactly as the next
    // tick would settle it (effects only: the skipped stage's send requests are dropped, which is what skipping it means).
    int rc = rg_settle_send(h);
    if (rc) return rc;
    rc = rg_stage_records(h, events, (size_t)n * sizeof(rg_progress_event));
    if (rc) return rc;
    hipLaunchKernelGGL(k_progress_events, dim3(rg_grid(n, 256)), dim3(256), 0, h->stream, h->st, h->ins_arena ? h->ins.meta : nullptr,
                       (const rg_progress_event *)h->d_recs, (u64)n, h->P);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return rg_fail(RG_ERR_NO_DEVICE, "rg_progress_events: launch failed: %s", hipGetErrorString(e));
    RG_HIP(hipStreamSynchronize(h->stream)); // control path, like rg_write_cells: the caller's array may be reused after return
    return RG_OK;
}

extern "C" int rg_progress_event_dense(rg_engine *h, uint32_t kind, const uint8_t *host_slot_plus1) {
    if (!h || !host_slot_plus1) return rg_fail(RG_ERR_INVALID_ARG, "rg_progress_event_dense: bad argument");
    if (kind < RG_EV_UNREACHABLE || kind > RG_EV_SNAPSHOT_FAILURE)
        return rg_fail(RG_ERR_INVALID_ARG, "rg_progress_event_dense: kind %u", kind);
    RG_ENTER(h);
    int rc = rg_settle_send(h); // (as in rg_progress_events: the last tick's Inflights effects come first)
    if (rc) return rc;
    rc = rg_stage_records(h, host_slot_plus1, (size_t)h->G);
    if (rc) return rc;
    hipLaunchKernelGGL(k_progress_event_dense, dim3(rg_grid(h->G, 256)), dim3(256), 0, h->stream, h->st,
                       h->ins_arena ? h->ins.meta : nullptr, (const u8 *)h->d_recs, (u32)kind, h->P);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return rg_fail(RG_ERR_NO_DEVICE, "rg_progress_event_dense: launch failed: %s", hipGetErrorString(e));
    RG_HIP(hipStreamSynchronize(h->stream)); // control path: the caller's array may be reused after return
    return RG_OK;
}

// After a dense stage the work items live in the columns; the compact list exists once somebody asks for it.
static int rg_send_materialize(rg_engine *h) {
    if (!h->send_cols_fresh) return RG_OK;
    RG_HIP(hipMemsetAsync(h->send_counter, 0, 4, h->stream));
    hipLaunchKernelGGL(k_send_compact, dim3(rg_grid(h->G, 256)), dim3(256), 0, h->stream, h->send_cols, (const u64 *)h->ins.tail, h->G, h->stride, h->P,
                       h->send_items, h->send_counter);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return rg_fail(RG_ERR_NO_DEVICE, "rg_send_items: compaction launch failed: %s", hipGetErrorString(e));
    h->send_cols_fresh = false;
    return RG_OK;
}

extern "C" int rg_send_items(rg_engine *h, rg_send_item *host_items, uint64_t cap, uint64_t *n) {
    if (!h || !n || (!host_items && cap)) return rg_fail(RG_ERR_INVALID_ARG, "rg_send_items: bad argument");
    if (!h->ins_arena) return rg_fail(RG_ERR_STATE, "rg_send_items: engine created with max_inflight = 0");
    if (h->host_items_valid) { // rg_flush_send already brought them over with the tick's results: no device access (and a
        *n = h->host_items.size(); // resident mailbox workgroup stays where it is)
        const u64 k = *n < cap ? *n : cap;
        if (k) memcpy(host_items, h->host_items.data(), k * sizeof(rg_send_item));
        return RG_OK;
    }
    RG_ENTER(h);
    {
        int mrc = rg_send_materialize(h);
        if (mrc) return mrc;
    }
    // small stages (the sparse path): counter and items come back together through pinned memory -- one round trip
    const u64 spec = h->send_bound < cap ? h->send_bound : cap;
    if (spec && spec <= RG_SEND_SPEC) {
        if (!h->pin_send)
            RG_HIP(hipHostMalloc(reinterpret_cast<void **>(&h->pin_send), 16 + RG_SEND_SPEC * sizeof(rg_send_item),
                                 hipHostMallocDefault));
        RG_HIP(hipMemcpyAsync(h->pin_send, h->send_counter, 4, hipMemcpyDeviceToHost, h->stream));
        RG_HIP(hipMemcpyAsync(h->pin_send + 16, h->send_items, spec * sizeof(rg_send_item), hipMemcpyDeviceToHost, h->stream));
        RG_HIP(hipStreamSynchronize(h->stream));
        const u32 cnt = *reinterpret_cast<const u32 *>(h->pin_send);
        *n = cnt;
        const u64 k = cnt < cap ? cnt : cap; // cnt <= send_bound, so k <= spec
        if (k) memcpy(host_items, h->pin_send + 16, k * sizeof(rg_send_item));
        return RG_OK;
    }
    u32 cnt = 0;
    RG_HIP(hipMemcpyAsync(&cnt, h->send_counter, 4, hipMemcpyDeviceToHost, h->stream));
    RG_HIP(hipStreamSynchronize(h->stream));
    *n = cnt;
    const u64 k = cnt < cap ? cnt : cap;
    if (k) {
        RG_HIP(hipMemcpyAsync(host_items, h->send_items, k * sizeof(rg_send_item), hipMemcpyDeviceToHost, h->stream));
        RG_HIP(hipStreamSynchronize(h->stream));
    }
    return RG_OK;
}

extern "C" const rg_send_item *rg_send_items_ptr(rg_engine *h) {
    if (!h || !h->ins_arena) return nullptr;
    if (hipSetDevice(h->cfg.device) != hipSuccess || rg_send_materialize(h) != RG_OK) return nullptr;
    return h->send_items;
}

extern "C" int rg_send_columns(rg_engine *h, const uint64_t **dev_prev_index, const uint64_t **dev_last_index,
                               const uint32_t **dev_n_kind) {
    if (!h) return rg_fail(RG_ERR_INVALID_ARG, "rg_send_columns: null engine");
    if (!h->ins_arena) return rg_fail(RG_ERR_STATE, "rg_send_columns: engine created with max_inflight = 0");
    if (!h->send_last_dense)
        return rg_fail(RG_ERR_STATE, "rg_send_columns: the last send stage was not a dense one (sparse stages produce the "
                                     "compact list only)");
    if (dev_prev_index) *dev_prev_index = h->send_cols.prev;
    if (dev_last_index) *dev_last_index = h->send_cols.last;
    if (dev_n_kind) *dev_n_kind = h->send_cols.n;
    return RG_OK;
}

extern "C" int rg_send_tail_column(rg_engine *h, const uint64_t **dev_newest_inflight) {
    if (!h || !dev_newest_inflight) return rg_fail(RG_ERR_INVALID_ARG, "rg_send_tail_column: bad argument");
    if (!h->ins_arena) return rg_fail(RG_ERR_STATE, "rg_send_tail_column: engine created with max_inflight = 0");
    *dev_newest_inflight = h->ins.tail;
    return RG_OK;
}
static_assert(RG_SEND_LAST_IS_TAIL == RG_SEND_NK_LAST_IS_TAIL && RG_SEND_LAST_IS_PREV == RG_SEND_NK_LAST_IS_PREV, "the header's bits are the kernels'");

extern "C" uint64_t rg_inflights_bytes(const rg_engine *h, int ring) {
    if (!h || !h->ins_arena) return 0;
    return ring ? (uint64_t)h->G * h->P * h->ins.cap * 8 : (uint64_t)h->P * h->stride * 4;
}

// The oldest and the newest inflight of a window live in the `head` / `tail` columns (rg_send.h); to the outside the
// ring is whole.
extern "C" int rg_read_inflights(rg_engine *h, uint32_t *host_meta, uint64_t *host_ring) {
    if (!h) return rg_fail(RG_ERR_INVALID_ARG, "rg_read_inflights: null engine");
    if (!h->ins_arena) return rg_fail(RG_ERR_STATE, "rg_read_inflights: engine created with max_inflight = 0");
    RG_ENTER(h);
    const u64 cells = (u64)h->P * h->stride;
    std::vector<u32> meta_tmp;
    std::vector<u64> head, tail;
    u32 *meta = host_meta;
    if (host_ring) {
        head.resize(cells);
        tail.resize(cells);
        if (!meta) {
            meta_tmp.resize(cells);
            meta = meta_tmp.data();
        }
        RG_HIP(hipMemcpyAsync(head.data(), h->ins.head, cells * 8, hipMemcpyDeviceToHost, h->stream));
        RG_HIP(hipMemcpyAsync(tail.data(), h->ins.tail, cells * 8, hipMemcpyDeviceToHost, h->stream));
        RG_HIP(hipMemcpyAsync(host_ring, h->ins.ring, rg_inflights_bytes(h, 1), hipMemcpyDeviceToHost, h->stream));
    }
    if (meta) RG_HIP(hipMemcpyAsync(meta, h->ins.meta, cells * 4, hipMemcpyDeviceToHost, h->stream));
    RG_HIP(hipStreamSynchronize(h->stream));
    if (host_ring)
        for (u32 p = 0; p < h->P; p++)
            for (u64 g = 0; g < h->G; g++) {
                const u32 m = meta[(u64)p * h->stride + g], start = m & 0xffffu, count = m >> 16;
                if (!count) continue;
                u64 *cell = host_ring + (g * h->P + p) * h->ins.cap;
                cell[start] = head[(u64)p * h->stride + g];
                cell[(start + count - 1) % h->ins.cap] = tail[(u64)p * h->stride + g];
            }
    return RG_OK;
}

extern "C" int rg_load_inflights(rg_engine *h, const uint32_t *host_meta, const uint64_t *host_ring) {
    if (!h || !host_meta || !host_ring) return rg_fail(RG_ERR_INVALID_ARG, "rg_load_inflights: meta and ring are both required");
    if (!h->ins_arena) return rg_fail(RG_ERR_STATE, "rg_load_inflights: engine created with max_inflight = 0");
    const u64 cells = (u64)h->P * h->stride;
    std::vector<u64> head(cells, 0), tail(cells, 0);
    for (u32 p = 0; p < h->P; p++)
        for (u64 g = 0; g < h->G; g++) { // start < cap, count <= cap for every cell
            const u64 o = (u64)p * h->stride + g;
            const u32 m = host_meta[o], start = m & 0xffffu, count = m >> 16;
            if (start >= h->ins.cap || count > h->ins.cap)
                return rg_fail(RG_ERR_INVALID_ARG, "rg_load_inflights: group %llu slot %u: start %u count %u outside cap %u",
                               (unsigned long long)g, p, start, count, h->ins.cap);
            if (!count) continue;
            const u64 *cell = host_ring + (g * h->P + p) * h->ins.cap;
            head[o] = cell[start];
            for (u32 i = 1; i < count; i++) // last indices of consecutive MsgAppends
                if (cell[(start + i) % h->ins.cap] <= cell[(start + i - 1) % h->ins.cap])
                    return rg_fail(RG_ERR_INVALID_ARG, "rg_load_inflights: group %llu slot %u: inflights must be strictly "
                                                       "increasing, oldest first", (unsigned long long)g, p);
            tail[o] = cell[(start + count - 1) % h->ins.cap];
        }
    RG_ENTER(h);
    RG_HIP(hipMemcpyAsync(h->ins.meta, host_meta, cells * 4, hipMemcpyHostToDevice, h->stream));
    RG_HIP(hipMemcpyAsync(h->ins.head, head.data(), cells * 8, hipMemcpyHostToDevice, h->stream));
    RG_HIP(hipMemcpyAsync(h->ins.tail, tail.data(), cells * 8, hipMemcpyHostToDevice, h->stream));
    RG_HIP(hipMemcpyAsync(h->ins.ring, host_ring, rg_inflights_bytes(h, 1), hipMemcpyHostToDevice, h->stream));
    // Inflights::full() of the loaded windows, for the next tick's is_paused() / free_first_one decisions
    hipLaunchKernelGGL(k_fix_ins_full, dim3(rg_grid(h->G, RG_BLOCK)), dim3(RG_BLOCK), 0, h->stream, h->st, h->ins, h->P);
    RG_HIP(hipStreamSynchronize(h->stream));
    return RG_OK;
}

extern "C" int rg_heartbeat_commits(rg_engine *h, uint64_t *dev_hb, uint64_t *host_hb) {
    if (!h || (!dev_hb && !host_hb)) return rg_fail(RG_ERR_INVALID_ARG, "rg_heartbeat_commits: no destination");
    RG_ENTER(h);
    u64 *tmp = nullptr;
    u64 *dst = (u64 *)dev_hb;
    const size_t bytes = (size_t)h->P * h->stride * 8;
    if (!dst) {
        RG_HIP(hipMalloc(&tmp, bytes));
        dst = tmp;
    }
    hipLaunchKernelGGL(k_heartbeat_commits, dim3(rg_grid(h->G, RG_BLOCK)), dim3(RG_BLOCK), 0, h->stream, h->st, h->P, dst);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && host_hb) {
        e = hipMemcpyAsync(host_hb, dst, bytes, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    }
    if (tmp) {
        if (!host_hb) (void)hipStreamSynchronize(h->stream);
        (void)hipFree(tmp);
    }
    if (e != hipSuccess) return rg_fail(RG_ERR_NO_DEVICE, "rg_heartbeat_commits: %s", hipGetErrorString(e));
    return RG_OK;
}

extern "C" int rg_results(rg_engine *h, uint64_t *host_commit, uint32_t *host_out) {
    if (!h) return rg_fail(RG_ERR_INVALID_ARG, "rg_results: null engine");
    if (!h->ticked) return rg_fail(RG_ERR_STATE, "rg_results: no tick has run yet");
    RG_ENTER(h);
    if (host_commit) RG_HIP(hipMemcpyAsync(host_commit, h->st.commit, h->G * 8, hipMemcpyDeviceToHost, h->stream));
    if (host_out) RG_HIP(hipMemcpyAsync(host_out, h->st.out, h->G * 4, hipMemcpyDeviceToHost, h->stream));
    RG_HIP(hipStreamSynchronize(h->stream));
    return RG_OK;
}

extern "C" int rg_result_counts(rg_engine *h, uint64_t *n_changed, uint64_t *n_fault) {
    if (!h) return rg_fail(RG_ERR_INVALID_ARG, "rg_result_counts: null engine");
    if (!h->ticked) return rg_fail(RG_ERR_STATE, "rg_result_counts: no tick has run yet");
    RG_ENTER(h);
    RG_HIP(hipMemsetAsync(h->d_counts, 0, 32, h->stream));
    const unsigned grid = rg_grid(h->G, RG_BLOCK) < 2048 ? rg_grid(h->G, RG_BLOCK) : 2048;
    hipLaunchKernelGGL(k_count_out, dim3(grid), dim3(RG_BLOCK), 0, h->stream, (const u32 *)h->st.out, h->G, h->d_counts);
    u64 c[2] = {0, 0};
    RG_HIP(hipMemcpyAsync(c, h->d_counts, 16, hipMemcpyDeviceToHost, h->stream));
    RG_HIP(hipStreamSynchronize(h->stream));
    if (n_changed) *n_changed = c[0];
    if (n_fault) *n_fault = c[1];
    return RG_OK;
}

extern "C" int rg_host_hints(rg_engine *h, rg_host_hint *host_items, uint64_t cap, uint64_t *n) {
    if (!h || !n || (cap && !host_items)) return rg_fail(RG_ERR_INVALID_ARG, "rg_host_hints: bad argument");
    *n = 0;
    if (!h->ticked) return rg_fail(RG_ERR_STATE, "rg_host_hints: no tick has run yet");
    RG_ENTER(h);
    RG_HIP(hipMemsetAsync(h->d_counts, 0, 8, h->stream));
    const unsigned grid = rg_grid(h->G, RG_BLOCK) < 2048 ? rg_grid(h->G, RG_BLOCK) : 2048;
    u64 *items = reinterpret_cast<u64 *>(h->d_scratch); // G x 8 B: one packed word per flagged group
    hipLaunchKernelGGL(k_host_hints, dim3(grid), dim3(RG_BLOCK), 0, h->stream, (const u32 *)h->st.out, (const u8 *)h->st.hhint,
                       h->G, items, h->d_counts);
    u64 cnt = 0;
    RG_HIP(hipMemcpyAsync(&cnt, h->d_counts, 8, hipMemcpyDeviceToHost, h->stream));
    RG_HIP(hipStreamSynchronize(h->stream));
    *n = cnt;
    const u64 k = cnt < cap ? cnt : cap;
    if (k) {
        std::vector<u64> packed(k);
        RG_HIP(hipMemcpy(packed.data(), items, k * 8, hipMemcpyDeviceToHost));
        for (u64 i = 0; i < k; i++) {
            host_items[i].group = packed[i] & ((1ULL << 56) - 1);
            host_items[i].slot_mask = (uint32_t)(packed[i] >> 56);
            host_items[i].reserved = 0;
        }
    }
    return RG_OK;
}

extern "C" int rg_resolve_host_hints(rg_engine *h, const rg_resolved_hint *items, uint64_t n, uint8_t *host_applied) {
    if (!h || (!items && n)) return rg_fail(RG_ERR_INVALID_ARG, "rg_resolve_host_hints: bad argument");
    if (!h->ticked) return rg_fail(RG_ERR_STATE, "rg_resolve_host_hints: no tick has run yet");
    if (n == 0) return RG_OK;
    for (u64 i = 0; i < n; i++)
        if (items[i].group >= h->G || items[i].slot >= h->P)
            return rg_fail(RG_ERR_INVALID_ARG, "rg_resolve_host_hints: record %llu names group %llu slot %u", (unsigned long long)i,
                           (unsigned long long)items[i].group, items[i].slot);
    RG_ENTER(h);
    // records, then one result byte per record, in the staging buffer
    const size_t rec_b = (size_t)n * sizeof(rg_resolved_hint);
    std::vector<char> stage(rec_b + (size_t)n, 0);
    memcpy(stage.data(), items, rec_b);
    int rc = rg_stage_records(h, stage.data(), stage.size());
    if (rc) return rc;
    u8 *d_applied = reinterpret_cast<u8 *>(h->d_recs) + rec_b;
    hipLaunchKernelGGL(k_resolve_apply, dim3(rg_grid(n, 256)), dim3(256), 0, h->stream, h->st, h->ins_arena ? h->ins.meta : nullptr,
                       (const rg_resolved_hint *)h->d_recs, (u64)n, h->P, d_applied);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return rg_fail(RG_ERR_NO_DEVICE, "rg_resolve_host_hints: launch failed: %s", hipGetErrorString(e));
    std::vector<u8> applied(n);
    RG_HIP(hipMemcpyAsync(applied.data(), d_applied, n, hipMemcpyDeviceToHost, h->stream));
    RG_HIP(hipStreamSynchronize(h->stream));
    if (host_applied)
        for (u64 i = 0; i < n; i++) host_applied[i] = applied[i] & RG_RESOLVE_APPLIED;
    h->host_res_valid = false; // (the host copy of a sparse tick's result words no longer matches RG_COL_OUT)
    std::vector<u64> groups; // the groups whose LAST waiting slot this call answered: their send requests are due now
    for (u64 i = 0; i < n; i++)
        if (applied[i] & RG_RESOLVE_RELEASED) groups.push_back(items[i].group);
    if (h->ins_arena && !h->send_ready && !groups.empty()) {
        // the send stage of this tick has run already and held these groups' requests back (rg_group_send / rg_group_tick_send):
        // serve them now, over exactly these groups, with that stage's limit and flags, and append the work items to the compact
        // list (send_ready still set: the stage is yet to come, rg_send_appends will find the completed result words)
        std::sort(groups.begin(), groups.end());
        groups.erase(std::unique(groups.begin(), groups.end()), groups.end());
        rc = rg_send_materialize(h); // (a dense stage's items: columns -> list, so that the list holds everything)
        if (rc) return rc;
        h->send_last_dense = false;
        rc = rg_stage_records(h, groups.data(), groups.size() * 8);
        if (rc) return rc;
        // (requests only: that stage applied the groups' Inflights effects when it skipped their requests)
        rc = rg_send_enqueue(h, h->stage_max_entries, h->stage_flags | RG_SEND_APPEND_LIST | RG_SEND_REQUESTS_ONLY, (const u64 *)h->d_recs,
                             groups.size(), nullptr);
        if (rc) return rc;
        h->send_bound += groups.size() * h->P;
        RG_HIP(hipStreamSynchronize(h->stream));
    }
    return RG_OK;
}

extern "C" int rg_msg_stats(rg_engine *h, const uint8_t *d_m_flags, uint64_t counts[5]) {
    if (!h || !d_m_flags || !counts) return rg_fail(RG_ERR_INVALID_ARG, "rg_msg_stats: bad argument");
    RG_ENTER(h);
    RG_HIP(hipMemsetAsync(h->d_counts, 0, 40, h->stream));
    const unsigned grid = rg_grid(h->G, RG_BLOCK) < 1024 ? rg_grid(h->G, RG_BLOCK) : 1024;
    hipLaunchKernelGGL(k_msg_stats, dim3(grid), dim3(RG_BLOCK), 0, h->stream, (const u64 *)d_m_flags,
                       (const u32 *)h->st.cfg, h->G, h->d_counts);
    RG_HIP(hipMemcpyAsync(counts, h->d_counts, 40, hipMemcpyDeviceToHost, h->stream));
    RG_HIP(hipStreamSynchronize(h->stream));
    return RG_OK;
}

// ------------------------------------------------------------------------------------------------
// votes / liveness
// ------------------------------------------------------------------------------------------------
extern "C" int rg_vote_result(rg_engine *h, const uint8_t *yes, const uint8_t *no, uint8_t *result) {
    if (!h || !yes || !no || !result) return rg_fail(RG_ERR_INVALID_ARG, "rg_vote_result: bad argument");
    RG_ENTER(h);
    u8 *d = reinterpret_cast<u8 *>(h->d_scratch); // 8*stride bytes: yes | no | result
    RG_HIP(hipMemcpyAsync(d, yes, h->G, hipMemcpyHostToDevice, h->stream));
    RG_HIP(hipMemcpyAsync(d + h->stride, no, h->G, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_vote, dim3(rg_grid(h->G, RG_BLOCK)), dim3(RG_BLOCK), 0, h->stream, h->st, d,
                       d + h->stride, d + 2 * h->stride, (u8 *)nullptr, (u8 *)nullptr);
    RG_HIP(hipMemcpyAsync(result, d + 2 * h->stride, h->G, hipMemcpyDeviceToHost, h->stream));
    RG_HIP(hipStreamSynchronize(h->stream));
    return RG_OK;
}

extern "C" int rg_tally_votes(rg_engine *h, const uint8_t *yes, const uint8_t *no, uint8_t *granted, uint8_t *rejected,
                              uint8_t *result) {
    if (!h || !yes || !no || !granted || !rejected || !result)
        return rg_fail(RG_ERR_INVALID_ARG, "rg_tally_votes: bad argument");
    RG_ENTER(h);
    u8 *d = reinterpret_cast<u8 *>(h->d_scratch); // 8*stride bytes: yes | no | result | granted | rejected
    RG_HIP(hipMemcpyAsync(d, yes, h->G, hipMemcpyHostToDevice, h->stream));
    RG_HIP(hipMemcpyAsync(d + h->stride, no, h->G, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_vote, dim3(rg_grid(h->G, RG_BLOCK)), dim3(RG_BLOCK), 0, h->stream, h->st, d,
                       d + h->stride, d + 2 * h->stride, d + 3 * h->stride, d + 4 * h->stride);
    RG_HIP(hipMemcpyAsync(result, d + 2 * h->stride, h->G, hipMemcpyDeviceToHost, h->stream));
    RG_HIP(hipMemcpyAsync(granted, d + 3 * h->stride, h->G, hipMemcpyDeviceToHost, h->stream));
    RG_HIP(hipMemcpyAsync(rejected, d + 4 * h->stride, h->G, hipMemcpyDeviceToHost, h->stream));
    RG_HIP(hipStreamSynchronize(h->stream));
    return RG_OK;
}

extern "C" int rg_quorum_recently_active(rg_engine *h, uint8_t *result) {
    if (!h || !result) return rg_fail(RG_ERR_INVALID_ARG, "rg_quorum_recently_active: bad argument");
    RG_ENTER(h);
    u8 *d = reinterpret_cast<u8 *>(h->d_scratch);
    hipLaunchKernelGGL(k_quorum_active, dim3(rg_grid(h->G, RG_BLOCK)), dim3(RG_BLOCK), 0, h->stream, h->st, d);
    RG_HIP(hipMemcpyAsync(result, d, h->G, hipMemcpyDeviceToHost, h->stream));
    RG_HIP(hipStreamSynchronize(h->stream));
    return RG_OK;
}

// ------------------------------------------------------------------------------------------------
// host mirror of RawNode::step for MsgAppendResponse
// ------------------------------------------------------------------------------------------------
static void rg_mirror_init(rg_engine *h) {
    if (h->host_mirror) return;
    h->peer_ids.assign(h->G * 8, 0);
    h->terms.assign(h->G, 0);
    const size_t n = (size_t)h->P * h->stride;
    h->q_mi.assign(n, 0);
    h->q_mc.assign(n, 0);
    h->q_mh.assign(n, 0);
    h->q_mrs.assign(n, 0);
    h->q_mlt.assign(n, 0);
    h->q_mf.assign(h->G * 8, 0);
    h->host_mirror = true;
}

extern "C" int rg_set_peers(rg_engine *h, uint64_t group, const uint64_t *peer_ids, uint32_t n, uint64_t term) {
    if (!h || !peer_ids || group >= h->G || n > h->P) return rg_fail(RG_ERR_INVALID_ARG, "rg_set_peers: bad argument");
    rg_mirror_init(h);
    for (u32 i = 0; i < 8; i++) h->peer_ids[group * 8 + i] = i < n ? peer_ids[i] : 0; // id 0 is illegal (raw_node.rs:303)
    h->terms[group] = term;
    return RG_OK;
}

static int rg_find_slot(rg_engine *h, u64 group, u64 id) {
    if (id == 0) return -1;
    for (u32 i = 0; i < h->P; i++)
        if (h->peer_ids[group * 8 + i] == id) return (int)i;
    return -1;
}

static void rg_touch(rg_engine *h, u64 group) {
    u64 row = 0;
    memcpy(&row, &h->q_mf[group * 8], 8);
    if (row == 0) h->q_dirty.push_back(group);
}

static int rg_self_slot(rg_engine *h, u64 group, u32 *slot);

// A response whose `from` is the leader's OWN id. No follower sends one; the reference would run it against the leader's
// own Progress, where a well-formed one changes nothing (matched = persisted = last_index: a reject is stale, an accept at
// or below matched is a no-op). Here the leader's slot carries the LOCAL events of the tick -- VALID is
// on_persist_entries, and the REJECT bit is RG_MF_BECOME_LEADER, whose m_hint is the new TERM: a spoofed or misrouted
// reject would run Raft::reset + become_leader with reject_hint as the term. So the mirror drops such a message (RG_OK,
// nothing queued); local events enter through rg_local_* only.
static int rg_from_self(rg_engine *h, u64 group, int slot, bool *is_self) {
    u32 self;
    int rc = rg_self_slot(h, group, &self);
    if (rc) return rc;
    *is_self = (u32)slot == self;
    return RG_OK;
}

extern "C" int rg_step(rg_engine *h, uint64_t group, const rg_append_response *m) {
    if (!h || !m || group >= h->G) return rg_fail(RG_ERR_INVALID_ARG, "rg_step: bad argument");
    if (!h->host_mirror) return rg_fail(RG_ERR_STATE, "rg_step: rg_set_peers was never called");
    // RawNode::step (src/raw_node.rs:402-411): MsgAppendResponse is not a local message type; a response from an id
    // without a Progress is rejected BEFORE Raft::step looks at the term, so a removed peer cannot depose the leader
    const int slot = rg_find_slot(h, group, m->from);
    if (slot < 0) return rg_fail(RG_ERR_STEP_PEER_NOT_FOUND, "rg_step: peer %llu not in group %llu (raw_node.rs:407-410)",
                                 (unsigned long long)m->from, (unsigned long long)group);
    // Raft::step term gate (src/raft.rs:1282-1411); term 0 skips the gate (":1282 local message") and falls
    // through to step_leader exactly as in the reference
    if (m->term != 0) {
        if (m->term > h->terms[group])
            return rg_fail(RG_ERR_HIGHER_TERM, "rg_step: message term %llu > leader term %llu: step down (raft.rs:1284-1348)",
                           (unsigned long long)m->term, (unsigned long long)h->terms[group]);
        if (m->term < h->terms[group]) return RG_OK; // stale term: ignored (raft.rs:1349-1411)
    }
    bool from_self;
    int src = rg_from_self(h, group, slot, &from_self);
    if (src) return src;
    if (from_self) return RG_OK; // (dropped: see rg_from_self)
    u8 &f = h->q_mf[group * 8 + slot];
    if (f & (RG_MF_VALID | RG_MF_HEARTBEAT)) return rg_fail(RG_ERR_SLOT_BUSY, "rg_step: peer %llu already has a message queued; rg_flush first",
                                        (unsigned long long)m->from);
    rg_touch(h, group);
    const size_t o = (size_t)slot * h->stride + group;
    h->q_mi[o] = m->index;
    h->q_mc[o] = m->commit;
    h->q_mh[o] = m->reject_hint;
    h->q_mrs[o] = m->request_snapshot;
    h->q_mlt[o] = m->log_term;
    if (m->reject && m->log_term) h->q_any_logterm = true;
    f |= RG_MF_VALID | (m->reject ? RG_MF_REJECT : 0) | (m->request_snapshot ? RG_MF_HAS_RS : 0) |
         (m->ins_full ? RG_MF_INS_FULL : 0) | ((m->reject && m->log_term) ? RG_MF_HAS_LOGTERM : 0);
    return RG_OK;
}

extern "C" int rg_step_heartbeat_response(rg_engine *h, uint64_t group, uint64_t from, uint64_t term, uint64_t commit,
                                          uint8_t ins_full) {
    if (!h || group >= h->G) return rg_fail(RG_ERR_INVALID_ARG, "rg_step_heartbeat_response: bad argument");
    if (!h->host_mirror) return rg_fail(RG_ERR_STATE, "rg_step_heartbeat_response: rg_set_peers was never called");
    const int slot = rg_find_slot(h, group, from); // raw_node.rs:407-410 comes before the term gate
    if (slot < 0) return rg_fail(RG_ERR_STEP_PEER_NOT_FOUND, "rg_step_heartbeat_response: peer %llu not in group %llu",
                                 (unsigned long long)from, (unsigned long long)group);
    if (term != 0) {
        if (term > h->terms[group]) return rg_fail(RG_ERR_HIGHER_TERM, "rg_step_heartbeat_response: higher term: step down");
        if (term < h->terms[group]) return RG_OK;
    }
    bool from_self;
    int src = rg_from_self(h, group, slot, &from_self);
    if (src) return src;
    if (from_self) return RG_OK; // (dropped: see rg_from_self)
    u8 &f = h->q_mf[group * 8 + slot];
    if (f & (RG_MF_VALID | RG_MF_HEARTBEAT))
        return rg_fail(RG_ERR_SLOT_BUSY, "rg_step_heartbeat_response: peer %llu already has a message queued", (unsigned long long)from);
    rg_touch(h, group);
    h->q_mc[(size_t)slot * h->stride + group] = commit;
    f |= RG_MF_HEARTBEAT | (ins_full ? RG_MF_INS_FULL : 0);
    return RG_OK;
}

static int rg_self_slot(rg_engine *h, u64 group, u32 *slot) {
    // the self slot lives in the device cfg word; the mirror keeps a host copy of the column, refreshed
    // whenever the column may have changed (rg_load_column / rg_set_config / rg_workload_init)
    if (!h->host_cfg_valid) {
        // (a resident mailbox workgroup sits on the stream: without this the copy below waits for its idle time-out)
        int qrc = rg_mailbox_quiesce(h);
        if (qrc) return qrc;
        h->host_cfg.resize(h->G);
        RG_HIP(hipMemcpyAsync(h->host_cfg.data(), h->st.cfg, h->G * 4, hipMemcpyDeviceToHost, h->stream));
        RG_HIP(hipStreamSynchronize(h->stream));
        h->host_cfg_valid = true;
    }
    *slot = RG_CFG_SELF(h->host_cfg[group]);
    return RG_OK;
}

// ---- eraftpb::Message off the wire: the decoder is rg_wire.h (host code, sanitiser-tested on its own) ----
extern "C" int rg_decode_message(const uint8_t *bytes, uint64_t len, rg_decoded_message *out) {
    if ((!bytes && len) || !out) return rg_fail(RG_ERR_INVALID_ARG, "rg_decode_message: bad argument");
    rg_wire_u64 bad = 0;
    if (!rg_wire_decode(bytes, len, out, &bad))
        return rg_fail(RG_ERR_INVALID_ARG, "rg_decode_message: not a protobuf-encoded eraftpb::Message (malformed at byte %llu)",
                       (unsigned long long)bad);
    return RG_OK;
}

// ---- ... and onto the wire: the encoder is rg_wire.h as well (host code; nothing here touches an engine) ----
extern "C" uint64_t rg_entry_size(const rg_entry *e) { return e ? rg_wire_entry_size(e) : 0; }

extern "C" uint64_t rg_limit_size(const rg_entry *entries, uint64_t n, uint64_t max_size) {
    return entries ? rg_wire_limit_size(entries, n, max_size) : 0;
}

extern "C" int rg_message_size(const rg_message *m, uint64_t *len) {
    if (!m || !len) return rg_fail(RG_ERR_INVALID_ARG, "rg_message_size: bad argument");
    if (!rg_wire_message_size(m, len))
        return rg_fail(RG_ERR_INVALID_ARG, "rg_message_size: a length without its pointer, or more than 2 GiB - 1 bytes");
    return RG_OK;
}

extern "C" int rg_encode_message(const rg_message *m, uint8_t *buf, uint64_t cap, uint64_t *len) {
    if (!m || !len || (!buf && cap)) return rg_fail(RG_ERR_INVALID_ARG, "rg_encode_message: bad argument");
    if (!rg_wire_message_size(m, len))
        return rg_fail(RG_ERR_INVALID_ARG, "rg_encode_message: a length without its pointer, or more than 2 GiB - 1 bytes");
    if (cap < *len)
        return rg_fail(RG_ERR_INVALID_ARG, "rg_encode_message: %llu bytes needed, the buffer holds %llu",
                       (unsigned long long)*len, (unsigned long long)cap);
    rg_wire_encode(m, buf);
    return RG_OK;
}

extern "C" int rg_step_bytes(rg_engine *h, uint64_t group, const uint8_t *bytes, uint64_t len, uint8_t ins_full) {
    if (!h || group >= h->G) return rg_fail(RG_ERR_INVALID_ARG, "rg_step_bytes: bad argument");
    rg_decoded_message m;
    int rc = rg_decode_message(bytes, len, &m);
    if (rc) return rc;
    switch (m.msg_type) {
    case 0: case 1: case 10: case 11: case 12: // MsgHup, MsgBeat, MsgUnreachable, MsgSnapStatus, MsgCheckQuorum: is_local_msg
        return rg_fail(RG_ERR_STEP_LOCAL_MSG, "rg_step_bytes: raft: cannot step raft local message (raw_node.rs:404-406)");
    case 4: { // MsgAppendResponse
        rg_append_response r;
        memset(&r, 0, sizeof(r));
        r.from = m.from;
        r.term = m.term;
        r.index = m.index;
        r.commit = m.commit;
        r.reject = (uint8_t)m.reject;
        r.reject_hint = m.reject_hint;
        r.log_term = m.log_term;
        r.request_snapshot = m.request_snapshot;
        r.ins_full = ins_full; // (not on the wire: the caller's Inflights::full() for m.from, as in rg_step)
        return rg_step(h, group, &r);
    }
    case 9: // MsgHeartbeatResponse
        return rg_step_heartbeat_response(h, group, m.from, m.term, m.commit, ins_full);
    default:
        return rg_fail(RG_ERR_NOT_ON_PATH, "rg_step_bytes: message type %u is not handled on this path (the host's Raft::step takes it)",
                       m.msg_type);
    }
}

extern "C" int rg_local_append(rg_engine *h, uint64_t group, uint64_t new_last_index) {
    if (!h || group >= h->G) return rg_fail(RG_ERR_INVALID_ARG, "rg_local_append: bad argument");
    if (!h->host_mirror) return rg_fail(RG_ERR_STATE, "rg_local_append: rg_set_peers was never called");
    u32 slot;
    int rc = rg_self_slot(h, group, &slot);
    if (rc) return rc;
    rg_touch(h, group);
    h->q_mc[(size_t)slot * h->stride + group] = new_last_index;
    h->q_mf[group * 8 + slot] |= RG_MF_APPEND;
    return RG_OK;
}

extern "C" int rg_local_persisted(rg_engine *h, uint64_t group, uint64_t index) {
    if (!h || group >= h->G) return rg_fail(RG_ERR_INVALID_ARG, "rg_local_persisted: bad argument");
    if (!h->host_mirror) return rg_fail(RG_ERR_STATE, "rg_local_persisted: rg_set_peers was never called");
    u32 slot;
    int rc = rg_self_slot(h, group, &slot);
    if (rc) return rc;
    u8 &f = h->q_mf[group * 8 + slot];
    if (f & RG_MF_VALID) return rg_fail(RG_ERR_SLOT_BUSY, "rg_local_persisted: already queued; rg_flush first");
    rg_touch(h, group);
    h->q_mi[(size_t)slot * h->stride + group] = index;
    f |= RG_MF_VALID;
    return RG_OK;
}

extern "C" int rg_local_become_leader(rg_engine *h, uint64_t group, uint64_t term) {
    if (!h || group >= h->G) return rg_fail(RG_ERR_INVALID_ARG, "rg_local_become_leader: bad argument");
    if (!h->host_mirror) return rg_fail(RG_ERR_STATE, "rg_local_become_leader: rg_set_peers was never called");
    if (term <= h->terms[group])
        return rg_fail(RG_ERR_INVALID_ARG, "rg_local_become_leader: term %llu is not above the group's term %llu",
                       (unsigned long long)term, (unsigned long long)h->terms[group]);
    u32 slot;
    int rc = rg_self_slot(h, group, &slot);
    if (rc) return rc;
    u64 row = 0;
    memcpy(&row, &h->q_mf[group * 8], 8);
    if (row) return rg_fail(RG_ERR_SLOT_BUSY, "rg_local_become_leader: the group already has events queued (they belong "
                                              "to the old term); rg_flush first");
    u8 &f = h->q_mf[group * 8 + slot];
    rg_touch(h, group);
    h->q_mh[(size_t)slot * h->stride + group] = term;
    f |= RG_MF_BECOME_LEADER;
    // responses of the new term pass the gate from now on (they may be queued behind the election in this very flush);
    // the flush checks the device's verdict and moves the gate BACK if the event was refused there (rg_settle_elections)
    h->q_elections.push_back({group, h->terms[group]});
    h->terms[group] = term;
    return RG_OK;
}

// RawNode::report_unreachable / report_snapshot (src/raw_node.rs:692-709): MsgUnreachable / MsgSnapStatus stepped at a leader
static int rg_report(rg_engine *h, uint64_t group, uint64_t peer_id, u32 kind, const char *who) {
    if (!h || group >= h->G) return rg_fail(RG_ERR_INVALID_ARG, "%s: bad argument", who);
    if (!h->host_mirror) return rg_fail(RG_ERR_STATE, "%s: rg_set_peers was never called", who);
    const int slot = rg_find_slot(h, group, peer_id);
    if (slot < 0) return RG_OK; // "no progress available for {}": ignored (the reference drops the step's result as well)
    u64 row = 0;
    memcpy(&row, &h->q_mf[group * 8], 8);
    if (row) return rg_fail(RG_ERR_SLOT_BUSY, "%s: group %llu has traffic queued; rg_flush first (local messages apply in call order)",
                            who, (unsigned long long)group);
    const rg_progress_event ev = {group, (u32)slot, kind};
    return rg_progress_events(h, &ev, 1);
}
extern "C" int rg_report_unreachable(rg_engine *h, uint64_t group, uint64_t peer_id) {
    return rg_report(h, group, peer_id, RG_EV_UNREACHABLE, "rg_report_unreachable");
}
extern "C" int rg_report_snapshot(rg_engine *h, uint64_t group, uint64_t peer_id, int failure) {
    return rg_report(h, group, peer_id, failure ? RG_EV_SNAPSHOT_FAILURE : RG_EV_SNAPSHOT_FINISH, "rg_report_snapshot");
}

extern "C" int rg_mark_sent(rg_engine *h, uint64_t group, uint64_t peer_id) {
    if (!h || group >= h->G) return rg_fail(RG_ERR_INVALID_ARG, "rg_mark_sent: bad argument");
    if (!h->host_mirror) return rg_fail(RG_ERR_STATE, "rg_mark_sent: rg_set_peers was never called");
    const int slot = rg_find_slot(h, group, peer_id);
    if (slot < 0) return rg_fail(RG_ERR_STEP_PEER_NOT_FOUND, "rg_mark_sent: peer %llu not in group %llu",
                                 (unsigned long long)peer_id, (unsigned long long)group);
    bool to_self;
    int src = rg_from_self(h, group, slot, &to_self);
    if (src) return src;
    if (to_self) return RG_OK; // the leader sends itself nothing (and SENT has no meaning on its slot)
    rg_touch(h, group);
    h->q_mf[group * 8 + slot] |= RG_MF_SENT;
    return RG_OK;
}

// One sparse tick in ONE host<->device round trip: records (the caller's, or built from the mirror's queues when
// `recs` is NULL) -> pinned staging -> ingest / clear / hint resolve / tick / gather back to back -> one packed D2H
// copy of (groups, duplicates, {group, commit, out}...) -> one synchronisation. Results stay cached on the host.
struct rg_send_req { // run the send stage inside the same round trip (engines with device Inflights)
    u64 max_entries;
    u32 flags;
};

static int rg_sparse_threecall(rg_engine *h, const rg_wire_msg *recs, u64 n, u32 *dup_out, const rg_send_req *send) {
    // big batches are throughput-bound, not latency-bound: the packed copy (one slot per RECORD, not per group)
    // and the host-side unpacking cost more than two extra synchronisations (profiles/r01_sparse_path...)
    uint64_t d64 = 0, ng = 0;
    int rc = rg_ingest(h, recs, n, &d64);
    if (rc == RG_OK) rc = rg_tick_ingested(h, &ng);
    if (rc == RG_OK && send) rc = rg_send_appends(h, send->max_entries, send->flags);
    if (dup_out) *dup_out = (u32)d64;
    return rc;
}

static int rg_mailbox_flush(rg_engine *h, u64 n, bool any_logterm, bool *served, const rg_send_req *send);
static int rg_sparse_roundtrip(rg_engine *h, const rg_wire_msg *recs, u64 n, bool any_logterm, u32 *dup_out,
                               const rg_send_req *send = nullptr) {
    RG_HIP(hipSetDevice(h->cfg.device)); // (not RG_ENTER: this is the one path the resident mailbox kernel serves)
    int rc;
    if (h->ins_arena && h->hint_check_due) { // (rare: a log-term tick came before; the check needs the stream to itself)
        rc = rg_mailbox_quiesce(h);
        if (rc) return rc;
        rc = rg_require_hints_resolved(h, "rg_flush / rg_ingest_tick");
        if (rc) return rc;
    }
    rc = rg_ensure_sparse(h);
    if (rc) return rc;
    if (recs && n > RG_ROUNDTRIP_MAX) return rg_sparse_threecall(h, recs, n, dup_out, send);
    if (!recs) {
        n = 0;
        for (u64 g : h->q_dirty)
            for (u32 p = 0; p < h->P; p++) n += h->q_mf[g * 8 + p] != 0;
    }
    if (n > RG_INGEST_BLOCK || (send && !h->ins_arena)) { // not a flush the resident mailbox workgroup can serve: it leaves
        rc = rg_mailbox_quiesce(h);                         // now, before anything below waits for the stream or replaces
        if (rc) return rc;                                  // a buffer it reads
    }
    if (n > h->pin_records_cap) {
        if (h->pin_records) {
            RG_HIP(hipStreamSynchronize(h->stream));
            (void)hipHostFree(h->pin_records);
            h->pin_records = nullptr;
        }
        u64 cap = 4096;
        while (cap < n) cap *= 2;
        RG_HIP(hipHostMalloc(reinterpret_cast<void **>(&h->pin_records), cap * sizeof(rg_wire_msg), hipHostMallocDefault));
        h->pin_records_cap = cap;
    }
    if (recs) {
        if (n) memcpy(h->pin_records, recs, n * sizeof(rg_wire_msg));
    } else {
        u64 k = 0;
        for (u64 g : h->q_dirty) {
            for (u32 p = 0; p < h->P; p++) {
                const u8 f = h->q_mf[g * 8 + p];
                if (!f) continue;
                const size_t o = (size_t)p * h->stride + g;
                rg_wire_msg &r = h->pin_records[k++];
                r.group = g;
                r.index = h->q_mi[o];
                r.commit = h->q_mc[o];
                r.hint = h->q_mh[o];
                r.rs = h->q_mrs[o];
                r.log_term = h->q_mlt[o];
                r.slot = p;
                r.flags = f;
                r.pad = 0;
            }
        }
        if (n > RG_ROUNDTRIP_MAX) return rg_sparse_threecall(h, h->pin_records, n, dup_out, send);
    }
    // the resident mailbox kernel, when it is on: no launch, no synchronisation (rg_mailbox_flush says whether it took it)
    bool served = false;
    if (h->mbox_on) {
        rc = rg_mailbox_flush(h, n, any_logterm, &served, send);
        if (rc) return rc;
    }
    if (!served) {
        rc = rg_mailbox_quiesce(h); // this flush goes through launches on the stream
        if (rc) return rc;
    }
    u32 n_groups = 0, dup = 0;
    u64 upper = 0;
    bool fetch_items = false;
    if (served) {
        h->out_is_dense = false; // (what rg_sparse_enqueue records)
        h->tick_launches++;
        rg_ctr_flip(h);
        n_groups = reinterpret_cast<const u32 *>(h->pin_packed)[0];
        dup = reinterpret_cast<const u32 *>(h->pin_packed)[1];
        if (send) { // the request ran the stage of every touched group: its items are in pin_send (rg_tick_send_listed)
            upper = n; // (only "something was walked", below)
            fetch_items = true;
            h->send_cols_fresh = false;
            h->send_last_dense = false;
            h->host_items_valid = false;
        }
    } else {
    if (n > h->d_records_cap) {
        if (h->d_records) {
            RG_HIP(hipStreamSynchronize(h->stream));
            (void)hipFree(h->d_records);
            h->d_records = nullptr;
        }
        u64 cap = h->d_records_cap ? h->d_records_cap : 4096;
        while (cap < n) cap *= 2;
        RG_HIP(hipMalloc(&h->d_records, (cap + RG_INGEST_BLOCK) * sizeof(rg_wire_msg)));
        h->d_records_cap = cap;
    }
    const u64 upper_all = h->ingested_upper + n; // device ingests of this window count too
    upper = upper_all < h->G ? upper_all : h->G;
    if (upper > h->packed_cap) {
        if (h->d_packed) {
            RG_HIP(hipStreamSynchronize(h->stream));
            (void)hipFree(h->d_packed);
            (void)hipHostFree(h->pin_packed);
            h->d_packed = h->pin_packed = nullptr;
        }
        u64 cap = 4096;
        while (cap < upper) cap *= 2;
        RG_HIP(hipMalloc(&h->d_packed, RG_PACKED_HDR + cap * sizeof(rg_res_rec)));
        RG_HIP(hipHostMalloc(reinterpret_cast<void **>(&h->pin_packed), RG_PACKED_HDR + cap * sizeof(rg_res_rec),
                             hipHostMallocDefault));
        h->packed_cap = cap;
    }
    // Small batches are latency-bound: every HIP call costs the host 3-5 us. The kernels then read the records straight
    // out of the pinned staging buffer and write the packed results straight into pinned host memory (both are mapped into
    // the device's address space; a few KB over PCIe inside a kernel cost less than a copy command each way), and the
    // ingest kernel also zeroes the previous sparse tick's result words: ingest + tick + counter reset + ONE
    // synchronisation instead of copy, ingest, clear, tick, copy, reset, synchronisation.
    const bool zero_copy = n && upper <= RG_ZEROCOPY_MAX;
    // ... and up to one workgroup's worth of records the whole flush is ONE launch (k_flush_small)
    const bool one_launch = zero_copy && n <= RG_INGEST_BLOCK && !(h->ins_arena && h->send_ready);
    bool out_cleared = false;
    RgIngest fused_args;
    if (one_launch) {
        RgClear clr = {nullptr, nullptr, 0u, rg_ctr_other(h)};
        if (!h->out_is_dense) {
            clr.list = h->res_list;
            clr.out = h->st.out;
            clr.n = (u32)h->last_sparse_n;
            out_cleared = true;
        }
        fused_args = rg_ingest_args(h, h->pin_records, n, clr);
    } else if (n) {
        RgClear clr = {nullptr, nullptr, 0u, rg_ctr_other(h)};
        // (with device Inflights and an unconsumed send stage the previous result words are still needed: rg_settle_send)
        if (zero_copy && !h->out_is_dense && !(h->ins_arena && h->send_ready)) {
            clr.list = h->res_list;
            clr.out = h->st.out;
            clr.n = (u32)h->last_sparse_n;
            out_cleared = true;
        }
        const rg_wire_msg *src = h->pin_records;
        if (!zero_copy) {
            RG_HIP(hipMemcpyAsync(h->d_records, h->pin_records, n * sizeof(rg_wire_msg), hipMemcpyHostToDevice, h->stream));
            src = h->d_records;
        }
        hipLaunchKernelGGL(k_ingest, dim3(rg_grid(n, RG_INGEST_BLOCK)), dim3(RG_INGEST_BLOCK), 0, h->stream,
                           rg_ingest_args(h, src, n, clr));
    }
    // (records ingested on the device in this window may carry log terms the host has not seen)
    // One launch AND a send request: the stage of every touched group runs behind its tick inside k_flush_small_send, on the
    // tick's registers; its work items land in the device list and, through the mapped pinned buffer, in host memory.
    const bool stage_inside = one_launch && send;
    RgSmallSend small_send;
    if (stage_inside) {
        if (!h->pin_send)
            RG_HIP(hipHostMalloc(reinterpret_cast<void **>(&h->pin_send), 16 + RG_SEND_SPEC * sizeof(rg_send_item),
                                 hipHostMallocDefault));
        small_send.ins = h->ins;
        small_send.max_entries = send->max_entries;
        small_send.flags = send->flags;
        h->stage_max_entries = send->max_entries;
        h->stage_flags = send->flags;
        small_send.items = h->send_items;
        small_send.counter = h->send_counter;
        small_send.pin = h->pin_send;
        h->send_cols_fresh = false; // (what rg_send_enqueue records for a stage over a list)
        h->send_last_dense = false;
        h->host_items_valid = false;
    }
    rc = rg_sparse_enqueue(h, upper, zero_copy ? h->pin_packed : h->d_packed, any_logterm || h->ingested_upper != 0, out_cleared,
                           one_launch ? &fused_args : nullptr, stage_inside ? &small_send : nullptr);
    if (rc) return rc;
    // the send stage rides along: it walks the gathered list, whose length is still only on the device
    const u64 item_bound = upper * h->P;
    fetch_items = send && upper && (stage_inside || item_bound <= RG_SEND_SPEC);
    if (send && upper && !stage_inside) {
        rc = rg_send_enqueue(h, send->max_entries, send->flags, h->res_list, upper, h->counters);
        if (rc) return rc;
        if (fetch_items) {
            if (!h->pin_send)
                RG_HIP(hipHostMalloc(reinterpret_cast<void **>(&h->pin_send), 16 + RG_SEND_SPEC * sizeof(rg_send_item),
                                     hipHostMallocDefault));
            RG_HIP(hipMemcpyAsync(h->pin_send, h->send_counter, 4, hipMemcpyDeviceToHost, h->stream));
            RG_HIP(hipMemcpyAsync(h->pin_send + 16, h->send_items, item_bound * sizeof(rg_send_item), hipMemcpyDeviceToHost,
                                  h->stream));
        }
    }
    if (upper) {
        if (!zero_copy)
            RG_HIP(hipMemcpyAsync(h->pin_packed, h->d_packed, RG_PACKED_HDR + upper * sizeof(rg_res_rec), hipMemcpyDeviceToHost,
                                  h->stream));
        RG_HIP(hipStreamSynchronize(h->stream));
        rg_ctr_flip(h); // (the next window's pair was reset by this window's ingest kernel)
        n_groups = reinterpret_cast<const u32 *>(h->pin_packed)[0];
        dup = reinterpret_cast<const u32 *>(h->pin_packed)[1];
    }
    } // (!served)
    rc = rg_sparse_finish(h, n_groups);
    if (rc) return rc;
    if (send) {
        h->send_ready = false; // the stage of this tick has run (or had nothing to walk)
        h->send_bound = (u64)n_groups * h->P;
        if (!upper) RG_HIP(hipMemsetAsync(h->send_counter, 0, 4, h->stream));
        if (!upper) {
            h->host_items.clear();
            h->host_items_valid = true;
        } else if (fetch_items) {
            const u32 cnt = *reinterpret_cast<const u32 *>(h->pin_send);
            const rg_send_item *it = reinterpret_cast<const rg_send_item *>(h->pin_send + 16);
            h->host_items.assign(it, it + cnt);
            h->host_items_valid = true;
        }
    }
    if (dup_out) *dup_out = dup;
    const rg_res_rec *rec = reinterpret_cast<const rg_res_rec *>(h->pin_packed + RG_PACKED_HDR);
    h->host_res_groups.resize(n_groups);
    h->host_res_commit.resize(n_groups);
    h->host_res_out.resize(n_groups);
    for (u32 i = 0; i < n_groups; i++) {
        h->host_res_groups[i] = rec[i].group;
        h->host_res_commit[i] = rec[i].commit;
        h->host_res_out[i] = rec[i].out;
    }
    h->host_res_valid = true;
    return RG_OK;
}

// ---- the resident small-batch path (rg_mailbox_start; kernel: k_mailbox in rg_tick_kernels.h) ----
#define RG_MBOX_TICKS_PER_US 100ull /* wall_clock64(): constant 100 MHz */
#define RG_MBOX_MAX_US 200000ull    /* one launch never stays longer than this, whatever the host does */

static int rg_mailbox_quiesce(rg_engine *h) {
    if (!h->mbox_running) return RG_OK;
    __atomic_store_n(&h->mbox->stop, 1u, __ATOMIC_RELEASE);
    hipError_t e = hipStreamSynchronize(h->stream);
    h->mbox_running = false;
    __atomic_store_n(&h->mbox->stop, 0u, __ATOMIC_RELEASE);
    if (e != hipSuccess) return rg_fail(RG_ERR_NO_DEVICE, "mailbox: %s", hipGetErrorString(e));
    return RG_OK;
}

static int rg_mailbox_launch(rg_engine *h) {
    RgMsgs ms = h->staged;
    ms.mhr = ms.mh;
    RgListOut lo;
    lo.rl = h->res_list;
    lo.rc = h->res_commit;
    lo.ro = h->res_out;
    lo.packed = h->pin_packed;
    RgClear clr = {h->res_list, h->st.out, 0u, nullptr};
    const RgIngest a0 = rg_ingest_args(h, h->pin_records, 0, clr);
    u64 *mf = (u64 *)h->staged.mflags;
    const u64 max_ticks = RG_MBOX_MAX_US * RG_MBOX_TICKS_PER_US;
    RgSmallSend ss0; // where a request's send stage (rg_flush_send) puts its work items; limit and flags come with the request
    memset(&ss0, 0, sizeof(ss0));
    if (h->ins_arena) {
        ss0.ins = h->ins;
        ss0.items = h->send_items;
        ss0.counter = h->send_counter;
        ss0.pin = h->pin_send;
    }
    switch (h->P) {
    case 1: rg_launch_mailbox_t<1>(h->stream, h->st, ms, h->any_group_commit, a0, h->counters_base, h->rhint, mf, lo, h->mbox, h->mbox_idle_ticks, max_ticks, ss0); break;
    case 2: rg_launch_mailbox_t<2>(h->stream, h->st, ms, h->any_group_commit, a0, h->counters_base, h->rhint, mf, lo, h->mbox, h->mbox_idle_ticks, max_ticks, ss0); break;
    case 3: rg_launch_mailbox_t<3>(h->stream, h->st, ms, h->any_group_commit, a0, h->counters_base, h->rhint, mf, lo, h->mbox, h->mbox_idle_ticks, max_ticks, ss0); break;
    case 4: rg_launch_mailbox_t<4>(h->stream, h->st, ms, h->any_group_commit, a0, h->counters_base, h->rhint, mf, lo, h->mbox, h->mbox_idle_ticks, max_ticks, ss0); break;
    case 5: rg_launch_mailbox_t<5>(h->stream, h->st, ms, h->any_group_commit, a0, h->counters_base, h->rhint, mf, lo, h->mbox, h->mbox_idle_ticks, max_ticks, ss0); break;
    case 6: rg_launch_mailbox_t<6>(h->stream, h->st, ms, h->any_group_commit, a0, h->counters_base, h->rhint, mf, lo, h->mbox, h->mbox_idle_ticks, max_ticks, ss0); break;
    case 7: rg_launch_mailbox_t<7>(h->stream, h->st, ms, h->any_group_commit, a0, h->counters_base, h->rhint, mf, lo, h->mbox, h->mbox_idle_ticks, max_ticks, ss0); break;
    default: rg_launch_mailbox_t<8>(h->stream, h->st, ms, h->any_group_commit, a0, h->counters_base, h->rhint, mf, lo, h->mbox, h->mbox_idle_ticks, max_ticks, ss0); break;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return rg_fail(RG_ERR_NO_DEVICE, "mailbox: launch failed: %s", hipGetErrorString(e));
    h->mbox_running = true;
    h->mbox_launches++;
    return RG_OK;
}

// One small flush through the mailbox: the records are in h->pin_records already. Returns RG_OK with *served = false
// when the request cannot go this way (the caller takes the launch path).
static int rg_mailbox_flush(rg_engine *h, u64 n, bool any_logterm, bool *served, const rg_send_req *send) {
    *served = false;
    // the body clears the PREVIOUS sparse tick's result words itself; a dense predecessor needs a memset on the stream
    if (!h->mbox_on || h->pub || h->out_is_dense || h->ingested_upper || n == 0 || n > RG_INGEST_BLOCK ||
        h->last_sparse_n > RG_ZEROCOPY_MAX || h->epoch == 0xffffffffu)
        return RG_OK;
    // device Inflights: only a flush that runs its send stage in the same request (rg_flush_send), with no stage of an
    // earlier tick left to settle (that one needs a launch), and a limit the request word can carry
    u32 lim = 0;
    if (h->ins_arena) {
        // (a tick that can raise RG_OUT_HOST_HINT, or one that follows such a tick, takes the launch path: rg_require_hints_resolved)
        if (any_logterm || h->hint_check_due) return RG_OK;
        if (!send || h->send_ready) return RG_OK;
        if (send->max_entries == ~0ULL) lim = 0xffffffffu;
        else if (send->max_entries >= 0xffffffffULL) return RG_OK;
        else lim = (u32)send->max_entries;
    } else if (send) {
        return RG_OK;
    }
    RgMbox *mb = h->mbox;
    if (h->mbox_running && !__atomic_load_n(&mb->alive, __ATOMIC_ACQUIRE) &&
        __atomic_load_n(&mb->seq_done, __ATOMIC_ACQUIRE) == h->mbox_seq) {
        // the instance has left (idle / lifetime): let the stream see it end before the next one goes on
        int rc = rg_mailbox_quiesce(h);
        if (rc) return rc;
    }
    if (send) {
        h->stage_max_entries = send->max_entries;
        h->stage_flags = send->flags;
    }
    const u32 s = ++h->mbox_seq;
    // (RgMbox: three self-validating words, one 8-byte store each; the records in pin_records are older stores)
    const u32 w0 = (u32)n | ((h->counters == h->counters_base ? 0u : 1u) << 16) | ((any_logterm ? 1u : 0u) << 17) |
                   ((send ? 1u : 0u) << 18) | ((send ? (send->flags & 3u) : 0u) << 19);
    __atomic_store_n(&mb->w[0], rg_mbox_word(s, w0), __ATOMIC_RELEASE);
    __atomic_store_n(&mb->w[1], rg_mbox_word(s, h->epoch), __ATOMIC_RELEASE);
    __atomic_store_n(&mb->w[2], rg_mbox_word(s, (u32)h->last_sparse_n), __ATOMIC_RELEASE);
    __atomic_store_n(&mb->w[3], rg_mbox_word(s, lim), __ATOMIC_RELEASE);
    if (!h->mbox_running) {
        int rc = rg_mailbox_launch(h);
        if (rc) return rc;
    }
    // spin on the answer; an instance that left without serving the request (it timed out as the request arrived) is
    // replaced -- the new one starts from seq_done and finds the request waiting
    u64 spins = 0;
    while (__atomic_load_n(&mb->seq_done, __ATOMIC_ACQUIRE) != s) {
        if ((++spins & 0xfffu) == 0) {
            if (hipStreamQuery(h->stream) != hipErrorNotReady) { // the kernel is gone (left, or failed)
                if (__atomic_load_n(&mb->seq_done, __ATOMIC_ACQUIRE) == s) break;
                h->mbox_running = false;
                hipError_t e = hipStreamSynchronize(h->stream);
                if (e != hipSuccess) return rg_fail(RG_ERR_NO_DEVICE, "mailbox: %s", hipGetErrorString(e));
                int rc = rg_mailbox_launch(h);
                if (rc) return rc;
            }
            if (spins > (1ull << 33)) return rg_fail(RG_ERR_NO_DEVICE, "mailbox: no answer from the device");
        }
        __builtin_ia32_pause();
    }
    *served = true;
    h->mbox_served++;
    return RG_OK;
}

extern "C" int rg_mailbox_start(rg_engine *h, uint32_t idle_timeout_us) {
    if (!h) return rg_fail(RG_ERR_INVALID_ARG, "rg_mailbox_start: null engine");
    RG_ENTER(h);
    int rc = rg_ensure_sparse(h);
    if (rc) return rc;
    if (!h->mbox) {
        RG_HIP(hipHostMalloc(reinterpret_cast<void **>(&h->mbox), sizeof(RgMbox), hipHostMallocCoherent | hipHostMallocMapped));
        memset(h->mbox, 0, sizeof(RgMbox));
    }
    // the kernel's arguments are fixed at launch: the staging buffers it reads / writes have to exist at their final size
    if (!h->pin_records) {
        RG_HIP(hipHostMalloc(reinterpret_cast<void **>(&h->pin_records), 4096 * sizeof(rg_wire_msg), hipHostMallocDefault));
        h->pin_records_cap = 4096;
    }
    if (!h->pin_packed) {
        RG_HIP(hipMalloc(&h->d_packed, RG_PACKED_HDR + 4096 * sizeof(rg_res_rec)));
        RG_HIP(hipHostMalloc(reinterpret_cast<void **>(&h->pin_packed), RG_PACKED_HDR + 4096 * sizeof(rg_res_rec), hipHostMallocDefault));
        h->packed_cap = 4096;
    }
    if (h->ins_arena && !h->pin_send) // device Inflights: rg_flush_send's work items come back through this buffer
        RG_HIP(hipHostMalloc(reinterpret_cast<void **>(&h->pin_send), 16 + RG_SEND_SPEC * sizeof(rg_send_item), hipHostMallocDefault));
    h->mbox_idle_ticks = (u64)(idle_timeout_us ? idle_timeout_us : 2000u) * RG_MBOX_TICKS_PER_US;
    h->mbox_on = true;
    return RG_OK;
}

extern "C" int rg_mailbox_stats(const rg_engine *h, uint64_t *flushes_served, uint64_t *launches) {
    if (!h) return rg_fail(RG_ERR_INVALID_ARG, "rg_mailbox_stats: null engine");
    if (flushes_served) *flushes_served = h->mbox_served;
    if (launches) *launches = h->mbox_launches;
    return RG_OK;
}

extern "C" int rg_mailbox_stop(rg_engine *h) {
    if (!h) return rg_fail(RG_ERR_INVALID_ARG, "rg_mailbox_stop: null engine");
    RG_ENTER(h);
    h->mbox_on = false;
    return RG_OK;
}

static int rg_flush_sparse(rg_engine *h, const rg_send_req *send) {
    u32 dup = 0;
    int rc = rg_sparse_roundtrip(h, nullptr, 0, h->q_any_logterm, &dup, send);
    if (rc) return rc;
    if (dup) return rg_fail(RG_ERR_STATE, "rg_flush: %u duplicate cells (internal error)", dup);
    return RG_OK;
}

extern "C" int rg_ingest_tick(rg_engine *h, const rg_wire_msg *records, uint64_t n, uint64_t *n_groups,
                              uint64_t *n_duplicates) {
    if (!h || (!records && n)) return rg_fail(RG_ERR_INVALID_ARG, "rg_ingest_tick: bad argument");
    if (n_groups) *n_groups = 0;
    if (n_duplicates) *n_duplicates = 0;
    static const rg_wire_msg none = {};
    u32 dup = 0;
    int rc = rg_sparse_roundtrip(h, records ? records : &none, n, true, &dup);
    if (rc) return rc;
    if (n_groups) *n_groups = h->last_sparse_n;
    if (n_duplicates) *n_duplicates = dup;
    return RG_OK;
}

// The device is the judge of an RG_MF_BECOME_LEADER event: it validates the new term against RG_COL_CUR_TERM (which the
// host may have reloaded or restored since rg_set_peers registered a term) and answers RG_OUT_BECAME_LEADER, or
// RG_OUT_FAULT with the event ignored. rg_local_become_leader had to move the host's term gate when the event was QUEUED
// (responses of the new term may follow in the same flush); where the device refused, the gate goes back to the old term --
// otherwise responses of the old term would be dropped and those of the new one applied to a Progress set that was never
// reset. RG_COL_CUR_TERM and the registered term of a group belong together: load one, register the other.
static int rg_settle_elections(rg_engine *h, bool results_available) {
    int rc = RG_OK;
    if (results_available && h->last_sparse_n) {
        const u64 n = h->last_sparse_n;
        std::vector<u64> groups(n);
        std::vector<u32> out(n);
        u64 got = 0;
        rc = rg_ingested_results(h, groups.data(), nullptr, out.data(), n, &got);
        if (rc == RG_OK) {
            std::unordered_map<u64, bool> became; // election groups of this flush -> did the device apply the event?
            for (const auto &e : h->q_elections) became[e.group] = false;
            for (u64 i = 0; i < n; i++) {
                auto it = became.find(groups[i]);
                if (it != became.end() && (out[i] & RG_OUT_BECAME_LEADER)) it->second = true;
            }
            // (newest first: should a group ever be listed twice, the OLDEST recorded term -- the registered one -- wins;
            // rg_local_become_leader refuses a second election of a group inside one flush, RG_ERR_SLOT_BUSY)
            for (auto e = h->q_elections.rbegin(); e != h->q_elections.rend(); ++e)
                if (!became[e->group]) h->terms[e->group] = e->old_term;
        }
    }
    if (!results_available || !h->last_sparse_n || rc != RG_OK) {
        // no verdict (the flush failed, or its results could not be read): the conservative side -- every gate goes back
        for (auto e = h->q_elections.rbegin(); e != h->q_elections.rend(); ++e) h->terms[e->group] = e->old_term;
    }
    h->q_elections.clear();
    return rc;
}

static int rg_flush_impl(rg_engine *h, const rg_send_req *send) {
    if (!h->host_mirror) return rg_fail(RG_ERR_STATE, "rg_flush: rg_set_peers was never called");
    int rc;
    const u64 launches0 = h->tick_launches;
    if (h->q_dirty.size() * 2 >= h->G) {
        // most groups have events: stream the whole columns through the dense tick (measured crossover with the
        // 64-B-record path is around 60 % of the groups: profiles/r01_sparse_path_and_recompute.txt, mirror_bench)
        rg_msgs m;
        m.m_index = h->q_mi.data();
        m.m_commit = h->q_mc.data();
        m.m_hint = h->q_mh.data();
        m.m_rs = h->q_mrs.data();
        m.m_logterm = h->q_any_logterm ? h->q_mlt.data() : nullptr;
        m.m_flags = h->q_mf.data();
        // (with a send request: the tick and its stage as ONE launch, k_tick_send)
        if (send) {
            const RgSendReq sr = {(u64)send->max_entries, (u32)send->flags};
            rc = rg_tick_host_impl(h, &m, &sr);
        } else {
            rc = rg_tick(h, &m);
        }
        // rg_ingested_results must work after ANY flush: gather the dirty groups' results compactly
        if (rc == RG_OK) rc = rg_ensure_sparse(h);
        if (rc == RG_OK) {
            const u32 n = (u32)h->q_dirty.size();
            hipError_t e = hipMemcpyAsync(h->list, h->q_dirty.data(), (size_t)n * 8, hipMemcpyHostToDevice, h->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(h->counters, &n, 4, hipMemcpyHostToDevice, h->stream);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(k_gather_results, dim3(rg_grid(n, 256)), dim3(256), 0, h->stream, h->list, h->counters,
                                   (const u64 *)h->st.commit, (const u32 *)h->st.out, h->res_list, h->res_commit, h->res_out,
                                   (char *)nullptr);
                e = hipMemsetAsync(h->counters, 0, 4, h->stream);
            }
            if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
            if (e != hipSuccess) rc = rg_fail(RG_ERR_NO_DEVICE, "rg_flush: %s", hipGetErrorString(e));
            else h->last_sparse_n = n;
        }
    } else {
        // few groups have events: ship only their records and tick only them -- ONE host<->device round trip
        // (pinned record staging, five back-to-back launches, one packed D2H copy, one synchronisation)
        rc = rg_flush_sparse(h, send);
    }
    // A flush that failed BEFORE its tick was enqueued changed nothing on the device: the queued events stay queued
    // and the call can be retried. Once the tick is enqueued the events are consumed (a retry would apply them
    // twice); an error after that point means results could not be fetched, not that the batch was lost.
    if (rc != RG_OK && h->tick_launches == launches0) return rc;
    for (u64 g : h->q_dirty) memset(&h->q_mf[g * 8], 0, 8);
    h->q_dirty.clear();
    h->q_any_logterm = false;
    if (!h->q_elections.empty()) {
        const int erc = rg_settle_elections(h, rc == RG_OK);
        if (rc == RG_OK) rc = erc;
    }
    return rc;
}

extern "C" int rg_flush(rg_engine *h) {
    if (!h) return rg_fail(RG_ERR_INVALID_ARG, "rg_flush: null engine");
    return rg_flush_impl(h, nullptr);
}

extern "C" int rg_flush_send(rg_engine *h, uint64_t max_entries_per_msg, uint32_t flags) {
    if (!h) return rg_fail(RG_ERR_INVALID_ARG, "rg_flush_send: null engine");
    if (!h->ins_arena)
        return rg_fail(RG_ERR_STATE, "rg_flush_send: engine created with max_inflight = 0 (Inflights are the host's)");
    if (flags & ~(RG_SEND_SKIP_BCAST_COMMIT | RG_SEND_BYTES)) return rg_fail(RG_ERR_INVALID_ARG, "rg_flush_send: unknown flags %#x", flags);
    if ((flags & RG_SEND_BYTES) && !h->esz)
        return rg_fail(RG_ERR_STATE, "rg_flush_send: RG_SEND_BYTES needs the entry sizes (rg_log_sizes_enable)");
    rg_send_req req;
    req.max_entries = max_entries_per_msg;
    req.flags = flags;
    return rg_flush_impl(h, &req);
}

// ------------------------------------------------------------------------------------------------
// synthetic stream
// ------------------------------------------------------------------------------------------------
extern "C" int rg_workload_init(rg_engine *h, const rg_workload *w, uint64_t first) {
    if (!h || !w) return rg_fail(RG_ERR_INVALID_ARG, "rg_workload_init: bad argument");
    if (w->workload != RG_WL_MAJORITY && w->workload != RG_WL_JOINT && w->workload != RG_WL_MIXED)
        return rg_fail(RG_ERR_INVALID_ARG, "rg_workload_init: unknown workload %u", w->workload);
    if ((w->reserved & 0xfu) > 8 || (w->reserved & ~0x1fu))
        return rg_fail(RG_ERR_INVALID_ARG, "rg_workload_init: fixed replica-set size %u / flags %#x", w->reserved & 0xfu, w->reserved & ~0xfu);
    RG_ENTER(h);
    hipLaunchKernelGGL(k_wl_init, dim3(rg_grid(h->G, RG_BLOCK)), dim3(RG_BLOCK), 0, h->stream, h->st, (u64)w->seed,
                       w->workload | (w->reserved << 8), h->P, (u64)first);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return rg_fail(RG_ERR_NO_DEVICE, "rg_workload_init: %s", hipGetErrorString(e));
    if (h->pub) h->pub->local_lost = true;
    h->host_cfg_valid = false;
    h->cls_stale = true;
    return RG_OK;
}

extern "C" int rg_workload_gen(rg_engine *h, const rg_workload *w, uint64_t first, uint64_t tick, uint64_t *mi,
                               uint64_t *mc, uint64_t *mh, uint64_t *mrs, uint8_t *mf) {
    if (!h || !w || !mi || !mc || !mh || !mrs || !mf) return rg_fail(RG_ERR_INVALID_ARG, "rg_workload_gen: bad argument");
    RG_ENTER(h);
    hipLaunchKernelGGL(k_wl_gen, dim3(rg_grid(h->G, RG_BLOCK)), dim3(RG_BLOCK), 0, h->stream, h->st, (u64)w->seed,
                       w->workload | (w->reserved << 8), h->P, (u64)first, (u64)tick, (u64 *)mi, (u64 *)mc, (u64 *)mh, (u64 *)mrs, mf);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return rg_fail(RG_ERR_NO_DEVICE, "rg_workload_gen: %s", hipGetErrorString(e));
    return RG_OK;
}

extern "C" int rg_workload_init_host(const rg_workload *w, uint64_t first, rg_host_state *s) {
    if (!w || !s || s->n_slots == 0 || s->n_slots > 8) return rg_fail(RG_ERR_INVALID_ARG, "rg_workload_init_host: bad argument");
    for (u64 g = 0; g < s->n_groups; g++)
        rg_wl_init_group(w->seed, w->workload | (w->reserved << 8), s->n_slots, s->stride, g,
                         first + rg_wl_place(w->workload | (w->reserved << 8), g, s->n_groups), (u64 *)s->match, (u64 *)s->next,
                         (u64 *)s->pr_commit, (u64 *)s->pend_snap, (u64 *)s->pend_rs, (u64 *)s->gid, s->pflags,
                         (u64 *)s->commit, (u64 *)s->term_lo, (u64 *)s->term_hi, s->cfg);
    return RG_OK;
}

extern "C" int rg_workload_gen_host(const rg_workload *w, uint64_t first, uint64_t tick, const rg_host_state *s,
                                    uint64_t *mi, uint64_t *mc, uint64_t *mh, uint64_t *mrs, uint8_t *mf) {
    if (!w || !s || !mi || !mc || !mh || !mrs || !mf) return rg_fail(RG_ERR_INVALID_ARG, "rg_workload_gen_host: bad argument");
    for (u64 g = 0; g < s->n_groups; g++)
        rg_wl_gen_group(w->seed, w->workload | (w->reserved << 8), s->n_slots, s->stride, g,
                        first + rg_wl_place(w->workload | (w->reserved << 8), g, s->n_groups), tick, (const u64 *)s->match,
                        (const u64 *)s->next, s->pflags, (const u64 *)s->commit, (const u64 *)s->term_lo,
                        (const u64 *)s->term_hi, (u64 *)mi,
                        (u64 *)mc, (u64 *)mh, (u64 *)mrs, mf);
    return RG_OK;
}

// ------------------------------------------------------------------------------------------------
// multi-GPU: publication of commit indices (encoding and replica kernels in rg_publish.h)
// ------------------------------------------------------------------------------------------------
// replica kernels (the arithmetic is rg_pub_apply8 in rg_publish.h, shared with the host twins)
__global__ __launch_bounds__(256) void k_pub_apply(u64 *replica, RgPubSlots sl, RgPubLayout l, u32 world) {
    const u64 per_rank = l.Gpad / 8;
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= per_rank * world) return;
    rg_pub_apply8(replica, sl, l, (u32)(i / per_rank), (i % per_rank) * 8);
}

// The exact-value lists of the same publications: one thread per (publication, rank, entry).
__global__ __launch_bounds__(256) void k_pub_apply_lists(u64 *replica, RgPubSlots sl, RgPubLayout l, u32 world, u32 *lost) {
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    const u64 per_slot = (u64)world * l.cap;
    if (i >= per_slot * sl.n) return;
    const u32 s = (u32)(i / per_slot), rank = (u32)((i % per_slot) / l.cap), k = (u32)(i % l.cap);
    const char *base = sl.slice[s] + (u64)rank * l.bytes_per_rank;
    const RgPubHdr *hdr = reinterpret_cast<const RgPubHdr *>(base);
    // a slice that asks for a resynchronisation (RG_PUB_LOST, or more list entries than fit)
    if (k == 0 && ((hdr->flags & RG_PUB_LOST) || hdr->n_overflow > l.cap)) atomicOr(lost, 1u);
    if (k >= hdr->n_overflow) return;
    const RgPubOvf e = reinterpret_cast<const RgPubOvf *>(base + l.off_list)[k];
    if (e.group < l.G) atomicAdd((unsigned long long *)&replica[(u64)rank * l.Gpad + e.group], (unsigned long long)e.extra);
}

static const char *rg_nccl_err(ncclResult_t r) { return g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?"; }

static int rg_pub_allgather(rg_engine *h, const void *send, void *recv, u64 bytes) {
    RgPub *p = h->pub;
    if (p->transport) {
        const int rc = p->transport(p->transport_user, send, recv, bytes, p->side);
        if (rc) return rg_fail(RG_ERR_NO_DEVICE, "rg_publish_commit: the custom all-gather transport failed (%d)", rc);
        return RG_OK;
    }
    const ncclResult_t r = g_rccl.AllGather(send, recv, (size_t)bytes, ncclUint8, p->comm, p->side);
    if (r != ncclSuccess) return rg_fail(RG_ERR_NO_DEVICE, "rg_publish_commit: ncclAllGather failed: %s", rg_nccl_err(r));
    return RG_OK;
}

// point the tick kernels at send buffer `b`
static void rg_pub_target(rg_engine *h, int b) {
    h->st.pub = h->pub->send[b];
    h->st.pub_off_delta = h->pub->lay.off_delta;
    h->st.pub_cap = h->pub->lay.cap;
}

// fold the buffered publications into the replica (side stream)
static int rg_pub_materialize(rg_engine *h) {
    RgPub *p = h->pub;
    if (!p->pending) return RG_OK;
    RgPubSlots sl;
    sl.n = p->pending;
    const u64 slot_bytes = (u64)p->world * p->lay.bytes_per_rank;
    for (u32 j = 0; j < p->pending; j++) // the `pending` most recent delta publications, ring order is irrelevant
        sl.slice[j] = p->ring_buf + (u64)j * slot_bytes;
    const u64 words = p->lay.Gpad / 8 * p->world;
    hipLaunchKernelGGL(k_pub_apply, dim3(rg_grid(words, 256)), dim3(256), 0, p->side, p->replica, sl, p->lay, p->world);
    const u64 entries = (u64)p->world * p->lay.cap * sl.n;
    hipLaunchKernelGGL(k_pub_apply_lists, dim3(rg_grid(entries, 256)), dim3(256), 0, p->side, p->replica, sl, p->lay,
                       p->world, p->d_lost);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return rg_fail(RG_ERR_NO_DEVICE, "rg_publish: replica update failed: %s", hipGetErrorString(e));
    p->pending = 0;
    p->stats.replica_updates++;
    return RG_OK;
}

static inline double rg_now_us() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

// One publication of one engine in three phases, so that a single host thread can drive several engines of ONE process
// through the same exchange (rg_publish_commit_all): `pre` of every engine (everything up to the exchange: the loss protocol's
// check point, the snapshot of a full publication, the side stream ordered behind the tick that completed the slice), the
// exchange of all of them together (RCCL: one ncclGroupStart / ncclAllGather x n / ncclGroupEnd -- the group is what keeps a
// single thread from blocking in rank 0's collective while rank 1's has not been issued; the in-process transport:
// device-to-device copies), `post` of every engine (the slice starts its next interval, rotation). rg_publish_commit is the
// three phases of one engine back to back.
struct RgPubStep {
    bool full;
    int b;
    const void *send;
    void *recv;
    u64 bytes;
    double t0, t1, t2;
};

static int rg_pub_pre(rg_engine *h, bool force_full, RgPubStep &s) {
    RgPub *p = h->pub;
    const u64 i = p->n_pub;
    const int b = (int)(i % RG_PUB_SEND);
    s.b = b;
    s.t0 = rg_now_us();
    // Loss protocol. Every `ring` publications is a CHECK POINT (the same publication numbers on every rank): all
    // buffered slices are folded into the replica first -- the update kernel raises d_lost for a slice that carries
    // RG_PUB_LOST or an overfull list -- and d_lost is copied to the host. The copy of the PREVIOUS check point
    // (finished long ago: no stall) decides whether this publication is a full snapshot. Every rank reads the same
    // gathered headers at the same publication numbers, so every rank decides the same without another collective.
    bool full = force_full;
    const bool check = (i % p->ring) == 0 && i != 0;
    if (check) {
        const int cb = (int)((i / p->ring) & 1), pb = cb ^ 1;
        if (p->chk_pending[pb]) {
            RG_HIP(hipEventSynchronize(p->ev_chk[pb]));
            p->chk_pending[pb] = false;
            if (p->pin_lost[pb]) full = true;
        }
        int rc = rg_pub_materialize(h);
        if (rc) return rc;
        RG_HIP(hipMemcpyAsync(&p->pin_lost[cb], p->d_lost, 4, hipMemcpyDeviceToHost, p->side));
        RG_HIP(hipMemsetAsync(p->d_lost, 0, 4, p->side));
        RG_HIP(hipEventRecord(p->ev_chk[cb], p->side));
        p->chk_pending[cb] = true;
    }
    if (p->local_lost && !p->lost_announced && !full) { // tell the other ranks (they act on it at a check point)
        static const u32 k_lost = RG_PUB_LOST;
        RG_HIP(hipMemcpyAsync(p->send[b] + offsetof(RgPubHdr, flags), &k_lost, 4, hipMemcpyHostToDevice, h->stream));
        p->lost_announced = true;
    }
    if (full) // snapshot the column before later ticks move it
        RG_HIP(hipMemcpyAsync(p->full_send, h->st.commit, h->G * 8, hipMemcpyDeviceToDevice, h->stream));
    RG_HIP(hipEventRecord(p->ev_tick[b], h->stream));
    RG_HIP(hipStreamWaitEvent(p->side, p->ev_tick[b], 0));
    s.full = full;
    if (full) {
        // the snapshot supersedes every buffered delta publication (and this interval's deltas)
        p->pending = 0;
        s.send = p->full_send;
        s.recv = p->replica;
        s.bytes = p->lay.Gpad * 8;
    } else {
        if (p->pending == p->ring) { // (reads between check points can leave the ring out of step with them)
            int rc = rg_pub_materialize(h);
            if (rc) return rc;
        }
        s.send = p->send[b];
        s.recv = p->ring_buf + (u64)p->pending * p->world * p->lay.bytes_per_rank;
        s.bytes = p->lay.bytes_per_rank;
    }
    s.t1 = rg_now_us();
    return RG_OK;
}

static int rg_pub_post(rg_engine *h, RgPubStep &s) {
    RgPub *p = h->pub;
    const int b = s.b;
#ifdef RG_PUB_DEBUG_BUILD /* measurement builds only (python -m raft_rs_amd.build --exp pubdbg -DRG_PUB_DEBUG_BUILD=1): the default library reads no environment */
    static const int dbg = getenv("RG_PUB_DEBUG") ? atoi(getenv("RG_PUB_DEBUG")) : 0; // measurement knobs (profiles/)
#else
    const int dbg = 0;
#endif
    if (s.full) {
        p->local_lost = false;
        p->lost_announced = false;
        p->stats.full_publications++;
        p->stats.bytes_per_rank_last = p->lay.Gpad * 8;
    } else {
        p->pending++;
        p->stats.bytes_per_rank_last = p->lay.bytes_per_rank;
    }
    s.t2 = rg_now_us();
    // this slice starts its next interval empty
    RG_HIP(hipMemsetAsync(p->send[b], 0, p->lay.bytes_per_rank, p->side));
    const double t3 = rg_now_us();
    RG_HIP(hipEventRecord(p->ev_done[b], p->side));
    p->done_pending[b] = true;
    // the ticks that follow accumulate into the next slice, once the exchange that last read it has let go of it
    const int nb = (b + 1) % RG_PUB_SEND;
    if (p->done_pending[nb]) {
        if (dbg & 1) RG_HIP(hipStreamWaitEvent(h->stream, p->ev_done[nb], 0)); // (the variant that was measured against)
        else RG_HIP(hipEventSynchronize(p->ev_done[nb]));
        p->done_pending[nb] = false;
    }
    rg_pub_target(h, nb);
    p->n_pub++;
    p->stats.publications++;
    const double t4 = rg_now_us();
    p->stats.host_us_events += (s.t1 - s.t0) + (t4 - t3);
    p->stats.host_us_allgather += s.t2 - s.t1;
    p->stats.host_us_memset += t3 - s.t2;
    return RG_OK;
}

static int rg_publish_impl(rg_engine *h, bool force_full) {
    if (h->pub->in_process)
        return rg_fail(RG_ERR_STATE, "rg_publish_commit: this engine is one of several ranks driven by ONE thread (rg_comm_init_all): "
                                     "publish through rg_publish_commit_all");
    RgPubStep s;
    int rc = rg_pub_pre(h, force_full, s);
    if (rc) return rc;
    rc = rg_pub_allgather(h, s.send, s.recv, s.bytes);
    if (rc) return rc;
    return rg_pub_post(h, s);
}

// ---- several engines of ONE process, driven by ONE thread: every engine is a rank of the same publication ----
// The exchange of all ranks in one go. RCCL: the n ncclAllGather calls inside one group (each on its engine's device and side
// stream). In-process transport (RG_COMM_ALL_LOCAL; engines that share a device, or a host that does not want RCCL): rank i's
// side stream copies every rank's slice into its gather buffer, device to device, behind the event that marks that slice
// complete; afterwards every rank's side stream waits for the others' copies of ITS slice, so that the slice is not reset
// (post) while somebody still reads it.
static int rg_pub_gather_all(rg_engine *const *e, uint32_t n, RgPubStep *st) {
    for (uint32_t i = 1; i < n; i++)
        if (st[i].bytes != st[0].bytes || st[i].full != st[0].full)
            return rg_fail(RG_ERR_STATE, "rg_publish_commit_all: the ranks disagree about the form of this publication "
                                         "(engine %u: %s, engine 0: %s) -- they must be published together, always", i,
                           st[i].full ? "full" : "delta", st[0].full ? "full" : "delta");
    if (e[0]->pub->comm) {
        ncclResult_t r = g_rccl.GroupStart();
        for (uint32_t i = 0; i < n && r == ncclSuccess; i++) {
            RG_HIP(hipSetDevice(e[i]->cfg.device));
            r = g_rccl.AllGather(st[i].send, st[i].recv, (size_t)st[i].bytes, ncclUint8, e[i]->pub->comm, e[i]->pub->side);
        }
        const ncclResult_t r2 = g_rccl.GroupEnd();
        if (r == ncclSuccess) r = r2;
        if (r != ncclSuccess) return rg_fail(RG_ERR_NO_DEVICE, "rg_publish_commit_all: grouped ncclAllGather failed: %s", rg_nccl_err(r));
        return RG_OK;
    }
    for (uint32_t i = 0; i < n; i++) {
        RgPub *p = e[i]->pub;
        RG_HIP(hipSetDevice(e[i]->cfg.device));
        for (uint32_t r = 0; r < n; r++) {
            if (r != i) RG_HIP(hipStreamWaitEvent(p->side, e[r]->pub->ev_tick[st[r].b], 0));
            RG_HIP(hipMemcpyAsync(reinterpret_cast<char *>(st[i].recv) + (u64)r * st[i].bytes, st[r].send, st[i].bytes,
                                  hipMemcpyDeviceToDevice, p->side));
        }
        RG_HIP(hipEventRecord(p->ev_read, p->side));
    }
    for (uint32_t i = 0; i < n; i++) {
        RG_HIP(hipSetDevice(e[i]->cfg.device));
        for (uint32_t r = 0; r < n; r++)
            if (r != i) RG_HIP(hipStreamWaitEvent(e[i]->pub->side, e[r]->pub->ev_read, 0));
    }
    return RG_OK;
}

static int rg_publish_all_impl(rg_engine *const *e, uint32_t n, bool force_full) {
    std::vector<RgPubStep> st(n);
    for (uint32_t i = 0; i < n; i++) {
        RG_HIP(hipSetDevice(e[i]->cfg.device));
        int rc = rg_mailbox_quiesce(e[i]);
        if (!rc) rc = rg_pub_pre(e[i], force_full, st[i]);
        if (rc) return rc;
    }
    int rc = rg_pub_gather_all(e, n, st.data());
    if (rc) return rc;
    for (uint32_t i = 0; i < n; i++) {
        RG_HIP(hipSetDevice(e[i]->cfg.device));
        rc = rg_pub_post(e[i], st[i]);
        if (rc) return rc;
    }
    return RG_OK;
}

extern "C" int rg_comm_unique_id(uint8_t id[RG_COMM_ID_BYTES]) {
    if (!id) return rg_fail(RG_ERR_INVALID_ARG, "rg_comm_unique_id: null argument");
    int rc = rg_rccl_load();
    if (rc) return rc;
    static_assert(RG_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "RG_COMM_ID_BYTES must match RCCL's unique id");
    ncclUniqueId u;
    const ncclResult_t r = g_rccl.GetUniqueId(&u);
    if (r != ncclSuccess) return rg_fail(RG_ERR_NO_DEVICE, "rg_comm_unique_id: ncclGetUniqueId failed: %s", rg_nccl_err(r));
    memcpy(id, u.internal, RG_COMM_ID_BYTES);
    return RG_OK;
}

extern "C" int rg_comm_destroy(rg_engine *h) {
    if (!h) return rg_fail(RG_ERR_INVALID_ARG, "rg_comm_destroy: null engine");
    RgPub *p = h->pub;
    if (!p) return RG_OK;
    (void)hipSetDevice(h->cfg.device);
    (void)rg_mailbox_quiesce(h);
    (void)hipStreamSynchronize(h->stream);
    if (p->side) (void)hipStreamSynchronize(p->side);
    if (p->comm) (void)g_rccl.CommDestroy(p->comm);
    for (int k = 0; k < RG_PUB_SEND; k++) {
        if (p->send[k]) (void)hipFree(p->send[k]);
        if (p->ev_tick[k]) (void)hipEventDestroy(p->ev_tick[k]);
        if (p->ev_done[k]) (void)hipEventDestroy(p->ev_done[k]);
    }
    for (int k = 0; k < 2; k++)
        if (p->ev_chk[k]) (void)hipEventDestroy(p->ev_chk[k]);
    if (p->ring_buf) (void)hipFree(p->ring_buf);
    if (p->replica) (void)hipFree(p->replica);
    if (p->full_send) (void)hipFree(p->full_send);
    if (p->d_lost) (void)hipFree(p->d_lost);
    if (p->pin_lost) (void)hipHostFree(p->pin_lost);
    if (p->ev_read) (void)hipEventDestroy(p->ev_read);
    if (p->side) (void)hipStreamDestroy(p->side);
    delete p;
    h->pub = nullptr;
    h->st.pub = nullptr;
    return RG_OK;
}

// Everything of rg_comm_init but the communicator and the first publication: buffers, streams, events.
// xdev: slices are read by OTHER devices (in-process transport across GPUs): the slice-complete events keep their system-scope fence.
static int rg_comm_setup(rg_engine *h, u32 rank, u32 world, u32 ring_ticks, u32 overflow_slots, rg_allgather_fn transport,
                         void *transport_user, bool in_process, bool xdev) {
    RgPub *p = new (std::nothrow) RgPub();
    if (!p) return rg_fail(RG_ERR_OUT_OF_MEMORY, "rg_comm_init: host allocation failed");
    memset(p, 0, sizeof(*p));
    p->rank = rank;
    p->world = world;
    p->transport = transport;
    p->transport_user = transport_user;
    p->in_process = in_process;
    p->ring = ring_ticks ? ring_ticks : 32;
    const u32 cap = overflow_slots ? overflow_slots : (u32)(h->G / 256 + 64);
    p->lay = rg_pub_layout(h->G, cap);
    h->pub = p;
#define RG_PUB_TRY(expr)                                                                                       \
    do {                                                                                                       \
        hipError_t e__ = (expr);                                                                               \
        if (e__ != hipSuccess) {                                                                               \
            (void)rg_comm_destroy(h);                                                                          \
            return rg_fail(e__ == hipErrorOutOfMemory ? RG_ERR_OUT_OF_MEMORY : RG_ERR_NO_DEVICE,               \
                           "rg_comm_init: %s failed: %s", #expr, hipGetErrorString(e__));                      \
        }                                                                                                      \
    } while (0)
    RG_PUB_TRY(hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking));
    // ev_tick orders the tick kernel before the exchange's first kernel ON THIS DEVICE (ncclAllGather reads the slice
    // with a kernel of this device; a host transport synchronises the device itself), so the system-scope fence a
    // recorded event normally implies -- an L2 write-back worth ~2 us per tick -- is not needed
    // (RG_PUB_DEBUG & 4 keeps it, for A/B measurements: profiles/r02_publish_overhead.txt; so does the in-process
    // transport between DIFFERENT devices, whose copies read the slice from the other GPU)
#ifdef RG_PUB_DEBUG_BUILD
    const unsigned evf = hipEventDisableTiming | ((xdev || (getenv("RG_PUB_DEBUG") && (atoi(getenv("RG_PUB_DEBUG")) & 4))) ? 0 : hipEventDisableSystemFence);
#else
    const unsigned evf = hipEventDisableTiming | (xdev ? 0u : (unsigned)hipEventDisableSystemFence);
#endif
    for (int k = 0; k < RG_PUB_SEND; k++) {
        RG_PUB_TRY(hipMalloc(&p->send[k], p->lay.bytes_per_rank));
        RG_PUB_TRY(hipMemsetAsync(p->send[k], 0, p->lay.bytes_per_rank, h->stream));
        RG_PUB_TRY(hipEventCreateWithFlags(&p->ev_tick[k], evf));
        RG_PUB_TRY(hipEventCreateWithFlags(&p->ev_done[k], hipEventDisableTiming));
    }
    for (int k = 0; k < 2; k++) RG_PUB_TRY(hipEventCreateWithFlags(&p->ev_chk[k], hipEventDisableTiming));
    RG_PUB_TRY(hipEventCreateWithFlags(&p->ev_read, hipEventDisableTiming));
    RG_PUB_TRY(hipMalloc(&p->ring_buf, (size_t)p->ring * p->world * p->lay.bytes_per_rank));
    RG_PUB_TRY(hipMalloc(&p->replica, (size_t)p->world * p->lay.Gpad * 8));
    RG_PUB_TRY(hipMemsetAsync(p->replica, 0, (size_t)p->world * p->lay.Gpad * 8, h->stream));
    RG_PUB_TRY(hipMalloc(&p->full_send, p->lay.Gpad * 8));
    RG_PUB_TRY(hipMemsetAsync(p->full_send, 0, p->lay.Gpad * 8, h->stream));
    RG_PUB_TRY(hipMalloc(&p->d_lost, 256));
    RG_PUB_TRY(hipMemsetAsync(p->d_lost, 0, 256, h->stream));
    RG_PUB_TRY(hipHostMalloc(reinterpret_cast<void **>(&p->pin_lost), 64, hipHostMallocDefault));
    memset(p->pin_lost, 0, 64);
    RG_PUB_TRY(hipStreamSynchronize(h->stream));
#undef RG_PUB_TRY
    h->dev.engine_bytes += RG_PUB_SEND * p->lay.bytes_per_rank + (u64)p->ring * p->world * p->lay.bytes_per_rank +
                           (u64)p->world * p->lay.Gpad * 8 + p->lay.Gpad * 8;
    rg_pub_target(h, 0);
    return RG_OK;
}

extern "C" int rg_comm_init(rg_engine *h, const rg_comm_config *cfg) {
    if (!h || !cfg) return rg_fail(RG_ERR_INVALID_ARG, "rg_comm_init: null argument");
    if (h->pub) return rg_fail(RG_ERR_STATE, "rg_comm_init: already initialised (rg_comm_destroy first)");
    if (cfg->world == 0 || cfg->rank >= cfg->world)
        return rg_fail(RG_ERR_INVALID_ARG, "rg_comm_init: rank %u of %u", cfg->rank, cfg->world);
    if (!cfg->transport && !cfg->unique_id)
        return rg_fail(RG_ERR_INVALID_ARG, "rg_comm_init: the RCCL transport needs the unique id of rg_comm_unique_id");
    if (cfg->ring_ticks > RG_PUB_MAX_RING)
        return rg_fail(RG_ERR_INVALID_ARG, "rg_comm_init: ring_ticks %u, at most %d", cfg->ring_ticks, RG_PUB_MAX_RING);
    RG_ENTER(h);
    int rc = rg_comm_setup(h, cfg->rank, cfg->world, cfg->ring_ticks, cfg->overflow_slots, cfg->transport, cfg->transport_user, false, false);
    if (rc) return rc;
    RgPub *p = h->pub;
    if (!cfg->transport) {
        rc = rg_rccl_load();
        if (rc) {
            (void)rg_comm_destroy(h);
            return rc;
        }
        ncclUniqueId u;
        memcpy(u.internal, cfg->unique_id, RG_COMM_ID_BYTES);
        const ncclResult_t r = g_rccl.CommInitRank(&p->comm, (int)p->world, u, (int)p->rank);
        if (r != ncclSuccess) {
            p->comm = nullptr;
            (void)rg_comm_destroy(h);
            return rg_fail(RG_ERR_NO_DEVICE, "rg_comm_init: ncclCommInitRank(rank %u of %u) failed: %s", cfg->rank, cfg->world,
                           rg_nccl_err(r));
        }
    }
    // every replica starts from the actual columns: one full publication (a collective: all ranks are in here)
    rc = rg_publish_impl(h, true);
    if (rc) {
        (void)rg_comm_destroy(h);
        return rc;
    }
    return RG_OK;
}

static int rg_all_check(rg_engine *const *engines, uint32_t n, const char *who, bool need_pub) {
    if (!engines || n == 0) return rg_fail(RG_ERR_INVALID_ARG, "%s: no engines", who);
    for (uint32_t i = 0; i < n; i++) {
        if (!engines[i]) return rg_fail(RG_ERR_INVALID_ARG, "%s: engine %u is null", who, i);
        for (uint32_t j = 0; j < i; j++)
            if (engines[j] == engines[i]) return rg_fail(RG_ERR_INVALID_ARG, "%s: engine %u is listed twice", who, i);
        if (engines[i]->G != engines[0]->G)
            return rg_fail(RG_ERR_INVALID_ARG, "%s: engine %u holds %llu groups, engine 0 %llu (equal shards: the all-gather moves equal slices)",
                           who, i, (unsigned long long)engines[i]->G, (unsigned long long)engines[0]->G);
        if (need_pub && (!engines[i]->pub || !engines[i]->pub->in_process || engines[i]->pub->world != n || engines[i]->pub->rank != i))
            return rg_fail(RG_ERR_STATE, "%s: engine %u is not rank %u of %u of an rg_comm_init_all communicator", who, i, i, n);
    }
    return RG_OK;
}

extern "C" int rg_comm_init_all(rg_engine *const *engines, uint32_t n, const rg_comm_all_config *cfg) {
    int rc = rg_all_check(engines, n, "rg_comm_init_all", false);
    if (rc) return rc;
    rg_comm_all_config c = {0, 0, RG_COMM_ALL_AUTO, 0};
    if (cfg) c = *cfg;
    if (c.transport > RG_COMM_ALL_LOCAL) return rg_fail(RG_ERR_INVALID_ARG, "rg_comm_init_all: unknown transport %u", c.transport);
    if (c.ring_ticks > RG_PUB_MAX_RING) return rg_fail(RG_ERR_INVALID_ARG, "rg_comm_init_all: ring_ticks %u, at most %d", c.ring_ticks, RG_PUB_MAX_RING);
    bool shared = false, xdev = false; // two engines on one device (RCCL refuses that) / engines on different devices
    for (uint32_t i = 0; i < n; i++) {
        if (engines[i]->pub) return rg_fail(RG_ERR_STATE, "rg_comm_init_all: engine %u already has a communicator (rg_comm_destroy first)", i);
        for (uint32_t j = 0; j < i; j++) {
            shared = shared || engines[j]->cfg.device == engines[i]->cfg.device;
            xdev = xdev || engines[j]->cfg.device != engines[i]->cfg.device;
        }
    }
    if (c.transport == RG_COMM_ALL_RCCL && shared)
        return rg_fail(RG_ERR_INVALID_ARG, "rg_comm_init_all: RCCL needs one device per rank; two of the engines share one (RG_COMM_ALL_LOCAL)");
    const bool rccl = c.transport == RG_COMM_ALL_RCCL || (c.transport == RG_COMM_ALL_AUTO && !shared && n > 1);
    auto undo = [&](uint32_t upto) {
        for (uint32_t i = 0; i < upto; i++) (void)rg_comm_destroy(engines[i]);
    };
    for (uint32_t i = 0; i < n; i++) {
        rg_engine *h = engines[i];
        RG_HIP(hipSetDevice(h->cfg.device));
        rc = rg_mailbox_quiesce(h);
        if (!rc) rc = rg_comm_setup(h, i, n, c.ring_ticks, c.overflow_slots, nullptr, nullptr, true, !rccl && xdev);
        if (rc) {
            undo(i);
            return rc;
        }
    }
    if (rccl) {
        rc = rg_rccl_load();
        ncclUniqueId u;
        ncclResult_t r = ncclSuccess;
        if (!rc) r = g_rccl.GetUniqueId(&u);
        if (!rc && r == ncclSuccess) {
            // one thread, n ranks: the initialisations of all of them inside ONE group (outside it the first
            // ncclCommInitRank would wait for ranks this very thread has not started yet)
            r = g_rccl.GroupStart();
            for (uint32_t i = 0; i < n && r == ncclSuccess; i++) {
                if (hipSetDevice(engines[i]->cfg.device) != hipSuccess) r = ncclUnhandledCudaError;
                else r = g_rccl.CommInitRank(&engines[i]->pub->comm, (int)n, u, (int)i);
            }
            const ncclResult_t r2 = g_rccl.GroupEnd();
            if (r == ncclSuccess) r = r2;
        }
        if (rc || r != ncclSuccess) {
            for (uint32_t i = 0; i < n; i++) engines[i]->pub->comm = nullptr; // (a failed group leaves no usable communicator)
            undo(n);
            return rc ? rc : rg_fail(RG_ERR_NO_DEVICE, "rg_comm_init_all: grouped ncclCommInitRank of %u ranks failed: %s", n, rg_nccl_err(r));
        }
    }
    rc = rg_publish_all_impl(engines, n, true); // every replica starts from the actual columns
    if (rc) undo(n);
    return rc;
}

extern "C" int rg_publish_commit_all(rg_engine *const *engines, uint32_t n, uint32_t flags) {
    if (flags & ~RG_PUBLISH_FULL) return rg_fail(RG_ERR_INVALID_ARG, "rg_publish_commit_all: unknown flags %#x", flags);
    int rc = rg_all_check(engines, n, "rg_publish_commit_all", true);
    if (rc) return rc;
    return rg_publish_all_impl(engines, n, (flags & RG_PUBLISH_FULL) != 0);
}

extern "C" int rg_publish_commit(rg_engine *h, uint32_t flags) {
    if (!h) return rg_fail(RG_ERR_INVALID_ARG, "rg_publish_commit: null engine");
    if (!h->pub) return rg_fail(RG_ERR_STATE, "rg_publish_commit: rg_comm_init was never called");
    if (flags & ~RG_PUBLISH_FULL) return rg_fail(RG_ERR_INVALID_ARG, "rg_publish_commit: unknown flags %#x", flags);
    RG_ENTER(h);
    return rg_publish_impl(h, (flags & RG_PUBLISH_FULL) != 0);
}

extern "C" int rg_publish_sync(rg_engine *h) {
    if (!h) return rg_fail(RG_ERR_INVALID_ARG, "rg_publish_sync: null engine");
    if (!h->pub) return rg_fail(RG_ERR_STATE, "rg_publish_sync: rg_comm_init was never called");
    RG_ENTER(h);
    int rc = rg_pub_materialize(h);
    if (rc) return rc;
    RG_HIP(hipStreamSynchronize(h->pub->side));
    return RG_OK;
}

extern "C" const uint64_t *rg_published_commit_ptr(rg_engine *h, uint64_t *stride) {
    if (!h || !h->pub) return nullptr;
    if (stride) *stride = h->pub->lay.Gpad;
    return h->pub->replica;
}

extern "C" int rg_published_commit(rg_engine *h, uint32_t rank, uint64_t first, uint64_t n, uint64_t *host_commit) {
    if (!h || (n && !host_commit)) return rg_fail(RG_ERR_INVALID_ARG, "rg_published_commit: bad argument");
    if (!h->pub) return rg_fail(RG_ERR_STATE, "rg_published_commit: rg_comm_init was never called");
    if (rank >= h->pub->world || first > h->G || n > h->G - first)
        return rg_fail(RG_ERR_INVALID_ARG, "rg_published_commit: rank %u groups [%llu, +%llu) outside %u ranks x %llu groups",
                       rank, (unsigned long long)first, (unsigned long long)n, h->pub->world, (unsigned long long)h->G);
    int rc = rg_publish_sync(h);
    if (rc || !n) return rc;
    RG_HIP(hipMemcpy(host_commit, h->pub->replica + (u64)rank * h->pub->lay.Gpad + first, n * 8, hipMemcpyDeviceToHost));
    return RG_OK;
}

extern "C" int rg_publish_stats_get(rg_engine *h, rg_publish_stats *out) {
    if (!h || !out) return rg_fail(RG_ERR_INVALID_ARG, "rg_publish_stats_get: bad argument");
    if (!h->pub) return rg_fail(RG_ERR_STATE, "rg_publish_stats_get: rg_comm_init was never called");
    *out = h->pub->stats;
    out->bytes_per_rank_delta = h->pub->lay.bytes_per_rank;
    out->bytes_per_rank_full = h->pub->lay.Gpad * 8;
    out->overflow_slots = h->pub->lay.cap;
    out->ring_ticks = h->pub->ring;
    return RG_OK;
}

// Host twins of the encoding (no GPU involved): what the tick kernels write and what the replica kernels add, over
// caller-provided buffers. CPU-only tests run the N > 1 exchange with these under gloo.
extern "C" uint64_t rg_pub_bytes_per_rank(uint64_t n_groups, uint32_t overflow_slots) {
    return rg_pub_layout(n_groups, overflow_slots ? overflow_slots : (u32)(n_groups / 256 + 64)).bytes_per_rank;
}

extern "C" int rg_pub_accumulate_host(uint64_t n_groups, uint32_t overflow_slots, const uint64_t *old_commit,
                                      const uint64_t *new_commit, uint8_t *slice) {
    if (!old_commit || !new_commit || !slice) return rg_fail(RG_ERR_INVALID_ARG, "rg_pub_accumulate_host: null argument");
    const RgPubLayout l = rg_pub_layout(n_groups, overflow_slots ? overflow_slots : (u32)(n_groups / 256 + 64));
    RgPubHdr *hdr = reinterpret_cast<RgPubHdr *>(slice);
    RgPubOvf *list = reinterpret_cast<RgPubOvf *>(slice + l.off_list);
    u8 *dlt = slice + l.off_delta;
    for (u64 g = 0; g < n_groups; g++) {
        if (new_commit[g] < old_commit[g]) return rg_fail(RG_ERR_INVALID_ARG, "rg_pub_accumulate_host: group %llu: the commit index decreased", (unsigned long long)g);
        if (new_commit[g] != old_commit[g]) dlt[g] = (u8)rg_pub_accumulate(dlt[g], old_commit[g], new_commit[g], g, hdr, list, l.cap);
    }
    return RG_OK;
}

extern "C" int rg_pub_apply_host(uint64_t n_groups, uint32_t overflow_slots, uint32_t world, const uint8_t *gathered,
                                 uint64_t *replica, uint32_t *lost_ranks) {
    if (!gathered || !replica) return rg_fail(RG_ERR_INVALID_ARG, "rg_pub_apply_host: null argument");
    const RgPubLayout l = rg_pub_layout(n_groups, overflow_slots ? overflow_slots : (u32)(n_groups / 256 + 64));
    RgPubSlots sl;
    sl.n = 1;
    sl.slice[0] = reinterpret_cast<const char *>(gathered);
    u32 lost = 0;
    for (u32 r = 0; r < world; r++) {
        for (u64 g8 = 0; g8 < l.Gpad; g8 += 8) rg_pub_apply8(replica, sl, l, r, g8);
        const char *base = sl.slice[0] + (u64)r * l.bytes_per_rank;
        const RgPubHdr *hdr = reinterpret_cast<const RgPubHdr *>(base);
        const RgPubOvf *list = reinterpret_cast<const RgPubOvf *>(base + l.off_list);
        for (u32 k = 0; k < hdr->n_overflow && k < l.cap; k++)
            if (list[k].group < l.G) replica[(u64)r * l.Gpad + list[k].group] += list[k].extra;
        if ((hdr->flags & RG_PUB_LOST) || hdr->n_overflow > l.cap) lost++;
    }
    if (lost_ranks) *lost_ranks = lost;
    return RG_OK;
}
