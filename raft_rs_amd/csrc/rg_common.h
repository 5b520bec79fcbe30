// rg_common.h -- shared host/device definitions for the MI355X multi-raft progress/commit engine.
//
// Everything here is the engine's own restatement of the reference semantics cited in
// include/raftgroups.h; file:line citations are relative to the pingcap/raft-rs v0.6.0 tree.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/raftgroups.h"

typedef uint64_t u64;
typedef unsigned int u32;
typedef unsigned char u8;

#define RG_HD __host__ __device__ __forceinline__
#define RG_D __device__ __forceinline__

// Column views handed to kernels (plain pointers, no ownership).
struct RgState {
    u64 *match, *next, *prc, *psnap, *prs, *gid; // [P][stride]
    u64 *pflags;                                 // [G] one byte per slot
    u64 *commit, *lo, *hi;                       // [G]
    u32 *cfg, *out;                              // [G]
    u64 *run_first, *run_term;                   // [RG_TERM_RUNS][stride] term-run table (cold)
    u64 *dummy_idx, *dummy_term, *cur_term;      // [G] (cold)
    u8 *hhint;                                   // [G] (cold) RG_COL_HOST_HINT: rejects of the last log-term tick left to the host
                                                 // ... and, `stride` bytes further on, RG_COL_RUN_COUNT (rg_run_n): the two byte columns
                                                 // share ONE base pointer -- a kernel argument more costs the send-stage and fused
                                                 // kernels, at the limit of their scalar registers, a spill to scratch
    u64 G, stride;
    // commit publication (rg_publish.h): this rank's slice under construction, nullptr = not publishing
    char *pub;          // [RgPubHdr | RgPubOvf list[pub_cap] | u8 delta[Gpad]]
    u64 pub_off_delta;
    u32 pub_cap;
    u32 ix64; // host side only: this engine's launches take the 64-bit-offset instantiations (rg_ix32; decided once, in rg_create)
};

// RG_COL_RUN_COUNT [G]: used runs of the term-run table (engine-owned, derived); laid out right behind RG_COL_HOST_HINT
static inline __host__ __device__ u8 *rg_run_n(const RgState &st) { return st.hhint + st.stride; }

struct RgMsgs {
    const u64 *mi, *mc, *mh, *mrs; // [P][stride]
    const u64 *mflags;             // [G] one byte per slot
    const u64 *mlt;                // [P][stride] Message.log_term (cold)
    const u64 *mhr;                // [P][stride] hints after find_conflict_by_term (written by the pre-pass;
                                   // == mh when no message of the tick carries a log term)
};

// Column access with the index type of the caller. IX = u32 (dense lane kernels, engines with P * stride * 8 < 4 GiB):
// the byte offset is computed in 32 bits and zero-extended onto the column pointer, which is exactly the
// `SGPR base + 32-bit VGPR offset` form of the global memory instructions -- one VGPR per slot addresses that slot's cell
// in EVERY column, instead of a 64-bit address pair per (column, slot) kept alive from the load to the store.
// IX = u64: plain indexing (gathers over arbitrary groups, the host).
// 32-bit offsets are usable when the farthest cell any kernel addresses this way -- slot P-1 of a peer-major column, or
// run RG_TERM_RUNS-1 of the term-run table, group stride-1, 8 bytes each -- lies below 4 GiB.
static inline bool rg_fits_u32_offsets(u64 n_slots, u64 stride) {
    const u64 rows = n_slots > RG_TERM_RUNS ? n_slots : RG_TERM_RUNS;
    return rows * stride * 8 <= 0xffffffffULL;
}
// What the launchers ask: 32-bit cell offsets unless the engine is too large -- or was CREATED with RG_CFGF_IX64 in
// rg_config.flags, which makes every launch of that engine take the 64-bit-offset instantiations (k_tick_lane / _list /
// _fused / _compact <..., u64>, k_send_dense<..., u64>) that no engine a test can afford to build would otherwise reach on
// a GPU. Decided once, by rg_create, into RgState::ix64.
static inline bool rg_ix32(const RgState &st, u64 n_slots) { return !st.ix64 && rg_fits_u32_offsets(n_slots, st.stride); }
// rg_u32o: a 32-bit cell index whose BYTE OFFSET is made opaque right before every access (rg_at below). `base + zext(offset)`
// then stays in the block of the access, where instruction selection turns it into the SGPR-base + VGPR-offset addressing
// mode; left alone, the compiler hoists the 64-bit sum out of the branches and keeps a VGPR address PAIR per cell alive
// (118 v_lshl_add_u64 in k_tick_lane<5>). Register allocation with the opaque form: lane<5> 123 -> 101 VGPRs, lane<7>
// 162 -> 135, fused<5> 156 -> 115 (3 -> 4 waves per SIMD), fused<7> 201 -> 145 (2 -> 3). Measured on one box
// (profiles/calls/gpu_r04_g.sh): the fused kernels gain 10-12 % (their occupancy step), the single-tick lane kernels, whose
// occupancy does not change, lose 0-3 % (the asm statements pin the order of the accesses) -- so it is the FUSED kernel's
// index type and nobody else's.
struct rg_u32o {
    u32 v;
    RG_HD rg_u32o() {}
    RG_HD rg_u32o(u64 x) : v((u32)x) {}
    RG_HD operator u32() const { return v; }
    friend RG_HD rg_u32o operator*(rg_u32o a, rg_u32o b) { return rg_u32o((u64)(a.v * b.v)); }
    friend RG_HD rg_u32o operator+(rg_u32o a, rg_u32o b) { return rg_u32o((u64)(a.v + b.v)); }
};
template <typename T, typename IX> RG_HD T &rg_at(T *base, IX i) {
    typedef typename std::conditional<std::is_const<T>::value, const char, char>::type B;
    if constexpr (std::is_same<IX, rg_u32o>::value) {
        u32 off = i.v * (u32)sizeof(T);
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(off));
#endif
        return *reinterpret_cast<T *>(reinterpret_cast<B *>(base) + off);
    } else {
        return *reinterpret_cast<T *>(reinterpret_cast<B *>(base) + (IX)(i * (IX)sizeof(T)));
    }
}

RG_HD void rg_swap64(u64 &a, u64 &b) {
    const u64 t = a;
    a = b;
    b = t;
}
RG_HD u64 rg_min(u64 a, u64 b) { return a < b ? a : b; }
RG_HD u64 rg_max(u64 a, u64 b) { return a > b ? a : b; }

// splitmix64 finaliser: the counter-based PRNG of the synthetic stream.
RG_HD u64 rg_splitmix64(u64 x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
RG_HD u64 rg_hash(u64 seed, u64 tick, u64 group, u64 slot) {
    return rg_splitmix64(seed ^ (tick << 40) ^ (group << 3) ^ slot);
}
