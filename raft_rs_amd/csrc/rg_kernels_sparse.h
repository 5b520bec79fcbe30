// rg_kernels_sparse.h -- kernels of abi_mirror.hip: ingest and the helpers of the sparse path
// Included by exactly one abi_*.hip unit (the kernels are not templates: one definition per library).
#pragma once
#include "rg_engine.h"

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


