// rg_kernels_placement.h -- kernels of abi_placement.hip: the gathers that move whole groups to new positions
// Included by exactly one abi_*.hip unit (the kernels are not templates over the slot count: one definition per library).
#pragma once
#include "rg_engine.h"

// A peer-major or per-group column of `rows` rows: dst[r][i] = src[r][perm[i]] for i < G. Lane = destination group, so the
// stores are whole lines; the loads follow the permutation (a placement by size class reads three interleaved streams).
template <typename T>
__global__ __launch_bounds__(256) void k_place_rows(const T *__restrict__ src, T *__restrict__ dst, const u64 *__restrict__ perm,
                                                    u64 G, u64 stride) {
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= G) return;
    const u64 r = blockIdx.y;
    dst[r * stride + i] = src[r * stride + perm[i]];
}

// A group-major array of `words` elements per group (the Inflights rings: P x cap u64; the entry-size windows: w u32):
// dst[i][w] = src[perm[i]][w]; lanes run along the record, so both sides move whole lines.
template <typename T>
__global__ __launch_bounds__(256) void k_place_records(const T *__restrict__ src, T *__restrict__ dst, const u64 *__restrict__ perm,
                                                       u64 G, u64 words) {
    const u64 n = G * words;
    for (u64 k = (u64)blockIdx.x * 256 + threadIdx.x; k < n; k += (u64)gridDim.x * 256) {
        const u64 i = k / words, w = k - i * words;
        dst[k] = src[perm[i] * words + w];
    }
}
