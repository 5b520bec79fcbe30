// rg_kernels_publish.h -- kernels of abi_publish.hip: replica updates
// Included by exactly one abi_*.hip unit (the kernels are not templates: one definition per library).
#pragma once
#include "rg_engine.h"

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


