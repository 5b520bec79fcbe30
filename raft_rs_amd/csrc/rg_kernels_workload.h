// rg_kernels_workload.h -- kernels of abi_wire.hip: the synthetic stream
// Included by exactly one abi_*.hip unit (the kernels are not templates: one definition per library).
#pragma once
#include "rg_engine.h"

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


