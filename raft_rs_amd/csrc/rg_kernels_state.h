// rg_kernels_state.h -- kernels of abi_state.hip: status read-back, sparse cell writes, engine-owned flag bits
// Included by exactly one abi_*.hip unit (the kernels are not templates: one definition per library).
#pragma once
#include "rg_engine.h"

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


