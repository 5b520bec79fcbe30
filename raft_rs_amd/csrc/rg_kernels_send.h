// rg_kernels_send.h -- kernels of abi_send.hip: the send stage, entry sizes, host-sent messages, progress events, RG_PF_INS_FULL
// Included by exactly one abi_*.hip unit (the kernels are not templates: one definition per library).
#pragma once
#include "rg_engine.h"

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


// RawNode::report_unreachable / report_snapshot applied to the cells in place (rg_progress_events). Lane i applies the whole
// run of records of its (group, slot), in order, if it holds the run's first record.
__global__ __launch_bounds__(256) void k_progress_events(RgState st, u32 *ins_meta, const rg_progress_event *ev, u64 n, u32 P) {
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i < n) rg_progress_events_at(st, ins_meta, ev, n, P, i);
}

// ... and one kind of event for every group that names a slot (rg_progress_event_dense): lane = group

__global__ __launch_bounds__(256) void k_progress_event_dense(RgState st, u32 *ins_meta, const u8 *slot_plus1, u32 kind, u32 P) {
    const u64 g = (u64)blockIdx.x * 256 + threadIdx.x;
    if (g >= st.G) return;
    const u32 s1 = slot_plus1[g];
    if (!s1) return;
    const rg_progress_event ev = {g, s1 - 1u, kind};
    rg_progress_events_at(st, ins_meta, &ev, 1, P, 0);
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


