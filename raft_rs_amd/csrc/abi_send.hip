// abi_send.hip -- the send stage: Inflights on the device and the maybe_send_append decision (include/raftgroups.h: "send stage")
// There is NO CPU fallback anywhere in this file: without a HIP device every entry point fails.
#include "rg_engine.h"
#include "rg_kernels_send.h"

// ------------------------------------------------------------------------------------------------
// send stage
// ------------------------------------------------------------------------------------------------
// Enqueue the send stage over `list[0..n)` (NULL = all groups); n_ptr != NULL: the length is read on the device and
// `n` only sizes the grid.
int rg_send_enqueue(rg_engine *h, uint64_t max_entries_per_msg, uint32_t flags, const u64 *list, u64 n,
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

extern "C" int rg_send_appends(rg_engine *h, uint64_t max_entries_per_msg, uint32_t flags) try {
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
} RG_ABI_GUARD

extern "C" int rg_log_sizes_enable(rg_engine *h, uint32_t window) try {
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
} RG_ABI_GUARD

// host records -> device staging buffer on the engine's stream (grown on demand; the copy is stream-ordered, the host
// array may be reused once the call returns: pageable copies are staged by the runtime)
int rg_stage_records(rg_engine *h, const void *recs, size_t bytes) {
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

extern "C" int rg_log_sizes_write(rg_engine *h, const rg_log_size *recs, uint64_t n) try {
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
} RG_ABI_GUARD

extern "C" int rg_workload_sizes(rg_engine *h, uint64_t seed, uint32_t min_bytes, uint32_t spread) try {
    if (!h) return rg_fail(RG_ERR_INVALID_ARG, "rg_workload_sizes: null engine");
    if (!h->esz) return rg_fail(RG_ERR_STATE, "rg_workload_sizes: rg_log_sizes_enable first");
    RG_ENTER(h);
    hipLaunchKernelGGL(k_wl_sizes, dim3(rg_grid(h->G, RG_BLOCK)), dim3(RG_BLOCK), 0, h->stream, (const u64 *)h->st.hi, h->G, h->esz,
                       h->ins.esz_w, (u64)seed, min_bytes, spread);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return rg_fail(RG_ERR_NO_DEVICE, "rg_workload_sizes: launch failed: %s", hipGetErrorString(e));
    return RG_OK;
} RG_ABI_GUARD

extern "C" int rg_update_state(rg_engine *h, const rg_sent_msg *msgs, uint64_t n) try {
    if (!h || (!msgs && n)) return rg_fail(RG_ERR_INVALID_ARG, "rg_update_state: bad argument");
    if (!h->ins_arena) return rg_fail(RG_ERR_STATE, "rg_update_state: engine created with max_inflight = 0 (use RG_MF_SENT events)");
    if (n == 0) return RG_OK;
    RG_ENTER(h);
    int rc = rg_send_materialize(h); // (ins.add moves the windows' tails: RG_SEND_LAST_IS_TAIL items are read out first)
    if (rc) return rc;
    rc = rg_stage_records(h, msgs, (size_t)n * sizeof(rg_sent_msg));
    if (rc) return rc;
    hipLaunchKernelGGL(k_update_state, dim3(rg_grid(n, 256)), dim3(256), 0, h->stream, h->st, h->ins, (const rg_sent_msg *)h->d_recs,
                       (u64)n, h->P);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return rg_fail(RG_ERR_NO_DEVICE, "rg_update_state: launch failed: %s", hipGetErrorString(e));
    return RG_OK;
} RG_ABI_GUARD

extern "C" int rg_progress_events(rg_engine *h, const rg_progress_event *events, uint64_t n) try {
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
} RG_ABI_GUARD

extern "C" int rg_progress_event_dense(rg_engine *h, uint32_t kind, const uint8_t *host_slot_plus1) try {
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
} RG_ABI_GUARD

// After a dense stage the work items live in the columns; the compact list exists once somebody asks for it.
int rg_send_materialize(rg_engine *h) {
    if (!h->send_cols_fresh) return RG_OK;
    RG_HIP(hipMemsetAsync(h->send_counter, 0, 4, h->stream));
    hipLaunchKernelGGL(k_send_compact, dim3(rg_grid(h->G, 256)), dim3(256), 0, h->stream, h->send_cols, (const u64 *)h->ins.tail, h->G, h->stride, h->P,
                       h->send_items, h->send_counter);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return rg_fail(RG_ERR_NO_DEVICE, "rg_send_items: compaction launch failed: %s", hipGetErrorString(e));
    h->send_cols_fresh = false;
    return RG_OK;
}

extern "C" int rg_send_items(rg_engine *h, rg_send_item *host_items, uint64_t cap, uint64_t *n) try {
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
} RG_ABI_GUARD

extern "C" const rg_send_item *rg_send_items_ptr(rg_engine *h) {
    if (!h || !h->ins_arena) return nullptr;
    if (hipSetDevice(h->cfg.device) != hipSuccess || rg_send_materialize(h) != RG_OK) return nullptr;
    return h->send_items;
}

extern "C" int rg_send_columns(rg_engine *h, const uint64_t **dev_prev_index, const uint64_t **dev_last_index,
                               const uint32_t **dev_n_kind) try {
    if (!h) return rg_fail(RG_ERR_INVALID_ARG, "rg_send_columns: null engine");
    if (!h->ins_arena) return rg_fail(RG_ERR_STATE, "rg_send_columns: engine created with max_inflight = 0");
    if (!h->send_last_dense)
        return rg_fail(RG_ERR_STATE, "rg_send_columns: the last send stage was not a dense one (sparse stages produce the "
                                     "compact list only)");
    if (dev_prev_index) *dev_prev_index = h->send_cols.prev;
    if (dev_last_index) *dev_last_index = h->send_cols.last;
    if (dev_n_kind) *dev_n_kind = h->send_cols.n;
    return RG_OK;
} RG_ABI_GUARD

extern "C" int rg_send_tail_column(rg_engine *h, const uint64_t **dev_newest_inflight) try {
    if (!h || !dev_newest_inflight) return rg_fail(RG_ERR_INVALID_ARG, "rg_send_tail_column: bad argument");
    if (!h->ins_arena) return rg_fail(RG_ERR_STATE, "rg_send_tail_column: engine created with max_inflight = 0");
    *dev_newest_inflight = h->ins.tail;
    return RG_OK;
} RG_ABI_GUARD
static_assert(RG_SEND_LAST_IS_TAIL == RG_SEND_NK_LAST_IS_TAIL && RG_SEND_LAST_IS_PREV == RG_SEND_NK_LAST_IS_PREV, "the header's bits are the kernels'");

extern "C" uint64_t rg_inflights_bytes(const rg_engine *h, int ring) {
    if (!h || !h->ins_arena) return 0;
    return ring ? (uint64_t)h->G * h->P * h->ins.cap * 8 : (uint64_t)h->P * h->stride * 4;
}

// The oldest and the newest inflight of a window live in the `head` / `tail` columns (rg_send.h); to the outside the
// ring is whole.
extern "C" int rg_read_inflights(rg_engine *h, uint32_t *host_meta, uint64_t *host_ring) try {
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
    // to the outside every window is a ring: start < cap, the entries oldest first from there. A compact window (rg_send.h:
    // up to four entries held as distances in the columns, start == RG_INS_COMPACT) is written out at start = 0.
    if (meta)
        for (u32 p = 0; p < h->P; p++)
            for (u64 g = 0; g < h->G; g++) {
                const u64 o = (u64)p * h->stride + g;
                const u32 m = meta[o], start = m & 0xffffu, count = m >> 16;
                if (start == RG_INS_COMPACT) meta[o] = count << 16;
                if (!count || !host_ring) continue;
                u64 *cell = host_ring + (g * h->P + p) * h->ins.cap;
                if (start == RG_INS_COMPACT) {
                    u64 e[4];
                    rg_ins_compact_entries(head[o], tail[o], e);
                    for (u32 i = 0; i < count && i < 4; i++) cell[i] = e[count - 1 - i];
                } else {
                    cell[start] = head[o];
                    cell[(start + count - 1) % h->ins.cap] = tail[o];
                }
            }
    return RG_OK;
} RG_ABI_GUARD

extern "C" int rg_load_inflights(rg_engine *h, const uint32_t *host_meta, const uint64_t *host_ring) try {
    if (!h || !host_meta || !host_ring) return rg_fail(RG_ERR_INVALID_ARG, "rg_load_inflights: meta and ring are both required");
    if (!h->ins_arena) return rg_fail(RG_ERR_STATE, "rg_load_inflights: engine created with max_inflight = 0");
    const u64 cells = (u64)h->P * h->stride;
    std::vector<u64> head(cells, 0), tail(cells, 0);
    std::vector<u32> meta_in(host_meta, host_meta + cells); // (windows that fit the columns are loaded compact: rg_send.h)
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
            if (count <= RG_INS_COMPACT_MAX) { // up to four entries whose distances fit 21 bits: the columns hold them all
                u64 hd = 0;
                bool fits = true;
                for (u32 i = 1; i < count; i++) { // d_i = e[i - 1] - e[i], newest first
                    const u64 d = cell[(start + count - i) % h->ins.cap] - cell[(start + count - 1 - i) % h->ins.cap];
                    fits = fits && d <= RG_INS_DMASK;
                    hd |= (d & RG_INS_DMASK) << (RG_INS_DBITS * (i - 1));
                }
                if (fits) {
                    head[o] = hd;
                    meta_in[o] = RG_INS_COMPACT | (count << 16);
                }
            }
        }
    RG_ENTER(h);
    {   // (the loaded windows replace the tail column RG_SEND_LAST_IS_TAIL items of the last dense stage point at)
        const int mrc = rg_send_materialize(h);
        if (mrc) return mrc;
    }
    RG_HIP(hipMemcpyAsync(h->ins.meta, meta_in.data(), cells * 4, hipMemcpyHostToDevice, h->stream));
    RG_HIP(hipMemcpyAsync(h->ins.head, head.data(), cells * 8, hipMemcpyHostToDevice, h->stream));
    RG_HIP(hipMemcpyAsync(h->ins.tail, tail.data(), cells * 8, hipMemcpyHostToDevice, h->stream));
    RG_HIP(hipMemcpyAsync(h->ins.ring, host_ring, rg_inflights_bytes(h, 1), hipMemcpyHostToDevice, h->stream));
    // Inflights::full() of the loaded windows, for the next tick's is_paused() / free_first_one decisions
    const int frc = rg_fix_ins_full(h);
    if (frc) return frc;
    RG_HIP(hipStreamSynchronize(h->stream));
    return RG_OK;
} RG_ABI_GUARD

int rg_fix_ins_full(rg_engine *h) {
    hipLaunchKernelGGL(k_fix_ins_full, dim3(rg_grid(h->G, RG_BLOCK)), dim3(RG_BLOCK), 0, h->stream, h->st, h->ins, h->P);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return rg_fail(RG_ERR_NO_DEVICE, "k_fix_ins_full launch failed: %s", hipGetErrorString(e));
    return RG_OK;
}


