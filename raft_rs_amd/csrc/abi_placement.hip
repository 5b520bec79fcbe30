// abi_placement.hip -- placement of groups inside a shard by replica-set size class (include/raftgroups.h: "placement")
// There is NO CPU fallback anywhere in this file: without a HIP device rg_permute_groups fails (rg_plan_placement is pure host
// arithmetic over the caller's cfg words, like rg_decode_message).
#include "rg_engine.h"
#include "rg_kernels_placement.h"

// Membership is the host's to change at any time (ProgressTracker::apply_conf, src/tracker.rs:380-397; the voter sets of
// src/tracker.rs:37-92): a shard whose groups ARRIVE in any order -- or drift there as conf changes accumulate -- has the cells
// of peers a group does not have in the same lines as cells that are used, and the dense tick moves them all (config 5,
// interleaved: 1.47 x the algorithmic bytes). The one-launch class kernel needs groups of one size in contiguous ranges; these
// two calls are how a host gets there: plan on the host from the cfg words, gather on the device.
extern "C" int rg_plan_placement(const uint32_t *cfg_words, uint64_t n_groups, uint32_t n_slots, uint64_t *perm,
                                 rg_size_class *classes, uint32_t cap, uint32_t *n_classes) try {
    if (!cfg_words || !perm || n_groups == 0 || n_slots == 0 || n_slots > RG_MAX_SLOTS || (cap && !classes))
        return rg_fail(RG_ERR_INVALID_ARG, "rg_plan_placement: bad argument");
    // the bodies k_tick_classes<P> has: 3, 5, 7 slots below P, and P itself (rg_class_body) -- a counting sort over them,
    // stable, so groups of one class keep their relative order (what keeps a host's own range bookkeeping simple)
    u64 count[RG_MAX_SLOTS + 1];
    memset(count, 0, sizeof(count));
    for (u64 g = 0; g < n_groups; g++) count[rg_class_body(rg_cfg_slots_named(cfg_words[g]), n_slots)]++;
    u64 start[RG_MAX_SLOTS + 1];
    u64 at = 0;
    for (u32 k = 0; k <= RG_MAX_SLOTS; k++) {
        start[k] = at;
        at += count[k];
    }
    u64 fill[RG_MAX_SLOTS + 1];
    memcpy(fill, start, sizeof(fill));
    for (u64 g = 0; g < n_groups; g++) perm[fill[rg_class_body(rg_cfg_slots_named(cfg_words[g]), n_slots)]++] = g;
    // what the engine will derive from the permuted cfg column: per block of RG_BLOCK groups the largest body among its
    // groups (a block that straddles a boundary runs the larger one), run-length encoded
    u32 k = 0;
    u64 b = 0;
    const u64 nb = (n_groups + RG_BLOCK - 1) / RG_BLOCK;
    auto body_of_block = [&](u64 blk) {
        const u64 last = rg_min((blk + 1) * RG_BLOCK, n_groups) - 1; // classes ascend along the shard: the block's last group decides
        u32 q = 0;
        for (u32 c = 0; c <= RG_MAX_SLOTS; c++)
            if (count[c] && last >= start[c]) q = c;
        return q;
    };
    while (b < nb) {
        const u32 q = body_of_block(b);
        u64 e = b + 1;
        while (e < nb && body_of_block(e) == q) e++;
        if (k < cap) {
            classes[k].first_group = b * RG_BLOCK;
            classes[k].n_groups = rg_min(e * RG_BLOCK, n_groups) - b * RG_BLOCK;
            classes[k].n_slots = q;
            classes[k].reserved = 0;
        }
        k++;
        b = e;
    }
    if (n_classes) *n_classes = k;
    return RG_OK;
} RG_ABI_GUARD

template <typename T>
static void rg_place_rows(rg_engine *h, const void *src, void *dst, const u64 *d_perm, u64 rows) {
    hipLaunchKernelGGL(k_place_rows<T>, dim3(rg_grid(h->G, 256), (unsigned)rows), dim3(256), 0, h->stream, (const T *)src, (T *)dst, d_perm,
                       h->G, h->stride);
}

extern "C" int rg_permute_groups(rg_engine *h, const uint64_t *host_perm) try {
    if (!h || !host_perm) return rg_fail(RG_ERR_INVALID_ARG, "rg_permute_groups: bad argument");
    {   // a permutation of [0, G): every old position exactly once
        std::vector<u8> seen(h->G, 0);
        for (u64 i = 0; i < h->G; i++) {
            const u64 o = host_perm[i];
            if (o >= h->G || seen[o]) return rg_fail(RG_ERR_INVALID_ARG, "rg_permute_groups: perm[%llu] = %llu: not a permutation of the shard's groups",
                                                     (unsigned long long)i, (unsigned long long)o);
            seen[o] = 1;
        }
    }
    if (h->host_mirror && !h->q_dirty.empty())
        return rg_fail(RG_ERR_SLOT_BUSY, "rg_permute_groups: messages are queued for the next flush (rg_step): flush first");
    if (h->ingested_upper)
        return rg_fail(RG_ERR_STATE, "rg_permute_groups: records are ingested and not ticked yet (rg_ingest): tick first");
    RG_ENTER(h);
    // the last tick's send stage, if the host skipped it: its Inflights effects are applied first (its requests are dropped, as
    // the next tick would do); an unanswered host hint refuses the call like any other next step
    int rc = rg_settle_send(h);
    if (rc) return rc;
    rc = rg_send_materialize(h); // (the compact list keeps the OLD positions in its `group` field: fetch it before this call)
    if (rc) return rc;
    char *tmp = nullptr, *itmp = nullptr;
    u64 *d_perm = nullptr;
    u32 *etmp = nullptr;
    hipError_t e = hipMalloc(&tmp, h->state_bytes);
    if (e == hipSuccess) e = hipMalloc(&d_perm, h->G * 8);
    if (e == hipSuccess && h->ins_arena) e = hipMalloc(&itmp, h->ins_state_bytes);
    if (e == hipSuccess && h->esz) e = hipMalloc(&etmp, (size_t)h->G * h->ins.esz_w * 4);
    if (e == hipSuccess) e = hipMemcpyAsync(d_perm, host_perm, h->G * 8, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(tmp, h->arena, h->state_bytes, hipMemcpyDeviceToDevice, h->stream); // (padding and all)
    if (e == hipSuccess) {
        for (int c = 0; c < RG_COL_COUNT; c++) {
            const u64 rows = rg_col_per_slot(c) ? h->P : rg_col_per_run(c) ? RG_TERM_RUNS : 1;
            const char *src = h->arena + h->col_off[c];
            char *dst = tmp + h->col_off[c];
            switch (rg_col_elem(c)) {
            case 8: rg_place_rows<u64>(h, src, dst, d_perm, rows); break;
            case 4: rg_place_rows<u32>(h, src, dst, d_perm, rows); break;
            default: rg_place_rows<u8>(h, src, dst, d_perm, rows); break;
            }
        }
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h->arena, tmp, h->state_bytes, hipMemcpyDeviceToDevice, h->stream);
    if (e == hipSuccess && h->ins_arena) { // the windows travel with their groups: meta | oldest | newest columns, the rings
        e = hipMemcpyAsync(itmp, h->ins_arena, h->ins_state_bytes, hipMemcpyDeviceToDevice, h->stream);
        if (e == hipSuccess) {
            rg_place_rows<u32>(h, h->ins.meta, itmp + ((char *)h->ins.meta - h->ins_arena), d_perm, h->P);
            rg_place_rows<u64>(h, h->ins.head, itmp + ((char *)h->ins.head - h->ins_arena), d_perm, h->P);
            rg_place_rows<u64>(h, h->ins.tail, itmp + ((char *)h->ins.tail - h->ins_arena), d_perm, h->P);
            const u64 words = (u64)h->P * h->ins.cap;
            const unsigned grid = (unsigned)rg_min((u64)rg_grid(h->G * words, 256), (u64)65536);
            hipLaunchKernelGGL(k_place_records<u64>, dim3(grid), dim3(256), 0, h->stream, (const u64 *)h->ins.ring,
                               reinterpret_cast<u64 *>(itmp + ((char *)h->ins.ring - h->ins_arena)), (const u64 *)d_perm, h->G, words);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(h->ins_arena, itmp, h->ins_state_bytes, hipMemcpyDeviceToDevice, h->stream);
    }
    if (e == hipSuccess && h->esz) { // ... and so do the entry-size windows of RG_SEND_BYTES
        const u64 words = h->ins.esz_w;
        const unsigned grid = (unsigned)rg_min((u64)rg_grid(h->G * words, 256), (u64)65536);
        hipLaunchKernelGGL(k_place_records<u32>, dim3(grid), dim3(256), 0, h->stream, (const u32 *)h->esz, etmp, (const u64 *)d_perm, h->G, words);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(h->esz, etmp, (size_t)h->G * words * 4, hipMemcpyDeviceToDevice, h->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (tmp) (void)hipFree(tmp);
    if (itmp) (void)hipFree(itmp);
    if (etmp) (void)hipFree(etmp);
    if (d_perm) (void)hipFree(d_perm);
    if (e != hipSuccess)
        return rg_fail(e == hipErrorOutOfMemory ? RG_ERR_OUT_OF_MEMORY : RG_ERR_NO_DEVICE, "rg_permute_groups: %s", hipGetErrorString(e));
    // everything that was keyed by position
    if (h->host_mirror) { // the mirror's tables: peer ids, term gates, the cfg copy
        std::vector<u64> ids(h->peer_ids.size());
        std::vector<u64> terms(h->terms.size());
        for (u64 i = 0; i < h->G; i++) {
            memcpy(&ids[i * 8], &h->peer_ids[host_perm[i] * 8], 64);
            terms[i] = h->terms[host_perm[i]];
        }
        h->peer_ids.swap(ids);
        h->terms.swap(terms);
    }
    h->host_cfg_valid = false;
    h->cls_stale = true;           // the point of it all: the next dense tick derives the size classes of the new placement
    h->host_res_valid = false;
    h->out_is_dense = true;        // (RG_COL_OUT moved wholesale: the compact result list of a sparse tick no longer names it)
    h->last_sparse_n = 0;
    h->send_bound = 0;
    h->host_items_valid = false;
    if (h->pub) h->pub->local_lost = true; // every position's commit index changed: the next check point publishes a full snapshot
    // a checkpoint taken before is a complete image of the OLD placement (device side only: the mirror's tables are not in it)
    if (h->ckpt) {
        (void)hipFree(h->ckpt);
        h->ckpt = nullptr;
    }
    if (h->ins_ckpt) {
        (void)hipFree(h->ins_ckpt);
        h->ins_ckpt = nullptr;
    }
    if (h->esz_ckpt) {
        (void)hipFree(h->esz_ckpt);
        h->esz_ckpt = nullptr;
    }
    return RG_OK;
} RG_ABI_GUARD
