// abi_tick.hip -- the hot path: dense ticks, fused ticks, recompute, results, host hints, votes (include/raftgroups.h: "the hot path", "vote / quorum-liveness bitmaps")
// There is NO CPU fallback anywhere in this file: without a HIP device every entry point fails.
#include "rg_engine.h"
#include "rg_kernels_quorum.h"

// ------------------------------------------------------------------------------------------------
// the hot path
// ------------------------------------------------------------------------------------------------


// Device Inflights: a tick's result word carries free_to / free_first_one / left-Replicate effects for the rings. If
// the host skipped rg_send_appends, apply those effects (and nothing else: the send requests are dropped, which
// is what skipping the stage means) before the next tick overwrites RG_COL_OUT, so no window is left stale.
// Device Inflights and RG_OUT_HOST_HINT: the reference runs a deferred reject's send_append BEFORE the group's other sends of
// the step, so the group's send requests wait for rg_resolve_host_hints, which serves them (its Inflights effects -- free_to,
// free_first_one, the window resets -- are applied by the stage either way). A host that moved on without resolving would drop
// those requests for good and leave `next` / the windows behind the reference's: exact or loud -- every entry point that
// starts the next step refuses while such a group exists. Checked only after a tick that carried log terms (nothing else can
// raise the bit): one reduction over RG_COL_OUT and one synchronisation on that rare path, nothing on the others.
int rg_require_hints_resolved(rg_engine *h, const char *who) {
    if (!h->ins_arena || !h->hint_check_due) return RG_OK;
    if (h->hint_probe_pending) {
        // the last log-term tick was a dense one: its pre-pass has told pinned memory whether it left any reject to the host.
        // Waiting for THAT (the pre-pass runs before its tick) costs the caller no synchronisation with the tick itself, and a
        // tick that raised nothing -- every tick of a host that simply always passes a log-term column -- ends the matter here
        h->hint_probe_pending = false;
        RG_HIP(hipEventSynchronize(h->ev_hint));
        if (*(volatile u32 *)h->pin_hint_raised == 0) {
            h->hint_check_due = false;
            return RG_OK;
        }
    }
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

// The find_conflict_by_term pre-pass of a dense tick whose messages carry a log-term column (k_resolve_hints over every group),
// with the probe described at rg_engine::d_hint_raised around it. `probe`: engines with device Inflights (the only ones whose
// next step depends on the answer) and the fused driver (which stops at a tick that raised a hint).
// `probed` (out): the probe was armed. It is not while the stream is being CAPTURED into a hipGraph (tests/test_graph_capture_gpu.py:
// a captured launch sequence is replayed without the host, which could neither wait for the event nor read the word): such a tick
// falls back to what round 5 did -- the counting check at the next entry point / no stop inside a fused call.
static int rg_hint_prepass(rg_engine *h, RgMsgs &ms, bool probe, bool *probed = nullptr) {
    if (probe) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(h->stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) probe = false;
    }
    if (probed) *probed = probe;
    if (probe) {
        if (!h->pin_hint_raised) {
            RG_HIP(hipHostMalloc(reinterpret_cast<void **>(&h->pin_hint_raised), 64, hipHostMallocDefault));
            RG_HIP(hipEventCreateWithFlags(&h->ev_hint, hipEventDisableTiming));
        }
        RG_HIP(hipMemsetAsync(h->d_hint_raised, 0, 4, h->stream));
    }
    hipLaunchKernelGGL(k_resolve_hints, dim3(rg_grid(h->G, RG_BLOCK)), dim3(RG_BLOCK), 0, h->stream, h->st, ms, h->P, h->rhint,
                       probe ? h->d_hint_raised : nullptr);
    ms.mhr = h->rhint;
    if (probe) {
        RG_HIP(hipMemcpyAsync(h->pin_hint_raised, h->d_hint_raised, 4, hipMemcpyDeviceToHost, h->stream));
        RG_HIP(hipEventRecord(h->ev_hint, h->stream));
    }
    return RG_OK;
}

int rg_settle_send(rg_engine *h) {
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
int rg_refresh_classes(rg_engine *h) {
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
int rg_tick_impl(rg_engine *h, const RgMsgs &ms, const RgSendReq *send) {
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
        h->dev.last_tick_offset_bits = rg_ix32(h->st, h->P) ? 32u : 64u;
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
    // commit publication: the event rg_publish_commit would record behind this tick rides on the tick's own dispatch packet
    // (RG_LAUNCH_TICK, rg_tick_kernels.h) -- not while the stream is being captured into a graph, not for the ranks of
    // rg_comm_init_all (their publication records its events itself)
    int evt_slot = -1;
#ifndef RG_NO_PUB_RIDE /* (measurement builds: python -m raft_rs_amd.build --exp noride -DRG_NO_PUB_RIDE) */
    if (h->pub && !h->pub->in_process) {
        hipStreamCaptureStatus pcs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(h->stream, &pcs) == hipSuccess && pcs == hipStreamCaptureStatusNone) evt_slot = (int)(h->pub->n_pub % RG_PUB_SEND);
    }
#endif
    struct RgStopEvt { // (armed for the launches of THIS call only, whatever way it returns)
        explicit RgStopEvt(hipEvent_t e) { rg_tls_stop_event = e; }
        ~RgStopEvt() { rg_tls_stop_event = nullptr; }
        bool went_out() const { return rg_tls_stop_event == nullptr; }
    } stop_evt(evt_slot >= 0 ? h->pub->ev_tick[evt_slot] : nullptr);
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
            if (evt_slot >= 0 && stop_evt.went_out()) h->pub_tick_evt = evt_slot;
            h->dev.last_tick_kernel = RG_KERNEL_CLASSES;
            h->dev.last_tick_offset_bits = 32u; // (the class bodies exist for 32-bit offsets only: the condition above)
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
    if (evt_slot >= 0 && stop_evt.went_out()) h->pub_tick_evt = evt_slot;
    h->dev.last_tick_kernel = kernel;
    h->dev.last_tick_offset_bits = rg_ix32(h->st, h->P) ? 32u : 64u;
    if (kernel == RG_KERNEL_LDS) h->dev.last_tick_offset_bits = 64u; // (the LDS-staged comparison kernels index with 64 bits at any size)
    // (the group-commit instantiation and the LDS / compact variants have no streaming twins: rg_launch_tick_gc)
    h->dev.last_tick_streaming = kernel == RG_KERNEL_SPLIT ? 2u : (kernel == RG_KERNEL_LANE && !h->any_group_commit) ? (h->nt_all ? 2u : h->nt_msgs ? 1u : 0u) : 0u;
    h->tick_launches++;
    h->ticked = true;
    h->out_is_dense = true;
    h->send_ready = true;
    h->host_res_valid = false;
    return RG_OK;
}

extern "C" int rg_size_classes(rg_engine *h, rg_size_class *out, uint32_t cap, uint32_t *n) try {
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
} RG_ABI_GUARD

int rg_send_check(rg_engine *h, uint32_t flags, const char *who) {
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
    bool probed = false;
    if (m->m_logterm) { // this tick may carry log terms: resolve the flagged hints first
        const int prc = rg_hint_prepass(h, ms, h->ins_arena != nullptr, &probed);
        if (prc) return prc;
    }
    const int trc = rg_tick_impl(h, ms, send);
    if (trc == RG_OK && m->m_logterm && h->ins_arena) { // (rg_require_hints_resolved: this tick CAN have raised RG_OUT_HOST_HINT;
        h->hint_check_due = true;                       //  whether it did is in the probe's word, read at the next entry point)
        h->hint_probe_pending = probed;
    }
    return trc;
}

extern "C" int rg_tick_device(rg_engine *h, const rg_msgs *m) try {
    if (!h || !m || !m->m_index || !m->m_commit || !m->m_flags)
        return rg_fail(RG_ERR_INVALID_ARG, "rg_tick_device: m_index, m_commit and m_flags are required");
    return rg_tick_device_impl(h, m, nullptr);
} RG_ABI_GUARD

extern "C" int rg_tick_device_send(rg_engine *h, const rg_msgs *m, uint64_t max_entries_per_msg, uint32_t flags) try {
    if (!h || !m || !m->m_index || !m->m_commit || !m->m_flags)
        return rg_fail(RG_ERR_INVALID_ARG, "rg_tick_device_send: m_index, m_commit and m_flags are required");
    int rc = rg_send_check(h, flags, "rg_tick_device_send");
    if (rc) return rc;
    const RgSendReq send = {(u64)max_entries_per_msg, (u32)flags};
    return rg_tick_device_impl(h, m, &send);
} RG_ABI_GUARD

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
                                    uint64_t *dev_commit_t) try {
    if (!h || !m || !dev_out_t || n_ticks == 0 || n_ticks > RG_MAX_FUSE)
        return rg_fail(RG_ERR_INVALID_ARG, "rg_tick_device_fused: need 1..%d ticks and an out buffer", RG_MAX_FUSE);
    if (h->ins_arena)
        return rg_fail(RG_ERR_STATE, "rg_tick_device_fused: engines with device Inflights (max_inflight > 0) need "
                                     "rg_send_appends after every tick; fused launches are not available");
    for (u32 t = 0; t < n_ticks; t++)
        if (!m[t].m_index || !m[t].m_commit || !m[t].m_flags)
            return rg_fail(RG_ERR_INVALID_ARG, "rg_tick_device_fused: tick %u lacks m_index/m_commit/m_flags", t);
    RG_ENTER(h);
    h->fused_done = 0;
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
            ms.mhr = ms.mh;
            bool probed = false;
            int rc = rg_hint_prepass(h, ms, true, &probed);
            if (rc) return rc;
            rc = rg_tick_impl(h, ms);
            if (rc) return rc;
            RG_HIP(hipMemcpyAsync(dev_out_t + (size_t)e * h->G, h->st.out, h->G * 4, hipMemcpyDeviceToDevice, h->stream));
            if (dev_commit_t)
                RG_HIP(hipMemcpyAsync(dev_commit_t + (size_t)e * h->G, h->st.commit, h->G * 8, hipMemcpyDeviceToDevice, h->stream));
            e++;
            h->fused_done = e;
            // Exact or loud: did this tick leave a reject to the host (RG_OUT_HOST_HINT)? The reference applies that reject
            // before anything later (raft_log.rs:209-235 -> raft.rs:1657-1660), so the call STOPS behind the tick -- the later
            // ticks are the host's to submit again once rg_resolve_host_hints (or the re-stepped reject) has answered. The
            // pre-pass's word says "none" without waiting for the tick; only a raised word costs the count and its wait.
            if (probed) RG_HIP(hipEventSynchronize(h->ev_hint));
            if (probed && e < n_ticks && *(volatile u32 *)h->pin_hint_raised != 0) {
                RG_HIP(hipMemsetAsync(h->d_counts, 0, 32, h->stream));
                const unsigned grid = rg_grid(h->G, RG_BLOCK) < 2048 ? rg_grid(h->G, RG_BLOCK) : 2048;
                hipLaunchKernelGGL(k_count_out, dim3(grid), dim3(RG_BLOCK), 0, h->stream, (const u32 *)h->st.out, h->G, h->d_counts);
                u64 c[3] = {0, 0, 0};
                RG_HIP(hipMemcpyAsync(c, h->d_counts, 24, hipMemcpyDeviceToHost, h->stream));
                RG_HIP(hipStreamSynchronize(h->stream));
                if (c[2]) {
                    h->ticked = true;
                    h->host_res_valid = false;
                    h->out_is_dense = true;
                    return rg_fail(RG_ERR_HOST_HINT, "rg_tick_device_fused: tick %u of %u left %llu group(s) with a reject for the host "
                                                     "(RG_OUT_HOST_HINT); the call stopped behind it -- %u tick(s) applied "
                                                     "(rg_fused_ticks_done), RG_COL_OUT / RG_COL_HOST_HINT are that tick's",
                                   e - 1, n_ticks, (unsigned long long)c[2], e);
                }
            }
        }
        t = e;
        h->fused_done = t;
    }
    h->ticked = true;
    h->host_res_valid = false;
    h->out_is_dense = true;
    return RG_OK;
} RG_ABI_GUARD

extern "C" int rg_fused_ticks_done(const rg_engine *h, uint32_t *n) try {
    if (!h || !n) return rg_fail(RG_ERR_INVALID_ARG, "rg_fused_ticks_done: bad argument");
    *n = h->fused_done;
    return RG_OK;
} RG_ABI_GUARD

int rg_ensure_msg_arena(rg_engine *h) {
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

int rg_tick_host_impl(rg_engine *h, const rg_msgs *m, const RgSendReq *send) {
    RG_ENTER(h);
    {   // (device Inflights: nothing of the next step is enqueued while a host hint of the last one is unanswered)
        const int hrc__ = rg_require_hints_resolved(h, "rg_tick");
        if (hrc__) return hrc__;
    }
    int rc = rg_ensure_msg_arena(h);
    if (rc) return rc;
    const size_t colb = (size_t)h->P * h->stride * 8;
    bool probed = false;
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
        rc = rg_hint_prepass(h, ms, h->ins_arena != nullptr, &probed);
        if (rc) return rc;
    }
    rc = rg_tick_impl(h, ms, send);
    if (rc) return rc;
    if (m->m_logterm && h->ins_arena) {
        h->hint_check_due = true;
        h->hint_probe_pending = probed;
    }
    // the engine-owned message columns must read "no events" outside a tick (sparse-path invariant)
    RG_HIP(hipMemsetAsync((void *)h->staged.mflags, 0, h->stride * 8, h->stream));
    RG_HIP(hipStreamSynchronize(h->stream)); // caller-owned host buffers may be reused after return
    return RG_OK;
}

extern "C" int rg_tick(rg_engine *h, const rg_msgs *m) try {
    if (!h || !m || !m->m_index || !m->m_commit || !m->m_flags)
        return rg_fail(RG_ERR_INVALID_ARG, "rg_tick: m_index, m_commit and m_flags are required");
    return rg_tick_host_impl(h, m, nullptr);
} RG_ABI_GUARD

extern "C" int rg_tick_send(rg_engine *h, const rg_msgs *m, uint64_t max_entries_per_msg, uint32_t flags) try {
    if (!h || !m || !m->m_index || !m->m_commit || !m->m_flags)
        return rg_fail(RG_ERR_INVALID_ARG, "rg_tick_send: m_index, m_commit and m_flags are required");
    int rc = rg_send_check(h, flags, "rg_tick_send");
    if (rc) return rc;
    const RgSendReq send = {(u64)max_entries_per_msg, (u32)flags};
    return rg_tick_host_impl(h, m, &send);
} RG_ABI_GUARD


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

extern "C" int rg_recompute(rg_engine *h) try {
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
} RG_ABI_GUARD

extern "C" int rg_maximal_committed_index(rg_engine *h, uint64_t *host_mci, uint8_t *host_gc) try {
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
} RG_ABI_GUARD


extern "C" int rg_heartbeat_commits(rg_engine *h, uint64_t *dev_hb, uint64_t *host_hb) try {
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
} RG_ABI_GUARD

extern "C" int rg_results(rg_engine *h, uint64_t *host_commit, uint32_t *host_out) try {
    if (!h) return rg_fail(RG_ERR_INVALID_ARG, "rg_results: null engine");
    if (!h->ticked) return rg_fail(RG_ERR_STATE, "rg_results: no tick has run yet");
    RG_ENTER(h);
    if (host_commit) RG_HIP(hipMemcpyAsync(host_commit, h->st.commit, h->G * 8, hipMemcpyDeviceToHost, h->stream));
    if (host_out) RG_HIP(hipMemcpyAsync(host_out, h->st.out, h->G * 4, hipMemcpyDeviceToHost, h->stream));
    RG_HIP(hipStreamSynchronize(h->stream));
    return RG_OK;
} RG_ABI_GUARD

extern "C" int rg_result_counts(rg_engine *h, uint64_t *n_changed, uint64_t *n_fault) try {
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
} RG_ABI_GUARD

extern "C" int rg_host_hints(rg_engine *h, rg_host_hint *host_items, uint64_t cap, uint64_t *n) try {
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
} RG_ABI_GUARD

extern "C" int rg_resolve_host_hints(rg_engine *h, const rg_resolved_hint *items, uint64_t n, uint8_t *host_applied) try {
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
} RG_ABI_GUARD

extern "C" int rg_msg_stats(rg_engine *h, const uint8_t *d_m_flags, uint64_t counts[5]) try {
    if (!h || !d_m_flags || !counts) return rg_fail(RG_ERR_INVALID_ARG, "rg_msg_stats: bad argument");
    RG_ENTER(h);
    RG_HIP(hipMemsetAsync(h->d_counts, 0, 40, h->stream));
    const unsigned grid = rg_grid(h->G, RG_BLOCK) < 1024 ? rg_grid(h->G, RG_BLOCK) : 1024;
    hipLaunchKernelGGL(k_msg_stats, dim3(grid), dim3(RG_BLOCK), 0, h->stream, (const u64 *)d_m_flags,
                       (const u32 *)h->st.cfg, h->G, h->d_counts);
    RG_HIP(hipMemcpyAsync(counts, h->d_counts, 40, hipMemcpyDeviceToHost, h->stream));
    RG_HIP(hipStreamSynchronize(h->stream));
    return RG_OK;
} RG_ABI_GUARD

// ------------------------------------------------------------------------------------------------
// votes / liveness
// ------------------------------------------------------------------------------------------------
extern "C" int rg_vote_result(rg_engine *h, const uint8_t *yes, const uint8_t *no, uint8_t *result) try {
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
} RG_ABI_GUARD

extern "C" int rg_tally_votes(rg_engine *h, const uint8_t *yes, const uint8_t *no, uint8_t *granted, uint8_t *rejected,
                              uint8_t *result) try {
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
} RG_ABI_GUARD

extern "C" int rg_quorum_recently_active(rg_engine *h, uint8_t *result) try {
    if (!h || !result) return rg_fail(RG_ERR_INVALID_ARG, "rg_quorum_recently_active: bad argument");
    RG_ENTER(h);
    u8 *d = reinterpret_cast<u8 *>(h->d_scratch);
    hipLaunchKernelGGL(k_quorum_active, dim3(rg_grid(h->G, RG_BLOCK)), dim3(RG_BLOCK), 0, h->stream, h->st, d);
    RG_HIP(hipMemcpyAsync(result, d, h->G, hipMemcpyDeviceToHost, h->stream));
    RG_HIP(hipStreamSynchronize(h->stream));
    return RG_OK;
} RG_ABI_GUARD


