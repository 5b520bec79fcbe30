// abi_mirror.hip -- the sparse path, the host mirror of RawNode::step, the resident mailbox, flushes (include/raftgroups.h: "message-at-a-time host mirror", "sparse path", "mailbox")
// There is NO CPU fallback anywhere in this file: without a HIP device every entry point fails.
#include "rg_engine.h"
#include "rg_kernels_sparse.h"

static RgIngest rg_ingest_args(rg_engine *h, const rg_wire_msg *rec, u64 n, const RgClear &clr) {
    RgIngest a;
    a.rec = rec;
    a.n = n;
    a.G = h->G;
    a.stride = h->stride;
    a.P = h->P;
    a.mi = (u64 *)h->staged.mi;
    a.mc = (u64 *)h->staged.mc;
    a.mh = (u64 *)h->staged.mh;
    a.mrs = (u64 *)h->staged.mrs;
    a.mlt = (u64 *)h->staged.mlt;
    a.mflags32 = (u32 *)h->staged.mflags;
    a.gmark = h->gmark;
    a.epoch = h->epoch;
    a.list = h->list;
    a.counters = h->counters;
    a.clr = clr;
    return a;
}

// Two {touched groups, dropped records} counter pairs take turns: a sparse tick uses one, the ingest kernel of the same
// window resets the other for the tick after it, rg_ctr_flip switches -- no memset command per tick.
static u32 *rg_ctr_other(rg_engine *h) { return h->counters == h->counters_base ? h->counters_base + 2 : h->counters_base; }
static void rg_ctr_flip(rg_engine *h) { h->counters = rg_ctr_other(h); }

int rg_ensure_sparse(rg_engine *h) {
    if (h->sparse_arena) return RG_OK;
    int rc = rg_ensure_msg_arena(h);
    if (rc) return rc;
    const size_t G = h->stride;
    const size_t o_gmark = 0, o_list = rg_align(G * 4), o_rl = o_list + rg_align(G * 8), o_rc = o_rl + rg_align(G * 8);
    const size_t o_ro = o_rc + rg_align(G * 8), o_cnt = o_ro + rg_align(G * 4), total = o_cnt + 256;
    RG_HIP(hipMalloc(&h->sparse_arena, total));
    RG_HIP(hipMemsetAsync(h->sparse_arena, 0, total, h->stream));
    h->gmark = (u32 *)(h->sparse_arena + o_gmark);
    h->list = (u64 *)(h->sparse_arena + o_list);
    h->res_list = (u64 *)(h->sparse_arena + o_rl);
    h->res_commit = (u64 *)(h->sparse_arena + o_rc);
    h->res_out = (u32 *)(h->sparse_arena + o_ro);
    h->counters_base = (u32 *)(h->sparse_arena + o_cnt);
    h->counters = h->counters_base;
    return RG_OK;
}

extern "C" int rg_ingest(rg_engine *h, const rg_wire_msg *records, uint64_t n, uint64_t *n_duplicates) try {
    if (!h || (!records && n)) return rg_fail(RG_ERR_INVALID_ARG, "rg_ingest: bad argument");
    if (n_duplicates) *n_duplicates = 0;
    if (n == 0) return RG_OK;
    RG_ENTER(h);
    {   // (device Inflights: nothing of the next step is enqueued while a host hint of the last one is unanswered)
        const int hrc__ = rg_require_hints_resolved(h, "rg_ingest");
        if (hrc__) return hrc__;
    }
    int rc = rg_ensure_sparse(h);
    if (rc) return rc;
    if (n > h->d_records_cap) {
        if (h->d_records) {
            RG_HIP(hipStreamSynchronize(h->stream));
            (void)hipFree(h->d_records);
            h->d_records = nullptr;
        }
        u64 cap = h->d_records_cap ? h->d_records_cap : 4096;
        while (cap < n) cap *= 2;
        RG_HIP(hipMalloc(&h->d_records, (cap + RG_INGEST_BLOCK) * sizeof(rg_wire_msg)));
        h->d_records_cap = cap;
    }
    u32 dup0 = 0; // duplicates so far in this tick window (device ingests included)
    RG_HIP(hipMemcpyAsync(&dup0, h->counters + 1, 4, hipMemcpyDeviceToHost, h->stream));
    RG_HIP(hipMemcpyAsync(h->d_records, records, n * sizeof(rg_wire_msg), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_ingest, dim3(rg_grid(n, RG_INGEST_BLOCK)), dim3(RG_INGEST_BLOCK), 0, h->stream,
                       rg_ingest_args(h, h->d_records, n, RgClear{nullptr, nullptr, 0u, rg_ctr_other(h)}));
    u32 dup = 0;
    RG_HIP(hipMemcpyAsync(&dup, h->counters + 1, 4, hipMemcpyDeviceToHost, h->stream));
    RG_HIP(hipStreamSynchronize(h->stream)); // the caller's record array may be reused after return
    if (n_duplicates) *n_duplicates = dup - dup0; // dup0 was read before the kernel ran (stream order)
    h->ingested_upper += n;
    return RG_OK;
} RG_ABI_GUARD

extern "C" int rg_ingest_device(rg_engine *h, const rg_wire_msg *dev_records, uint64_t n) try {
    if (!h || (!dev_records && n)) return rg_fail(RG_ERR_INVALID_ARG, "rg_ingest_device: bad argument");
    if (n == 0) return RG_OK;
    RG_ENTER(h);
    {   // (device Inflights: nothing of the next step is enqueued while a host hint of the last one is unanswered)
        const int hrc__ = rg_require_hints_resolved(h, "rg_ingest_device");
        if (hrc__) return hrc__;
    }
    int rc = rg_ensure_sparse(h);
    if (rc) return rc;
    hipLaunchKernelGGL(k_ingest, dim3(rg_grid(n, RG_INGEST_BLOCK)), dim3(RG_INGEST_BLOCK), 0, h->stream,
                       rg_ingest_args(h, dev_records, n, RgClear{nullptr, nullptr, 0u, rg_ctr_other(h)}));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return rg_fail(RG_ERR_NO_DEVICE, "rg_ingest_device: %s", hipGetErrorString(e));
    h->ingested_upper += n;
    return RG_OK;
} RG_ABI_GUARD

extern "C" int rg_ingested_duplicates(rg_engine *h, uint64_t *n_duplicates) try {
    if (!h || !n_duplicates) return rg_fail(RG_ERR_INVALID_ARG, "rg_ingested_duplicates: bad argument");
    *n_duplicates = 0;
    if (!h->sparse_arena) return RG_OK;
    RG_ENTER(h);
    u32 dup = 0;
    RG_HIP(hipMemcpyAsync(&dup, h->counters + 1, 4, hipMemcpyDeviceToHost, h->stream));
    RG_HIP(hipStreamSynchronize(h->stream));
    *n_duplicates = dup;
    return RG_OK;
} RG_ABI_GUARD

// Everything of a sparse tick that needs no host round trip: clear the previous results, resolve hints, tick the
// listed groups, gather their results (also into `packed` when given). `upper` bounds the list length.
int rg_sparse_enqueue(rg_engine *h, u64 upper, char *packed, bool any_logterm, bool out_cleared = false,
                             const RgIngest *one_launch = nullptr, const RgSmallSend *small_send = nullptr) {
    int src = rg_settle_send(h); // (walks the PREVIOUS tick's result list, before it is cleared below)
    if (src) return src;
    // RG_COL_OUT must hold zeros for every group this tick does not touch
    if (out_cleared) {
        // (the ingest kernel of this flush has done it)
    } else if (h->out_is_dense) {
        RG_HIP(hipMemsetAsync(h->st.out, 0, h->stride * 4, h->stream));
    } else if (h->last_sparse_n) {
        hipLaunchKernelGGL(k_clear_out, dim3(rg_grid(h->last_sparse_n, 256)), dim3(256), 0, h->stream, h->res_list,
                           h->last_sparse_n, h->st.out);
    }
    h->out_is_dense = false;
    h->last_sparse_n = 0;
    h->host_res_valid = false;
    if (!upper) return RG_OK;
    if (any_logterm) { // (rg_require_hints_resolved: a sparse tick has no probe, the check counts)
        h->hint_check_due = true;
        h->hint_probe_pending = false;
    }
    RgMsgs ms = h->staged;
    ms.mhr = ms.mh;
    u64 *mf = (u64 *)h->staged.mflags;
    RgListOut lo; // the tick gathers its own results (one launch less than a separate gather kernel)
    lo.rl = h->res_list;
    lo.rc = h->res_commit;
    lo.ro = h->res_out;
    lo.packed = packed;
    if (one_launch && small_send) { // ... and the touched groups' send stage as well (rg_flush_send)
        if (any_logterm) ms.mhr = h->rhint;
        switch (h->P) {
        case 1: rg_launch_flush_small_send_t<1>(h->stream, h->st, ms, h->any_group_commit, *one_launch, h->rhint, mf, lo, *small_send); break;
        case 2: rg_launch_flush_small_send_t<2>(h->stream, h->st, ms, h->any_group_commit, *one_launch, h->rhint, mf, lo, *small_send); break;
        case 3: rg_launch_flush_small_send_t<3>(h->stream, h->st, ms, h->any_group_commit, *one_launch, h->rhint, mf, lo, *small_send); break;
        case 4: rg_launch_flush_small_send_t<4>(h->stream, h->st, ms, h->any_group_commit, *one_launch, h->rhint, mf, lo, *small_send); break;
        case 5: rg_launch_flush_small_send_t<5>(h->stream, h->st, ms, h->any_group_commit, *one_launch, h->rhint, mf, lo, *small_send); break;
        case 6: rg_launch_flush_small_send_t<6>(h->stream, h->st, ms, h->any_group_commit, *one_launch, h->rhint, mf, lo, *small_send); break;
        case 7: rg_launch_flush_small_send_t<7>(h->stream, h->st, ms, h->any_group_commit, *one_launch, h->rhint, mf, lo, *small_send); break;
        default: rg_launch_flush_small_send_t<8>(h->stream, h->st, ms, h->any_group_commit, *one_launch, h->rhint, mf, lo, *small_send); break;
        }
        hipError_t e1 = hipGetLastError();
        if (e1 != hipSuccess) return rg_fail(RG_ERR_NO_DEVICE, "sparse tick + send stage: launch failed: %s", hipGetErrorString(e1));
        h->tick_launches++;
        return RG_OK;
    }
    if (one_launch) { // <= 256 records: ingest, hint resolution, tick and results in ONE single-workgroup launch
        if (any_logterm) ms.mhr = h->rhint;
        switch (h->P) {
        case 1: rg_launch_flush_small_t<1>(h->stream, h->st, ms, h->any_group_commit, *one_launch, h->rhint, mf, lo); break;
        case 2: rg_launch_flush_small_t<2>(h->stream, h->st, ms, h->any_group_commit, *one_launch, h->rhint, mf, lo); break;
        case 3: rg_launch_flush_small_t<3>(h->stream, h->st, ms, h->any_group_commit, *one_launch, h->rhint, mf, lo); break;
        case 4: rg_launch_flush_small_t<4>(h->stream, h->st, ms, h->any_group_commit, *one_launch, h->rhint, mf, lo); break;
        case 5: rg_launch_flush_small_t<5>(h->stream, h->st, ms, h->any_group_commit, *one_launch, h->rhint, mf, lo); break;
        case 6: rg_launch_flush_small_t<6>(h->stream, h->st, ms, h->any_group_commit, *one_launch, h->rhint, mf, lo); break;
        case 7: rg_launch_flush_small_t<7>(h->stream, h->st, ms, h->any_group_commit, *one_launch, h->rhint, mf, lo); break;
        default: rg_launch_flush_small_t<8>(h->stream, h->st, ms, h->any_group_commit, *one_launch, h->rhint, mf, lo); break;
        }
        hipError_t e1 = hipGetLastError();
        if (e1 != hipSuccess) return rg_fail(RG_ERR_NO_DEVICE, "sparse tick: launch failed: %s", hipGetErrorString(e1));
        h->tick_launches++;
        return RG_OK;
    }
    if (any_logterm) { // records may carry log terms: resolve the touched groups' flagged hints first
        ms.mhr = h->rhint;
        hipLaunchKernelGGL(k_resolve_hints_list, dim3(rg_grid(upper, RG_BLOCK)), dim3(RG_BLOCK), 0, h->stream, h->st, ms,
                           h->P, h->rhint, (const u64 *)h->list, (const u32 *)h->counters);
    }
    switch (h->P) {
    case 1: rg_launch_tick_list_t<1>(h->stream, h->st, ms, h->any_group_commit, h->list, h->counters, upper, mf, lo); break;
    case 2: rg_launch_tick_list_t<2>(h->stream, h->st, ms, h->any_group_commit, h->list, h->counters, upper, mf, lo); break;
    case 3: rg_launch_tick_list_t<3>(h->stream, h->st, ms, h->any_group_commit, h->list, h->counters, upper, mf, lo); break;
    case 4: rg_launch_tick_list_t<4>(h->stream, h->st, ms, h->any_group_commit, h->list, h->counters, upper, mf, lo); break;
    case 5: rg_launch_tick_list_t<5>(h->stream, h->st, ms, h->any_group_commit, h->list, h->counters, upper, mf, lo); break;
    case 6: rg_launch_tick_list_t<6>(h->stream, h->st, ms, h->any_group_commit, h->list, h->counters, upper, mf, lo); break;
    case 7: rg_launch_tick_list_t<7>(h->stream, h->st, ms, h->any_group_commit, h->list, h->counters, upper, mf, lo); break;
    default: rg_launch_tick_list_t<8>(h->stream, h->st, ms, h->any_group_commit, h->list, h->counters, upper, mf, lo); break;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return rg_fail(RG_ERR_NO_DEVICE, "sparse tick: launch failed: %s", hipGetErrorString(e));
    h->tick_launches++;
    return RG_OK;
}

// Bookkeeping once the sparse tick's group count is known on the host.
int rg_sparse_finish(rg_engine *h, u64 n_groups) {
    h->last_sparse_n = n_groups;
    h->ingested_upper = 0;
    h->epoch++;
    if (h->epoch == 0) { // epoch wrapped: the marks are ambiguous, reset them
        RG_HIP(hipMemsetAsync(h->gmark, 0, h->stride * 4, h->stream));
        h->epoch = 1;
    }
    h->ticked = true;
    h->send_ready = true;
    return RG_OK;
}

extern "C" int rg_tick_ingested(rg_engine *h, uint64_t *n_groups) try {
    if (!h) return rg_fail(RG_ERR_INVALID_ARG, "rg_tick_ingested: null engine");
    if (n_groups) *n_groups = 0;
    RG_ENTER(h);
    {   // (device Inflights: nothing of the next step is enqueued while a host hint of the last one is unanswered)
        const int hrc__ = rg_require_hints_resolved(h, "rg_tick_ingested");
        if (hrc__) return hrc__;
    }
    int rc = rg_ensure_sparse(h);
    if (rc) return rc;
    const u64 upper = h->ingested_upper < h->G ? h->ingested_upper : h->G;
    rc = rg_sparse_enqueue(h, upper, nullptr, true); // device-side ingests may carry log terms
    if (rc) return rc;
    u32 n = 0;
    if (upper) {
        RG_HIP(hipMemcpyAsync(&n, h->counters, 4, hipMemcpyDeviceToHost, h->stream));
        RG_HIP(hipStreamSynchronize(h->stream));
        rg_ctr_flip(h); // (the next window's pair was reset by this window's ingest kernels)
    }
    rc = rg_sparse_finish(h, n);
    if (rc) return rc;
    if (n_groups) *n_groups = h->last_sparse_n;
    return RG_OK;
} RG_ABI_GUARD

extern "C" int rg_ingested_results(rg_engine *h, uint64_t *groups, uint64_t *commit, uint32_t *out, uint64_t cap,
                                   uint64_t *n) try {
    if (!h || !n) return rg_fail(RG_ERR_INVALID_ARG, "rg_ingested_results: bad argument");
    *n = h->last_sparse_n;
    const u64 k = h->last_sparse_n < cap ? h->last_sparse_n : cap;
    if (k == 0) return RG_OK;
    if (h->host_res_valid) { // the single-copy flush already brought them over
        if (groups) memcpy(groups, h->host_res_groups.data(), k * 8);
        if (commit) memcpy(commit, h->host_res_commit.data(), k * 8);
        if (out) memcpy(out, h->host_res_out.data(), k * 4);
        return RG_OK;
    }
    RG_ENTER(h);
    if (groups) RG_HIP(hipMemcpyAsync(groups, h->res_list, k * 8, hipMemcpyDeviceToHost, h->stream));
    if (commit) RG_HIP(hipMemcpyAsync(commit, h->res_commit, k * 8, hipMemcpyDeviceToHost, h->stream));
    if (out) RG_HIP(hipMemcpyAsync(out, h->res_out, k * 4, hipMemcpyDeviceToHost, h->stream));
    RG_HIP(hipStreamSynchronize(h->stream));
    return RG_OK;
} RG_ABI_GUARD


// ------------------------------------------------------------------------------------------------
// host mirror of RawNode::step for MsgAppendResponse
// ------------------------------------------------------------------------------------------------
static void rg_mirror_init(rg_engine *h) {
    if (h->host_mirror) return;
    h->peer_ids.assign(h->G * 8, 0);
    h->terms.assign(h->G, 0);
    const size_t n = (size_t)h->P * h->stride;
    h->q_mi.assign(n, 0);
    h->q_mc.assign(n, 0);
    h->q_mh.assign(n, 0);
    h->q_mrs.assign(n, 0);
    h->q_mlt.assign(n, 0);
    h->q_mf.assign(h->G * 8, 0);
    h->host_mirror = true;
}

extern "C" int rg_set_peers(rg_engine *h, uint64_t group, const uint64_t *peer_ids, uint32_t n, uint64_t term) try {
    if (!h || !peer_ids || group >= h->G || n > h->P) return rg_fail(RG_ERR_INVALID_ARG, "rg_set_peers: bad argument");
    rg_mirror_init(h);
    for (u32 i = 0; i < 8; i++) h->peer_ids[group * 8 + i] = i < n ? peer_ids[i] : 0; // id 0 is illegal (raw_node.rs:303)
    h->terms[group] = term;
    return RG_OK;
} RG_ABI_GUARD

static int rg_find_slot(rg_engine *h, u64 group, u64 id) {
    if (id == 0) return -1;
    for (u32 i = 0; i < h->P; i++)
        if (h->peer_ids[group * 8 + i] == id) return (int)i;
    return -1;
}

static void rg_touch(rg_engine *h, u64 group) {
    u64 row = 0;
    memcpy(&row, &h->q_mf[group * 8], 8);
    if (row == 0) h->q_dirty.push_back(group);
}

static int rg_self_slot(rg_engine *h, u64 group, u32 *slot);

// A response whose `from` is the leader's OWN id. No follower sends one; the reference would run it against the leader's
// own Progress, where a well-formed one changes nothing (matched = persisted = last_index: a reject is stale, an accept at
// or below matched is a no-op). Here the leader's slot carries the LOCAL events of the tick -- VALID is
// on_persist_entries, and the REJECT bit is RG_MF_BECOME_LEADER, whose m_hint is the new TERM: a spoofed or misrouted
// reject would run Raft::reset + become_leader with reject_hint as the term. So the mirror drops such a message (RG_OK,
// nothing queued); local events enter through rg_local_* only.
static int rg_from_self(rg_engine *h, u64 group, int slot, bool *is_self) {
    u32 self;
    int rc = rg_self_slot(h, group, &self);
    if (rc) return rc;
    *is_self = (u32)slot == self;
    return RG_OK;
}

extern "C" int rg_step(rg_engine *h, uint64_t group, const rg_append_response *m) try {
    if (!h || !m || group >= h->G) return rg_fail(RG_ERR_INVALID_ARG, "rg_step: bad argument");
    if (!h->host_mirror) return rg_fail(RG_ERR_STATE, "rg_step: rg_set_peers was never called");
    // RawNode::step (src/raw_node.rs:402-411): MsgAppendResponse is not a local message type; a response from an id
    // without a Progress is rejected BEFORE Raft::step looks at the term, so a removed peer cannot depose the leader
    const int slot = rg_find_slot(h, group, m->from);
    if (slot < 0) return rg_fail(RG_ERR_STEP_PEER_NOT_FOUND, "rg_step: peer %llu not in group %llu (raw_node.rs:407-410)",
                                 (unsigned long long)m->from, (unsigned long long)group);
    // Raft::step term gate (src/raft.rs:1282-1411); term 0 skips the gate (":1282 local message") and falls
    // through to step_leader exactly as in the reference
    if (m->term != 0) {
        if (m->term > h->terms[group])
            return rg_fail(RG_ERR_HIGHER_TERM, "rg_step: message term %llu > leader term %llu: step down (raft.rs:1284-1348)",
                           (unsigned long long)m->term, (unsigned long long)h->terms[group]);
        if (m->term < h->terms[group]) return RG_OK; // stale term: ignored (raft.rs:1349-1411)
    }
    bool from_self;
    int src = rg_from_self(h, group, slot, &from_self);
    if (src) return src;
    if (from_self) return RG_OK; // (dropped: see rg_from_self)
    u8 &f = h->q_mf[group * 8 + slot];
    if (f & (RG_MF_VALID | RG_MF_HEARTBEAT)) return rg_fail(RG_ERR_SLOT_BUSY, "rg_step: peer %llu already has a message queued; rg_flush first",
                                        (unsigned long long)m->from);
    rg_touch(h, group);
    const size_t o = (size_t)slot * h->stride + group;
    h->q_mi[o] = m->index;
    h->q_mc[o] = m->commit;
    h->q_mh[o] = m->reject_hint;
    h->q_mrs[o] = m->request_snapshot;
    h->q_mlt[o] = m->log_term;
    if (m->reject && m->log_term) h->q_any_logterm = true;
    f |= RG_MF_VALID | (m->reject ? RG_MF_REJECT : 0) | (m->request_snapshot ? RG_MF_HAS_RS : 0) |
         (m->ins_full ? RG_MF_INS_FULL : 0) | ((m->reject && m->log_term) ? RG_MF_HAS_LOGTERM : 0);
    return RG_OK;
} RG_ABI_GUARD

extern "C" int rg_step_heartbeat_response(rg_engine *h, uint64_t group, uint64_t from, uint64_t term, uint64_t commit,
                                          uint8_t ins_full) try {
    if (!h || group >= h->G) return rg_fail(RG_ERR_INVALID_ARG, "rg_step_heartbeat_response: bad argument");
    if (!h->host_mirror) return rg_fail(RG_ERR_STATE, "rg_step_heartbeat_response: rg_set_peers was never called");
    const int slot = rg_find_slot(h, group, from); // raw_node.rs:407-410 comes before the term gate
    if (slot < 0) return rg_fail(RG_ERR_STEP_PEER_NOT_FOUND, "rg_step_heartbeat_response: peer %llu not in group %llu",
                                 (unsigned long long)from, (unsigned long long)group);
    if (term != 0) {
        if (term > h->terms[group]) return rg_fail(RG_ERR_HIGHER_TERM, "rg_step_heartbeat_response: higher term: step down");
        if (term < h->terms[group]) return RG_OK;
    }
    bool from_self;
    int src = rg_from_self(h, group, slot, &from_self);
    if (src) return src;
    if (from_self) return RG_OK; // (dropped: see rg_from_self)
    u8 &f = h->q_mf[group * 8 + slot];
    if (f & (RG_MF_VALID | RG_MF_HEARTBEAT))
        return rg_fail(RG_ERR_SLOT_BUSY, "rg_step_heartbeat_response: peer %llu already has a message queued", (unsigned long long)from);
    rg_touch(h, group);
    h->q_mc[(size_t)slot * h->stride + group] = commit;
    f |= RG_MF_HEARTBEAT | (ins_full ? RG_MF_INS_FULL : 0);
    return RG_OK;
} RG_ABI_GUARD

static int rg_self_slot(rg_engine *h, u64 group, u32 *slot) {
    // the self slot lives in the device cfg word; the mirror keeps a host copy of the column, refreshed
    // whenever the column may have changed (rg_load_column / rg_set_config / rg_workload_init)
    if (!h->host_cfg_valid) {
        // (a resident mailbox workgroup sits on the stream: without this the copy below waits for its idle time-out)
        int qrc = rg_mailbox_quiesce(h);
        if (qrc) return qrc;
        h->host_cfg.resize(h->G);
        RG_HIP(hipMemcpyAsync(h->host_cfg.data(), h->st.cfg, h->G * 4, hipMemcpyDeviceToHost, h->stream));
        RG_HIP(hipStreamSynchronize(h->stream));
        h->host_cfg_valid = true;
    }
    *slot = RG_CFG_SELF(h->host_cfg[group]);
    return RG_OK;
}


extern "C" int rg_step_bytes(rg_engine *h, uint64_t group, const uint8_t *bytes, uint64_t len, uint8_t ins_full) try {
    if (!h || group >= h->G) return rg_fail(RG_ERR_INVALID_ARG, "rg_step_bytes: bad argument");
    rg_decoded_message m;
    int rc = rg_decode_message(bytes, len, &m);
    if (rc) return rc;
    switch (m.msg_type) {
    case 0: case 1: case 10: case 11: case 12: // MsgHup, MsgBeat, MsgUnreachable, MsgSnapStatus, MsgCheckQuorum: is_local_msg
        return rg_fail(RG_ERR_STEP_LOCAL_MSG, "rg_step_bytes: raft: cannot step raft local message (raw_node.rs:404-406)");
    case 4: { // MsgAppendResponse
        rg_append_response r;
        memset(&r, 0, sizeof(r));
        r.from = m.from;
        r.term = m.term;
        r.index = m.index;
        r.commit = m.commit;
        r.reject = (uint8_t)m.reject;
        r.reject_hint = m.reject_hint;
        r.log_term = m.log_term;
        r.request_snapshot = m.request_snapshot;
        r.ins_full = ins_full; // (not on the wire: the caller's Inflights::full() for m.from, as in rg_step)
        return rg_step(h, group, &r);
    }
    case 9: // MsgHeartbeatResponse
        return rg_step_heartbeat_response(h, group, m.from, m.term, m.commit, ins_full);
    default:
        return rg_fail(RG_ERR_NOT_ON_PATH, "rg_step_bytes: message type %u is not handled on this path (the host's Raft::step takes it)",
                       m.msg_type);
    }
} RG_ABI_GUARD

extern "C" int rg_local_append(rg_engine *h, uint64_t group, uint64_t new_last_index) try {
    if (!h || group >= h->G) return rg_fail(RG_ERR_INVALID_ARG, "rg_local_append: bad argument");
    if (!h->host_mirror) return rg_fail(RG_ERR_STATE, "rg_local_append: rg_set_peers was never called");
    u32 slot;
    int rc = rg_self_slot(h, group, &slot);
    if (rc) return rc;
    rg_touch(h, group);
    h->q_mc[(size_t)slot * h->stride + group] = new_last_index;
    h->q_mf[group * 8 + slot] |= RG_MF_APPEND;
    return RG_OK;
} RG_ABI_GUARD

extern "C" int rg_local_persisted(rg_engine *h, uint64_t group, uint64_t index) try {
    if (!h || group >= h->G) return rg_fail(RG_ERR_INVALID_ARG, "rg_local_persisted: bad argument");
    if (!h->host_mirror) return rg_fail(RG_ERR_STATE, "rg_local_persisted: rg_set_peers was never called");
    u32 slot;
    int rc = rg_self_slot(h, group, &slot);
    if (rc) return rc;
    u8 &f = h->q_mf[group * 8 + slot];
    if (f & RG_MF_VALID) return rg_fail(RG_ERR_SLOT_BUSY, "rg_local_persisted: already queued; rg_flush first");
    rg_touch(h, group);
    h->q_mi[(size_t)slot * h->stride + group] = index;
    f |= RG_MF_VALID;
    return RG_OK;
} RG_ABI_GUARD

extern "C" int rg_local_become_leader(rg_engine *h, uint64_t group, uint64_t term) try {
    if (!h || group >= h->G) return rg_fail(RG_ERR_INVALID_ARG, "rg_local_become_leader: bad argument");
    if (!h->host_mirror) return rg_fail(RG_ERR_STATE, "rg_local_become_leader: rg_set_peers was never called");
    if (term <= h->terms[group])
        return rg_fail(RG_ERR_INVALID_ARG, "rg_local_become_leader: term %llu is not above the group's term %llu",
                       (unsigned long long)term, (unsigned long long)h->terms[group]);
    u32 slot;
    int rc = rg_self_slot(h, group, &slot);
    if (rc) return rc;
    u64 row = 0;
    memcpy(&row, &h->q_mf[group * 8], 8);
    if (row) return rg_fail(RG_ERR_SLOT_BUSY, "rg_local_become_leader: the group already has events queued (they belong "
                                              "to the old term); rg_flush first");
    u8 &f = h->q_mf[group * 8 + slot];
    rg_touch(h, group);
    h->q_mh[(size_t)slot * h->stride + group] = term;
    f |= RG_MF_BECOME_LEADER;
    // responses of the new term pass the gate from now on (they may be queued behind the election in this very flush);
    // the flush checks the device's verdict and moves the gate BACK if the event was refused there (rg_settle_elections)
    h->q_elections.push_back({group, h->terms[group]});
    h->terms[group] = term;
    return RG_OK;
} RG_ABI_GUARD

// RawNode::report_unreachable / report_snapshot (src/raw_node.rs:692-709): MsgUnreachable / MsgSnapStatus stepped at a leader
static int rg_report(rg_engine *h, uint64_t group, uint64_t peer_id, u32 kind, const char *who) {
    if (!h || group >= h->G) return rg_fail(RG_ERR_INVALID_ARG, "%s: bad argument", who);
    if (!h->host_mirror) return rg_fail(RG_ERR_STATE, "%s: rg_set_peers was never called", who);
    const int slot = rg_find_slot(h, group, peer_id);
    if (slot < 0) return RG_OK; // "no progress available for {}": ignored (the reference drops the step's result as well)
    u64 row = 0;
    memcpy(&row, &h->q_mf[group * 8], 8);
    if (row) return rg_fail(RG_ERR_SLOT_BUSY, "%s: group %llu has traffic queued; rg_flush first (local messages apply in call order)",
                            who, (unsigned long long)group);
    const rg_progress_event ev = {group, (u32)slot, kind};
    return rg_progress_events(h, &ev, 1);
}
extern "C" int rg_report_unreachable(rg_engine *h, uint64_t group, uint64_t peer_id) try {
    return rg_report(h, group, peer_id, RG_EV_UNREACHABLE, "rg_report_unreachable");
} RG_ABI_GUARD
extern "C" int rg_report_snapshot(rg_engine *h, uint64_t group, uint64_t peer_id, int failure) try {
    return rg_report(h, group, peer_id, failure ? RG_EV_SNAPSHOT_FAILURE : RG_EV_SNAPSHOT_FINISH, "rg_report_snapshot");
} RG_ABI_GUARD

extern "C" int rg_mark_sent(rg_engine *h, uint64_t group, uint64_t peer_id) try {
    if (!h || group >= h->G) return rg_fail(RG_ERR_INVALID_ARG, "rg_mark_sent: bad argument");
    if (!h->host_mirror) return rg_fail(RG_ERR_STATE, "rg_mark_sent: rg_set_peers was never called");
    const int slot = rg_find_slot(h, group, peer_id);
    if (slot < 0) return rg_fail(RG_ERR_STEP_PEER_NOT_FOUND, "rg_mark_sent: peer %llu not in group %llu",
                                 (unsigned long long)peer_id, (unsigned long long)group);
    bool to_self;
    int src = rg_from_self(h, group, slot, &to_self);
    if (src) return src;
    if (to_self) return RG_OK; // the leader sends itself nothing (and SENT has no meaning on its slot)
    rg_touch(h, group);
    h->q_mf[group * 8 + slot] |= RG_MF_SENT;
    return RG_OK;
} RG_ABI_GUARD

// One sparse tick in ONE host<->device round trip: records (the caller's, or built from the mirror's queues when
// `recs` is NULL) -> pinned staging -> ingest / clear / hint resolve / tick / gather back to back -> one packed D2H
// copy of (groups, duplicates, {group, commit, out}...) -> one synchronisation. Results stay cached on the host.
struct rg_send_req { // run the send stage inside the same round trip (engines with device Inflights)
    u64 max_entries;
    u32 flags;
};

static int rg_sparse_threecall(rg_engine *h, const rg_wire_msg *recs, u64 n, u32 *dup_out, const rg_send_req *send) {
    // big batches are throughput-bound, not latency-bound: the packed copy (one slot per RECORD, not per group)
    // and the host-side unpacking cost more than two extra synchronisations (profiles/r01_sparse_path...)
    uint64_t d64 = 0, ng = 0;
    int rc = rg_ingest(h, recs, n, &d64);
    if (rc == RG_OK) rc = rg_tick_ingested(h, &ng);
    if (rc == RG_OK && send) rc = rg_send_appends(h, send->max_entries, send->flags);
    if (dup_out) *dup_out = (u32)d64;
    return rc;
}

static int rg_mailbox_flush(rg_engine *h, u64 n, bool any_logterm, bool *served, const rg_send_req *send);
static int rg_sparse_roundtrip(rg_engine *h, const rg_wire_msg *recs, u64 n, bool any_logterm, u32 *dup_out,
                               const rg_send_req *send = nullptr) {
    RG_HIP(hipSetDevice(h->cfg.device)); // (not RG_ENTER: this is the one path the resident mailbox kernel serves)
    int rc;
    if (h->ins_arena && h->hint_check_due) { // (rare: a log-term tick came before; the check needs the stream to itself)
        rc = rg_mailbox_quiesce(h);
        if (rc) return rc;
        rc = rg_require_hints_resolved(h, "rg_flush / rg_ingest_tick");
        if (rc) return rc;
    }
    rc = rg_ensure_sparse(h);
    if (rc) return rc;
    if (recs && n > RG_ROUNDTRIP_MAX) return rg_sparse_threecall(h, recs, n, dup_out, send);
    if (!recs) {
        n = 0;
        for (u64 g : h->q_dirty)
            for (u32 p = 0; p < h->P; p++) n += h->q_mf[g * 8 + p] != 0;
    }
    if (n > RG_INGEST_BLOCK || (send && !h->ins_arena)) { // not a flush the resident mailbox workgroup can serve: it leaves
        rc = rg_mailbox_quiesce(h);                         // now, before anything below waits for the stream or replaces
        if (rc) return rc;                                  // a buffer it reads
    }
    if (n > h->pin_records_cap) {
        if (h->pin_records) {
            RG_HIP(hipStreamSynchronize(h->stream));
            (void)hipHostFree(h->pin_records);
            h->pin_records = nullptr;
        }
        u64 cap = 4096;
        while (cap < n) cap *= 2;
        RG_HIP(hipHostMalloc(reinterpret_cast<void **>(&h->pin_records), cap * sizeof(rg_wire_msg), hipHostMallocDefault));
        h->pin_records_cap = cap;
    }
    if (recs) {
        if (n) memcpy(h->pin_records, recs, n * sizeof(rg_wire_msg));
    } else {
        u64 k = 0;
        for (u64 g : h->q_dirty) {
            for (u32 p = 0; p < h->P; p++) {
                const u8 f = h->q_mf[g * 8 + p];
                if (!f) continue;
                const size_t o = (size_t)p * h->stride + g;
                rg_wire_msg &r = h->pin_records[k++];
                r.group = g;
                r.index = h->q_mi[o];
                r.commit = h->q_mc[o];
                r.hint = h->q_mh[o];
                r.rs = h->q_mrs[o];
                r.log_term = h->q_mlt[o];
                r.slot = p;
                r.flags = f;
                r.pad = 0;
            }
        }
        if (n > RG_ROUNDTRIP_MAX) return rg_sparse_threecall(h, h->pin_records, n, dup_out, send);
    }
    // the resident mailbox kernel, when it is on: no launch, no synchronisation (rg_mailbox_flush says whether it took it)
    bool served = false;
    if (h->mbox_on) {
        rc = rg_mailbox_flush(h, n, any_logterm, &served, send);
        if (rc) return rc;
    }
    if (!served) {
        rc = rg_mailbox_quiesce(h); // this flush goes through launches on the stream
        if (rc) return rc;
    }
    u32 n_groups = 0, dup = 0;
    u64 upper = 0;
    bool fetch_items = false;
    if (served) {
        h->out_is_dense = false; // (what rg_sparse_enqueue records)
        h->tick_launches++;
        rg_ctr_flip(h);
        n_groups = reinterpret_cast<const u32 *>(h->pin_packed)[0];
        dup = reinterpret_cast<const u32 *>(h->pin_packed)[1];
        if (send) { // the request ran the stage of every touched group: its items are in pin_send (rg_tick_send_listed)
            upper = n; // (only "something was walked", below)
            fetch_items = true;
            h->send_cols_fresh = false;
            h->send_last_dense = false;
            h->host_items_valid = false;
        }
    } else {
    if (n > h->d_records_cap) {
        if (h->d_records) {
            RG_HIP(hipStreamSynchronize(h->stream));
            (void)hipFree(h->d_records);
            h->d_records = nullptr;
        }
        u64 cap = h->d_records_cap ? h->d_records_cap : 4096;
        while (cap < n) cap *= 2;
        RG_HIP(hipMalloc(&h->d_records, (cap + RG_INGEST_BLOCK) * sizeof(rg_wire_msg)));
        h->d_records_cap = cap;
    }
    const u64 upper_all = h->ingested_upper + n; // device ingests of this window count too
    upper = upper_all < h->G ? upper_all : h->G;
    if (upper > h->packed_cap) {
        if (h->d_packed) {
            RG_HIP(hipStreamSynchronize(h->stream));
            (void)hipFree(h->d_packed);
            (void)hipHostFree(h->pin_packed);
            h->d_packed = h->pin_packed = nullptr;
        }
        u64 cap = 4096;
        while (cap < upper) cap *= 2;
        RG_HIP(hipMalloc(&h->d_packed, RG_PACKED_HDR + cap * sizeof(rg_res_rec)));
        RG_HIP(hipHostMalloc(reinterpret_cast<void **>(&h->pin_packed), RG_PACKED_HDR + cap * sizeof(rg_res_rec),
                             hipHostMallocDefault));
        h->packed_cap = cap;
    }
    // Small batches are latency-bound: every HIP call costs the host 3-5 us. The kernels then read the records straight
    // out of the pinned staging buffer and write the packed results straight into pinned host memory (both are mapped into
    // the device's address space; a few KB over PCIe inside a kernel cost less than a copy command each way), and the
    // ingest kernel also zeroes the previous sparse tick's result words: ingest + tick + counter reset + ONE
    // synchronisation instead of copy, ingest, clear, tick, copy, reset, synchronisation.
    const bool zero_copy = n && upper <= RG_ZEROCOPY_MAX;
    // ... and up to one workgroup's worth of records the whole flush is ONE launch (k_flush_small)
    const bool one_launch = zero_copy && n <= RG_INGEST_BLOCK && !(h->ins_arena && h->send_ready);
    bool out_cleared = false;
    RgIngest fused_args;
    if (one_launch) {
        RgClear clr = {nullptr, nullptr, 0u, rg_ctr_other(h)};
        if (!h->out_is_dense) {
            clr.list = h->res_list;
            clr.out = h->st.out;
            clr.n = (u32)h->last_sparse_n;
            out_cleared = true;
        }
        fused_args = rg_ingest_args(h, h->pin_records, n, clr);
    } else if (n) {
        RgClear clr = {nullptr, nullptr, 0u, rg_ctr_other(h)};
        // (with device Inflights and an unconsumed send stage the previous result words are still needed: rg_settle_send)
        if (zero_copy && !h->out_is_dense && !(h->ins_arena && h->send_ready)) {
            clr.list = h->res_list;
            clr.out = h->st.out;
            clr.n = (u32)h->last_sparse_n;
            out_cleared = true;
        }
        const rg_wire_msg *src = h->pin_records;
        if (!zero_copy) {
            RG_HIP(hipMemcpyAsync(h->d_records, h->pin_records, n * sizeof(rg_wire_msg), hipMemcpyHostToDevice, h->stream));
            src = h->d_records;
        }
        hipLaunchKernelGGL(k_ingest, dim3(rg_grid(n, RG_INGEST_BLOCK)), dim3(RG_INGEST_BLOCK), 0, h->stream,
                           rg_ingest_args(h, src, n, clr));
    }
    // (records ingested on the device in this window may carry log terms the host has not seen)
    // One launch AND a send request: the stage of every touched group runs behind its tick inside k_flush_small_send, on the
    // tick's registers; its work items land in the device list and, through the mapped pinned buffer, in host memory.
    const bool stage_inside = one_launch && send;
    RgSmallSend small_send;
    if (stage_inside) {
        if (!h->pin_send)
            RG_HIP(hipHostMalloc(reinterpret_cast<void **>(&h->pin_send), 16 + RG_SEND_SPEC * sizeof(rg_send_item),
                                 hipHostMallocDefault));
        small_send.ins = h->ins;
        small_send.max_entries = send->max_entries;
        small_send.flags = send->flags;
        h->stage_max_entries = send->max_entries;
        h->stage_flags = send->flags;
        small_send.items = h->send_items;
        small_send.counter = h->send_counter;
        small_send.pin = h->pin_send;
        h->send_cols_fresh = false; // (what rg_send_enqueue records for a stage over a list)
        h->send_last_dense = false;
        h->host_items_valid = false;
    }
    rc = rg_sparse_enqueue(h, upper, zero_copy ? h->pin_packed : h->d_packed, any_logterm || h->ingested_upper != 0, out_cleared,
                           one_launch ? &fused_args : nullptr, stage_inside ? &small_send : nullptr);
    if (rc) return rc;
    // the send stage rides along: it walks the gathered list, whose length is still only on the device
    const u64 item_bound = upper * h->P;
    fetch_items = send && upper && (stage_inside || item_bound <= RG_SEND_SPEC);
    if (send && upper && !stage_inside) {
        rc = rg_send_enqueue(h, send->max_entries, send->flags, h->res_list, upper, h->counters);
        if (rc) return rc;
        if (fetch_items) {
            if (!h->pin_send)
                RG_HIP(hipHostMalloc(reinterpret_cast<void **>(&h->pin_send), 16 + RG_SEND_SPEC * sizeof(rg_send_item),
                                     hipHostMallocDefault));
            RG_HIP(hipMemcpyAsync(h->pin_send, h->send_counter, 4, hipMemcpyDeviceToHost, h->stream));
            RG_HIP(hipMemcpyAsync(h->pin_send + 16, h->send_items, item_bound * sizeof(rg_send_item), hipMemcpyDeviceToHost,
                                  h->stream));
        }
    }
    if (upper) {
        if (!zero_copy)
            RG_HIP(hipMemcpyAsync(h->pin_packed, h->d_packed, RG_PACKED_HDR + upper * sizeof(rg_res_rec), hipMemcpyDeviceToHost,
                                  h->stream));
        RG_HIP(hipStreamSynchronize(h->stream));
        rg_ctr_flip(h); // (the next window's pair was reset by this window's ingest kernel)
        n_groups = reinterpret_cast<const u32 *>(h->pin_packed)[0];
        dup = reinterpret_cast<const u32 *>(h->pin_packed)[1];
    }
    } // (!served)
    rc = rg_sparse_finish(h, n_groups);
    if (rc) return rc;
    if (send) {
        h->send_ready = false; // the stage of this tick has run (or had nothing to walk)
        h->send_bound = (u64)n_groups * h->P;
        if (!upper) RG_HIP(hipMemsetAsync(h->send_counter, 0, 4, h->stream));
        if (!upper) {
            h->host_items.clear();
            h->host_items_valid = true;
        } else if (fetch_items) {
            const u32 cnt = *reinterpret_cast<const u32 *>(h->pin_send);
            const rg_send_item *it = reinterpret_cast<const rg_send_item *>(h->pin_send + 16);
            h->host_items.assign(it, it + cnt);
            h->host_items_valid = true;
        }
    }
    if (dup_out) *dup_out = dup;
    const rg_res_rec *rec = reinterpret_cast<const rg_res_rec *>(h->pin_packed + RG_PACKED_HDR);
    h->host_res_groups.resize(n_groups);
    h->host_res_commit.resize(n_groups);
    h->host_res_out.resize(n_groups);
    for (u32 i = 0; i < n_groups; i++) {
        h->host_res_groups[i] = rec[i].group;
        h->host_res_commit[i] = rec[i].commit;
        h->host_res_out[i] = rec[i].out;
    }
    h->host_res_valid = true;
    return RG_OK;
}

// ---- the resident small-batch path (rg_mailbox_start; kernel: k_mailbox in rg_tick_kernels.h) ----
#define RG_MBOX_TICKS_PER_US 100ull /* wall_clock64(): constant 100 MHz */
#define RG_MBOX_MAX_US 200000ull    /* one launch never stays longer than this, whatever the host does */

int rg_mailbox_quiesce(rg_engine *h) {
    if (!h->mbox_running) return RG_OK;
    __atomic_store_n(&h->mbox->stop, 1u, __ATOMIC_RELEASE);
    hipError_t e = hipStreamSynchronize(h->stream);
    h->mbox_running = false;
    __atomic_store_n(&h->mbox->stop, 0u, __ATOMIC_RELEASE);
    if (e != hipSuccess) return rg_fail(RG_ERR_NO_DEVICE, "mailbox: %s", hipGetErrorString(e));
    return RG_OK;
}

static int rg_mailbox_launch(rg_engine *h) {
    RgMsgs ms = h->staged;
    ms.mhr = ms.mh;
    RgListOut lo;
    lo.rl = h->res_list;
    lo.rc = h->res_commit;
    lo.ro = h->res_out;
    lo.packed = h->pin_packed;
    RgClear clr = {h->res_list, h->st.out, 0u, nullptr};
    const RgIngest a0 = rg_ingest_args(h, h->pin_records, 0, clr);
    u64 *mf = (u64 *)h->staged.mflags;
    const u64 max_ticks = RG_MBOX_MAX_US * RG_MBOX_TICKS_PER_US;
    RgSmallSend ss0; // where a request's send stage (rg_flush_send) puts its work items; limit and flags come with the request
    memset(&ss0, 0, sizeof(ss0));
    if (h->ins_arena) {
        ss0.ins = h->ins;
        ss0.items = h->send_items;
        ss0.counter = h->send_counter;
        ss0.pin = h->pin_send;
    }
    switch (h->P) {
    case 1: rg_launch_mailbox_t<1>(h->stream, h->st, ms, h->any_group_commit, a0, h->counters_base, h->rhint, mf, lo, h->mbox, h->mbox_idle_ticks, max_ticks, ss0); break;
    case 2: rg_launch_mailbox_t<2>(h->stream, h->st, ms, h->any_group_commit, a0, h->counters_base, h->rhint, mf, lo, h->mbox, h->mbox_idle_ticks, max_ticks, ss0); break;
    case 3: rg_launch_mailbox_t<3>(h->stream, h->st, ms, h->any_group_commit, a0, h->counters_base, h->rhint, mf, lo, h->mbox, h->mbox_idle_ticks, max_ticks, ss0); break;
    case 4: rg_launch_mailbox_t<4>(h->stream, h->st, ms, h->any_group_commit, a0, h->counters_base, h->rhint, mf, lo, h->mbox, h->mbox_idle_ticks, max_ticks, ss0); break;
    case 5: rg_launch_mailbox_t<5>(h->stream, h->st, ms, h->any_group_commit, a0, h->counters_base, h->rhint, mf, lo, h->mbox, h->mbox_idle_ticks, max_ticks, ss0); break;
    case 6: rg_launch_mailbox_t<6>(h->stream, h->st, ms, h->any_group_commit, a0, h->counters_base, h->rhint, mf, lo, h->mbox, h->mbox_idle_ticks, max_ticks, ss0); break;
    case 7: rg_launch_mailbox_t<7>(h->stream, h->st, ms, h->any_group_commit, a0, h->counters_base, h->rhint, mf, lo, h->mbox, h->mbox_idle_ticks, max_ticks, ss0); break;
    default: rg_launch_mailbox_t<8>(h->stream, h->st, ms, h->any_group_commit, a0, h->counters_base, h->rhint, mf, lo, h->mbox, h->mbox_idle_ticks, max_ticks, ss0); break;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return rg_fail(RG_ERR_NO_DEVICE, "mailbox: launch failed: %s", hipGetErrorString(e));
    h->mbox_running = true;
    h->mbox_launches++;
    return RG_OK;
}

// One small flush through the mailbox: the records are in h->pin_records already. Returns RG_OK with *served = false
// when the request cannot go this way (the caller takes the launch path).
static int rg_mailbox_flush(rg_engine *h, u64 n, bool any_logterm, bool *served, const rg_send_req *send) {
    *served = false;
    // the body clears the PREVIOUS sparse tick's result words itself; a dense predecessor needs a memset on the stream
    if (!h->mbox_on || h->pub || h->out_is_dense || h->ingested_upper || n == 0 || n > RG_INGEST_BLOCK ||
        h->last_sparse_n > RG_ZEROCOPY_MAX || h->epoch == 0xffffffffu)
        return RG_OK;
    // device Inflights: only a flush that runs its send stage in the same request (rg_flush_send), with no stage of an
    // earlier tick left to settle (that one needs a launch), and a limit the request word can carry
    u32 lim = 0;
    if (h->ins_arena) {
        // (a tick that can raise RG_OUT_HOST_HINT, or one that follows such a tick, takes the launch path: rg_require_hints_resolved)
        if (any_logterm || h->hint_check_due) return RG_OK;
        if (!send || h->send_ready) return RG_OK;
        if (send->max_entries == ~0ULL) lim = 0xffffffffu;
        else if (send->max_entries >= 0xffffffffULL) return RG_OK;
        else lim = (u32)send->max_entries;
    } else if (send) {
        return RG_OK;
    }
    RgMbox *mb = h->mbox;
    if (h->mbox_running && !__atomic_load_n(&mb->alive, __ATOMIC_ACQUIRE) &&
        __atomic_load_n(&mb->seq_done, __ATOMIC_ACQUIRE) == h->mbox_seq) {
        // the instance has left (idle / lifetime): let the stream see it end before the next one goes on
        int rc = rg_mailbox_quiesce(h);
        if (rc) return rc;
    }
    if (send) {
        h->stage_max_entries = send->max_entries;
        h->stage_flags = send->flags;
    }
    const u32 s = ++h->mbox_seq;
    // (RgMbox: three self-validating words, one 8-byte store each; the records in pin_records are older stores)
    const u32 w0 = (u32)n | ((h->counters == h->counters_base ? 0u : 1u) << 16) | ((any_logterm ? 1u : 0u) << 17) |
                   ((send ? 1u : 0u) << 18) | ((send ? (send->flags & 3u) : 0u) << 19);
    __atomic_store_n(&mb->w[0], rg_mbox_word(s, w0), __ATOMIC_RELEASE);
    __atomic_store_n(&mb->w[1], rg_mbox_word(s, h->epoch), __ATOMIC_RELEASE);
    __atomic_store_n(&mb->w[2], rg_mbox_word(s, (u32)h->last_sparse_n), __ATOMIC_RELEASE);
    __atomic_store_n(&mb->w[3], rg_mbox_word(s, lim), __ATOMIC_RELEASE);
    if (!h->mbox_running) {
        int rc = rg_mailbox_launch(h);
        if (rc) return rc;
    }
    // spin on the answer; an instance that left without serving the request (it timed out as the request arrived) is
    // replaced -- the new one starts from seq_done and finds the request waiting
    u64 spins = 0;
    while (__atomic_load_n(&mb->seq_done, __ATOMIC_ACQUIRE) != s) {
        if ((++spins & 0xfffu) == 0) {
            if (hipStreamQuery(h->stream) != hipErrorNotReady) { // the kernel is gone (left, or failed)
                if (__atomic_load_n(&mb->seq_done, __ATOMIC_ACQUIRE) == s) break;
                h->mbox_running = false;
                hipError_t e = hipStreamSynchronize(h->stream);
                if (e != hipSuccess) return rg_fail(RG_ERR_NO_DEVICE, "mailbox: %s", hipGetErrorString(e));
                int rc = rg_mailbox_launch(h);
                if (rc) return rc;
            }
            if (spins > (1ull << 33)) return rg_fail(RG_ERR_NO_DEVICE, "mailbox: no answer from the device");
        }
        __builtin_ia32_pause();
    }
    *served = true;
    h->mbox_served++;
    return RG_OK;
}

extern "C" int rg_mailbox_start(rg_engine *h, uint32_t idle_timeout_us) try {
    if (!h) return rg_fail(RG_ERR_INVALID_ARG, "rg_mailbox_start: null engine");
    RG_ENTER(h);
    int rc = rg_ensure_sparse(h);
    if (rc) return rc;
    if (!h->mbox) {
        RG_HIP(hipHostMalloc(reinterpret_cast<void **>(&h->mbox), sizeof(RgMbox), hipHostMallocCoherent | hipHostMallocMapped));
        memset(h->mbox, 0, sizeof(RgMbox));
    }
    // the kernel's arguments are fixed at launch: the staging buffers it reads / writes have to exist at their final size
    if (!h->pin_records) {
        RG_HIP(hipHostMalloc(reinterpret_cast<void **>(&h->pin_records), 4096 * sizeof(rg_wire_msg), hipHostMallocDefault));
        h->pin_records_cap = 4096;
    }
    if (!h->pin_packed) {
        RG_HIP(hipMalloc(&h->d_packed, RG_PACKED_HDR + 4096 * sizeof(rg_res_rec)));
        RG_HIP(hipHostMalloc(reinterpret_cast<void **>(&h->pin_packed), RG_PACKED_HDR + 4096 * sizeof(rg_res_rec), hipHostMallocDefault));
        h->packed_cap = 4096;
    }
    if (h->ins_arena && !h->pin_send) // device Inflights: rg_flush_send's work items come back through this buffer
        RG_HIP(hipHostMalloc(reinterpret_cast<void **>(&h->pin_send), 16 + RG_SEND_SPEC * sizeof(rg_send_item), hipHostMallocDefault));
    h->mbox_idle_ticks = (u64)(idle_timeout_us ? idle_timeout_us : 2000u) * RG_MBOX_TICKS_PER_US;
    h->mbox_on = true;
    return RG_OK;
} RG_ABI_GUARD

extern "C" int rg_mailbox_stats(const rg_engine *h, uint64_t *flushes_served, uint64_t *launches) try {
    if (!h) return rg_fail(RG_ERR_INVALID_ARG, "rg_mailbox_stats: null engine");
    if (flushes_served) *flushes_served = h->mbox_served;
    if (launches) *launches = h->mbox_launches;
    return RG_OK;
} RG_ABI_GUARD

extern "C" int rg_mailbox_stop(rg_engine *h) try {
    if (!h) return rg_fail(RG_ERR_INVALID_ARG, "rg_mailbox_stop: null engine");
    RG_ENTER(h);
    h->mbox_on = false;
    return RG_OK;
} RG_ABI_GUARD

static int rg_flush_sparse(rg_engine *h, const rg_send_req *send) {
    u32 dup = 0;
    int rc = rg_sparse_roundtrip(h, nullptr, 0, h->q_any_logterm, &dup, send);
    if (rc) return rc;
    if (dup) return rg_fail(RG_ERR_STATE, "rg_flush: %u duplicate cells (internal error)", dup);
    return RG_OK;
}

extern "C" int rg_ingest_tick(rg_engine *h, const rg_wire_msg *records, uint64_t n, uint64_t *n_groups,
                              uint64_t *n_duplicates) try {
    if (!h || (!records && n)) return rg_fail(RG_ERR_INVALID_ARG, "rg_ingest_tick: bad argument");
    if (n_groups) *n_groups = 0;
    if (n_duplicates) *n_duplicates = 0;
    static const rg_wire_msg none = {};
    u32 dup = 0;
    int rc = rg_sparse_roundtrip(h, records ? records : &none, n, true, &dup);
    if (rc) return rc;
    if (n_groups) *n_groups = h->last_sparse_n;
    if (n_duplicates) *n_duplicates = dup;
    return RG_OK;
} RG_ABI_GUARD

// The device is the judge of an RG_MF_BECOME_LEADER event: it validates the new term against RG_COL_CUR_TERM (which the
// host may have reloaded or restored since rg_set_peers registered a term) and answers RG_OUT_BECAME_LEADER, or
// RG_OUT_FAULT with the event ignored. rg_local_become_leader had to move the host's term gate when the event was QUEUED
// (responses of the new term may follow in the same flush); where the device refused, the gate goes back to the old term --
// otherwise responses of the old term would be dropped and those of the new one applied to a Progress set that was never
// reset. RG_COL_CUR_TERM and the registered term of a group belong together: load one, register the other.
static int rg_settle_elections(rg_engine *h, bool results_available) {
    int rc = RG_OK;
    if (results_available && h->last_sparse_n) {
        const u64 n = h->last_sparse_n;
        std::vector<u64> groups(n);
        std::vector<u32> out(n);
        u64 got = 0;
        rc = rg_ingested_results(h, groups.data(), nullptr, out.data(), n, &got);
        if (rc == RG_OK) {
            std::unordered_map<u64, bool> became; // election groups of this flush -> did the device apply the event?
            for (const auto &e : h->q_elections) became[e.group] = false;
            for (u64 i = 0; i < n; i++) {
                auto it = became.find(groups[i]);
                if (it != became.end() && (out[i] & RG_OUT_BECAME_LEADER)) it->second = true;
            }
            // (newest first: should a group ever be listed twice, the OLDEST recorded term -- the registered one -- wins;
            // rg_local_become_leader refuses a second election of a group inside one flush, RG_ERR_SLOT_BUSY)
            for (auto e = h->q_elections.rbegin(); e != h->q_elections.rend(); ++e)
                if (!became[e->group]) h->terms[e->group] = e->old_term;
        }
    }
    if (!results_available || !h->last_sparse_n || rc != RG_OK) {
        // no verdict (the flush failed, or its results could not be read): the conservative side -- every gate goes back
        for (auto e = h->q_elections.rbegin(); e != h->q_elections.rend(); ++e) h->terms[e->group] = e->old_term;
    }
    h->q_elections.clear();
    return rc;
}

static int rg_flush_impl(rg_engine *h, const rg_send_req *send) {
    if (!h->host_mirror) return rg_fail(RG_ERR_STATE, "rg_flush: rg_set_peers was never called");
    int rc;
    const u64 launches0 = h->tick_launches;
    if (h->q_dirty.size() * 2 >= h->G) {
        // most groups have events: stream the whole columns through the dense tick (measured crossover with the
        // 64-B-record path is around 60 % of the groups: profiles/r01_sparse_path_and_recompute.txt, mirror_bench)
        rg_msgs m;
        m.m_index = h->q_mi.data();
        m.m_commit = h->q_mc.data();
        m.m_hint = h->q_mh.data();
        m.m_rs = h->q_mrs.data();
        m.m_logterm = h->q_any_logterm ? h->q_mlt.data() : nullptr;
        m.m_flags = h->q_mf.data();
        // (with a send request: the tick and its stage as ONE launch, k_tick_send)
        if (send) {
            const RgSendReq sr = {(u64)send->max_entries, (u32)send->flags};
            rc = rg_tick_host_impl(h, &m, &sr);
        } else {
            rc = rg_tick(h, &m);
        }
        // rg_ingested_results must work after ANY flush: gather the dirty groups' results compactly
        if (rc == RG_OK) rc = rg_ensure_sparse(h);
        if (rc == RG_OK) {
            const u32 n = (u32)h->q_dirty.size();
            hipError_t e = hipMemcpyAsync(h->list, h->q_dirty.data(), (size_t)n * 8, hipMemcpyHostToDevice, h->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(h->counters, &n, 4, hipMemcpyHostToDevice, h->stream);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(k_gather_results, dim3(rg_grid(n, 256)), dim3(256), 0, h->stream, h->list, h->counters,
                                   (const u64 *)h->st.commit, (const u32 *)h->st.out, h->res_list, h->res_commit, h->res_out,
                                   (char *)nullptr);
                e = hipMemsetAsync(h->counters, 0, 4, h->stream);
            }
            if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
            if (e != hipSuccess) rc = rg_fail(RG_ERR_NO_DEVICE, "rg_flush: %s", hipGetErrorString(e));
            else h->last_sparse_n = n;
        }
    } else {
        // few groups have events: ship only their records and tick only them -- ONE host<->device round trip
        // (pinned record staging, five back-to-back launches, one packed D2H copy, one synchronisation)
        rc = rg_flush_sparse(h, send);
    }
    // A flush that failed BEFORE its tick was enqueued changed nothing on the device: the queued events stay queued
    // and the call can be retried. Once the tick is enqueued the events are consumed (a retry would apply them
    // twice); an error after that point means results could not be fetched, not that the batch was lost.
    if (rc != RG_OK && h->tick_launches == launches0) return rc;
    for (u64 g : h->q_dirty) memset(&h->q_mf[g * 8], 0, 8);
    h->q_dirty.clear();
    h->q_any_logterm = false;
    if (!h->q_elections.empty()) {
        const int erc = rg_settle_elections(h, rc == RG_OK);
        if (rc == RG_OK) rc = erc;
    }
    return rc;
}

extern "C" int rg_flush(rg_engine *h) try {
    if (!h) return rg_fail(RG_ERR_INVALID_ARG, "rg_flush: null engine");
    return rg_flush_impl(h, nullptr);
} RG_ABI_GUARD

extern "C" int rg_flush_send(rg_engine *h, uint64_t max_entries_per_msg, uint32_t flags) try {
    if (!h) return rg_fail(RG_ERR_INVALID_ARG, "rg_flush_send: null engine");
    if (!h->ins_arena)
        return rg_fail(RG_ERR_STATE, "rg_flush_send: engine created with max_inflight = 0 (Inflights are the host's)");
    if (flags & ~(RG_SEND_SKIP_BCAST_COMMIT | RG_SEND_BYTES)) return rg_fail(RG_ERR_INVALID_ARG, "rg_flush_send: unknown flags %#x", flags);
    if ((flags & RG_SEND_BYTES) && !h->esz)
        return rg_fail(RG_ERR_STATE, "rg_flush_send: RG_SEND_BYTES needs the entry sizes (rg_log_sizes_enable)");
    rg_send_req req;
    req.max_entries = max_entries_per_msg;
    req.flags = flags;
    return rg_flush_impl(h, &req);
} RG_ABI_GUARD


