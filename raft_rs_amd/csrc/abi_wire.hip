// abi_wire.hip -- eraftpb::Message on and off the wire (pure host code) and the synthetic AppendResponse stream
// There is NO CPU fallback anywhere in this file: without a HIP device every entry point fails.
#include "rg_engine.h"
#include "rg_kernels_workload.h"

// ---- eraftpb::Message off the wire: the decoder is rg_wire.h (host code, sanitiser-tested on its own) ----
extern "C" int rg_decode_message(const uint8_t *bytes, uint64_t len, rg_decoded_message *out) try {
    if ((!bytes && len) || !out) return rg_fail(RG_ERR_INVALID_ARG, "rg_decode_message: bad argument");
    rg_wire_u64 bad = 0;
    if (!rg_wire_decode(bytes, len, out, &bad))
        return rg_fail(RG_ERR_INVALID_ARG, "rg_decode_message: not a protobuf-encoded eraftpb::Message (malformed at byte %llu)",
                       (unsigned long long)bad);
    return RG_OK;
} RG_ABI_GUARD

// ---- ... and onto the wire: the encoder is rg_wire.h as well (host code; nothing here touches an engine) ----
extern "C" uint64_t rg_entry_size(const rg_entry *e) { return e ? rg_wire_entry_size(e) : 0; }

extern "C" uint64_t rg_limit_size(const rg_entry *entries, uint64_t n, uint64_t max_size) {
    return entries ? rg_wire_limit_size(entries, n, max_size) : 0;
}

extern "C" int rg_message_size(const rg_message *m, uint64_t *len) try {
    if (!m || !len) return rg_fail(RG_ERR_INVALID_ARG, "rg_message_size: bad argument");
    if (!rg_wire_message_size(m, len))
        return rg_fail(RG_ERR_INVALID_ARG, "rg_message_size: a length without its pointer, or more than 2 GiB - 1 bytes");
    return RG_OK;
} RG_ABI_GUARD

extern "C" int rg_encode_message(const rg_message *m, uint8_t *buf, uint64_t cap, uint64_t *len) try {
    if (!m || !len || (!buf && cap)) return rg_fail(RG_ERR_INVALID_ARG, "rg_encode_message: bad argument");
    if (!rg_wire_message_size(m, len))
        return rg_fail(RG_ERR_INVALID_ARG, "rg_encode_message: a length without its pointer, or more than 2 GiB - 1 bytes");
    if (cap < *len)
        return rg_fail(RG_ERR_INVALID_ARG, "rg_encode_message: %llu bytes needed, the buffer holds %llu",
                       (unsigned long long)*len, (unsigned long long)cap);
    rg_wire_encode(m, buf);
    return RG_OK;
} RG_ABI_GUARD


// ------------------------------------------------------------------------------------------------
// synthetic stream
// ------------------------------------------------------------------------------------------------
extern "C" int rg_workload_init(rg_engine *h, const rg_workload *w, uint64_t first) try {
    if (!h || !w) return rg_fail(RG_ERR_INVALID_ARG, "rg_workload_init: bad argument");
    if (w->workload != RG_WL_MAJORITY && w->workload != RG_WL_JOINT && w->workload != RG_WL_MIXED)
        return rg_fail(RG_ERR_INVALID_ARG, "rg_workload_init: unknown workload %u", w->workload);
    if ((w->reserved & 0xfu) > 8 || (w->reserved & ~0x3fu))
        return rg_fail(RG_ERR_INVALID_ARG, "rg_workload_init: fixed replica-set size %u / flags %#x", w->reserved & 0xfu, w->reserved & ~0xfu);
    RG_ENTER(h);
    hipLaunchKernelGGL(k_wl_init, dim3(rg_grid(h->G, RG_BLOCK)), dim3(RG_BLOCK), 0, h->stream, h->st, (u64)w->seed,
                       w->workload | (w->reserved << 8), h->P, (u64)first);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return rg_fail(RG_ERR_NO_DEVICE, "rg_workload_init: %s", hipGetErrorString(e));
    if (h->pub) h->pub->local_lost = true;
    h->host_cfg_valid = false;
    h->cls_stale = true;
    h->any_group_commit = (w->reserved & RG_WL_GROUP_COMMIT) != 0; // (every cfg word of the shard was just written)
    return RG_OK;
} RG_ABI_GUARD

extern "C" int rg_workload_gen(rg_engine *h, const rg_workload *w, uint64_t first, uint64_t tick, uint64_t *mi,
                               uint64_t *mc, uint64_t *mh, uint64_t *mrs, uint8_t *mf) try {
    if (!h || !w || !mi || !mc || !mh || !mrs || !mf) return rg_fail(RG_ERR_INVALID_ARG, "rg_workload_gen: bad argument");
    RG_ENTER(h);
    hipLaunchKernelGGL(k_wl_gen, dim3(rg_grid(h->G, RG_BLOCK)), dim3(RG_BLOCK), 0, h->stream, h->st, (u64)w->seed,
                       w->workload | (w->reserved << 8), h->P, (u64)first, (u64)tick, (u64 *)mi, (u64 *)mc, (u64 *)mh, (u64 *)mrs, mf);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return rg_fail(RG_ERR_NO_DEVICE, "rg_workload_gen: %s", hipGetErrorString(e));
    return RG_OK;
} RG_ABI_GUARD

extern "C" int rg_workload_init_host(const rg_workload *w, uint64_t first, rg_host_state *s) try {
    if (!w || !s || s->n_slots == 0 || s->n_slots > 8) return rg_fail(RG_ERR_INVALID_ARG, "rg_workload_init_host: bad argument");
    for (u64 g = 0; g < s->n_groups; g++)
        rg_wl_init_group(w->seed, w->workload | (w->reserved << 8), s->n_slots, s->stride, g,
                         first + rg_wl_place(w->workload | (w->reserved << 8), g, s->n_groups), (u64 *)s->match, (u64 *)s->next,
                         (u64 *)s->pr_commit, (u64 *)s->pend_snap, (u64 *)s->pend_rs, (u64 *)s->gid, s->pflags,
                         (u64 *)s->commit, (u64 *)s->term_lo, (u64 *)s->term_hi, s->cfg);
    return RG_OK;
} RG_ABI_GUARD

extern "C" int rg_workload_gen_host(const rg_workload *w, uint64_t first, uint64_t tick, const rg_host_state *s,
                                    uint64_t *mi, uint64_t *mc, uint64_t *mh, uint64_t *mrs, uint8_t *mf) try {
    if (!w || !s || !mi || !mc || !mh || !mrs || !mf) return rg_fail(RG_ERR_INVALID_ARG, "rg_workload_gen_host: bad argument");
    for (u64 g = 0; g < s->n_groups; g++)
        rg_wl_gen_group(w->seed, w->workload | (w->reserved << 8), s->n_slots, s->stride, g,
                        first + rg_wl_place(w->workload | (w->reserved << 8), g, s->n_groups), tick, (const u64 *)s->match,
                        (const u64 *)s->next, s->pflags, (const u64 *)s->commit, (const u64 *)s->term_lo,
                        (const u64 *)s->term_hi, (u64 *)mi,
                        (u64 *)mc, (u64 *)mh, (u64 *)mrs, mf);
    return RG_OK;
} RG_ABI_GUARD


