// abi_state.hip -- lifecycle, error plumbing, state in / out (include/raftgroups.h: "lifecycle", "state in/out")
// There is NO CPU fallback anywhere in this file: without a HIP device every entry point fails.
#include "rg_engine.h"
#include "rg_kernels_state.h"

static std::mutex g_live_mu;
static int g_live_on_device[RG_MAX_DEVICES];

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local char g_last_error[512]; // (a fixed buffer: reporting an allocation failure must not allocate)

int rg_fail(int code, const char *fmt, ...) {
    // HIP keeps the last error until somebody reads it: a failed hipMalloc must not resurface later as the
    // "launch error" of an unrelated kernel (every launch site checks hipGetLastError)
    (void)hipGetLastError();
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char *rg_version(void) { return "raftgroups 0.1 (gfx950, opt " RG_STR(RG_OPT) ")"; }
extern "C" uint32_t rg_abi_version(void) { return RG_ABI_VERSION; }
extern "C" const char *rg_last_error(void) { return g_last_error; }

extern "C" int rg_device_count(void) try {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
} RG_ABI_GUARD

extern "C" uint64_t rg_column_bytes(const rg_engine *h, int c) {
    if (!h || c < 0 || c >= RG_COL_COUNT) return 0;
    if (rg_col_per_slot(c)) return (uint64_t)h->P * h->stride * 8;
    if (rg_col_per_run(c)) return (uint64_t)RG_TERM_RUNS * h->stride * 8;
    return (uint64_t)h->G * rg_col_elem(c);
}



// (rg_create's failure paths and rg_destroy)
void rg_drop(rg_engine *h) {
    if (h->counted_live) {
        std::lock_guard<std::mutex> lk(g_live_mu);
        g_live_on_device[h->cfg.device]--;
    }
    delete h;
}

// The size of the device's Infinity Cache (the memory-side "L3" / MALL: 256 MiB on MI355X), ASKED of the device instead of
// assumed: HIP's device properties stop at the L2, the HSA runtime underneath enumerates every cache of an agent
// (hsa_agent_iterate_caches: level 3 is the one). The runtime is in the process already (HIP sits on it); it is bound at run
// time like RCCL, and the agent is matched to the HIP device by its PCI address. 0 = could not be asked (the caller falls back to
// the MI355X constant and says so in rg_device_info.infinity_cache_queried).
#include <dlfcn.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
namespace {
struct RgHsaQuery {
    decltype(&hsa_agent_get_info) agent_get_info;
    decltype(&hsa_agent_iterate_caches) iterate_caches;
    decltype(&hsa_cache_get_info) cache_get_info;
    uint32_t want_bdf, want_domain;
    uint64_t l3_bytes;
};
hsa_status_t rg_hsa_cache_cb(hsa_cache_t cache, void *data) {
    RgHsaQuery *q = static_cast<RgHsaQuery *>(data);
    uint8_t level = 0;
    uint32_t size = 0;
    if (q->cache_get_info(cache, HSA_CACHE_INFO_LEVEL, &level) == HSA_STATUS_SUCCESS &&
        q->cache_get_info(cache, HSA_CACHE_INFO_SIZE, &size) == HSA_STATUS_SUCCESS && level == 3) {
        // (ROCr hands on what the kernel driver's topology reports, and that is KiB -- 262 144 for MI355X's 256 MiB, as rocminfo
        //  prints it -- although the HSA specification says bytes: anything below 1 Mi is taken as KiB)
        const uint64_t bytes = size < (1u << 20) ? (uint64_t)size << 10 : (uint64_t)size;
        if (bytes > q->l3_bytes) q->l3_bytes = bytes;
    }
    return HSA_STATUS_SUCCESS;
}
hsa_status_t rg_hsa_agent_cb(hsa_agent_t agent, void *data) {
    RgHsaQuery *q = static_cast<RgHsaQuery *>(data);
    hsa_device_type_t type;
    if (q->agent_get_info(agent, HSA_AGENT_INFO_DEVICE, &type) != HSA_STATUS_SUCCESS || type != HSA_DEVICE_TYPE_GPU) return HSA_STATUS_SUCCESS;
    uint32_t bdf = 0, domain = 0;
    if (q->agent_get_info(agent, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
    (void)q->agent_get_info(agent, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &domain);
    if ((bdf & 0xffffu) != q->want_bdf || domain != q->want_domain) return HSA_STATUS_SUCCESS;
    (void)q->iterate_caches(agent, rg_hsa_cache_cb, q);
    return HSA_STATUS_INFO_BREAK;
}
} // namespace
static uint64_t rg_query_infinity_cache(const hipDeviceProp_t &prop) {
    void *lib = dlopen("libhsa-runtime64.so.1", RTLD_NOW | RTLD_NOLOAD); // (the instance HIP runs on)
    if (!lib) lib = dlopen("libhsa-runtime64.so.1", RTLD_NOW);
    if (!lib) return 0;
    auto init = reinterpret_cast<decltype(&hsa_init)>(dlsym(lib, "hsa_init"));
    auto shut = reinterpret_cast<decltype(&hsa_shut_down)>(dlsym(lib, "hsa_shut_down"));
    auto iter = reinterpret_cast<decltype(&hsa_iterate_agents)>(dlsym(lib, "hsa_iterate_agents"));
    RgHsaQuery q;
    q.agent_get_info = reinterpret_cast<decltype(&hsa_agent_get_info)>(dlsym(lib, "hsa_agent_get_info"));
    q.iterate_caches = reinterpret_cast<decltype(&hsa_agent_iterate_caches)>(dlsym(lib, "hsa_agent_iterate_caches"));
    q.cache_get_info = reinterpret_cast<decltype(&hsa_cache_get_info)>(dlsym(lib, "hsa_cache_get_info"));
    q.want_bdf = ((uint32_t)prop.pciBusID << 8) | ((uint32_t)prop.pciDeviceID << 3);
    q.want_domain = (uint32_t)prop.pciDomainID;
    q.l3_bytes = 0;
    if (init && shut && iter && q.agent_get_info && q.iterate_caches && q.cache_get_info && init() == HSA_STATUS_SUCCESS) {
        (void)iter(rg_hsa_agent_cb, &q);
        (void)shut(); // (reference-counted: HIP's own use of the runtime is untouched)
    }
    return q.l3_bytes;
}

extern "C" int rg_create(const rg_config *cfg, rg_engine **out) try {
    if (!cfg || !out) return rg_fail(RG_ERR_INVALID_ARG, "rg_create: null argument");
    if (cfg->n_groups == 0 || cfg->n_slots == 0 || cfg->n_slots > RG_MAX_SLOTS)
        return rg_fail(RG_ERR_INVALID_ARG, "rg_create: n_groups=%llu n_slots=%u out of range",
                       (unsigned long long)cfg->n_groups, cfg->n_slots);
    if (cfg->variant > RG_VARIANT_COMPACT) return rg_fail(RG_ERR_INVALID_ARG, "rg_create: unknown variant %u", cfg->variant);
    if (cfg->cache_policy > RG_CACHE_RESIDENT) return rg_fail(RG_ERR_INVALID_ARG, "rg_create: unknown cache_policy %u", cfg->cache_policy);
    if (cfg->flags & ~(RG_CFGF_NO_SIZE_CLASSES | RG_CFGF_CLASS_BLOCK_ORDER | RG_CFGF_IX64))
        return rg_fail(RG_ERR_INVALID_ARG, "rg_create: unknown flags %#x", cfg->flags);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return rg_fail(RG_ERR_NO_DEVICE, "rg_create: no HIP device visible (this engine has no CPU fallback)");
    if (cfg->device < 0 || cfg->device >= ndev || cfg->device >= RG_MAX_DEVICES)
        return rg_fail(RG_ERR_INVALID_ARG, "rg_create: device %d of %d", cfg->device, ndev);
    RG_HIP(hipSetDevice(cfg->device));
    hipDeviceProp_t prop;
    RG_HIP(hipGetDeviceProperties(&prop, cfg->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return rg_fail(RG_ERR_NO_DEVICE, "rg_create: device %d is %s; this library carries gfx950 (CDNA4) kernels only",
                       cfg->device, prop.gcnArchName);
    if (prop.warpSize != 64)
        return rg_fail(RG_ERR_NO_DEVICE, "rg_create: wavefront size %d, the kernels are written for 64", prop.warpSize);
    rg_engine *h = new (std::nothrow) rg_engine();
    if (!h) return rg_fail(RG_ERR_OUT_OF_MEMORY, "rg_create: host allocation failed");
    memset(&h->dev, 0, sizeof(h->dev));
    for (int i = 0; i < 31 && prop.gcnArchName[i] && prop.gcnArchName[i] != ':'; i++) h->dev.arch[i] = prop.gcnArchName[i];
    h->dev.compute_units = (uint32_t)prop.multiProcessorCount;
    h->dev.wavefront = (uint32_t)prop.warpSize;
    h->dev.lds_per_workgroup = prop.sharedMemPerBlock;
    h->dev.hbm_bytes = prop.totalGlobalMem;
    h->dev.l2_bytes = (uint64_t)prop.l2CacheSize;
    h->dev.infinity_cache_bytes = rg_query_infinity_cache(prop);
    h->dev.infinity_cache_queried = h->dev.infinity_cache_bytes ? 1u : 0u;
    if (!h->dev.infinity_cache_bytes) h->dev.infinity_cache_bytes = 256ull << 20; // (MI355X; rg_device_info says that it was not asked)
    h->cfg = *cfg;
    h->G = cfg->n_groups;
    h->P = cfg->n_slots;
    h->stride = (h->G + 255) / 256 * 256;
    h->stream = nullptr;
    h->ckpt = nullptr;
    h->msg_arena = nullptr;
    h->ticked = false;
    h->pub_tick_evt = -1;
    h->tick_launches = 0;
    h->sparse_arena = nullptr;
    h->d_records = nullptr;
    h->d_records_cap = 0;
    h->d_cells = nullptr;
    h->d_cells_cap = 0;
    h->pin_records = nullptr;
    h->pin_records_cap = 0;
    h->d_packed = nullptr;
    h->pin_packed = nullptr;
    h->packed_cap = 0;
    h->host_res_valid = false;
    h->epoch = 1;
    h->ingested_upper = 0;
    h->last_sparse_n = 0;
    h->out_is_dense = true;
    h->any_group_commit = false;
    h->host_mirror = false;
    h->host_cfg_valid = false;
    h->q_any_logterm = false;
    h->ins_arena = nullptr;
    h->ins_ckpt = nullptr;
    h->esz = h->esz_ckpt = nullptr;
    h->d_recs = nullptr;
    h->d_recs_cap = 0;
    h->mbox = nullptr;
    h->mbox_on = h->mbox_running = false;
    h->mbox_seq = 0;
    h->mbox_idle_ticks = 0;
    h->mbox_served = h->mbox_launches = 0;
    h->ins.esz = nullptr;
    h->ins.esz_w = 0;
    h->ins_state_bytes = 0;
    h->ins.meta = nullptr;
    h->ins.head = nullptr;
    h->ins.tail = nullptr;
    h->ins.ring = nullptr;
    h->ins.cap = 0;
    h->send_items = nullptr;
    h->send_counter = nullptr;
    h->send_cols.prev = h->send_cols.last = nullptr;
    h->send_cols.n = nullptr;
    h->send_cols_fresh = false;
    h->send_last_dense = false;
    h->send_ready = false;
    h->hint_check_due = false;
    h->d_hint_raised = h->pin_hint_raised = nullptr;
    h->ev_hint = nullptr;
    h->hint_probe_pending = false;
    h->fused_done = 0;
    h->cls_need = nullptr;
    h->cls_on = false;
    h->cls_order = nullptr;
    h->cls_stale = true;
    h->cls_off = (cfg->flags & RG_CFGF_NO_SIZE_CLASSES) != 0;
    h->cls_block_order = (cfg->flags & RG_CFGF_CLASS_BLOCK_ORDER) != 0;
    h->stage_max_entries = 0;
    h->stage_flags = 0;
    // ---- cache policy (rg_config.cache_policy; include/raftgroups.h: RG_CACHE_*), decided here and nowhere else ----
    // Infinity Cache (256 MB on MI355X). AUTO, by footprint:
    //  * STREAM_MSGS when the state a dense tick re-reads (24 P + 40 B per group) and the message columns of ONE tick
    //    (16 P + 8 B per group, read once) do not fit together (with the Inflights on the device a step also touches the
    //    window and work-item columns: 40 P B per group more);
    //  * STREAM_ALL when the state alone is more than 1.5 x the cache -- by the time a launch comes back to a line the cache
    //    has turned over, so allocating there only costs -- and the shard holds at most 13 M groups. Both ends are measured, at
    //    3, 5 and 7 slots (profiles/r04_nt_state.txt; profiles/r05_cache_policy_sweep.txt): the lower one follows the BYTES of
    //    state (1.3 x: streamed loses 1-5 %; 1.5 x: wins 6-9 % at every slot count; 8 M x 5: 528 -> 483 us, fraction 0.68 ->
    //    0.75), the upper one the NUMBER of groups -- at 12 M groups everything streamed wins at 3, 5 and 7 slots alike (529 /
    //    786 / 1046 us against 559 / 812 / 1068), at 14 M it loses or ties (705 / 1006 / 1275 against 664 / 975 / 1270), at 16 M
    //    it loses 6-8 % -- although the state of those engines spans 1.3 to 3.2 GB (round 4 had put that end at 7.5 x the cache
    //    in bytes, from 5 slots alone: right there, 12 % wrong at 3 slots);
    //  * RESIDENT (k_tick_split): a leading range of the groups keeps its state in the cache, the rest is streamed. Measured
    //    (profiles/r04_resident.txt, r05_cache_policy_sweep.txt): with 176 MB of state resident 2.4 M x 5 runs in 134 us instead
    //    of 150 (all streamed; 155 plain), 4 M x 5 in 227-233 instead of 246; at 3 / 7 slots it is the fastest policy from 1.3 x
    //    (114 / 126 us against 125 / 128) through 2.5 x the cache (237 / 248 against 243 / 259) and loses beyond (3.5 x at 3
    //    slots: 393 against 339; 8 M x 5: 491 -> 556) -- over a launch that long the resident lines are gone before the next one
    //    comes back to them; below 1.25 x (1.1 x: 95 / 109 against 94 / 102) the plain accesses with streamed messages win.
    //    So: 1.25 x cache < state <= 2.5 x cache. The cache is ONE per device: three size-class engines of config 5 at
    //    8 M groups, two of them with a resident range, took 893 us instead of 724 -- so AUTO grants the range only to an
    //    engine that is ALONE on its device when it is created (engines_on_device == 1 in rg_device_info); a later engine on the
    //    same device gets the streaming policy of its size and the first one keeps what it was given. A host that knows better
    //    says so: an explicit policy is honoured as given.
    {
        std::lock_guard<std::mutex> lk(g_live_mu);
        h->dev.engines_on_device = (uint32_t)++g_live_on_device[cfg->device];
        h->counted_live = true;
    }
    {
        const double mall = (double)h->dev.infinity_cache_bytes; // (asked of the device: rg_query_infinity_cache)
        const double per_group = (double)(24u * h->P + 40u), state = (double)h->G * per_group;
        const double with_msgs = (double)h->G * (double)(40u * h->P + 48u + (cfg->max_inflight ? 40u * h->P : 0u));
        const bool lane = cfg->variant == RG_VARIANT_DEFAULT || cfg->variant == RG_VARIANT_LANE || cfg->variant == RG_VARIANT_COOP;
        u32 pol = cfg->cache_policy;
        if (pol == RG_CACHE_AUTO) {
            pol = with_msgs > mall ? RG_CACHE_STREAM_MSGS : RG_CACHE_PLAIN;
            if (!cfg->max_inflight && state > 1.5 * mall && h->G <= 13000000ull) pol = RG_CACHE_STREAM_ALL;
            if (!cfg->max_inflight && lane && state > 1.25 * mall && state <= 2.5 * mall && h->dev.engines_on_device == 1)
                pol = RG_CACHE_RESIDENT;
        }
        if (cfg->max_inflight && pol == RG_CACHE_RESIDENT) { // (k_tick_split has no send stage)
            rg_drop(h);
            return rg_fail(RG_ERR_INVALID_ARG, "rg_create: RG_CACHE_RESIDENT needs max_inflight = 0 (engines with device Inflights: "
                                               "RG_CACHE_PLAIN, RG_CACHE_STREAM_MSGS or RG_CACHE_STREAM_ALL)");
        }
        h->nt_msgs = pol >= RG_CACHE_STREAM_MSGS;
        h->nt_all = pol >= RG_CACHE_STREAM_ALL;
        h->nt_resident = 0;
        if (pol == RG_CACHE_RESIDENT) {
            const u64 groups = cfg->cache_resident_groups ? cfg->cache_resident_groups : (u64)(0.6875 * mall / per_group) /* 176 of 256 MiB */;
            h->nt_resident = rg_min(groups, h->G) / RG_BLOCK;
            if (h->nt_resident == 0) pol = RG_CACHE_STREAM_ALL; // (less than one workgroup: nothing to keep)
        }
        h->dev.cache_policy = pol;
        h->dev.resident_groups = h->nt_resident * RG_BLOCK;
    }
    h->send_bound = 0;
    h->pin_send = nullptr;
    h->host_items_valid = false;
    h->ckpt_send_ready = false;
    h->ckpt_hint_check_due = false;
    h->ckpt_any_group_commit = false;
    if (cfg->max_inflight > 65535u) {
        rg_drop(h);
        return rg_fail(RG_ERR_INVALID_ARG, "rg_create: max_inflight=%u, at most 65535", cfg->max_inflight);
    }
    size_t off = 0;
    for (int c = 0; c < RG_COL_COUNT; c++) {
        h->col_off[c] = off;
        const size_t bytes = rg_col_per_slot(c)  ? (size_t)h->P * h->stride * 8
                             : rg_col_per_run(c) ? (size_t)RG_TERM_RUNS * h->stride * 8
                                                 : (size_t)h->stride * rg_col_elem(c);
        off += rg_align(bytes);
    }
    h->state_bytes = off;
    const size_t zero_bytes = rg_align((size_t)h->P * h->stride * 8);
    hipError_t e = hipMalloc(&h->arena, off + 2 * zero_bytes + 256 + rg_align(h->stride * 8));
    if (e != hipSuccess) {
        rg_drop(h);
        return rg_fail(RG_ERR_OUT_OF_MEMORY, "rg_create: hipMalloc(%zu) failed: %s", off, hipGetErrorString(e));
    }
    e = hipMemset(h->arena, 0, off + 2 * zero_bytes + 256 + rg_align(h->stride * 8));
    // hipMemset of device memory returns before the fill has run (it is queued on the NULL stream), and a caller's
    // stream created non-blocking (every torch.cuda.Stream) is not ordered behind the NULL stream: without this wait a
    // first kernel on such a stream races the fill (seen at 8 M groups: 0.1 % of the groups zeroed again after
    // rg_workload_init had written them)
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) {
        (void)hipFree(h->arena);
        rg_drop(h);
        return rg_fail(RG_ERR_NO_DEVICE, "rg_create: hipMemset failed: %s", hipGetErrorString(e));
    }
    h->dev.engine_bytes = off + 2 * zero_bytes + 256 + rg_align(h->stride * 8);
    h->zero_col = reinterpret_cast<u64 *>(h->arena + off);
    h->d_counts = reinterpret_cast<u64 *>(h->arena + off + zero_bytes);
    h->d_hint_raised = reinterpret_cast<u32 *>(h->arena + off + zero_bytes + 128); // (the reductions use the first 40 of these 256 bytes)
    h->d_scratch = h->arena + off + zero_bytes + 256;
    h->rhint = reinterpret_cast<u64 *>(h->arena + off + zero_bytes + 256 + rg_align(h->stride * 8));
    RgState &s = h->st;
    s.match = (u64 *)rg_col(h, RG_COL_MATCH);
    s.next = (u64 *)rg_col(h, RG_COL_NEXT);
    s.prc = (u64 *)rg_col(h, RG_COL_PR_COMMIT);
    s.psnap = (u64 *)rg_col(h, RG_COL_PEND_SNAP);
    s.prs = (u64 *)rg_col(h, RG_COL_PEND_RS);
    s.gid = (u64 *)rg_col(h, RG_COL_GID);
    s.pflags = (u64 *)rg_col(h, RG_COL_PFLAGS);
    s.commit = (u64 *)rg_col(h, RG_COL_COMMIT);
    s.lo = (u64 *)rg_col(h, RG_COL_TERM_LO);
    s.hi = (u64 *)rg_col(h, RG_COL_TERM_HI);
    s.cfg = (u32 *)rg_col(h, RG_COL_CFG);
    s.out = (u32 *)rg_col(h, RG_COL_OUT);
    s.run_first = (u64 *)rg_col(h, RG_COL_RUN_FIRST);
    s.run_term = (u64 *)rg_col(h, RG_COL_RUN_TERM);
    s.dummy_idx = (u64 *)rg_col(h, RG_COL_DUMMY_INDEX);
    s.dummy_term = (u64 *)rg_col(h, RG_COL_DUMMY_TERM);
    s.cur_term = (u64 *)rg_col(h, RG_COL_CUR_TERM);
    s.hhint = (u8 *)rg_col(h, RG_COL_HOST_HINT);
    s.G = h->G;
    s.stride = h->stride;
    // (consecutive byte columns, strides of 256: RG_COL_RUN_COUNT sits `stride` bytes behind RG_COL_HOST_HINT by construction)
    if ((u8 *)rg_col(h, RG_COL_RUN_COUNT) != rg_run_n(s)) {
        (void)hipFree(h->arena);
        rg_drop(h);
        return rg_fail(RG_ERR_INVALID_ARG, "rg_create: column layout");
    }
    s.ix64 = (cfg->flags & RG_CFGF_IX64) ? 1u : 0u; // (rg_common.h: rg_ix32)
    s.pub = nullptr;
    s.pub_off_delta = 0;
    s.pub_cap = 0;
    h->pub = nullptr;
    if (cfg->max_inflight) { // Inflights rings + the send stage's work-item list
        const size_t meta_b = rg_align((size_t)h->P * h->stride * 4) + 2 * rg_align((size_t)h->P * h->stride * 8); // meta | head | tail
        const size_t ring_b = rg_align((size_t)h->G * h->P * cfg->max_inflight * 8);
        const size_t items_b = rg_align((size_t)h->G * h->P * sizeof(rg_send_item));
        const size_t col8_b = rg_align((size_t)h->P * h->stride * 8), col4_b = rg_align((size_t)h->P * h->stride * 4);
        const size_t cols_b = 2 * col8_b + col4_b; // RgSendCols: prev | last | n
        e = hipMalloc(&h->ins_arena, meta_b + ring_b + items_b + 256 + cols_b);
        if (e == hipSuccess) e = hipMemset(h->ins_arena, 0, meta_b + ring_b + items_b + 256 + cols_b);
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr); // (as above: the fill must have run before rg_create returns)
        if (e != hipSuccess) {
            if (h->ins_arena) (void)hipFree(h->ins_arena);
            (void)hipFree(h->arena);
            rg_drop(h);
            return rg_fail(RG_ERR_OUT_OF_MEMORY, "rg_create: %zu bytes of Inflights (cap %u): %s", meta_b + ring_b + items_b + cols_b,
                           cfg->max_inflight, hipGetErrorString(e));
        }
        h->ins.meta = reinterpret_cast<u32 *>(h->ins_arena);
        h->ins.head = reinterpret_cast<u64 *>(h->ins_arena + rg_align((size_t)h->P * h->stride * 4));
        h->ins.tail = reinterpret_cast<u64 *>(h->ins_arena + rg_align((size_t)h->P * h->stride * 4) +
                                              rg_align((size_t)h->P * h->stride * 8));
        h->ins.ring = reinterpret_cast<u64 *>(h->ins_arena + meta_b);
        h->ins.cap = cfg->max_inflight;
        h->ins_state_bytes = meta_b + ring_b;
        h->send_items = reinterpret_cast<rg_send_item *>(h->ins_arena + meta_b + ring_b);
        h->send_counter = reinterpret_cast<u32 *>(h->ins_arena + meta_b + ring_b + items_b);
        char *cols = h->ins_arena + meta_b + ring_b + items_b + 256;
        h->send_cols.prev = reinterpret_cast<u64 *>(cols);
        h->send_cols.last = reinterpret_cast<u64 *>(cols + col8_b);
        h->send_cols.n = reinterpret_cast<u32 *>(cols + 2 * col8_b);
        h->dev.engine_bytes += meta_b + ring_b + items_b + 256 + cols_b;
    }
    *out = h;
    return RG_OK;
} RG_ABI_GUARD

extern "C" int rg_comm_destroy(rg_engine *h);

extern "C" void rg_destroy(rg_engine *h) {
    if (!h) return;
    (void)hipSetDevice(h->cfg.device);
    (void)rg_mailbox_quiesce(h);
    (void)hipStreamSynchronize(h->stream);
    if (h->mbox) (void)hipHostFree(h->mbox);
    if (h->pub) (void)rg_comm_destroy(h);
    if (h->arena) (void)hipFree(h->arena);
    if (h->ckpt) (void)hipFree(h->ckpt);
    if (h->cls_need) (void)hipFree(h->cls_need);
    if (h->cls_order) (void)hipFree(h->cls_order);
    if (h->ins_arena) (void)hipFree(h->ins_arena);
    if (h->ins_ckpt) (void)hipFree(h->ins_ckpt);
    if (h->esz) (void)hipFree(h->esz);
    if (h->esz_ckpt) (void)hipFree(h->esz_ckpt);
    if (h->d_recs) (void)hipFree(h->d_recs);
    if (h->msg_arena) (void)hipFree(h->msg_arena);
    if (h->sparse_arena) (void)hipFree(h->sparse_arena);
    if (h->d_records) (void)hipFree(h->d_records);
    if (h->d_cells) (void)hipFree(h->d_cells);
    if (h->pin_records) (void)hipHostFree(h->pin_records);
    if (h->pin_send) (void)hipHostFree(h->pin_send);
    if (h->d_packed) (void)hipFree(h->d_packed);
    if (h->pin_packed) (void)hipHostFree(h->pin_packed);
    if (h->pin_hint_raised) (void)hipHostFree(h->pin_hint_raised);
    if (h->ev_hint) (void)hipEventDestroy(h->ev_hint);
    rg_drop(h);
}

extern "C" uint64_t rg_stride(const rg_engine *h) { return h ? h->stride : 0; }

extern "C" int rg_get_device_info(const rg_engine *h, rg_device_info *info) try {
    if (!h || !info) return rg_fail(RG_ERR_INVALID_ARG, "rg_get_device_info: bad argument");
    *info = h->dev;
    return RG_OK;
} RG_ABI_GUARD

extern "C" int rg_set_stream(rg_engine *h, void *hip_stream) try {
    if (!h) return rg_fail(RG_ERR_INVALID_ARG, "rg_set_stream: null engine");
    RG_ENTER(h); // (a resident mailbox workgroup sits on the OLD stream: it has to leave before the engine moves)
    h->stream = reinterpret_cast<hipStream_t>(hip_stream);
    return RG_OK;
} RG_ABI_GUARD

extern "C" int rg_sync(rg_engine *h) try {
    if (!h) return rg_fail(RG_ERR_INVALID_ARG, "rg_sync: null engine");
    RG_ENTER(h);
    RG_HIP(hipStreamSynchronize(h->stream));
    return RG_OK;
} RG_ABI_GUARD

extern "C" int rg_load_column(rg_engine *h, int c, const void *src, uint64_t bytes) try {
    if (!h || !src || c < 0 || c >= RG_COL_COUNT) return rg_fail(RG_ERR_INVALID_ARG, "rg_load_column: bad argument");
    if (c == RG_COL_HOST_HINT) return rg_fail(RG_ERR_INVALID_ARG, "rg_load_column: RG_COL_HOST_HINT is written by the ticks only");
    if (c == RG_COL_RUN_COUNT) return rg_fail(RG_ERR_INVALID_ARG, "rg_load_column: RG_COL_RUN_COUNT is derived from RG_COL_RUN_FIRST by the engine");
    if (bytes != rg_column_bytes(h, c))
        return rg_fail(RG_ERR_INVALID_ARG, "rg_load_column(%d): %llu bytes given, %llu expected", c,
                       (unsigned long long)bytes, (unsigned long long)rg_column_bytes(h, c));
    if (c == RG_COL_CFG) { // a configuration word may only name slots the engine has
        const u32 *w = static_cast<const u32 *>(src);
        for (u64 g = 0; g < h->G; g++) {
            const u32 x = w[g];
            if (RG_CFG_SELF(x) >= h->P || (RG_CFG_PRESENT(x) >> h->P) || (RG_CFG_INCOMING(x) >> h->P) ||
                (RG_CFG_OUTGOING(x) >> h->P) || RG_CFG_TRANSFEREE(x) > h->P)
                return rg_fail(RG_ERR_INVALID_ARG, "rg_load_column(CFG): group %llu: word %#x names a slot >= %u",
                               (unsigned long long)g, x, h->P);
        }
    }
    RG_ENTER(h);
    RG_HIP(hipMemcpyAsync(rg_col(h, c), src, bytes, hipMemcpyHostToDevice, h->stream));
    if (c == RG_COL_PFLAGS && h->ins_arena) { // the FULL bit is the engine's: re-derive it from the windows
        const int frc = rg_fix_ins_full(h);
        if (frc) return frc;
    }
    if (c == RG_COL_RUN_FIRST) // ... and the table's fill count
        hipLaunchKernelGGL(k_fix_run_count, dim3((unsigned)((h->G + RG_BLOCK - 1) / RG_BLOCK)), dim3(RG_BLOCK), 0, h->stream, h->st);
    if (c == RG_COL_PFLAGS || c == RG_COL_PEND_SNAP || c == RG_COL_PEND_RS) // ... and so is RG_PF_PENDING
        hipLaunchKernelGGL(k_fix_pending, dim3((unsigned)((h->G + RG_BLOCK - 1) / RG_BLOCK)), dim3(RG_BLOCK), 0, h->stream,
                           h->st, h->P);
    RG_HIP(hipStreamSynchronize(h->stream));
    if (c == RG_COL_COMMIT && h->pub) h->pub->local_lost = true;
    if (c == RG_COL_CFG) {
        h->host_cfg_valid = false;
        h->cls_stale = true;
        const u32 *w = static_cast<const u32 *>(src);
        bool any = false;
        for (u64 g = 0; g < h->G && !any; g++) any = (w[g] & RG_CFG_GROUP_COMMIT) != 0;
        h->any_group_commit = any;
    }
    return RG_OK;
} RG_ABI_GUARD

extern "C" int rg_read_column(rg_engine *h, int c, void *dst, uint64_t bytes) try {
    if (!h || !dst || c < 0 || c >= RG_COL_COUNT) return rg_fail(RG_ERR_INVALID_ARG, "rg_read_column: bad argument");
    if (bytes != rg_column_bytes(h, c))
        return rg_fail(RG_ERR_INVALID_ARG, "rg_read_column(%d): %llu bytes given, %llu expected", c,
                       (unsigned long long)bytes, (unsigned long long)rg_column_bytes(h, c));
    RG_ENTER(h);
    RG_HIP(hipMemcpyAsync(dst, rg_col(h, c), bytes, hipMemcpyDeviceToHost, h->stream));
    RG_HIP(hipStreamSynchronize(h->stream));
    return RG_OK;
} RG_ABI_GUARD

extern "C" void *rg_column_ptr(rg_engine *h, int c) {
    if (!h || c < 0 || c >= RG_COL_COUNT) return nullptr;
    if (c == RG_COL_CFG) h->cls_off = true; // whoever holds this pointer can rewrite cfg words behind the engine's back
    return rg_col(h, c);
}

extern "C" int rg_checkpoint(rg_engine *h) try {
    if (!h) return rg_fail(RG_ERR_INVALID_ARG, "rg_checkpoint: null engine");
    RG_ENTER(h);
    if (!h->ckpt) RG_HIP(hipMalloc(&h->ckpt, h->state_bytes));
    RG_HIP(hipMemcpyAsync(h->ckpt, h->arena, h->state_bytes, hipMemcpyDeviceToDevice, h->stream));
    h->ckpt_any_group_commit = h->any_group_commit;
    if (h->ins_arena) {
        if (!h->ins_ckpt) RG_HIP(hipMalloc(&h->ins_ckpt, h->ins_state_bytes));
        RG_HIP(hipMemcpyAsync(h->ins_ckpt, h->ins_arena, h->ins_state_bytes, hipMemcpyDeviceToDevice, h->stream));
        h->ckpt_send_ready = h->send_ready;
        // RG_COL_OUT / RG_COL_HOST_HINT are part of the snapshot, so "a hint may be unanswered" is too: a restore brings the
        // flagged groups back, and the next step must refuse them again (rg_require_hints_resolved)
        h->ckpt_hint_check_due = h->hint_check_due;
    }
    if (h->esz) {
        const size_t b = (size_t)h->G * h->ins.esz_w * 4;
        if (!h->esz_ckpt) RG_HIP(hipMalloc(&h->esz_ckpt, b));
        RG_HIP(hipMemcpyAsync(h->esz_ckpt, h->esz, b, hipMemcpyDeviceToDevice, h->stream));
    }
    return RG_OK;
} RG_ABI_GUARD

extern "C" int rg_restore(rg_engine *h) try {
    if (!h) return rg_fail(RG_ERR_INVALID_ARG, "rg_restore: null engine");
    if (!h->ckpt) return rg_fail(RG_ERR_STATE, "rg_restore: no checkpoint taken");
    RG_ENTER(h);
    RG_HIP(hipMemcpyAsync(h->arena, h->ckpt, h->state_bytes, hipMemcpyDeviceToDevice, h->stream));
    h->host_res_valid = false;
    if (h->pub) h->pub->local_lost = true; // the published advances no longer describe this commit column
    h->out_is_dense = true; // RG_COL_OUT is whatever it was at the checkpoint: the next sparse tick clears all of it
    h->host_cfg_valid = false; // RG_COL_CFG came back too: the mirror re-reads its copy
    h->cls_stale = true;
    if (h->ckpt_any_group_commit) h->any_group_commit = true; // ... and so may group-commit configurations
    if (h->ins_arena && h->ins_ckpt) {
        // (work items of the last dense stage that name the window's tail -- RG_SEND_LAST_IS_TAIL -- are read against the LIVE
        //  tail column: the compact list is made from them before the windows change under it)
        const int mrc = rg_send_materialize(h);
        if (mrc) return mrc;
        RG_HIP(hipMemcpyAsync(h->ins_arena, h->ins_ckpt, h->ins_state_bytes, hipMemcpyDeviceToDevice, h->stream));
        h->send_ready = h->ckpt_send_ready; // RG_COL_OUT is part of the state: the tick's requests are back too
        h->hint_check_due = h->ckpt_hint_check_due; // ... and so are its unanswered host hints (counted, not probed)
        h->hint_probe_pending = false;
    }
    if (h->esz && h->esz_ckpt)
        RG_HIP(hipMemcpyAsync(h->esz, h->esz_ckpt, (size_t)h->G * h->ins.esz_w * 4, hipMemcpyDeviceToDevice, h->stream));
    return RG_OK;
} RG_ABI_GUARD



extern "C" int rg_write_cells(rg_engine *h, const rg_cell_write *cells, uint64_t n) try {
    if (!h || (!cells && n)) return rg_fail(RG_ERR_INVALID_ARG, "rg_write_cells: bad argument");
    if (n == 0) return RG_OK;
    RG_ENTER(h);
    if (n > h->d_cells_cap) { // engine-owned staging, grown geometrically (this call sits between ticks)
        if (h->d_cells) {
            RG_HIP(hipStreamSynchronize(h->stream));
            (void)hipFree(h->d_cells);
            h->d_cells = nullptr;
            h->d_cells_cap = 0;
        }
        u64 cap = 1024;
        while (cap < n) cap *= 2;
        RG_HIP(hipMalloc(&h->d_cells, cap * sizeof(rg_cell_write)));
        h->d_cells_cap = cap;
    }
    RG_HIP(hipMemcpyAsync(h->d_cells, cells, n * sizeof(rg_cell_write), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_write_cells, dim3(rg_grid(n, 256)), dim3(256), 0, h->stream, h->st, h->d_cells, (u64)n, h->P,
                       h->ins.meta);
    RG_HIP(hipStreamSynchronize(h->stream)); // the caller's array may be reused after return
    return RG_OK;
} RG_ABI_GUARD

extern "C" int rg_read_groups(rg_engine *h, const uint64_t *groups, uint64_t n, rg_group_status *host_out) try {
    if (!h || (n && (!groups || !host_out))) return rg_fail(RG_ERR_INVALID_ARG, "rg_read_groups: bad argument");
    if (n == 0) return RG_OK;
    RG_ENTER(h);
    char *d = nullptr; // [n x u64 group ids | n x rg_group_status]
    const size_t ids_b = rg_align(n * 8);
    RG_HIP(hipMalloc(&d, ids_b + n * sizeof(rg_group_status)));
    hipError_t e = hipMemcpyAsync(d, groups, n * 8, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_read_groups, dim3(rg_grid(n, 128)), dim3(128), 0, h->stream, h->st, (const u64 *)d, (u64)n, h->P,
                           (const u32 *)h->ins.meta, reinterpret_cast<rg_group_status *>(d + ids_b));
        e = hipMemcpyAsync(host_out, d + ids_b, n * sizeof(rg_group_status), hipMemcpyDeviceToHost, h->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return rg_fail(RG_ERR_NO_DEVICE, "rg_read_groups: %s", hipGetErrorString(e));
    for (u64 i = 0; i < n; i++)
        if (host_out[i].group == ~0ULL)
            return rg_fail(RG_ERR_INVALID_ARG, "rg_read_groups: group %llu does not exist (engine holds %llu)",
                           (unsigned long long)groups[i], (unsigned long long)h->G);
    return RG_OK;
} RG_ABI_GUARD

extern "C" int rg_set_config(rg_engine *h, uint64_t group, uint32_t cfg_word) try {
    if (!h || group >= h->G) return rg_fail(RG_ERR_INVALID_ARG, "rg_set_config: bad argument");
    if (RG_CFG_SELF(cfg_word) >= h->P || (RG_CFG_PRESENT(cfg_word) >> h->P) || (RG_CFG_INCOMING(cfg_word) >> h->P) ||
        (RG_CFG_OUTGOING(cfg_word) >> h->P) || RG_CFG_TRANSFEREE(cfg_word) > h->P)
        return rg_fail(RG_ERR_INVALID_ARG, "rg_set_config: cfg word %#x names a slot >= %u", cfg_word, h->P);
    RG_ENTER(h);
    RG_HIP(hipMemcpyAsync(h->st.cfg + group, &cfg_word, 4, hipMemcpyHostToDevice, h->stream));
    RG_HIP(hipStreamSynchronize(h->stream));
    if (cfg_word & RG_CFG_GROUP_COMMIT) h->any_group_commit = true; // (stays set: the GC kernel is a superset)
    if (h->host_cfg_valid) h->host_cfg[group] = cfg_word;
    // (a word that stays inside its block's class changes nothing: the class is an upper bound)
    if (!h->cls_stale && h->cls_on && rg_cfg_slots_named(cfg_word) > h->cls_host[group / RG_BLOCK]) h->cls_stale = true;
    return RG_OK;
} RG_ABI_GUARD


