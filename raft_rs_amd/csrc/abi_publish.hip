// abi_publish.hip -- multi-GPU: publication of commit indices (encoding and replica arithmetic in rg_publish.h)
// There is NO CPU fallback anywhere in this file: without a HIP device every entry point fails.
#include "rg_engine.h"
#include "rg_kernels_publish.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

struct RgRccl {
    void *lib;
    decltype(&ncclGetUniqueId) GetUniqueId;
    decltype(&ncclCommInitRank) CommInitRank;
    decltype(&ncclCommDestroy) CommDestroy;
    decltype(&ncclAllGather) AllGather;
    decltype(&ncclGetErrorString) GetErrorString;
    decltype(&ncclGroupStart) GroupStart;
    decltype(&ncclGroupEnd) GroupEnd;
    decltype(&ncclCommCount) CommCount;       // (rg_comm_info: what the COMMUNICATOR says about itself)
    decltype(&ncclCommUserRank) CommUserRank;
};
static RgRccl g_rccl = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
static std::mutex g_rccl_mu; // (engines of one process may be driven by one thread each: the first loads, the others wait)

int rg_rccl_load() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.lib) return RG_OK;
    static const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *lib = nullptr;
    for (const char *n : names)
        if ((lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!lib) return rg_fail(RG_ERR_NO_DEVICE, "rg_comm: cannot load RCCL (librccl.so.1): %s", dlerror());
    g_rccl.GetUniqueId = reinterpret_cast<decltype(&ncclGetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(&ncclCommInitRank)>(dlsym(lib, "ncclCommInitRank"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(&ncclCommDestroy)>(dlsym(lib, "ncclCommDestroy"));
    g_rccl.AllGather = reinterpret_cast<decltype(&ncclAllGather)>(dlsym(lib, "ncclAllGather"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(&ncclGetErrorString)>(dlsym(lib, "ncclGetErrorString"));
    g_rccl.GroupStart = reinterpret_cast<decltype(&ncclGroupStart)>(dlsym(lib, "ncclGroupStart"));
    g_rccl.GroupEnd = reinterpret_cast<decltype(&ncclGroupEnd)>(dlsym(lib, "ncclGroupEnd"));
    g_rccl.CommCount = reinterpret_cast<decltype(&ncclCommCount)>(dlsym(lib, "ncclCommCount"));
    g_rccl.CommUserRank = reinterpret_cast<decltype(&ncclCommUserRank)>(dlsym(lib, "ncclCommUserRank"));
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllGather || !g_rccl.GetErrorString ||
        !g_rccl.GroupStart || !g_rccl.GroupEnd || !g_rccl.CommCount || !g_rccl.CommUserRank) {
        dlclose(lib);
        return rg_fail(RG_ERR_NO_DEVICE, "rg_comm: the RCCL library lacks an expected symbol");
    }
    g_rccl.lib = lib;
    return RG_OK;
}


// ------------------------------------------------------------------------------------------------
// multi-GPU: publication of commit indices (encoding and replica kernels in rg_publish.h)
// ------------------------------------------------------------------------------------------------

static const char *rg_nccl_err(ncclResult_t r) { return g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?"; }

// Measurement knobs of EXPERIMENT builds only (python -m raft_rs_amd.build --exp pubdbg -DRG_PUB_DEBUG_BUILD=1; the default
// library reads no environment): RG_PUB_DEBUG bit 0 = the slice re-use gate as a cross-stream wait on the engine's stream
// (round 2's form), bit 1 = no reset of the slice, bit 2 = no exchange (both make the replica wrong: what each part costs,
// profiles/r02_publish_overhead.txt, r06_publish_event.txt)
static int rg_pub_dbg() {
#ifdef RG_PUB_DEBUG_BUILD
    static const int dbg = getenv("RG_PUB_DEBUG") ? atoi(getenv("RG_PUB_DEBUG")) : 0;
    return dbg;
#else
    return 0;
#endif
}

static int rg_pub_allgather(rg_engine *h, const void *send, void *recv, u64 bytes) {
    RgPub *p = h->pub;
    if (rg_pub_dbg() & 4) return RG_OK;
    if (p->transport) {
        const int rc = p->transport(p->transport_user, send, recv, bytes, p->side);
        if (rc) return rg_fail(RG_ERR_NO_DEVICE, "rg_publish_commit: the custom all-gather transport failed (%d)", rc);
        return RG_OK;
    }
    const ncclResult_t r = g_rccl.AllGather(send, recv, (size_t)bytes, ncclUint8, p->comm, p->side);
    if (r != ncclSuccess) return rg_fail(RG_ERR_NO_DEVICE, "rg_publish_commit: ncclAllGather failed: %s", rg_nccl_err(r));
    return RG_OK;
}

// point the tick kernels at send buffer `b`
static void rg_pub_target(rg_engine *h, int b) {
    h->st.pub = h->pub->send[b];
    h->st.pub_off_delta = h->pub->lay.off_delta;
    h->st.pub_cap = h->pub->lay.cap;
}

// fold the buffered publications into the replica (side stream)
static int rg_pub_materialize(rg_engine *h) {
    RgPub *p = h->pub;
    if (!p->pending) return RG_OK;
    RgPubSlots sl;
    sl.n = p->pending;
    const u64 slot_bytes = (u64)p->world * p->lay.bytes_per_rank;
    for (u32 j = 0; j < p->pending; j++) // the `pending` most recent delta publications, ring order is irrelevant
        sl.slice[j] = p->ring_buf + (u64)j * slot_bytes;
    const u64 words = p->lay.Gpad / 8 * p->world;
    hipLaunchKernelGGL(k_pub_apply, dim3(rg_grid(words, 256)), dim3(256), 0, p->side, p->replica, sl, p->lay, p->world);
    const u64 entries = (u64)p->world * p->lay.cap * sl.n;
    hipLaunchKernelGGL(k_pub_apply_lists, dim3(rg_grid(entries, 256)), dim3(256), 0, p->side, p->replica, sl, p->lay,
                       p->world, p->d_lost);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return rg_fail(RG_ERR_NO_DEVICE, "rg_publish: replica update failed: %s", hipGetErrorString(e));
    p->pending = 0;
    p->stats.replica_updates++;
    return RG_OK;
}

static inline double rg_now_us() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

// One publication of one engine in three phases, so that a single host thread can drive several engines of ONE process
// through the same exchange (rg_publish_commit_all): `pre` of every engine (everything up to the exchange: the loss protocol's
// check point, the snapshot of a full publication, the side stream ordered behind the tick that completed the slice), the
// exchange of all of them together (RCCL: one ncclGroupStart / ncclAllGather x n / ncclGroupEnd -- the group is what keeps a
// single thread from blocking in rank 0's collective while rank 1's has not been issued; the in-process transport:
// device-to-device copies), `post` of every engine (the slice starts its next interval, rotation). rg_publish_commit is the
// three phases of one engine back to back.
struct RgPubStep {
    bool full;
    int b;
    const void *send;
    void *recv;
    u64 bytes;
    double t0, t1, t2;
};

static int rg_pub_pre(rg_engine *h, bool force_full, RgPubStep &s) {
    RgPub *p = h->pub;
    const u64 i = p->n_pub;
    const int b = (int)(i % RG_PUB_SEND);
    s.b = b;
    s.t0 = rg_now_us();
    // Loss protocol. Every `ring` publications is a CHECK POINT (the same publication numbers on every rank): all
    // buffered slices are folded into the replica first -- the update kernel raises d_lost for a slice that carries
    // RG_PUB_LOST or an overfull list -- and d_lost is copied to the host. The copy of the PREVIOUS check point
    // (finished long ago: no stall) decides whether this publication is a full snapshot. Every rank reads the same
    // gathered headers at the same publication numbers, so every rank decides the same without another collective.
    bool full = force_full;
    const bool check = (i % p->ring) == 0 && i != 0;
    if (check) {
        const int cb = (int)((i / p->ring) & 1), pb = cb ^ 1;
        if (p->chk_pending[pb]) {
            RG_HIP(hipEventSynchronize(p->ev_chk[pb]));
            p->chk_pending[pb] = false;
            if (p->pin_lost[pb]) full = true;
        }
        int rc = rg_pub_materialize(h);
        if (rc) return rc;
        RG_HIP(hipMemcpyAsync(&p->pin_lost[cb], p->d_lost, 4, hipMemcpyDeviceToHost, p->side));
        RG_HIP(hipMemsetAsync(p->d_lost, 0, 4, p->side));
        RG_HIP(hipEventRecord(p->ev_chk[cb], p->side));
        p->chk_pending[cb] = true;
    }
    bool announced = false;
    if (p->local_lost && !p->lost_announced && !full) { // tell the other ranks (they act on it at a check point)
        announced = true;
        static const u32 k_lost = RG_PUB_LOST;
        RG_HIP(hipMemcpyAsync(p->send[b] + offsetof(RgPubHdr, flags), &k_lost, 4, hipMemcpyHostToDevice, h->stream));
        p->lost_announced = true;
    }
    if (full) // snapshot the column before later ticks move it
        RG_HIP(hipMemcpyAsync(p->full_send, h->st.commit, h->G * 8, hipMemcpyDeviceToDevice, h->stream));
    // the event that marks the slice complete: recorded here -- or already on its way, on the dispatch packet of the dense tick
    // this publication follows (rg_tick_impl: h->pub_rode; nothing else has been put on the engine's stream since, this call's
    // own copies above included)
    if (!(p->rode_slot == b && !full && !announced)) RG_HIP(hipEventRecord(p->ev_tick[b], h->stream));
    else p->stats.events_on_tick_packets++;
    p->rode_slot = -1;
    RG_HIP(hipStreamWaitEvent(p->side, p->ev_tick[b], 0));
    s.full = full;
    if (full) {
        // the snapshot supersedes every buffered delta publication (and this interval's deltas)
        p->pending = 0;
        s.send = p->full_send;
        s.recv = p->replica;
        s.bytes = p->lay.Gpad * 8;
    } else {
        if (p->pending == p->ring) { // (reads between check points can leave the ring out of step with them)
            int rc = rg_pub_materialize(h);
            if (rc) return rc;
        }
        s.send = p->send[b];
        s.recv = p->ring_buf + (u64)p->pending * p->world * p->lay.bytes_per_rank;
        s.bytes = p->lay.bytes_per_rank;
    }
    s.t1 = rg_now_us();
    return RG_OK;
}

static int rg_pub_post(rg_engine *h, RgPubStep &s) {
    RgPub *p = h->pub;
    const int b = s.b;
    const int dbg = rg_pub_dbg();
    if (s.full) {
        p->local_lost = false;
        p->lost_announced = false;
        p->stats.full_publications++;
        p->stats.bytes_per_rank_last = p->lay.Gpad * 8;
    } else {
        p->pending++;
        p->stats.bytes_per_rank_last = p->lay.bytes_per_rank;
    }
    s.t2 = rg_now_us();
    // this slice starts its next interval empty
    if (!(dbg & 2)) RG_HIP(hipMemsetAsync(p->send[b], 0, p->lay.bytes_per_rank, p->side));
    const double t3 = rg_now_us();
    RG_HIP(hipEventRecord(p->ev_done[b], p->side));
    p->done_pending[b] = true;
    // the ticks that follow accumulate into the next slice, once the exchange that last read it has let go of it
    const int nb = (b + 1) % RG_PUB_SEND;
    if (p->done_pending[nb]) {
        if (dbg & 1) RG_HIP(hipStreamWaitEvent(h->stream, p->ev_done[nb], 0)); // (the variant that was measured against)
        else RG_HIP(hipEventSynchronize(p->ev_done[nb]));
        p->done_pending[nb] = false;
    }
    rg_pub_target(h, nb);
    p->n_pub++;
    p->stats.publications++;
    const double t4 = rg_now_us();
    p->stats.host_us_events += (s.t1 - s.t0) + (t4 - t3);
    p->stats.host_us_allgather += s.t2 - s.t1;
    p->stats.host_us_memset += t3 - s.t2;
    return RG_OK;
}

static int rg_publish_impl(rg_engine *h, bool force_full) {
    if (h->pub->in_process)
        return rg_fail(RG_ERR_STATE, "rg_publish_commit: this engine is one of several ranks driven by ONE thread (rg_comm_init_all): "
                                     "publish through rg_publish_commit_all");
    RgPubStep s;
    int rc = rg_pub_pre(h, force_full, s);
    if (rc) return rc;
    rc = rg_pub_allgather(h, s.send, s.recv, s.bytes);
    if (rc) return rc;
    return rg_pub_post(h, s);
}

// ---- several engines of ONE process, driven by ONE thread: every engine is a rank of the same publication ----
// The exchange of all ranks in one go. RCCL: the n ncclAllGather calls inside one group (each on its engine's device and side
// stream). In-process transport (RG_COMM_ALL_LOCAL; engines that share a device, or a host that does not want RCCL): rank i's
// side stream copies every rank's slice into its gather buffer, device to device, behind the event that marks that slice
// complete; afterwards every rank's side stream waits for the others' copies of ITS slice, so that the slice is not reset
// (post) while somebody still reads it.
static int rg_pub_gather_all(rg_engine *const *e, uint32_t n, RgPubStep *st) {
    for (uint32_t i = 1; i < n; i++)
        if (st[i].bytes != st[0].bytes || st[i].full != st[0].full)
            return rg_fail(RG_ERR_STATE, "rg_publish_commit_all: the ranks disagree about the form of this publication "
                                         "(engine %u: %s, engine 0: %s) -- they must be published together, always", i,
                           st[i].full ? "full" : "delta", st[0].full ? "full" : "delta");
    if (e[0]->pub->comm) {
        ncclResult_t r = g_rccl.GroupStart();
        hipError_t he = hipSuccess;
        for (uint32_t i = 0; i < n && r == ncclSuccess && he == hipSuccess; i++) {
            he = hipSetDevice(e[i]->cfg.device);
            if (he == hipSuccess)
                r = g_rccl.AllGather(st[i].send, st[i].recv, (size_t)st[i].bytes, ncclUint8, e[i]->pub->comm, e[i]->pub->side);
        }
        const ncclResult_t r2 = g_rccl.GroupEnd(); // (always: an open group would swallow every later RCCL call of the thread)
        if (he != hipSuccess) return rg_fail(RG_ERR_NO_DEVICE, "rg_publish_commit_all: hipSetDevice failed: %s", hipGetErrorString(he));
        if (r == ncclSuccess) r = r2;
        if (r != ncclSuccess) return rg_fail(RG_ERR_NO_DEVICE, "rg_publish_commit_all: grouped ncclAllGather failed: %s", rg_nccl_err(r));
        return RG_OK;
    }
    for (uint32_t i = 0; i < n; i++) {
        RgPub *p = e[i]->pub;
        RG_HIP(hipSetDevice(e[i]->cfg.device));
        for (uint32_t r = 0; r < n; r++) {
            if (r != i) RG_HIP(hipStreamWaitEvent(p->side, e[r]->pub->ev_tick[st[r].b], 0));
            RG_HIP(hipMemcpyAsync(reinterpret_cast<char *>(st[i].recv) + (u64)r * st[i].bytes, st[r].send, st[i].bytes,
                                  hipMemcpyDeviceToDevice, p->side));
        }
        RG_HIP(hipEventRecord(p->ev_read, p->side));
    }
    for (uint32_t i = 0; i < n; i++) {
        RG_HIP(hipSetDevice(e[i]->cfg.device));
        for (uint32_t r = 0; r < n; r++)
            if (r != i) RG_HIP(hipStreamWaitEvent(e[i]->pub->side, e[r]->pub->ev_read, 0));
    }
    return RG_OK;
}

// What rg_pub_pre WILL decide for this engine's next publication -- full snapshot or delta slice -- without changing anything
// (it waits, at most, for the previous check point's 4-byte copy, which pre would wait for anyway).
static int rg_pub_peek_full(rg_engine *h, bool force_full, bool *full) {
    RgPub *p = h->pub;
    *full = force_full;
    const u64 i = p->n_pub;
    if ((i % p->ring) == 0 && i != 0) {
        const int pb = (int)((i / p->ring) & 1) ^ 1;
        if (p->chk_pending[pb]) {
            RG_HIP(hipSetDevice(h->cfg.device));
            RG_HIP(hipEventSynchronize(p->ev_chk[pb]));
            if (p->pin_lost[pb]) *full = true;
        }
    }
    return RG_OK;
}

static int rg_publish_all_impl(rg_engine *const *e, uint32_t n, bool force_full) {
    std::vector<RgPubStep> st(n);
    // the ranks must agree on the publication's number and form BEFORE any of them commits its pre-state (the check point's
    // copies, pending, ev_tick): a disagreement found after rg_pub_pre would leave the later publications out of step
    bool full0 = false;
    for (uint32_t i = 0; i < n; i++) {
        bool full = false;
        const int prc = rg_pub_peek_full(e[i], force_full, &full);
        if (prc) return prc;
        if (i == 0) full0 = full;
        if (e[i]->pub->n_pub != e[0]->pub->n_pub || full != full0)
            return rg_fail(RG_ERR_STATE, "rg_publish_commit_all: the ranks disagree about this publication (engine %u: #%llu %s, engine 0: "
                                         "#%llu %s) -- they must be published together, always; nothing was published", i,
                           (unsigned long long)e[i]->pub->n_pub, full ? "full" : "delta", (unsigned long long)e[0]->pub->n_pub,
                           full0 ? "full" : "delta");
    }
    for (uint32_t i = 0; i < n; i++) {
        RG_HIP(hipSetDevice(e[i]->cfg.device));
        int rc = rg_mailbox_quiesce(e[i]);
        if (!rc) rc = rg_pub_pre(e[i], force_full, st[i]);
        if (rc) return rc;
    }
    int rc = rg_pub_gather_all(e, n, st.data());
    if (rc) return rc;
    for (uint32_t i = 0; i < n; i++) {
        RG_HIP(hipSetDevice(e[i]->cfg.device));
        rc = rg_pub_post(e[i], st[i]);
        if (rc) return rc;
    }
    return RG_OK;
}

extern "C" int rg_comm_unique_id(uint8_t id[RG_COMM_ID_BYTES]) try {
    if (!id) return rg_fail(RG_ERR_INVALID_ARG, "rg_comm_unique_id: null argument");
    int rc = rg_rccl_load();
    if (rc) return rc;
    static_assert(RG_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "RG_COMM_ID_BYTES must match RCCL's unique id");
    ncclUniqueId u;
    const ncclResult_t r = g_rccl.GetUniqueId(&u);
    if (r != ncclSuccess) return rg_fail(RG_ERR_NO_DEVICE, "rg_comm_unique_id: ncclGetUniqueId failed: %s", rg_nccl_err(r));
    memcpy(id, u.internal, RG_COMM_ID_BYTES);
    return RG_OK;
} RG_ABI_GUARD

extern "C" int rg_comm_destroy(rg_engine *h) try {
    if (!h) return rg_fail(RG_ERR_INVALID_ARG, "rg_comm_destroy: null engine");
    RgPub *p = h->pub;
    if (!p) return RG_OK;
    (void)hipSetDevice(h->cfg.device);
    (void)rg_mailbox_quiesce(h);
    (void)hipStreamSynchronize(h->stream);
    if (p->side) (void)hipStreamSynchronize(p->side);
    if (p->comm) (void)g_rccl.CommDestroy(p->comm);
    for (int k = 0; k < RG_PUB_SEND; k++) {
        if (p->send[k]) (void)hipFree(p->send[k]);
        if (p->ev_tick[k]) (void)hipEventDestroy(p->ev_tick[k]);
        if (p->ev_done[k]) (void)hipEventDestroy(p->ev_done[k]);
    }
    for (int k = 0; k < 2; k++)
        if (p->ev_chk[k]) (void)hipEventDestroy(p->ev_chk[k]);
    if (p->ring_buf) (void)hipFree(p->ring_buf);
    if (p->replica) (void)hipFree(p->replica);
    if (p->full_send) (void)hipFree(p->full_send);
    if (p->d_lost) (void)hipFree(p->d_lost);
    if (p->pin_lost) (void)hipHostFree(p->pin_lost);
    if (p->ev_read) (void)hipEventDestroy(p->ev_read);
    if (p->side) (void)hipStreamDestroy(p->side);
    delete p;
    h->pub = nullptr;
    h->st.pub = nullptr;
    return RG_OK;
} RG_ABI_GUARD

extern "C" int rg_comm_info_get(rg_engine *h, rg_comm_info *out) try {
    if (!h || !out) return rg_fail(RG_ERR_INVALID_ARG, "rg_comm_info_get: bad argument");
    memset(out, 0, sizeof(*out));
    RgPub *p = h->pub;
    if (!p) return RG_OK; // (transport RG_TRANSPORT_NONE: a single engine)
    out->rank = p->rank;
    out->world = p->world;
    out->in_process = p->in_process ? 1u : 0u;
    out->transport = p->comm ? RG_TRANSPORT_RCCL : p->transport ? RG_TRANSPORT_CALLBACK : RG_TRANSPORT_LOCAL;
    if (p->comm) { // not what the engine was TOLD (rank / world above) but what the RCCL communicator reports
        int count = 0, urank = -1;
        ncclResult_t r = g_rccl.CommCount(p->comm, &count);
        if (r == ncclSuccess) r = g_rccl.CommUserRank(p->comm, &urank);
        if (r != ncclSuccess) return rg_fail(RG_ERR_NO_DEVICE, "rg_comm_info_get: ncclCommCount / ncclCommUserRank failed: %s", rg_nccl_err(r));
        out->rccl_ranks = (uint32_t)count;
        out->rccl_rank = (uint32_t)urank;
    }
    return RG_OK;
} RG_ABI_GUARD

// RCCL's first use in a process is slow -- the library is ~0.5 GB to map (seconds to minutes on a cold box:
// profiles/r05_rccl_cold_load.txt) and the first communicator sets up its transports. A host that bounds its start-up steps
// with timeouts calls this once, early, on the thread and device it will use: it loads the library and creates and destroys a
// one-rank communicator, so that the first real rg_comm_init finds everything mapped.
extern "C" int rg_comm_warmup(void) try {
    int rc = rg_rccl_load();
    if (rc) return rc;
    ncclUniqueId u;
    ncclResult_t r = g_rccl.GetUniqueId(&u);
    ncclComm_t comm = nullptr;
    if (r == ncclSuccess) r = g_rccl.CommInitRank(&comm, 1, u, 0);
    if (r == ncclSuccess) {
        int count = 0;
        r = g_rccl.CommCount(comm, &count);
        if (r == ncclSuccess && count != 1) r = ncclInternalError;
    }
    if (comm) (void)g_rccl.CommDestroy(comm);
    if (r != ncclSuccess) return rg_fail(RG_ERR_NO_DEVICE, "rg_comm_warmup: RCCL failed: %s", rg_nccl_err(r));
    return RG_OK;
} RG_ABI_GUARD

// Everything of rg_comm_init but the communicator and the first publication: buffers, streams, events.
// xdev: slices are read by OTHER devices (in-process transport across GPUs): the slice-complete events keep their system-scope fence.
static int rg_comm_setup(rg_engine *h, u32 rank, u32 world, u32 ring_ticks, u32 overflow_slots, rg_allgather_fn transport,
                         void *transport_user, bool in_process, bool xdev) {
    RgPub *p = new (std::nothrow) RgPub();
    if (!p) return rg_fail(RG_ERR_OUT_OF_MEMORY, "rg_comm_init: host allocation failed");
    memset(p, 0, sizeof(*p));
    p->rode_slot = -1;
    p->rank = rank;
    p->world = world;
    p->transport = transport;
    p->transport_user = transport_user;
    p->in_process = in_process;
    p->ring = ring_ticks ? ring_ticks : 32;
    const u32 cap = overflow_slots ? overflow_slots : (u32)(h->G / 256 + 64);
    p->lay = rg_pub_layout(h->G, cap);
    h->pub = p;
#define RG_PUB_TRY(expr)                                                                                       \
    do {                                                                                                       \
        hipError_t e__ = (expr);                                                                               \
        if (e__ != hipSuccess) {                                                                               \
            (void)rg_comm_destroy(h);                                                                          \
            return rg_fail(e__ == hipErrorOutOfMemory ? RG_ERR_OUT_OF_MEMORY : RG_ERR_NO_DEVICE,               \
                           "rg_comm_init: %s failed: %s", #expr, hipGetErrorString(e__));                      \
        }                                                                                                      \
    } while (0)
    RG_PUB_TRY(hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking));
    // ev_tick orders the tick kernel before the exchange's first kernel ON THIS DEVICE (ncclAllGather reads the slice
    // with a kernel of this device; a host transport synchronises the device itself), so the system-scope fence a
    // recorded event normally implies -- an L2 write-back worth ~2 us per tick -- is not needed
    // (RG_PUB_DEBUG & 4 keeps it, for A/B measurements: profiles/r02_publish_overhead.txt; so does the in-process
    // transport between DIFFERENT devices, whose copies read the slice from the other GPU)
#ifdef RG_PUB_DEBUG_BUILD
    const unsigned evf = hipEventDisableTiming | ((xdev || (getenv("RG_PUB_DEBUG") && (atoi(getenv("RG_PUB_DEBUG")) & 4))) ? 0 : hipEventDisableSystemFence);
#else
    const unsigned evf = hipEventDisableTiming | (xdev ? 0u : (unsigned)hipEventDisableSystemFence);
#endif
    for (int k = 0; k < RG_PUB_SEND; k++) {
        RG_PUB_TRY(hipMalloc(&p->send[k], p->lay.bytes_per_rank));
        RG_PUB_TRY(hipMemsetAsync(p->send[k], 0, p->lay.bytes_per_rank, h->stream));
        RG_PUB_TRY(hipEventCreateWithFlags(&p->ev_tick[k], evf));
        RG_PUB_TRY(hipEventCreateWithFlags(&p->ev_done[k], hipEventDisableTiming));
    }
    for (int k = 0; k < 2; k++) RG_PUB_TRY(hipEventCreateWithFlags(&p->ev_chk[k], hipEventDisableTiming));
    RG_PUB_TRY(hipEventCreateWithFlags(&p->ev_read, hipEventDisableTiming));
    RG_PUB_TRY(hipMalloc(&p->ring_buf, (size_t)p->ring * p->world * p->lay.bytes_per_rank));
    RG_PUB_TRY(hipMalloc(&p->replica, (size_t)p->world * p->lay.Gpad * 8));
    RG_PUB_TRY(hipMemsetAsync(p->replica, 0, (size_t)p->world * p->lay.Gpad * 8, h->stream));
    RG_PUB_TRY(hipMalloc(&p->full_send, p->lay.Gpad * 8));
    RG_PUB_TRY(hipMemsetAsync(p->full_send, 0, p->lay.Gpad * 8, h->stream));
    RG_PUB_TRY(hipMalloc(&p->d_lost, 256));
    RG_PUB_TRY(hipMemsetAsync(p->d_lost, 0, 256, h->stream));
    RG_PUB_TRY(hipHostMalloc(reinterpret_cast<void **>(&p->pin_lost), 64, hipHostMallocDefault));
    memset(p->pin_lost, 0, 64);
    RG_PUB_TRY(hipStreamSynchronize(h->stream));
#undef RG_PUB_TRY
    h->dev.engine_bytes += RG_PUB_SEND * p->lay.bytes_per_rank + (u64)p->ring * p->world * p->lay.bytes_per_rank +
                           (u64)p->world * p->lay.Gpad * 8 + p->lay.Gpad * 8;
    rg_pub_target(h, 0);
    return RG_OK;
}

extern "C" int rg_comm_init(rg_engine *h, const rg_comm_config *cfg) try {
    if (!h || !cfg) return rg_fail(RG_ERR_INVALID_ARG, "rg_comm_init: null argument");
    if (h->pub) return rg_fail(RG_ERR_STATE, "rg_comm_init: already initialised (rg_comm_destroy first)");
    if (cfg->world == 0 || cfg->rank >= cfg->world)
        return rg_fail(RG_ERR_INVALID_ARG, "rg_comm_init: rank %u of %u", cfg->rank, cfg->world);
    if (!cfg->transport && !cfg->unique_id)
        return rg_fail(RG_ERR_INVALID_ARG, "rg_comm_init: the RCCL transport needs the unique id of rg_comm_unique_id");
    if (cfg->ring_ticks > RG_PUB_MAX_RING)
        return rg_fail(RG_ERR_INVALID_ARG, "rg_comm_init: ring_ticks %u, at most %d", cfg->ring_ticks, RG_PUB_MAX_RING);
    RG_ENTER(h);
    int rc = rg_comm_setup(h, cfg->rank, cfg->world, cfg->ring_ticks, cfg->overflow_slots, cfg->transport, cfg->transport_user, false, false);
    if (rc) return rc;
    RgPub *p = h->pub;
    if (!cfg->transport) {
        rc = rg_rccl_load();
        if (rc) {
            (void)rg_comm_destroy(h);
            return rc;
        }
        ncclUniqueId u;
        memcpy(u.internal, cfg->unique_id, RG_COMM_ID_BYTES);
        const ncclResult_t r = g_rccl.CommInitRank(&p->comm, (int)p->world, u, (int)p->rank);
        if (r != ncclSuccess) {
            p->comm = nullptr;
            (void)rg_comm_destroy(h);
            return rg_fail(RG_ERR_NO_DEVICE, "rg_comm_init: ncclCommInitRank(rank %u of %u) failed: %s", cfg->rank, cfg->world,
                           rg_nccl_err(r));
        }
    }
    // every replica starts from the actual columns: one full publication (a collective: all ranks are in here)
    rc = rg_publish_impl(h, true);
    if (rc) {
        (void)rg_comm_destroy(h);
        return rc;
    }
    return RG_OK;
} RG_ABI_GUARD

static int rg_all_check(rg_engine *const *engines, uint32_t n, const char *who, bool need_pub) {
    if (!engines || n == 0) return rg_fail(RG_ERR_INVALID_ARG, "%s: no engines", who);
    for (uint32_t i = 0; i < n; i++) {
        if (!engines[i]) return rg_fail(RG_ERR_INVALID_ARG, "%s: engine %u is null", who, i);
        for (uint32_t j = 0; j < i; j++)
            if (engines[j] == engines[i]) return rg_fail(RG_ERR_INVALID_ARG, "%s: engine %u is listed twice", who, i);
        if (engines[i]->G != engines[0]->G)
            return rg_fail(RG_ERR_INVALID_ARG, "%s: engine %u holds %llu groups, engine 0 %llu (equal shards: the all-gather moves equal slices)",
                           who, i, (unsigned long long)engines[i]->G, (unsigned long long)engines[0]->G);
        if (need_pub && (!engines[i]->pub || !engines[i]->pub->in_process || engines[i]->pub->world != n || engines[i]->pub->rank != i))
            return rg_fail(RG_ERR_STATE, "%s: engine %u is not rank %u of %u of an rg_comm_init_all communicator", who, i, i, n);
    }
    return RG_OK;
}

extern "C" int rg_comm_init_all(rg_engine *const *engines, uint32_t n, const rg_comm_all_config *cfg) try {
    int rc = rg_all_check(engines, n, "rg_comm_init_all", false);
    if (rc) return rc;
    rg_comm_all_config c = {0, 0, RG_COMM_ALL_AUTO, 0};
    if (cfg) c = *cfg;
    if (c.transport > RG_COMM_ALL_LOCAL) return rg_fail(RG_ERR_INVALID_ARG, "rg_comm_init_all: unknown transport %u", c.transport);
    if (c.ring_ticks > RG_PUB_MAX_RING) return rg_fail(RG_ERR_INVALID_ARG, "rg_comm_init_all: ring_ticks %u, at most %d", c.ring_ticks, RG_PUB_MAX_RING);
    bool shared = false, xdev = false; // two engines on one device (RCCL refuses that) / engines on different devices
    for (uint32_t i = 0; i < n; i++) {
        if (engines[i]->pub) return rg_fail(RG_ERR_STATE, "rg_comm_init_all: engine %u already has a communicator (rg_comm_destroy first)", i);
        for (uint32_t j = 0; j < i; j++) {
            shared = shared || engines[j]->cfg.device == engines[i]->cfg.device;
            xdev = xdev || engines[j]->cfg.device != engines[i]->cfg.device;
        }
    }
    if (c.transport == RG_COMM_ALL_RCCL && shared)
        return rg_fail(RG_ERR_INVALID_ARG, "rg_comm_init_all: RCCL needs one device per rank; two of the engines share one (RG_COMM_ALL_LOCAL)");
    const bool rccl = c.transport == RG_COMM_ALL_RCCL || (c.transport == RG_COMM_ALL_AUTO && !shared && n > 1);
    auto undo = [&](uint32_t upto) {
        for (uint32_t i = 0; i < upto; i++) (void)rg_comm_destroy(engines[i]);
    };
    for (uint32_t i = 0; i < n; i++) {
        rg_engine *h = engines[i];
        const hipError_t he = hipSetDevice(h->cfg.device);
        rc = he == hipSuccess ? rg_mailbox_quiesce(h)
                              : rg_fail(RG_ERR_NO_DEVICE, "rg_comm_init_all: hipSetDevice(%d) failed: %s", h->cfg.device, hipGetErrorString(he));
        if (!rc) rc = rg_comm_setup(h, i, n, c.ring_ticks, c.overflow_slots, nullptr, nullptr, true, !rccl && xdev);
        if (rc) {
            undo(i);
            return rc;
        }
    }
    if (rccl) {
        rc = rg_rccl_load();
        ncclUniqueId u;
        ncclResult_t r = ncclSuccess;
        if (!rc) r = g_rccl.GetUniqueId(&u);
        if (!rc && r == ncclSuccess) {
            // one thread, n ranks: the initialisations of all of them inside ONE group (outside it the first
            // ncclCommInitRank would wait for ranks this very thread has not started yet)
            r = g_rccl.GroupStart();
            for (uint32_t i = 0; i < n && r == ncclSuccess; i++) {
                if (hipSetDevice(engines[i]->cfg.device) != hipSuccess) r = ncclUnhandledCudaError;
                else r = g_rccl.CommInitRank(&engines[i]->pub->comm, (int)n, u, (int)i);
            }
            const ncclResult_t r2 = g_rccl.GroupEnd();
            if (r == ncclSuccess) r = r2;
        }
        if (rc || r != ncclSuccess) {
            // a failed group leaves no usable communicator: whatever it did create is destroyed (rg_comm_destroy below does
            // that for every non-null comm), nothing is leaked and nothing half-initialised survives
            undo(n);
            return rc ? rc : rg_fail(RG_ERR_NO_DEVICE, "rg_comm_init_all: grouped ncclCommInitRank of %u ranks failed: %s", n, rg_nccl_err(r));
        }
    }
    rc = rg_publish_all_impl(engines, n, true); // every replica starts from the actual columns
    if (rc) undo(n);
    return rc;
} RG_ABI_GUARD

extern "C" int rg_publish_commit_all(rg_engine *const *engines, uint32_t n, uint32_t flags) try {
    if (flags & ~RG_PUBLISH_FULL) return rg_fail(RG_ERR_INVALID_ARG, "rg_publish_commit_all: unknown flags %#x", flags);
    int rc = rg_all_check(engines, n, "rg_publish_commit_all", true);
    if (rc) return rc;
    return rg_publish_all_impl(engines, n, (flags & RG_PUBLISH_FULL) != 0);
} RG_ABI_GUARD

extern "C" int rg_publish_commit(rg_engine *h, uint32_t flags) try {
    if (!h) return rg_fail(RG_ERR_INVALID_ARG, "rg_publish_commit: null engine");
    if (!h->pub) return rg_fail(RG_ERR_STATE, "rg_publish_commit: rg_comm_init was never called");
    if (flags & ~RG_PUBLISH_FULL) return rg_fail(RG_ERR_INVALID_ARG, "rg_publish_commit: unknown flags %#x", flags);
    const int rode = h->pub_tick_evt; // (RG_ENTER forgets it: read first)
    RG_ENTER(h);
    h->pub->rode_slot = rode;
    return rg_publish_impl(h, (flags & RG_PUBLISH_FULL) != 0);
} RG_ABI_GUARD

extern "C" int rg_publish_sync(rg_engine *h) try {
    if (!h) return rg_fail(RG_ERR_INVALID_ARG, "rg_publish_sync: null engine");
    if (!h->pub) return rg_fail(RG_ERR_STATE, "rg_publish_sync: rg_comm_init was never called");
    RG_ENTER(h);
    int rc = rg_pub_materialize(h);
    if (rc) return rc;
    RG_HIP(hipStreamSynchronize(h->pub->side));
    return RG_OK;
} RG_ABI_GUARD

extern "C" const uint64_t *rg_published_commit_ptr(rg_engine *h, uint64_t *stride) {
    if (!h || !h->pub) return nullptr;
    if (stride) *stride = h->pub->lay.Gpad;
    return h->pub->replica;
}

extern "C" int rg_published_commit(rg_engine *h, uint32_t rank, uint64_t first, uint64_t n, uint64_t *host_commit) try {
    if (!h || (n && !host_commit)) return rg_fail(RG_ERR_INVALID_ARG, "rg_published_commit: bad argument");
    if (!h->pub) return rg_fail(RG_ERR_STATE, "rg_published_commit: rg_comm_init was never called");
    if (rank >= h->pub->world || first > h->G || n > h->G - first)
        return rg_fail(RG_ERR_INVALID_ARG, "rg_published_commit: rank %u groups [%llu, +%llu) outside %u ranks x %llu groups",
                       rank, (unsigned long long)first, (unsigned long long)n, h->pub->world, (unsigned long long)h->G);
    int rc = rg_publish_sync(h);
    if (rc || !n) return rc;
    RG_HIP(hipMemcpy(host_commit, h->pub->replica + (u64)rank * h->pub->lay.Gpad + first, n * 8, hipMemcpyDeviceToHost));
    return RG_OK;
} RG_ABI_GUARD

extern "C" int rg_publish_stats_get(rg_engine *h, rg_publish_stats *out) try {
    if (!h || !out) return rg_fail(RG_ERR_INVALID_ARG, "rg_publish_stats_get: bad argument");
    if (!h->pub) return rg_fail(RG_ERR_STATE, "rg_publish_stats_get: rg_comm_init was never called");
    *out = h->pub->stats;
    out->bytes_per_rank_delta = h->pub->lay.bytes_per_rank;
    out->bytes_per_rank_full = h->pub->lay.Gpad * 8;
    out->overflow_slots = h->pub->lay.cap;
    out->ring_ticks = h->pub->ring;
    return RG_OK;
} RG_ABI_GUARD

// Host twins of the encoding (no GPU involved): what the tick kernels write and what the replica kernels add, over
// caller-provided buffers. CPU-only tests run the N > 1 exchange with these under gloo.
extern "C" uint64_t rg_pub_bytes_per_rank(uint64_t n_groups, uint32_t overflow_slots) {
    return rg_pub_layout(n_groups, overflow_slots ? overflow_slots : (u32)(n_groups / 256 + 64)).bytes_per_rank;
}

extern "C" int rg_pub_accumulate_host(uint64_t n_groups, uint32_t overflow_slots, const uint64_t *old_commit,
                                      const uint64_t *new_commit, uint8_t *slice) try {
    if (!old_commit || !new_commit || !slice) return rg_fail(RG_ERR_INVALID_ARG, "rg_pub_accumulate_host: null argument");
    const RgPubLayout l = rg_pub_layout(n_groups, overflow_slots ? overflow_slots : (u32)(n_groups / 256 + 64));
    RgPubHdr *hdr = reinterpret_cast<RgPubHdr *>(slice);
    RgPubOvf *list = reinterpret_cast<RgPubOvf *>(slice + l.off_list);
    u8 *dlt = slice + l.off_delta;
    for (u64 g = 0; g < n_groups; g++) {
        if (new_commit[g] < old_commit[g]) return rg_fail(RG_ERR_INVALID_ARG, "rg_pub_accumulate_host: group %llu: the commit index decreased", (unsigned long long)g);
        if (new_commit[g] != old_commit[g]) dlt[g] = (u8)rg_pub_accumulate(dlt[g], old_commit[g], new_commit[g], g, hdr, list, l.cap);
    }
    return RG_OK;
} RG_ABI_GUARD

extern "C" int rg_pub_apply_host(uint64_t n_groups, uint32_t overflow_slots, uint32_t world, const uint8_t *gathered,
                                 uint64_t *replica, uint32_t *lost_ranks) try {
    if (!gathered || !replica) return rg_fail(RG_ERR_INVALID_ARG, "rg_pub_apply_host: null argument");
    const RgPubLayout l = rg_pub_layout(n_groups, overflow_slots ? overflow_slots : (u32)(n_groups / 256 + 64));
    RgPubSlots sl;
    sl.n = 1;
    sl.slice[0] = reinterpret_cast<const char *>(gathered);
    u32 lost = 0;
    for (u32 r = 0; r < world; r++) {
        for (u64 g8 = 0; g8 < l.Gpad; g8 += 8) rg_pub_apply8(replica, sl, l, r, g8);
        const char *base = sl.slice[0] + (u64)r * l.bytes_per_rank;
        const RgPubHdr *hdr = reinterpret_cast<const RgPubHdr *>(base);
        const RgPubOvf *list = reinterpret_cast<const RgPubOvf *>(base + l.off_list);
        for (u32 k = 0; k < hdr->n_overflow && k < l.cap; k++)
            if (list[k].group < l.G) replica[(u64)r * l.Gpad + list[k].group] += list[k].extra;
        if ((hdr->flags & RG_PUB_LOST) || hdr->n_overflow > l.cap) lost++;
    }
    if (lost_ranks) *lost_ranks = lost;
    return RG_OK;
} RG_ABI_GUARD

