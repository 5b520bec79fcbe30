// rg_engine.h -- what the ABI units (abi_*.hip) share: the engine object, error plumbing, entry discipline, and the internal
// functions one unit provides to another. Nothing here is exported: the library is built with -fvisibility=hidden and only
// the entry points include/raftgroups.h declares are visible (tests/test_abi.py compares the two lists).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <unordered_map>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <vector>

#pragma GCC visibility push(default)
#include "../../include/raftgroups.h"
#pragma GCC visibility pop

#include "rg_group.h"
#include "rg_send.h"
#include "rg_wire.h"
#include "rg_workload.h"

#include "rg_tick_kernels.h"

// live engines of this process PER DEVICE: the Infinity Cache is one per device, and a range of ONE engine only stays
// resident there (k_tick_split) while no other engine's traffic goes through it. Looked at ONCE, by rg_create, when the cache
// policy of the new engine is decided (RG_CACHE_AUTO grants a resident range only to an engine that is alone on its device);
// a live engine's kernel never changes because another engine comes or goes.
#define RG_MAX_DEVICES 64

// ---- error plumbing (abi_state.hip): rg_fail, RG_ABI_GUARD ----
#include "rg_abi_guard.h"

#define RG_HIP(expr)                                                                               \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return rg_fail(e__ == hipErrorOutOfMemory ? RG_ERR_OUT_OF_MEMORY : RG_ERR_NO_DEVICE,   \
                           "%s failed: %s", #expr, hipGetErrorString(e__));                        \
    } while (0)

struct rg_engine;
int rg_mailbox_quiesce(rg_engine *h);            // abi_mirror.hip
int rg_require_hints_resolved(rg_engine *h, const char *who); // abi_tick.hip
// Every entry point that puts work on the engine's stream starts here: select the device and, if the resident mailbox
// kernel is on the stream (rg_mailbox_start), tell it to leave -- stream order would make the call wait for it anyway
// (until its idle timeout), this makes the wait a few microseconds.
#define RG_ENTER(h)                                                                                \
    do {                                                                                           \
        (h)->pub_tick_evt = -1; /* (whatever this call enqueues comes behind the last tick's event) */ \
        RG_HIP(hipSetDevice((h)->cfg.device));                                                     \
        int rc__ = rg_mailbox_quiesce(h);                                                          \
        if (rc__) return rc__;                                                                     \
    } while (0)

// rg_refresh_classes: per block of RG_BLOCK groups (= one workgroup of the lane kernels), the number of slots the block's cfg
// words name: 1 + the highest slot that is present, a voter of either majority, the leader's own, or the transferee.
RG_HD u32 rg_cfg_slots_named(u32 cfg) {
    const u32 tr = RG_CFG_TRANSFEREE(cfg);
    const u32 m = RG_CFG_PRESENT(cfg) | RG_CFG_INCOMING(cfg) | RG_CFG_OUTGOING(cfg) | (1u << RG_CFG_SELF(cfg)) | (tr ? 1u << (tr - 1u) : 0u);
    return 32u - (u32)__builtin_clz(m | 1u);
}
// the smallest body k_tick_classes<P> has for k slots (3, 5, 7 below P; P)
RG_HD u32 rg_class_body(u32 k, u32 P) { return (P > 3 && k <= 3) ? 3u : (P > 5 && k <= 5) ? 5u : (P > 7 && k <= 7) ? 7u : P; }

// ------------------------------------------------------------------------------------------------
// engine object
// ------------------------------------------------------------------------------------------------
struct rg_engine {
    rg_config cfg;
    rg_device_info dev;
    u64 G, stride;
    u32 P;
    hipStream_t stream;
    char *arena;      // state columns
    size_t state_bytes;
    char *ckpt;       // checkpoint copy of the state columns (lazy)
    char *msg_arena;  // device staging for rg_tick(host msgs) / rg_flush (lazy)
    u64 *zero_col;    // [P][stride] zeros, substituted for NULL m_hint / m_rs
    u64 *rhint;       // [P][stride] reject hints after find_conflict_by_term (pre-pass output)
    u64 *d_counts;    // 4 x u64 scratch for reductions
    void *d_scratch;  // G x 8 B scratch for host<->device result shuttles
    size_t col_off[RG_COL_COUNT];
    RgState st;
    RgMsgs staged;    // views into msg_arena
    bool ticked;
    // commit publication: the slot b whose ev_tick[b] rode on the dispatch packet of the LAST dense tick (RG_LAUNCH_TICK), -1 if
    // none or if any entry point has run since (RG_ENTER): rg_publish_commit right behind that tick need not record an event
    int pub_tick_evt;
    u64 tick_launches; // ticks enqueued so far (rg_flush: did a failed flush already change device state?)
    // sparse path (rg_ingest / rg_tick_ingested)
    char *sparse_arena;       // gmark | list | res_list | res_commit | res_out | counters
    u32 *gmark, *counters, *counters_base, *res_out;
    u64 *list, *res_list, *res_commit;
    // single-sync flush of the host mirror: pinned staging for the records, one packed D2H copy of the results
    rg_wire_msg *pin_records; // hipHostMalloc
    u64 pin_records_cap;
    char *d_packed, *pin_packed; // device / pinned host: header + rg_res_rec[]
    u64 packed_cap;           // records
    std::vector<u64> host_res_groups, host_res_commit;
    std::vector<u32> host_res_out;
    bool host_res_valid;      // the vectors hold the results of the last tick (served by rg_ingested_results)
    rg_cell_write *d_cells;   // device staging for rg_write_cells
    u64 d_cells_cap;
    rg_wire_msg *d_records;   // device staging for records
    u64 d_records_cap;
    u32 epoch;
    u64 ingested_upper;       // records accepted for upload since the last sparse tick (>= touched groups)
    u64 last_sparse_n;        // groups of the last rg_tick_ingested (result arrays are valid for them)
    bool out_is_dense;        // RG_COL_OUT was last written by a dense tick
    // send stage (rg_config.max_inflight > 0): Inflights rings, work items
    char *ins_arena;   // meta | head | tail | ring | items | counter
    char *ins_ckpt;    // checkpoint copy of meta | ring (lazy)
    u32 *esz, *esz_ckpt; // entry sizes for RG_SEND_BYTES (rg_log_sizes_enable), u32 [G][esz_w]; checkpoint copy (lazy)
    void *d_recs;      // staging for rg_log_sizes_write / rg_update_state records
    size_t d_recs_cap;
    // resident small-batch path (rg_mailbox_start): request / answer block in pinned host memory, whether the feature is
    // on, whether the host has launched an instance it has not seen leave, the last request number
    RgMbox *mbox;
    bool mbox_on, mbox_running;
    u32 mbox_seq;
    u64 mbox_idle_ticks;
    u64 mbox_served, mbox_launches; // flushes the resident workgroup answered / times it was (re)launched
    size_t ins_state_bytes;
    RgIns ins;
    rg_send_item *send_items;
    u32 *send_counter;
    RgSendCols send_cols;  // work items of a dense stage (peer-major columns)
    bool send_cols_fresh;  // ... hold the last stage's items and the compact list has not been materialised from them
    bool send_last_dense;  // the last stage was a dense one (the columns are its output)
    u64 send_bound;    // upper bound of the last stage's work items (groups it walked x peers)
    std::vector<rg_send_item> host_items; // items of the last stage when rg_flush_send fetched them
    bool host_items_valid;
    char *pin_send;    // pinned host: u32 count | pad | rg_send_item[RG_SEND_SPEC] (small stages: one round trip)
    // size classes (k_tick_classes): derived from RG_COL_CFG, lazily, by the first dense tick after anything wrote the column
    u8 *cls_need;      // device: one byte per block of RG_BLOCK groups (k_block_slots), padded to whole words
    std::vector<u8> cls_host; // its host copy
    bool cls_on;       // some block names fewer slots than the engine has: the dense lane tick runs k_tick_classes
    u32 *cls_order;    // device: one word per workgroup of that kernel, in launch order: block | slots << 28 (RgClasses::order)
    bool cls_stale;    // RG_COL_CFG may have changed since the bytes were derived
    bool cls_off;      // never use them: RG_CFGF_NO_SIZE_CLASSES at rg_create, or the cfg column's device pointer was handed out
    bool nt_msgs;      // dense ticks stream their message columns (non-temporal loads): state + one tick's messages > Infinity Cache
    bool nt_all;       // ... and the state columns, loads and stores: the state ALONE is far beyond the cache
    u64 nt_resident;   // ... except those of the first nt_resident workgroups' groups, which stay in the cache (k_tick_split); 0 = off
    bool counted_live;   // this engine is in g_live_on_device
    bool cls_block_order; // RG_CFGF_CLASS_BLOCK_ORDER
    bool hint_check_due; // device Inflights: a tick that may have raised RG_OUT_HOST_HINT (it carried log terms) ran and nobody has
                         // verified since that every hint was resolved (rg_require_hints_resolved)
    // ... and what makes that check free for the ticks that raised none: the pre-pass of a dense log-term tick ORs 1 into
    // d_hint_raised when it leaves a reject to the host; the word is copied to pinned memory behind the pre-pass (BEFORE the
    // tick kernel) and ev_hint recorded, so the next entry point waits for the pre-pass only, never for the tick
    u32 *d_hint_raised, *pin_hint_raised;
    hipEvent_t ev_hint;
    u32 fused_done;          // ticks the last rg_tick_device_fused applied (rg_fused_ticks_done)
    bool hint_probe_pending; // the last log-term tick went through the probe and nobody has looked at its word yet
    bool send_ready;   // a tick ran since the last rg_send_appends
    u64 stage_max_entries; // limit and flags of the last send stage (any form): rg_resolve_host_hints runs the stage of the
    u32 stage_flags;       // groups that stage skipped (RG_OUT_HOST_HINT) with the same ones
    bool ckpt_send_ready;
    bool ckpt_hint_check_due;
    bool ckpt_any_group_commit;
    bool any_group_commit; // some group's cfg word has RG_CFG_GROUP_COMMIT (tracked on cfg loads)
    // host mirror of RawNode::step (rg_set_peers / rg_step / rg_flush)
    std::vector<u64> peer_ids; // [G][8], 0 = unused
    std::vector<u64> terms;    // [G]
    std::vector<u64> q_mi, q_mc, q_mh, q_mrs, q_mlt; // [P][stride] host queues
    std::vector<u8> q_mf;                      // [G][8]
    std::vector<u64> q_dirty;                  // groups touched since the last flush
    std::vector<rg_wire_msg> q_records;        // flush staging (wire-order records of the dirty groups)
    bool q_any_logterm;                        // some queued message carries Message.log_term
    struct RgQueuedElection { u64 group, old_term; };
    std::vector<RgQueuedElection> q_elections; // rg_local_become_leader calls of the pending flush: group, the term its gate had
    std::vector<u32> host_cfg;                 // host copy of RG_COL_CFG for the mirror (self slots)
    bool host_cfg_valid;
    bool host_mirror;
    struct RgPub *pub; // commit publication across ranks (rg_comm_init), nullptr = single engine
};

// RCCL is bound lazily (the library is ~0.5 GB; single-GPU users never load it). The types come from its header.

#include <rccl/rccl.h>

// Send slices in rotation: the ticks of interval i accumulate into slice i % RG_PUB_SEND while the exchanges of the
// previous intervals still read theirs. Re-use is gated on the HOST (hipEventSynchronize on the exchange that last
// read the slice, three publications back: normally long finished), so the engine's stream carries no cross-stream
// wait -- a barrier packet per tick costs ~5 us of a ~58 us tick (profiles/r02_publish_overhead.txt).
#define RG_PUB_SEND 4
struct RgPub {
    u32 rank, world;
    ncclComm_t comm;          // RCCL transport (nullptr with a custom transport)
    rg_allgather_fn transport;
    void *transport_user;
    RgPubLayout lay;
    u32 ring;                 // publications buffered before the replica is brought up to date
    hipStream_t side;         // the exchange runs here; the engine's stream only records / waits events
    char *send[RG_PUB_SEND];  // this rank's slice under construction (a small ring), bytes_per_rank each
    char *ring_buf;           // [ring][world][bytes_per_rank]
    u64 *replica;             // [world][Gpad]
    u64 *full_send;           // [Gpad] snapshot of the commit column for a full publication
    u32 *d_lost;              // device: some gathered slice asked for a resynchronisation (set by the replica update)
    u32 *pin_lost;            // pinned host: [2] copies of d_lost taken at the last two check points
    hipEvent_t ev_tick[RG_PUB_SEND], ev_done[RG_PUB_SEND], ev_chk[2];
    bool done_pending[RG_PUB_SEND], chk_pending[2];
    u64 n_pub;                // publications so far
    int rode_slot;            // rg_publish_commit: the engine's pub_tick_evt as the call found it (-1: record the event)
    u32 pending;              // ring slots gathered and not yet folded into the replica
    bool in_process;          // one of several ranks of ONE process driven by one thread (rg_comm_init_all / rg_publish_commit_all)
    hipEvent_t ev_read;       // ... in-process transport: this rank's side stream has read every rank's slice of the current publication
    bool local_lost;          // this rank's deltas no longer describe its commit column (restore / column load)
    bool lost_announced;      // ... and a slice carrying RG_PUB_LOST has gone out (the full snapshot follows)
    rg_publish_stats stats;
};

static inline size_t rg_align(size_t x) { return (x + 255) & ~(size_t)255; }

static inline size_t rg_col_elem(int c) {
    if (c == RG_COL_PFLAGS) return 8; // one u64 row per group
    if (c == RG_COL_CFG || c == RG_COL_OUT) return 4;
    if (c == RG_COL_HOST_HINT || c == RG_COL_RUN_COUNT) return 1;
    return 8;
}
static inline bool rg_col_per_slot(int c) { return c <= RG_COL_GID; }
static inline bool rg_col_per_run(int c) { return c == RG_COL_RUN_FIRST || c == RG_COL_RUN_TERM; }

#define RG_STR2(x) #x
#define RG_STR(x) RG_STR2(x)
static inline void *rg_col(rg_engine *h, int c) { return h->arena + h->col_off[c]; }
static inline unsigned rg_grid(u64 n, unsigned per_block) { return (unsigned)((n + per_block - 1) / per_block); }

#define RG_SEND_SPEC 16384 /* work items copied speculatively with their count (512 KB of pinned memory) */
struct RgSendReq { // a tick that runs its send stage in the same launch (rg_tick_send, rg_tick_device_send)
    u64 max_entries;
    u32 flags;
};

// ---- internal functions one unit provides to the others (hidden visibility: not part of the ABI) ----
// abi_state.hip
void rg_drop(rg_engine *h);
// abi_tick.hip
int rg_settle_send(rg_engine *h);
int rg_refresh_classes(rg_engine *h);
int rg_tick_impl(rg_engine *h, const RgMsgs &ms, const RgSendReq *send = nullptr);
int rg_send_check(rg_engine *h, uint32_t flags, const char *who);
int rg_ensure_msg_arena(rg_engine *h);
int rg_tick_host_impl(rg_engine *h, const rg_msgs *m, const RgSendReq *send);
// abi_send.hip
int rg_send_enqueue(rg_engine *h, uint64_t max_entries_per_msg, uint32_t flags, const u64 *list, u64 n, const u32 *n_ptr);
int rg_stage_records(rg_engine *h, const void *recs, size_t bytes);
int rg_send_materialize(rg_engine *h);
int rg_fix_ins_full(rg_engine *h); // k_fix_ins_full over every group, on the engine's stream
// abi_mirror.hip
int rg_ensure_sparse(rg_engine *h);
// abi_publish.hip
int rg_rccl_load();

