/*
 * raftgroups.h -- C ABI of the MI355X multi-raft progress/commit engine.
 *
 * This is the drop-in boundary for ONE hot path of pingcap/raft-rs v0.6.0 (all
 * file:line citations are relative to the reference tree): what a leader does for
 * every MsgAppendResponse -- Raft::handle_append_response (src/raft.rs:1559-1775) ->
 * Progress::{update_committed,maybe_decr_to,maybe_update,become_*} (src/tracker/progress.rs)
 * -> Raft::maybe_commit (src/raft.rs:893-904) -> ProgressTracker::maximal_committed_index
 * (src/tracker.rs:294-298) -> JointConfig/MajorityConfig::committed_index
 * (src/quorum/joint.rs:47-51, src/quorum/majority.rs:70-124) -> RaftLog::maybe_commit
 * (src/raft_log.rs:487-499) -- evaluated for N independent raft groups per call on the GPU.
 *
 * The reference has no FFI: everything is generic Rust (Raft<T: Storage>, src/raft.rs:267).
 * The seam chosen is the ProgressTracker + RaftLog.committed state behind
 * handle_append_response; INTEGRATION.md shows the Rust `extern "C"` block that binds
 * these entry points and where Raft::step would call them.
 *
 * Conventions
 *  - plain C types only; all buffers are caller-owned; the engine keeps no host pointer
 *    after a call returns (async H2D copies are completed or staged before return);
 *  - every call returns RG_OK (0) or a negative rg_status; rg_last_error() has the text;
 *  - one caller thread per handle at a time (mirrors "thread-unsafe" RawNode, src/raw_node.rs:284); different handles may be
 *    driven by different threads of one process concurrently (see "several ranks in ONE process" below);
 *  - a group has up to 8 peer slots; slot s of a group is one Progress (src/tracker/progress.rs:8-56);
 *  - device state is struct-of-arrays: per-slot u64 columns are peer-major [P][stride]
 *    (stride = n_groups rounded up to 256), per-slot flag bytes are one u64 row per group
 *    ([G][8], byte s = slot s), per-group scalars are [G].
 */
#ifndef RAFTGROUPS_H
#define RAFTGROUPS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RG_MAX_SLOTS 8
/* The ABI is SOURCE-compatible across versions, not binary-compatible: rg_config and rg_device_info grow in place (zero-initialised
 * new members keep the old meaning), so a caller is compiled against the header of the library it loads. RG_ABI_VERSION is bumped
 * whenever a struct layout, an enum value or a signature changes; rg_abi_version() returns what the library was built with --
 * compare the two at start-up (raftgroups.hpp and the Python / Rust bindings do). */
#define RG_ABI_VERSION 7u

/* ---- status codes; the negative values mirror src/errors.rs:6-50 where one applies ----
 * Every entry point that returns int returns one of these and leaves the text in rg_last_error() (per thread). Nothing unwinds
 * through this boundary: a C++ exception inside the library -- a host allocation that fails -- is caught at the entry point
 * (RG_ERR_OUT_OF_MEMORY; anything else RG_ERR_STATE), the way a failed device allocation is reported, never thrown or fatal. */
typedef enum {
    RG_OK = 0,
    RG_ERR_INVALID_ARG = -1,
    RG_ERR_NO_DEVICE = -2,       /* no HIP device / HIP runtime error: there is NO CPU fallback */
    RG_ERR_OUT_OF_MEMORY = -3,
    RG_ERR_STEP_LOCAL_MSG = -4,  /* Error::StepLocalMsg   (src/errors.rs:19, src/raw_node.rs:404-406) */
    RG_ERR_STEP_PEER_NOT_FOUND = -5, /* Error::StepPeerNotFound (src/errors.rs:22, src/raw_node.rs:407-410) */
    RG_ERR_SLOT_BUSY = -6,       /* a second message for the same (group, peer) before rg_tick */
    RG_ERR_HIGHER_TERM = -7,     /* m.term > term: the host must step down (src/raft.rs:1284-1348) */
    RG_ERR_STATE = -8,           /* call sequence error (e.g. results before any tick) */
    RG_ERR_NOT_ON_PATH = -9,     /* rg_step_bytes: a well-formed message of a type this path does not handle (MsgAppend,
                                    votes, ...): the host's own Raft::step takes it */
    RG_ERR_HOST_HINT = -10       /* rg_tick_device_fused: a tick of the call raised RG_OUT_HOST_HINT; the call stopped BEHIND that
                                    tick (state, RG_COL_OUT and RG_COL_HOST_HINT are that tick's; rg_fused_ticks_done) */
} rg_status;

/* ---- Progress flag byte (one per slot; src/tracker/progress.rs:8-56, src/tracker/state.rs:22-29) ---- */
#define RG_PF_STATE_MASK 0x03u /* 0 Probe, 1 Replicate, 2 Snapshot */
#define RG_STATE_PROBE 0u
#define RG_STATE_REPLICATE 1u
#define RG_STATE_SNAPSHOT 2u
#define RG_PF_PAUSED 0x04u        /* Progress.paused */
#define RG_PF_RECENT_ACTIVE 0x08u /* Progress.recent_active */
#define RG_PF_PENDING_CONF 0x20u   /* host-owned, meaningful on the LEADER'S OWN slot only: Raft::has_pending_conf()
                                     (src/raft.rs:2684-2686), read by rg_send_appends(RG_SEND_SKIP_BCAST_COMMIT) */
#define RG_PF_INS_FULL 0x10u      /* engine-owned, only with rg_config.max_inflight > 0: Inflights::full() of the
                                     device-side ring (rg_send_appends maintains it; OR-ed with RG_MF_INS_FULL) */
#define RG_PF_PEND_SNAP 0x40u     /* engine-owned, exact: this Progress's pending_snapshot (RG_COL_PEND_SNAP) is non-zero */
#define RG_PF_PEND_RS 0x80u       /* engine-owned, exact: ... its pending_request_snapshot (RG_COL_PEND_RS) is non-zero.
                                     Both fields are zero almost always; the two bits let the kernels skip the cold
                                     columns -- Progress::reset's stores on an election, reset_state's, the
                                     pending_request_snapshot tests of the heartbeat response and of the send decision --
                                     without reading them. Whatever a caller passes for them is ignored: they are
                                     re-derived when the flag column or one of the two columns is loaded and when
                                     rg_write_cells touches the cell; reads report them. */
#define RG_PF_PENDING (RG_PF_PEND_SNAP | RG_PF_PEND_RS)

/* ---- message flag byte (one per slot per tick) ---- */
#define RG_MF_VALID 0x01u    /* an AppendResponse from this peer (on the self slot: on_persist_entries(m_index), src/raft.rs:994-1016) */
#define RG_MF_REJECT 0x02u   /* Message.reject */
#define RG_MF_HAS_RS 0x04u   /* Message.request_snapshot != 0 (value in m_rs) */
#define RG_MF_INS_FULL 0x08u /* host's Inflights::full() for this peer (Progress::is_paused, progress.rs:210-216);
                                not needed when the engine holds the Inflights (rg_config.max_inflight > 0) */
#define RG_MF_SENT 0x10u     /* the host sent a MsgAppend up to last_index since the previous tick:
                                apply Progress::update_state(last) first (progress.rs:231-243, src/raft.rs:726-729) */
#define RG_MF_APPEND 0x20u   /* self slot only: leader appended entries, new last_index in m_commit
                                (Raft::append_entry, src/raft.rs:976-991) */
#define RG_MF_HAS_LOGTERM 0x80u /* a reject with Message.log_term > 0 (value in m_logterm): the engine runs
                                RaftLog::find_conflict_by_term(reject_hint, log_term) (src/raft_log.rs:209-235,
                                src/raft.rs:1562,1657-1660) against the group's term-run table (RG_COL_RUN_*) */
#define RG_MF_BECOME_LEADER 0x02u /* leader's OWN slot only (where RG_MF_REJECT has no meaning): this node has just won the
                                group's election at term m_hint[self slot] -- Raft::reset(term) + Raft::become_leader
                                (src/raft.rs:942-971,1151-1202) run BEFORE every other event of the tick: every Progress is
                                reset (Progress::reset(last_index + 1), progress.rs:82-92), the leader's own keeps
                                matched = persisted, takes committed_index = committed and becomes Replicate, a leader
                                transfer is aborted, the new leader's empty entry is appended (last_index += 1; the new
                                term's index range starts there: RG_COL_TERM_LO = RG_COL_TERM_HI = last_index), the old
                                leader's range becomes one more run of RG_COL_RUN_* and RG_COL_CUR_TERM = the new term.
                                Result: RG_OUT_BECAME_LEADER | RG_OUT_APPENDED (the bcast_append that follows,
                                raft.rs:2190-2191). The term must
                                be above RG_COL_CUR_TERM, else RG_OUT_FAULT and the event is ignored; matched != last_index
                                on the own slot (raft.rs:1170 asserts persisted == last_index) raises RG_OUT_FAULT too. */
#define RG_MF_HEARTBEAT 0x40u /* a MsgHeartbeatResponse from this peer (m_commit = Message.commit), handled as
                                Raft::handle_heartbeat_response (src/raft.rs:1777-1803); exclusive with RG_MF_VALID.
                                Result bits: RG_OUT_SEND_APPEND(slot) = send_append (matched < last_index or a
                                pending snapshot request), RG_OUT_FREE_TO(slot) = ins.free_first_one() */

/* ---- per-group configuration word ---- */
#define RG_CFG_INCOMING(c) ((uint32_t)(c) & 0xffu)        /* voters.incoming as a slot bitmask (tracker.rs:37-40) */
#define RG_CFG_OUTGOING(c) (((uint32_t)(c) >> 8) & 0xffu) /* voters.outgoing; 0 = not joint */
#define RG_CFG_SELF(c) (((uint32_t)(c) >> 16) & 0x7u)     /* slot of the leader itself */
#define RG_CFG_GROUP_COMMIT 0x00080000u                   /* ProgressTracker.group_commit (tracker.rs:207) */
#define RG_CFG_TRANSFEREE(c) (((uint32_t)(c) >> 20) & 0xfu) /* lead_transferee slot + 1, 0 = none */
#define RG_CFG_PRESENT(c) (((uint32_t)(c) >> 24) & 0xffu) /* slots that have a Progress (voters + learners) */
#define RG_CFG_MAKE(incoming, outgoing, self_slot, group_commit, transferee_plus1, present)            \
    (((uint32_t)(incoming)&0xffu) | (((uint32_t)(outgoing)&0xffu) << 8) |                              \
     (((uint32_t)(self_slot)&0x7u) << 16) | ((group_commit) ? RG_CFG_GROUP_COMMIT : 0u) |              \
     (((uint32_t)(transferee_plus1)&0xfu) << 20) | (((uint32_t)(present)&0xffu) << 24))

/* ---- per-group result word written by every tick ---- */
#define RG_OUT_CHANGED 0x1u     /* some maybe_commit() returned true (src/raft.rs:1745): host runs bcast_append if should_bcast_commit() */
#define RG_OUT_FAULT 0x2u       /* a precondition of the path was violated (where the reference panics or input is malformed) */
#define RG_OUT_TIMEOUT_NOW 0x4u /* send_timeout_now(transferee) (src/raft.rs:1764-1774) */
#define RG_OUT_APPENDED 0x8u    /* the leader's log grew in this tick (RG_MF_APPEND): the bcast_append that follows a proposal is due (src/raft.rs:2049-2053) */
#define RG_OUT_BECAME_LEADER 0x10u /* an RG_MF_BECOME_LEADER event was applied in this tick (Raft::become_leader): every Progress was reset, so the send stage empties the group's Inflights before anything else */
#define RG_OUT_HOST_HINT 0x20u /* a reject with RG_MF_HAS_LOGTERM was NOT applied: RaftLog::find_conflict_by_term (src/raft_log.rs:209-235)
                                 needed terms of log entries the device's bounded term-run table no longer holds (RG_COL_RUN_*:
                                 more than RG_TERM_RUNS older terms since the last snapshot, and the walk reaches below them). The
                                 peer's recent_active and committed_index are updated (src/raft.rs:1674-1677: they do not depend on
                                 the hint), maybe_decr_to and everything behind it is left to the host: it resolves the hint
                                 against its own log and steps the same reject again with RG_MF_HAS_LOGTERM clear (log_term = 0)
                                 and WITHOUT RG_MF_SENT (a SENT event of the slot was applied). Which slots: RG_COL_HOST_HINT /
                                 rg_host_hints. A reject touches only its own peer's Progress and never `matched`, so taking it
                                 after the other messages of the tick changes no other result.
                                 Engines with device Inflights (max_inflight > 0): the re-step form is NOT available there -- the
                                 reference sends the deferred reject's MsgAppend before the group's other sends of the step, so the
                                 group's send requests wait for rg_resolve_host_hints, which serves them; the tick's Inflights
                                 effects (free_to, free_first_one, window resets) are applied by the stage either way. Until every
                                 flagged (group, slot) has been answered, every call that would start the next step (a tick, a
                                 flush, rg_progress_events ...) fails with RG_ERR_STATE and changes nothing. */
#define RG_OUT_SEND_APPEND(o) (((uint32_t)(o) >> 8) & 0xffu) /* per slot: send_append(from) (raft.rs:1719, :1750) */
#define RG_OUT_SEND_MORE(o) (((uint32_t)(o) >> 16) & 0xffu)  /* per slot: the maybe_send_append loop (raft.rs:1761) */
#define RG_OUT_FREE_TO(o) (((uint32_t)(o) >> 24) & 0xffu)    /* per slot: ins.free_to(m.index) (raft.rs:1742) */

/* ---- columns (for rg_load_column / rg_read_column / rg_column_ptr) ---- */
typedef enum {
    RG_COL_MATCH = 0,     /* u64 [P][stride]  Progress.matched */
    RG_COL_NEXT = 1,      /* u64 [P][stride]  Progress.next_idx */
    RG_COL_PR_COMMIT = 2, /* u64 [P][stride]  Progress.committed_index */
    RG_COL_PEND_SNAP = 3, /* u64 [P][stride]  Progress.pending_snapshot */
    RG_COL_PEND_RS = 4,   /* u64 [P][stride]  Progress.pending_request_snapshot */
    RG_COL_GID = 5,       /* u64 [P][stride]  Progress.commit_group_id */
    RG_COL_PFLAGS = 6,    /* u8  [G][8]       RG_PF_* */
    RG_COL_COMMIT = 7,    /* u64 [G]          RaftLog.committed */
    RG_COL_TERM_LO = 8,   /* u64 [G]          first index whose term == current term */
    RG_COL_TERM_HI = 9,   /* u64 [G]          last_index (last index whose term == current term) */
    RG_COL_CFG = 10,      /* u32 [G]          RG_CFG_* */
    RG_COL_OUT = 11,      /* u32 [G]          RG_OUT_* of the last tick */
    /* compact log-term table for find_conflict_by_term: the leader's log is a dummy entry
     * (index, term) = (first_index - 1, snapshot term), then up to RG_TERM_RUNS runs of equal-term entries of
     * OLDER terms (run k covers [run_first[k], run_first[k+1]), the last one up to term_lo - 1; unused runs have
     * run_first = 0, used runs come first in ascending order), then the entries [term_lo, term_hi] of the leader's
     * own term RG_COL_CUR_TERM (this last run grows with every RG_MF_APPEND without touching the table; an
     * RG_MF_BECOME_LEADER event pushes it into the table). The table is BOUNDED, the reference's log is not: when all
     * RG_TERM_RUNS runs are in use a push drops the OLDEST run, and the table then starts above the dummy entry. The
     * terms of the entries in between -- (dummy_index, run_first[0]), or (dummy_index, term_lo) with an empty table --
     * are not on the device; a find_conflict_by_term walk whose answer depends on one of them is not answered but handed
     * back (RG_OUT_HOST_HINT), so every hint the engine DOES apply is the reference's. A host that loads a table
     * therefore loads a contiguous one (run_first[0] = dummy_index + 1) or accepts host hints for the gap.
     * Only read for rejects with RG_MF_HAS_LOGTERM and by RG_MF_BECOME_LEADER. */
    RG_COL_RUN_FIRST = 12,   /* u64 [RG_TERM_RUNS][stride] */
    RG_COL_RUN_TERM = 13,    /* u64 [RG_TERM_RUNS][stride] */
    RG_COL_DUMMY_INDEX = 14, /* u64 [G] */
    RG_COL_DUMMY_TERM = 15,  /* u64 [G] */
    RG_COL_CUR_TERM = 16,    /* u64 [G] the leader's term (Raft.term) */
    RG_COL_HOST_HINT = 17,   /* u8 [G] engine-owned, read-only for the host: bit s = slot s's reject of the last tick was left to
                                the host (RG_OUT_HOST_HINT). Meaningful only for groups whose RG_COL_OUT word of that tick has the
                                bit; other bytes are stale. rg_host_hints gathers the flagged groups. */
    RG_COL_RUN_COUNT = 18,   /* u8 [G] engine-owned, read-only for the host: how many runs of RG_COL_RUN_FIRST / _TERM are in use
                                (derived when RG_COL_RUN_FIRST is loaded, kept by the elections): what lets an election file
                                its run without reading the table */
    RG_COL_COUNT = 19
} rg_column;
#define RG_TERM_RUNS 8

/* ---- a tick's messages, struct-of-arrays, HOST or DEVICE memory (see rg_tick / rg_tick_device) ---- */
typedef struct {
    const uint64_t *m_index;  /* [P][stride] Message.index (self slot: persisted index) */
    const uint64_t *m_commit; /* [P][stride] Message.commit (self slot with RG_MF_APPEND: new last_index) */
    const uint64_t *m_hint;   /* [P][stride] Message.reject_hint, after find_conflict_by_term when log_term>0 (raft.rs:1562,1657-1660); read only for rejects; may be NULL if no rejects */
    const uint64_t *m_rs;     /* [P][stride] Message.request_snapshot; read only when RG_MF_HAS_RS; may be NULL */
    const uint8_t *m_flags;   /* [G][8] RG_MF_* */
    const uint64_t *m_logterm; /* [P][stride] Message.log_term; read only when RG_MF_HAS_LOGTERM; may be NULL */
} rg_msgs;

typedef struct rg_engine rg_engine;

typedef struct {
    uint64_t n_groups; /* groups held by THIS engine (= this rank's shard) */
    uint32_t n_slots;  /* peer slots per group, 1..8 */
    int32_t device;    /* HIP device ordinal */
    uint32_t variant;  /* kernel variant: 0 = default (RG_VARIANT_*) */
    uint32_t max_inflight; /* 0 = Inflights stay with the host (RG_MF_INS_FULL in, RG_OUT_FREE_TO out);
                              1..65535 = Config::max_inflight_msgs (src/config.rs:112): one ring of that many u64
                              per Progress lives in HBM and rg_send_appends runs the send decision on the device */
    /* ---- how the dense ticks go through the 256 MB Infinity Cache (all zero = the engine decides; zero-initialised
     *      rg_config structs of older callers keep meaning what they meant). Decided ONCE, here: nothing outside this struct
     *      -- no environment variable, no other engine created or destroyed later -- changes which kernel a live engine
     *      runs; rg_get_device_info reports the decision. ---- */
    uint32_t cache_policy;          /* RG_CACHE_* */
    uint32_t flags;                 /* RG_CFGF_* */
    uint64_t cache_resident_groups; /* RG_CACHE_RESIDENT: the state of the first this-many groups (rounded down to whole
                                       workgroups of 64) stays in the cache, the rest is streamed; 0 = the engine's own
                                       sizing (176 MB of state). Ignored by the other policies. */
} rg_config;

/* rg_config.cache_policy. What a dense tick re-reads on every launch is its STATE (24 P + 40 bytes per group); the message
 * columns (16 P + 8) are read once. Measured windows (profiles/r04_nt_state.txt, r04_resident.txt at 5 slots;
 * profiles/r05_cache_policy_sweep.txt at 3 and 7): */
#define RG_CACHE_AUTO 0u        /* by footprint: PLAIN while state + one tick of messages fit the cache; STREAM_MSGS beyond;
                                   RESIDENT where 1.25 x cache < state <= 2.5 x cache and no other engine lives on the device
                                   at rg_create (the cache is one per device); else STREAM_ALL where state > 1.5 x cache and
                                   the shard holds at most 13 M groups; engines with device Inflights: PLAIN / STREAM_MSGS
                                   (STREAM_ALL on request, RESIDENT never) */
#define RG_CACHE_PLAIN 1u       /* every access allocates in the cache */
#define RG_CACHE_STREAM_MSGS 2u /* the read-once message columns are streamed past it (non-temporal loads) */
#define RG_CACHE_STREAM_ALL 3u  /* ... and the state columns, loads and stores: nothing of the launch is allocated (with device
                                   Inflights: meant for the one-launch form rg_tick_device_send; a separate rg_send_appends
                                   re-reads what the tick has just streamed out) */
#define RG_CACHE_RESIDENT 4u    /* STREAM_ALL except for a leading range of groups whose state stays resident (one launch,
                                   two bodies: k_tick_split); needs max_inflight = 0, no group commit, the lane variant --
                                   where those do not hold the launch falls back to STREAM_ALL and the report says so */
/* rg_config.flags */
#define RG_CFGF_NO_SIZE_CLASSES 0x1u   /* a shard placed by replica-set size class runs the plain kernel (rg_size_classes
                                          reports 0 ranges): the A/B switch of the class-placed layout */
#define RG_CFGF_CLASS_BLOCK_ORDER 0x2u /* k_tick_classes launches its workgroups in block order instead of dealing the
                                          classes out proportionally (measurement) */
#define RG_CFGF_IX64 0x4u              /* 64-bit cell offsets on an engine small enough for 32-bit ones: the instantiations
                                          only engines beyond 4 GiB per column reach otherwise (tests run both widths) */

#define RG_VARIANT_DEFAULT 0u
#define RG_VARIANT_LANE 1u /* one lane per group, columns straight into registers */
#define RG_VARIANT_LDS 2u  /* one wave per 128-group batch, peer columns staged through LDS */
#define RG_VARIANT_LDS_DMA 4u /* RG_VARIANT_LDS with the stage-in done by gfx950's LDS-DMA (global_load_lds_dwordx4: global
                                 memory -> LDS without passing through VGPRs). A measured comparison point, like LDS. */
#define RG_VARIANT_COMPACT 5u /* the lane kernel in 256-thread workgroups that, after the loads, gather the groups about to
                                 leave the steady path (an election, a reject, a Probe / Snapshot transition, a full window)
                                 into ONE wave per workgroup -- the loaded registers travel through LDS -- so that the other
                                 waves execute the steady path only. For streams with leader-term rollover (BASELINE config
                                 5), where the lane kernel is instruction-bound; same results, a few per cent slower than
                                 RG_VARIANT_LANE on a steady stream. */
#define RG_VARIANT_COOP 3u /* rg_recompute / rg_maximal_committed_index only: 8 lanes per group, one peer per lane,
                              wave-level rank-select of the quorum index with cross-lane shuffles (ticks run the
                              lane kernel). A measured comparison point, not the default. */

/* ---- lifecycle ---- */
const char *rg_version(void);
uint32_t rg_abi_version(void); /* RG_ABI_VERSION of the build */
const char *rg_last_error(void);
int rg_device_count(void);
int rg_create(const rg_config *cfg, rg_engine **out);
void rg_destroy(rg_engine *h);
uint64_t rg_stride(const rg_engine *h); /* column stride in elements */
/* What hipGetDeviceProperties reports for the engine's device (queried once in rg_create, which refuses
 * devices other than gfx950: the library carries CDNA4 code objects only). */
typedef struct {
    char arch[32];            /* gcnArchName up to the first ':' ("gfx950") */
    uint32_t compute_units;   /* multiProcessorCount (256 on MI355X) */
    uint32_t wavefront;       /* warpSize (64) */
    uint64_t lds_per_workgroup; /* sharedMemPerBlock, bytes */
    uint64_t hbm_bytes;       /* totalGlobalMem */
    uint64_t l2_bytes;        /* l2CacheSize (one XCD's L2) */
    uint64_t engine_bytes;    /* device memory this engine allocated in rg_create */
    /* the cache policy rg_create settled on (never RG_CACHE_AUTO) and why */
    uint32_t cache_policy;      /* RG_CACHE_PLAIN .. RG_CACHE_RESIDENT */
    uint32_t engines_on_device; /* live engines of this process on the engine's device when it was created, itself included
                                   (AUTO grants a resident range only to an engine that is alone) */
    uint64_t resident_groups;   /* RG_CACHE_RESIDENT: groups whose state stays in the cache (a multiple of 64), else 0 */
    /* the dense tick kernel of the LAST rg_tick / rg_tick_device(_send) launch (0 before the first): what actually ran */
    uint32_t last_tick_kernel;  /* RG_KERNEL_* */
    uint32_t last_tick_streaming; /* 0 = plain accesses, 1 = message columns streamed, 2 = state columns too
                                     (RG_KERNEL_SPLIT: 2 beyond the resident range, 1 inside) */
    /* the cache the policy is sized against: the device's memory-side (level-3) cache as the HSA runtime enumerates it
     * (hsa_agent_iterate_caches; the agent matched to the HIP device by PCI address) -- 256 MiB on MI355X. The policy's windows
     * (1.25 x / 1.5 x / 2.5 x this size, the 13 M-group bound, 11/16 of it for a resident range) are MEASURED on MI355X
     * (profiles/r04_nt_state.txt, r04_resident.txt, r05_cache_policy_sweep.txt) and scale with the size the device reports. */
    uint64_t infinity_cache_bytes;
    uint32_t infinity_cache_queried; /* 1 = asked of the device; 0 = the query failed and the MI355X constant stands in */
    /* the cell-offset width of that last launch: 32 while every cell of a column lies within 4 GiB of the column's start
     * (max(n_slots, RG_TERM_RUNS) x stride x 8 < 4 GiB: up to 67 108 608 groups), 64 beyond that or under RG_CFGF_IX64; 0 before
     * the first tick */
    uint32_t last_tick_offset_bits;
} rg_device_info;
#define RG_KERNEL_NONE 0u
#define RG_KERNEL_LANE 1u      /* k_tick_lane: one lane per group */
#define RG_KERNEL_CLASSES 2u   /* k_tick_classes: the lane kernel, workgroups instantiated per replica-set size class */
#define RG_KERNEL_SPLIT 3u     /* k_tick_split: the lane kernel with a cache-resident leading range */
#define RG_KERNEL_LDS 4u       /* k_tick_lds (RG_VARIANT_LDS / _LDS_DMA) */
#define RG_KERNEL_COMPACT 5u   /* k_tick_compact (RG_VARIANT_COMPACT) */
#define RG_KERNEL_TICK_SEND 6u /* k_tick_send: the tick and its send stage in one launch */
int rg_get_device_info(const rg_engine *h, rg_device_info *info);
/* Run all engine work on this hipStream_t (e.g. torch.cuda.current_stream().cuda_stream). */
int rg_set_stream(rg_engine *h, void *hip_stream);
int rg_sync(rg_engine *h);

/* ---- state in/out (parity checks, checkpoint/restore; ProgressTracker::get / Status) ---- */
/* Copy a whole column host->device / device->host. `bytes` must equal rg_column_bytes(). */
uint64_t rg_column_bytes(const rg_engine *h, int column);
int rg_load_column(rg_engine *h, int column, const void *host_src, uint64_t bytes);
int rg_read_column(rg_engine *h, int column, void *host_dst, uint64_t bytes);
/* Device address of a column (for zero-copy READERS, e.g. an RCCL all-gather of RG_COL_COMMIT). Writing through it bypasses
 * what rg_load_column derives on the way in -- the engine-owned flag bits, RG_COL_RUN_COUNT from RG_COL_RUN_FIRST, the size
 * classes from RG_COL_CFG (handing out that column's pointer switches k_tick_classes off for good) -- so state goes in
 * through rg_load_column / rg_write_cells. */
void *rg_column_ptr(rg_engine *h, int column);
/* Snapshot / restore the complete device state inside the engine (bench replays, rollbacks). */
int rg_checkpoint(rg_engine *h);
int rg_restore(rg_engine *h);

/* Sparse read-back of whole groups -- what Status / ProgressTracker::get hand out for ONE raft
 * (src/status.rs:25-52, src/tracker.rs:261-287) without copying 40 MB columns: one record per requested group. */
typedef struct {
    uint64_t group;
    uint64_t commit, term_lo, last_index; /* RaftLog.committed, first index of the leader's term, last_index */
    uint32_t cfg, out;                    /* RG_CFG_* word, RG_OUT_* word of the last tick */
    uint64_t match[RG_MAX_SLOTS], next[RG_MAX_SLOTS], pr_commit[RG_MAX_SLOTS];
    uint64_t pend_snap[RG_MAX_SLOTS], pend_rs[RG_MAX_SLOTS];
    uint8_t pflags[RG_MAX_SLOTS];         /* RG_PF_* per slot */
    uint8_t inflights[RG_MAX_SLOTS];      /* Inflights.count per slot, saturated at 255 (0 without device Inflights) */
} rg_group_status;
int rg_read_groups(rg_engine *h, const uint64_t *groups, uint64_t n, rg_group_status *host_out);

/* Sparse overwrite of Progress cells between ticks -- the send path and the other host-side
 * writers co-own these cells (Progress::update_state/become_snapshot/reset, heartbeat response,
 * unreachable: src/tracker/progress.rs:82-121,231-243; src/raft.rs:664-712,1791-1798,1945-1947). */
typedef struct {
    uint64_t group;
    uint32_t slot;
    uint32_t field_mask; /* bit i = write column i (RG_COL_MATCH..RG_COL_PFLAGS) */
    uint64_t match, next, pr_commit, pend_snap, pend_rs, gid;
    uint8_t pflags;
    uint8_t pad[7];
} rg_cell_write;
int rg_write_cells(rg_engine *h, const rg_cell_write *cells, uint64_t n);

/* Membership of ONE group (ProgressTracker::apply_conf, src/tracker.rs:380-397, after
 * Changer::{simple,enter_joint,leave_joint}, src/confchange/changer.rs:66-157): rewrite the group's
 * configuration word (RG_CFG_MAKE). New peers' Progress cells (Progress::new(last_index+1), recent_active)
 * are written with rg_write_cells; follow with rg_recompute for post_conf_change's maybe_commit (raft.rs:2630). */
int rg_set_config(rg_engine *h, uint64_t group, uint32_t cfg_word);

/* RawNode::report_unreachable / report_snapshot (src/raw_node.rs:692-709): the two LOCAL messages that write a Progress,
 * applied to the device cell in place (no read-back), in array order -- the records of one (group, slot) must be adjacent:
 *   RG_EV_UNREACHABLE      handle_unreachable (src/raft.rs:1931-1954): a Replicate peer becomes Probe (become_probe:
 *                          next = matched + 1, paused = false, pending_snapshot = 0, ins.reset()); any other state: nothing
 *   RG_EV_SNAPSHOT_FINISH  handle_snapshot_status (src/raft.rs:1891-1929), reject = false: a Snapshot peer becomes Probe
 *                          with next = max(matched + 1, pending_snapshot + 1), then pause() and pending_request_snapshot = 0
 *   RG_EV_SNAPSHOT_FAILURE ... reject = true: snapshot_failure() first, so next = matched + 1
 * A peer that is not in Snapshot ignores both snapshot events; a slot without a Progress (RG_CFG_PRESENT) or beyond the
 * engine's, and a group beyond the shard, are ignored ("no progress available"). With device
 * Inflights the window is reset whenever the state changes. ORDER with device Inflights: call order is event order, so run
 * the last tick's send stage (rg_send_appends) BEFORE these calls -- if it has not run, the tick's Inflights effects
 * (free_to, free_first_one, left Replicate) are applied first and its send requests are dropped, exactly as the next tick
 * would do with a skipped stage; the event never lands between a tick and that tick's own effects.
 * Synchronises (a control-path call, like rg_write_cells). */
typedef struct {
    uint64_t group;
    uint32_t slot;
    uint32_t kind; /* RG_EV_* */
} rg_progress_event;
#define RG_EV_UNREACHABLE 1u
#define RG_EV_SNAPSHOT_FINISH 2u
#define RG_EV_SNAPSHOT_FAILURE 3u
int rg_progress_events(rg_engine *h, const rg_progress_event *events, uint64_t n);
/* One kind of event for a whole shard at once -- what a lost connection to one store is: MsgUnreachable for that store's
 * peer in every group that has one. `host_slot_plus1`: u8 [G], 0 = nothing for the group, s + 1 = the event goes to its
 * slot s (values beyond the engine's slots are ignored). 1 B per group over PCIe instead of a 16 B record: 1 M groups in
 * 77 us instead of 5.5 ms (profiles/r03_progress_events.txt). Same arithmetic as rg_progress_events. Synchronises. */
int rg_progress_event_dense(rg_engine *h, uint32_t kind, const uint8_t *host_slot_plus1);

/* Size classes: replica sets of different sizes in one shard (BASELINE config 5: 3 / 5 / 7 peers). Peers a group does not
 * have still occupy cells of the engine's P slots; where the host places groups of one size in CONTIGUOUS ranges, the dense
 * tick skips the absent slots of a whole range -- their loads, stores and instructions -- in ONE launch (k_tick_classes).
 * Nothing to declare: the engine derives it from RG_COL_CFG by itself (per block of 64 groups, the highest slot any cfg word
 * names, rounded up to 3 / 5 / 7 / P slots: one byte per block, so a conf change that grows one group costs its own block
 * the shortcut and nobody else), re-derives it after anything wrote the column (rg_load_column, rg_set_config beyond its
 * block's class, rg_restore, rg_workload_init) and runs the plain kernel wherever the layout does not qualify (every block
 * names every slot -- sizes interleaved --, group commit on, the LDS / compact variants, an engine beyond 32-bit cell
 * offsets, RG_COL_CFG's device pointer handed out through rg_column_ptr). Results never depend on it.
 * This call reports what the next dense tick will use, as ranges of equal blocks: *n = number of ranges (0 = the plain
 * kernel; it may exceed cap), out[k] for k < cap. Re-deriving synchronises, so it cannot happen inside a stream capture: a
 * tick captured into a hipGraph while RG_COL_CFG has changed since the last derivation runs the plain kernel -- call this
 * (or run one tick) before the capture begins. */
typedef struct {
    uint64_t first_group, n_groups;
    uint32_t n_slots; /* slots the groups of the range use at most */
    uint32_t reserved;
} rg_size_class;
int rg_size_classes(rg_engine *h, rg_size_class *out, uint32_t cap, uint32_t *n);

/* ---- placement: getting a shard INTO that layout ----
 * Membership is the host's to change at any time (ProgressTracker::apply_conf, src/tracker.rs:380-397 over the voter sets of
 * src/tracker.rs:37-92), and groups arrive in whatever order the host learns of them: a shard whose replica-set sizes are
 * interleaved runs the plain kernel and moves the cells of absent peers with everything else (config 5 interleaved: 1.47 x the
 * algorithmic bytes, 0.41 of the roofline against 0.55 placed). Two calls fix the layout:
 *   rg_plan_placement   pure host arithmetic (no engine, no device): from the groups' cfg words, the permutation that places
 *                       them by size class -- the bodies k_tick_classes has: 3, 5, 7 slots below n_slots, and n_slots --
 *                       ascending, STABLE inside a class; perm[i] = the current position of the group that goes to position i.
 *                       `classes` (capacity `cap`, may be NULL with cap = 0) receives the ranges the engine will derive from
 *                       the permuted column, *n_classes their number (may exceed cap); a block of 64 groups that straddles a
 *                       boundary belongs to the larger class.
 *   rg_permute_groups   the device gather: EVERY column of the engine follows the permutation -- Progress cells, flag rows,
 *                       log ranges, term-run tables, result words and host-hint bytes, the Inflights windows and rings, the
 *                       entry-size windows -- and so do the host mirror's peer-id / term tables. Afterwards group i of every call
 *                       is the group that was at host_perm[i]; the next dense tick derives the size classes of the new layout.
 *                       Control path (synchronises; needs the state's size in free device memory once more while it runs).
 *                       Before the call: flush queued steps (RG_ERR_SLOT_BUSY), tick ingested records (RG_ERR_STATE), answer
 *                       host hints, and fetch the last stage's work items (their `group` fields name the old positions).
 *                       A checkpoint taken earlier is dropped (it images the old placement); with commit publication active
 *                       the rank's slice is marked lost, so the next check point publishes a full snapshot (as rg_restore).
 * Re-placing pays when conf changes have accumulated: a group that grew past its range's class costs only its own block of
 * 64 the shortcut (see above), so a host re-plans when rg_size_classes shows the ranges fraying, not per change. */
int rg_plan_placement(const uint32_t *cfg_words, uint64_t n_groups, uint32_t n_slots, uint64_t *perm, rg_size_class *classes,
                      uint32_t cap, uint32_t *n_classes);
int rg_permute_groups(rg_engine *h, const uint64_t *host_perm);

/* ---- the hot path ---- */
/* One tick: for every group, apply its <=1 message per slot in slot order exactly as
 * handle_append_response would (commit re-evaluated after every accepted ack), update state in
 * place and write RG_COL_OUT. Host-buffer form: copies the message columns H2D, then launches. */
int rg_tick(rg_engine *h, const rg_msgs *host_msgs);
/* Same, message columns already in device memory (layout identical). Asynchronous. */
int rg_tick_device(rg_engine *h, const rg_msgs *dev_msgs);
/* Temporal fusion: n_ticks (1..8) consecutive ticks in ONE launch. `dev_msgs` is a HOST array of n_ticks
 * rg_msgs whose pointers are DEVICE memory, in tick order. Each group's state stays in registers across the
 * ticks and is written back once, so per tick only the message columns plus 1/n of the state traffic move.
 * Results are bit-identical to n_ticks calls of rg_tick_device. `dev_out_t` (u32 [n_ticks][G], device,
 * required) receives every tick's RG_OUT_* word, `dev_commit_t` (u64 [n_ticks][G], device, may be NULL) the
 * commit index after every tick; RG_COL_OUT / RG_COL_COMMIT hold the last tick's. RG_MF_BECOME_LEADER events are applied as
 * in a single-tick launch. A tick whose m_logterm is not NULL needs find_conflict_by_term against the log as it stands
 * BEFORE that tick, so the library runs it as a single-tick launch (behind its pre-pass) between the fused launches of the
 * ticks around it -- same results, same arrays. With commit publication active (rg_comm_init) the call's total advance of
 * every group lands in its publication byte, exactly as n_ticks single launches without a publication in between would
 * leave it.
 * Exact or loud: a log-term tick that raises RG_OUT_HOST_HINT (a reject whose find_conflict_by_term needs terms the bounded
 * term-run table no longer holds) ENDS the call -- the reference applies that reject before anything later
 * (src/raft_log.rs:209-235 -> src/raft.rs:1657-1660), so the ticks behind it are not run. The call returns RG_ERR_HOST_HINT;
 * rg_fused_ticks_done reports how many ticks were applied (the hinted tick included: everything of it that does not depend on
 * the hint is in place, as after a single rg_tick_device), their rows of dev_out_t / dev_commit_t are written, and RG_COL_OUT /
 * RG_COL_HOST_HINT are the hinted tick's, so rg_host_hints / rg_resolve_host_hints (or re-stepping the reject) work as after a
 * single tick; the host then submits the remaining ticks again. A hinted LAST tick of the call returns RG_OK (nothing was run
 * behind it; the host sees the bit in RG_COL_OUT as usual). What it costs: behind every log-term tick of the call the host waits
 * for that tick's pre-pass (not for the tick); only a pre-pass that deferred something waits for the tick and counts.
 * Not available with device Inflights (rg_send_appends has to follow every tick).
 * Asynchronous. Use it to
 * work off a backlog of queued ticks or to replay a log of ticks; a single tick has no fusion to exploit. */
#define RG_MAX_FUSE 8
int rg_tick_device_fused(rg_engine *h, const rg_msgs *dev_msgs, uint32_t n_ticks, uint32_t *dev_out_t,
                         uint64_t *dev_commit_t);
/* Ticks the last rg_tick_device_fused applied: n_ticks after RG_OK, fewer after RG_ERR_HOST_HINT (or a launch failure). */
int rg_fused_ticks_done(const rg_engine *h, uint32_t *n);
/* Raft::maybe_commit() for every group with no messages (post_conf_change src/raft.rs:2630,
 * enable_group_commit :513-518, assign_commit_groups :531-544). Asynchronous. */
int rg_recompute(rg_engine *h);
/* ProgressTracker::maximal_committed_index for every group -> device/host u64[G] (no gate, no
 * state change); used_gc (u8[G], may be NULL) receives the group-commit flag. Host destination. */
int rg_maximal_committed_index(rg_engine *h, uint64_t *host_mci, uint8_t *host_used_gc);
/* Raft::bcast_heartbeat (src/raft.rs:885-891 -> send_heartbeat :822-844): the commit index each
 * MsgHeartbeat carries, min(pr.matched, raft_log.committed), for every slot -> u64 [P][stride] in DEVICE
 * memory (`dev_hb_commit`) or, when `host_hb_commit` is not NULL, copied to the host. Slots without a
 * Progress get 0. Asynchronous unless a host destination is given. */
int rg_heartbeat_commits(rg_engine *h, uint64_t *dev_hb_commit, uint64_t *host_hb_commit);
/* Results of the last tick: commit[G] and out[G] to host memory (either may be NULL). Synchronises. */
int rg_results(rg_engine *h, uint64_t *host_commit, uint32_t *host_out);
/* Sum of RG_OUT_CHANGED / RG_OUT_FAULT bits over all groups for the last tick (device reduction). */
int rg_result_counts(rg_engine *h, uint64_t *n_changed, uint64_t *n_fault);

/* Groups whose last tick raised RG_OUT_HOST_HINT, with the slots concerned (bit s of slot_mask) -> host array of capacity
 * `cap`; *n = number of such groups (only cap are written if it is larger). One device scan of RG_COL_OUT; synchronises.
 * For each (group, slot) the host runs its own RaftLog::find_conflict_by_term(reject_hint, log_term) and steps the reject
 * again with log_term = 0 (rg_step / RG_MF_VALID | RG_MF_REJECT without RG_MF_HAS_LOGTERM and without RG_MF_SENT). */
typedef struct {
    uint64_t group;
    uint32_t slot_mask;
    uint32_t reserved;
} rg_host_hint;
int rg_host_hints(rg_engine *h, rg_host_hint *host_items, uint64_t cap, uint64_t *n);
/* The host's answer: for each record, the rest of handle_append_response's reject branch (src/raft.rs:1679-1721) on the
 * device cell -- Progress::maybe_decr_to(index, hint, 0) (src/tracker/progress.rs:168-206) with `hint` =
 * find_conflict_by_term(reject_hint, log_term).0 from the host's log (src/raft.rs:1657-1660), become_probe when that leaves
 * Replicate -- and the group's RG_COL_OUT word completed: RG_OUT_SEND_APPEND(slot) where maybe_decr_to returned true
 * (host_applied[i] = 1, may be NULL); the group's RG_OUT_HOST_HINT bit falls with the LAST of its flagged slots
 * (bit s of RG_COL_HOST_HINT[g] is cleared per record; a record for a slot that is not waiting changes nothing). Pass the rejects
 * of a group before anything else touches the group. With device Inflights the order of the reference is kept by making the GROUP's sends wait for
 * this call: after rg_tick(_device) call it BEFORE rg_send_appends; the one-launch forms (rg_tick_send, rg_tick_device_send,
 * rg_flush_send) do not run the stage of a group that raised the bit -- this call runs it, with the limit and flags of that
 * launch, and appends the work items to the compact list (rg_send_items; rg_send_columns then no longer holds everything).
 * The compact results rg_ingested_results hands out are not rewritten. Synchronises (control path). */
typedef struct {
    uint64_t group;
    uint64_t index; /* Message.index of the reject */
    uint64_t hint;  /* the resolved hint */
    uint32_t slot;
    uint32_t reserved;
} rg_resolved_hint;
int rg_resolve_host_hints(rg_engine *h, const rg_resolved_hint *items, uint64_t n, uint8_t *host_applied);

/* Census of a tick's message flags in DEVICE memory: counts[0] = VALID messages, [1] = rejects,
 * [2] = slots with a Progress, [3] = groups with at least one event, [4] = RG_MF_BECOME_LEADER events
 * (bench: algorithmic bytes, rejects and elections per group). */
int rg_msg_stats(rg_engine *h, const uint8_t *dev_m_flags, uint64_t counts[5]);

/* ---- vote / quorum-liveness bitmaps (src/quorum/majority.rs:130-154, src/quorum/joint.rs:56-67,
 *      src/tracker.rs:313-372) ---- */
/* yes/no: u8[G] slot bitmasks of recorded votes (record_vote keeps the first, tracker.rs:307-309);
 * result u8[G]: 0 Pending, 1 Lost, 2 Won. Host buffers. */
int rg_vote_result(rg_engine *h, const uint8_t *host_yes, const uint8_t *host_no, uint8_t *host_result);
/* ProgressTracker::tally_votes (src/tracker.rs:313-333): additionally the number of granted / rejected votes among
 * the current voters (incoming or outgoing; votes of ids that left the configuration do not count) -> u8[G] each.
 * has_quorum(set) (tracker.rs:367-372) is rg_vote_result(yes = set, no = 0) == 2. */
int rg_tally_votes(rg_engine *h, const uint8_t *host_yes, const uint8_t *host_no, uint8_t *host_granted,
                   uint8_t *host_rejected, uint8_t *host_result);
/* quorum_recently_active for every group: result u8[G] (1 = active quorum); clears recent_active of
 * every other slot and sets the self slot's, exactly as tracker.rs:346-361. */
int rg_quorum_recently_active(rg_engine *h, uint8_t *host_result);

/* ---- message-at-a-time host mirror of RawNode::step for MsgAppendResponse
 *      (src/raw_node.rs:402-411 -> src/raft.rs:1280-1411 term gate -> :2096-2098) ---- */
typedef struct {
    uint64_t from;             /* Message.from (peer id) */
    uint64_t term;             /* Message.term */
    uint64_t index;            /* Message.index */
    uint64_t commit;           /* Message.commit */
    uint64_t reject_hint;      /* Message.reject_hint (resolved through find_conflict_by_term by the caller when log_term>0) */
    uint64_t request_snapshot; /* Message.request_snapshot */
    uint8_t reject;            /* Message.reject */
    uint8_t ins_full;          /* caller's Inflights::full() for `from` (leave 0 when the engine holds the Inflights) */
    uint8_t pad[6];
    uint64_t log_term;         /* Message.log_term: if > 0 on a reject, reject_hint is passed through
                                  find_conflict_by_term ON THE DEVICE (needs RG_COL_RUN_*); 0 = reject_hint is final */
} rg_append_response;
/* Register peer ids and the leader term of a group so rg_step can map Message.from to a slot. */
int rg_set_peers(rg_engine *h, uint64_t group, const uint64_t *peer_ids, uint32_t n, uint64_t term);
/* Queue one MsgAppendResponse for the next rg_flush. Errors mirror RawNode::step / Raft::step. */
int rg_step(rg_engine *h, uint64_t group, const rg_append_response *m);
/* Queue one MsgHeartbeatResponse (src/raft.rs:2099-2101 -> handle_heartbeat_response :1777-1803). */
int rg_step_heartbeat_response(rg_engine *h, uint64_t group, uint64_t from, uint64_t term, uint64_t commit,
                               uint8_t ins_full);
/* RawNode::step on the BYTES a transport delivers: one protobuf-encoded eraftpb::Message (proto3 wire format of
 * proto/proto/eraftpb.proto:71-92 -- what Message::parse_from_bytes reads before RawNode::step sees the message). The
 * fields of the path are decoded (msg_type 1, from 3, term 4, log_term 5, index 6, commit 8, reject 10, reject_hint 11,
 * request_snapshot 13; `to`, entries, snapshot, context, priority, commit_term and unknown fields are skipped) and the
 * message is stepped: MsgAppendResponse like rg_step, MsgHeartbeatResponse like rg_step_heartbeat_response, a local
 * message type (is_local_msg, src/raw_node.rs:57-66) is RG_ERR_STEP_LOCAL_MSG, every other type RG_ERR_NOT_ON_PATH (nothing
 * queued), bytes that are not a protobuf message RG_ERR_INVALID_ARG. `ins_full` is the one input of the step that is NOT on
 * the wire: the caller's Inflights::full() for the sender, exactly as rg_append_response.ins_full and the last argument
 * of rg_step_heartbeat_response (is_paused of a Replicate peer, src/tracker/progress.rs:209-216; the free_first_one /
 * `old_paused` decisions of src/raft.rs:1742-1751, :1786-1797) -- leave it 0 when the engine holds the Inflights
 * (max_inflight > 0). rg_decode_message is the decoder alone (pure host code: no engine, no device). */
typedef struct {
    uint32_t msg_type; /* eraftpb::MessageType */
    uint32_t reject;   /* Message.reject */
    uint64_t to, from, term, log_term, index, commit, commit_term, reject_hint, request_snapshot, priority;
    uint64_t n_entries;     /* repeated Entry entries = 7: how many (their bytes are skipped) */
    uint32_t has_snapshot;  /* Snapshot snapshot = 9 present */
    uint32_t context_len;   /* bytes context = 12 */
} rg_decoded_message;
int rg_decode_message(const uint8_t *bytes, uint64_t len, rg_decoded_message *out);
int rg_step_bytes(rg_engine *h, uint64_t group, const uint8_t *bytes, uint64_t len, uint8_t ins_full);

/* ---- the other side of the path: the messages a leader SENDS, as the bytes a transport takes ----
 * The send stage (below) decides WHAT goes to each peer -- `prev_index`, `last_index`, how many messages -- and leaves
 * the building to the host, which owns the log: Raft::maybe_send_append -> prepare_send_entries (src/raft.rs:714-731:
 * msg_type = MsgAppend, index = next_idx - 1, log_term = term(index), entries, commit = raft_log.committed) -> Raft::send
 * (:602-662: from = self.id, term = self.term) -> Message::write_to_bytes. These three calls are that last step and the
 * arithmetic both sides have to agree on; pure host code (no engine, no device), like rg_decode_message.
 * Serialisation is canonical proto3 (fields in field-number order, defaults omitted): byte for byte what the protobuf
 * runtime writes for eraftpb.proto:23-31, :71-92, and what any protobuf parser -- rust-protobuf and prost included -- reads
 * back into the same Message. */
typedef struct {
    uint32_t entry_type; /* eraftpb::EntryType: 0 EntryNormal, 1 EntryConfChange, 2 EntryConfChangeV2 */
    uint32_t sync_log;   /* Entry.sync_log (bool) */
    uint64_t term, index;
    const uint8_t *data;    /* Entry.data (may be NULL when data_len == 0) */
    uint64_t data_len;
    const uint8_t *context; /* Entry.context */
    uint64_t context_len;
} rg_entry;
/* Entry::compute_size(): the number util::limit_size adds up (src/util.rs:52-76) -- what rg_log_sizes_write wants
 * accumulated per group for RG_SEND_BYTES. Entry::default() is 0 bytes. */
uint64_t rg_entry_size(const rg_entry *e);
/* util::limit_size(entries, Some(max_size)) (src/util.rs:52-76; RaftLog::entries / Storage::entries apply it): how many of
 * the n entries ONE message keeps. n <= 1 and UINT64_MAX (NO_LIMIT) keep all; the first entry always stays, and so does
 * whatever follows while the running total is still 0. The device's RG_SEND_BYTES stage counts messages with the same rule. */
uint64_t rg_limit_size(const rg_entry *entries, uint64_t n, uint64_t max_size);
typedef struct {
    uint32_t msg_type; /* eraftpb::MessageType (MsgAppend 3, MsgSnapshot 7, MsgHeartbeat 8, MsgTimeoutNow 14, ...) */
    uint32_t reject;
    uint64_t to, from, term, log_term, index, commit, commit_term, reject_hint, request_snapshot, priority;
    const rg_entry *entries; /* repeated Entry entries = 7 */
    uint64_t n_entries;
    const uint8_t *snapshot; /* Snapshot snapshot = 9: an already serialised eraftpb::Snapshot (the storage's), NULL = none */
    uint64_t snapshot_len;
    const uint8_t *context;  /* bytes context = 12 */
    uint64_t context_len;
} rg_message;
/* Message::compute_size() -> *len. RG_ERR_INVALID_ARG: a length without its pointer, or more than a protobuf message may
 * hold (2 GiB - 1). */
int rg_message_size(const rg_message *m, uint64_t *len);
/* Message::write_to_bytes into buf[cap]. *len (required) always receives the size; nothing is written and
 * RG_ERR_INVALID_ARG is returned when cap is smaller (call with cap = 0 to size a buffer). */
int rg_encode_message(const rg_message *m, uint8_t *buf, uint64_t cap, uint64_t *len);
/* Leader-local events, queued the same way (src/raft.rs:976-1016). */
int rg_local_append(rg_engine *h, uint64_t group, uint64_t new_last_index);
int rg_local_persisted(rg_engine *h, uint64_t group, uint64_t index);
int rg_mark_sent(rg_engine *h, uint64_t group, uint64_t peer_id);
/* This node won the group's election at `term` (Raft::become_leader, src/raft.rs:1151-1202): queues
 * RG_MF_BECOME_LEADER for the group -- applied before every other event of the flush -- and moves the group's
 * term gate (rg_step) to `term`. Errors: RG_ERR_INVALID_ARG when `term` is not above the registered term. */
int rg_local_become_leader(rg_engine *h, uint64_t group, uint64_t term);
/* RawNode::report_unreachable(id) / report_snapshot(id, status) (src/raw_node.rs:692-709) through the mirror's peer ids:
 * rg_progress_events on the peer's slot, applied at once. Like the reference (which drops the step's result) an id without
 * a Progress is ignored. RG_ERR_SLOT_BUSY while the group has traffic queued for the next flush: the reference applies
 * local messages in call order, so flush first. `failure`: SnapshotStatus::Failure (non-zero) / Finish (0). */
int rg_report_unreachable(rg_engine *h, uint64_t group, uint64_t peer_id);
int rg_report_snapshot(rg_engine *h, uint64_t group, uint64_t peer_id, int failure);
/* Run one tick over everything queued since the last flush and clear the queue. */
int rg_flush(rg_engine *h);

/* ---- send stage (engines created with max_inflight > 0): Inflights on the device and the
 *      maybe_send_append DECISION (src/raft.rs:773-819, prepare_send_entries :722-731, bcast_append :850-857,
 *      the loop at :1761; src/tracker/inflights.rs:42-125) ----
 * Call once after every tick. For every group it (1) applies the tick's Inflights effects -- ins.free_to(m.index)
 * for an accepted ack in Replicate, ins.free_first_one() for a heartbeat response on a full window, ins.reset()
 * when the Progress left Replicate -- and (2) serves the tick's send requests in slot order: bcast_append when the
 * commit index moved (RG_OUT_CHANGED) or the leader appended (RG_OUT_APPENDED), send_append(from) and the
 * `while maybe_send_append(from, false)` loop. Each message sent applies Progress::update_state(last) (Replicate: next = last+1, ins.add(last); Probe: paused) in
 * place, so the host writes no RG_MF_SENT events in this mode. Messages are NOT built: the result is one work item
 * per peer that has something to send. `max_entries_per_msg` models Config::max_size_per_msg for equal-sized
 * entries (util::limit_size keeps at least one entry), 0 = NO_LIMIT; with RG_SEND_BYTES it IS max_size_per_msg, in
 * bytes, over the real entry sizes (below). A peer whose entries are compacted away
 * (next_idx < first_index = RG_COL_DUMMY_INDEX + 1) or that has a pending snapshot request yields
 * RG_SEND_SNAPSHOT (if recent_active, raft.rs:665-672; last_index then carries the requested snapshot index, 0 =
 * any); the host then fetches the snapshot and applies
 * Progress::become_snapshot with rg_write_cells -- the device leaves that Progress untouched.
 * Sends requested by message k of a tick happen after the whole tick; flush after every step where
 * that matters. Not available for fused launches. Asynchronous. */
typedef struct {
    uint64_t group;
    uint64_t prev_index; /* Message.index = next_idx - 1 of the first message */
    uint64_t last_index; /* index of the last entry sent (== prev_index: one empty MsgAppend) */
    uint32_t slot;
    uint16_t n_msgs;     /* messages of max_entries_per_msg entries each (the last may be shorter) */
    uint16_t kind;       /* RG_SEND_* */
} rg_send_item;
#define RG_SEND_APPEND 1u
#define RG_SEND_SNAPSHOT 2u
#define RG_SEND_HOST 3u /* RG_SEND_BYTES only: the peer needs entries whose sizes have left the device's window
                           (rg_log_sizes_enable): the host, which owns the log, runs maybe_send_append for it and reports
                           what it sent with rg_update_state; prev_index = next_idx - 1, last_index = the leader's
                           last_index, the Progress is untouched */
#define RG_SEND_SKIP_BCAST_COMMIT 0x1u /* Config::skip_bcast_commit (src/config.rs:87): a commit advance is broadcast
                                          only by groups with a pending conf change -- should_bcast_commit(),
                                          src/raft.rs:2684-2686; RG_PF_PENDING_CONF on the leader's slot */
#define RG_SEND_BYTES 0x2u /* the limit is Config::max_size_per_msg in BYTES (src/config.rs:58-63), applied as
                              RaftLog::entries does: util::limit_size over Entry::compute_size() of every entry
                              (src/util.rs:52-76; UINT64_MAX = NO_LIMIT, 0 = one entry per message). Needs the entry sizes
                              on the device: rg_log_sizes_enable / rg_log_sizes_write. */
int rg_send_appends(rg_engine *h, uint64_t max_entries_per_msg, uint32_t flags);
/* The dense tick AND its send stage as ONE launch: rg_tick(_device) immediately followed by rg_send_appends, same
 * results in the same columns (RG_COL_*, the Inflights, rg_send_items / rg_send_columns), but the stage runs on the registers
 * the tick leaves instead of re-reading what the tick has just stored (the result word, cfg, the flag row, last_index,
 * `matched`, `next`) one kernel boundary later -- the host's Ready loop (RawNode::ready, src/raw_node.rs:402-441, collects the
 * messages of a whole batch of steps anyway). With RG_SEND_BYTES the size records of the entries this tick's
 * local-append events announce must have been written (rg_log_sizes_write) BEFORE the call, since the stage reads them in
 * the same launch. rg_tick_send synchronises like rg_tick; rg_tick_device_send is asynchronous like rg_tick_device. */
int rg_tick_send(rg_engine *h, const rg_msgs *host_msgs, uint64_t max_entries_per_msg, uint32_t flags);
int rg_tick_device_send(rg_engine *h, const rg_msgs *dev_msgs, uint64_t max_entries_per_msg, uint32_t flags);

/* ---- entry sizes for RG_SEND_BYTES (byte-accurate Config::max_size_per_msg) ----
 * The send decision needs Entry::compute_size() of the entries it is about to attach (util::limit_size keeps the first
 * entry and then as many as fit `max` bytes). The device keeps, per group, the CUMULATIVE size of its last `window` log
 * entries (u32 [G][window], 4 * window bytes per group; window a power of two, 8..4096) -- what a leader replicating to
 * live followers reads. The host writes one record per appended entry (any order, any batch; before the send stage that
 * may send it): cum_bytes = the total size of the group's entries up to and including `index`, counted from wherever
 * the host likes (only differences are used, modulo 2^32: a window's total has to stay below 4 GiB). Entries whose
 * record has left the window (a follower more than window - 1 entries behind) are served by the host: RG_SEND_HOST.
 * EVERY entry the leader's log gains needs its record before a stage may send it -- including the empty entry an
 * RG_MF_BECOME_LEADER event appends on the device (index = the group's last_index once RG_OUT_BECAME_LEADER is reported; its
 * size is what Entry::compute_size() gives for {term, index} alone): the ring is indexed by `index & (window - 1)`, so a
 * missing record silently yields the size of the entry `window` places earlier. The engine cannot tell a missing record
 * from a written one; a host that cannot guarantee the order runs such a group's sends itself (rg_update_state).
 * Checkpoints include the table. Asynchronous (engine stream). */
typedef struct {
    uint64_t group;
    uint64_t index;     /* log index of the entry */
    uint64_t cum_bytes; /* sum of Entry::compute_size() over the group's entries <= index */
} rg_log_size;
int rg_log_sizes_enable(rg_engine *h, uint32_t window);
int rg_log_sizes_write(rg_engine *h, const rg_log_size *recs, uint64_t n);
/* Synthetic sizes for the benchmark / tests: every group's window (last_index - window, last_index] is filled from
 * size(group, index) = min_bytes + hash(seed, group, index) % (spread + 1) (rg_hash of BASELINE.md's generator). */
int rg_workload_sizes(rg_engine *h, uint64_t seed, uint32_t min_bytes, uint32_t spread);

/* The host sent MsgAppends itself (RG_SEND_HOST items; or a host that builds some messages on its own): apply
 * Progress::update_state(last) (src/tracker/progress.rs:231-243) for each -- Replicate: next_idx = last + 1 and
 * ins.add(last) on the device window; Probe: paused; Snapshot: RG_ERR_STATE is NOT raised (the reference panics), the
 * record is ignored; so is a message that no longer fits the peer's window (Inflights::add on a full window panics in the
 * reference, inflights.rs:66-68). Records are applied in array order; the records of one (group, slot) must be adjacent.
 * Only for engines with device Inflights. Asynchronous. */
typedef struct {
    uint64_t group;
    uint64_t last;  /* index of the last entry of the message */
    uint32_t slot;
    uint32_t reserved;
} rg_sent_msg;
int rg_update_state(rg_engine *h, const rg_sent_msg *msgs, uint64_t n);
/* rg_flush followed by rg_send_appends, and for small batches in the SAME host<->device round trip (the work items
 * come back with the tick's results; rg_send_items / rg_ingested_results then read host memory). */
int rg_flush_send(rg_engine *h, uint64_t max_entries_per_msg, uint32_t flags);
/* Work items of the last rg_send_appends (order unspecified) -> host array of capacity `cap`; *n = number of
 * items (only cap are written if it is larger). Synchronises. rg_send_items_ptr: the same list in device memory. */
int rg_send_items(rg_engine *h, rg_send_item *host_items, uint64_t cap, uint64_t *n);
const rg_send_item *rg_send_items_ptr(rg_engine *h);
/* A stage over EVERY group (after a dense tick) writes its work items as peer-major columns, like every other column of
 * the engine -- u64 prev_index / last_index [P][stride], u32 n_msgs | kind << 16 [P][stride], 0 = nothing for that peer --
 * with coalesced stores and no compaction; a device-side consumer (a message builder) reads them in place through
 * these pointers. rg_send_items / rg_send_items_ptr materialise the compact list from them on request. Stages over a
 * few touched groups (the sparse path) produce the compact list directly and leave these columns alone. */
int rg_send_columns(rg_engine *h, const uint64_t **dev_prev_index, const uint64_t **dev_last_index,
                    const uint32_t **dev_n_kind);
/* Bits 31 / 30 of an n / kind word (the kind is bits 16-29): the item's last_index was NOT written to its cell of
 * dev_last_index, because the value is at hand elsewhere:
 *   RG_SEND_LAST_IS_TAIL  it is the peer's NEWEST INFLIGHT, which the stage has just stored (every MsgAppend that carries entries
 *                         to a Replicate peer ends in Progress::update_state(last) -> ins.add(last), progress.rs:231-243: the
 *                         steady case) -- read it from the window's tail column, same layout (u64 [P][stride]): rg_send_tail_column;
 *   RG_SEND_LAST_IS_PREV  an empty MsgAppend (send_append to a peer that already has every entry): last_index == prev_index.
 * A wave of the stage writes a line of dev_last_index only when one of its 64 groups has an item of neither kind (8 B less per
 * item otherwise). The compact list (rg_send_items) always carries last_index itself. The tail column is LIVE state: a consumer
 * of the columns reads it before the next call that moves a window (a tick's stage, rg_update_state, rg_load_inflights,
 * rg_restore -- those three make the compact list from the columns first, so rg_send_items stays right across them). */
#define RG_SEND_LAST_IS_TAIL 0x80000000u
#define RG_SEND_LAST_IS_PREV 0x40000000u
int rg_send_tail_column(rg_engine *h, const uint64_t **dev_newest_inflight);
/* Inflights in/out (parity, checkpoints): meta u32 [P][stride] = start | count << 16; ring u64 [G][P][cap]. */
uint64_t rg_inflights_bytes(const rg_engine *h, int ring);
int rg_read_inflights(rg_engine *h, uint32_t *host_meta, uint64_t *host_ring); /* either may be NULL */
int rg_load_inflights(rg_engine *h, const uint32_t *host_meta, const uint64_t *host_ring); /* both required */

/* ---- resident small-batch path ("mailbox") ----
 * What one RawNode::step -> ready() costs a host is the round trip of a SMALL flush, and most of that is the HIP runtime:
 * one launch, one synchronisation, one wake-up (~15 of the ~21 us). With the mailbox on, ONE workgroup stays resident on
 * the device and serves small flushes (rg_flush / rg_ingest_tick with <= 256 records that follow another sparse tick)
 * out of pinned host memory: the host writes the records and a request word and spins on the answer word -- no launch, no
 * stream synchronisation. Same results as the launch path (it runs the same code). The workgroup leaves when any other
 * entry point needs the engine's stream (automatically), when it has been idle for idle_timeout_us (0 = 2000), or after
 * 200 ms, and is relaunched by the next small flush. With device Inflights it serves rg_flush_send -- the send stage of the
 * touched groups runs inside the same request -- and leaves plain rg_flush to the launch path; not with commit publication. The caller's thread spins while it waits: meant for a latency-bound host loop. */
int rg_mailbox_start(rg_engine *h, uint32_t idle_timeout_us);
int rg_mailbox_stop(rg_engine *h);
/* Flushes the resident workgroup has answered so far / how often it was (re)launched (either may be NULL). */
int rg_mailbox_stats(const rg_engine *h, uint64_t *flushes_served, uint64_t *launches);

/* ---- sparse path: wire-order records -> slot matrix -> tick over the touched groups only ----
 * For realistic traffic (a small fraction of the groups has events in a tick) the dense sweep of rg_tick
 * wastes bandwidth. rg_ingest scatters array-of-structs records (what a transport thread produces from
 * eraftpb::Message, proto/proto/eraftpb.proto:71-92) into the engine-owned message columns on the device
 * (LDS-staged AoS->SoA transpose) and collects the set of touched groups; rg_tick_ingested runs the same
 * per-group arithmetic as rg_tick over exactly those groups and consumes the events. At most ONE record per
 * (group, slot) between two rg_tick_ingested calls: a second one is dropped and counted in n_duplicates (the
 * caller re-submits it after the tick, which preserves per-peer order). rg_flush uses this path. */
typedef struct {
    uint64_t group;  /* engine-local group index */
    uint64_t index;  /* Message.index (self slot: persisted index) */
    uint64_t commit; /* Message.commit (self slot with RG_MF_APPEND: new last_index) */
    uint64_t hint;   /* Message.reject_hint (after find_conflict_by_term) */
    uint64_t rs;     /* Message.request_snapshot */
    uint64_t log_term; /* Message.log_term (with RG_MF_HAS_LOGTERM) */
    uint32_t slot;   /* peer slot 0..P-1 */
    uint32_t flags;  /* RG_MF_* */
    uint64_t pad;    /* 64-byte records */
} rg_wire_msg;
int rg_ingest(rg_engine *h, const rg_wire_msg *host_records, uint64_t n, uint64_t *n_duplicates);
/* Same, records already in DEVICE memory (a device-side transport / decoder); the array must hold
 * n records (any alignment of 16 B). Asynchronous: duplicates are accumulated and reported by the next
 * rg_ingest (host) call or readable through rg_ingested_duplicates after rg_sync. */
int rg_ingest_device(rg_engine *h, const rg_wire_msg *dev_records, uint64_t n);
int rg_ingested_duplicates(rg_engine *h, uint64_t *n_duplicates);
int rg_tick_ingested(rg_engine *h, uint64_t *n_groups);
/* rg_ingest + rg_tick_ingested in ONE host<->device round trip (pinned record staging, back-to-back launches, one
 * packed copy of the results, one synchronisation): the low-latency form for small batches (33 us for a handful
 * of groups against 118 us for the three-call sequence). *n_duplicates counts dropped records of the whole window
 * (earlier rg_ingest / rg_ingest_device calls included). rg_ingested_results then returns from host memory. */
int rg_ingest_tick(rg_engine *h, const rg_wire_msg *host_records, uint64_t n, uint64_t *n_groups,
                   uint64_t *n_duplicates);
/* Groups touched by the last rg_tick_ingested with their commit index and result word (host arrays of
 * capacity `cap`; *n receives the number of groups, which may exceed cap: then only cap are written). */
int rg_ingested_results(rg_engine *h, uint64_t *groups, uint64_t *commit, uint32_t *out, uint64_t cap, uint64_t *n);

/* ---- multi-GPU: disjoint group ranges, one engine per rank, and the ONE exchange of the path -- the publication of
 *      commit indices. What crosses xGMI is what RawNode::advance_append surfaces per group
 *      (src/raw_node.rs:643-651, fed by Raft::maybe_commit, src/raft.rs:893-904): the new commit index, which never
 *      decreases (RaftLog::commit_to, src/raft_log.rs:286-300). The ticks therefore record, per group, how far the index
 *      ADVANCED since the last publication -- one byte, fused into the tick's store path; advances >= 255 go to a short
 *      exact-value list -- and rg_publish_commit all-gathers that ~1 B/group slice (ncclAllGather over xGMI, RCCL bound at
 *      run time) on a side stream instead of the 8 B/group column: every tick can be published at 8 ranks (1 MB instead
 *      of 8 MB per rank and tick at 1 M groups). Each rank keeps a replica of ALL commit indices [world][stride] u64 in
 *      HBM; gathered slices are folded into it lazily (when `ring_ticks` publications are buffered, or when it is read).
 *      A full exact-value list (or rg_restore / a reloaded commit column) marks the rank's slice "lost": all ranks
 *      see that in the gathered headers when they fold the slices into their replicas, which they all do at the same
 *      publication numbers (every `ring_ticks`-th), and the check point after that one publishes a full 8 B/group
 *      snapshot -- decided identically everywhere without another collective (the replica of a lost rank is inexact
 *      for at most 2 x ring_ticks publications; a host that knows it rolled back passes RG_PUBLISH_FULL). All ranks must hold the same number of groups and call
 *      rg_comm_init / rg_publish_commit in the same order (they are collectives). ---- */
#define RG_COMM_ID_BYTES 128
/* ncclGetUniqueId: rank 0 creates the id, the host distributes it to the other ranks (any channel). */
int rg_comm_unique_id(uint8_t id[RG_COMM_ID_BYTES]);
/* A host-provided all-gather (MPI, a test harness): gather `bytes_per_rank` bytes from every rank's dev_send into
 * dev_recv ([world][bytes_per_rank], rank order), enqueued on (or completed before returning from) hip_stream.
 * Returns 0 on success. */
typedef int (*rg_allgather_fn)(void *user, const void *dev_send, void *dev_recv, uint64_t bytes_per_rank, void *hip_stream);
typedef struct {
    uint32_t rank, world;
    const uint8_t *unique_id;   /* RG_COMM_ID_BYTES from rg_comm_unique_id (RCCL transport); ignored with `transport` */
    rg_allgather_fn transport;  /* NULL = RCCL's ncclAllGather; else the exchange goes through this callback */
    void *transport_user;
    uint32_t ring_ticks;        /* gathered publications buffered before the replica is updated; 0 = 32, at most 64 */
    uint32_t overflow_slots;    /* capacity of a slice's exact-value list; 0 = n_groups / 256 + 64 */
} rg_comm_config;
/* Collective. Allocates the publication buffers and the replica, creates the communicator and runs one FULL
 * publication so every replica starts from the actual commit columns. */
int rg_comm_init(rg_engine *h, const rg_comm_config *cfg);
int rg_comm_destroy(rg_engine *h);
/* RCCL's first use in a process is slow: the library is ~0.5 GB to map (seconds, up to minutes on a cold machine) and the
 * first communicator sets up its transports. rg_comm_warmup loads it and creates and destroys a ONE-rank communicator on the
 * calling thread's current device -- call it at start-up, before any step a timeout bounds. Not a collective. */
int rg_comm_warmup(void);
/* What the engine's publication runs over -- and, for RCCL, what the COMMUNICATOR reports about itself (ncclCommCount /
 * ncclCommUserRank), as opposed to what the engine was told: a multi-GPU run proves its rank count with this. */
typedef struct {
    uint32_t rank, world;  /* as given to rg_comm_init / rg_comm_init_all (0 / 0 without a communicator) */
    uint32_t transport;    /* RG_TRANSPORT_* */
    uint32_t in_process;   /* 1 = one of several ranks driven by one thread (rg_comm_init_all) */
    uint32_t rccl_ranks;   /* ncclCommCount of the engine's communicator (0 unless RG_TRANSPORT_RCCL) */
    uint32_t rccl_rank;    /* ncclCommUserRank */
} rg_comm_info;
#define RG_TRANSPORT_NONE 0u     /* no communicator: a single engine */
#define RG_TRANSPORT_RCCL 1u     /* ncclAllGather */
#define RG_TRANSPORT_CALLBACK 2u /* the host's rg_allgather_fn */
#define RG_TRANSPORT_LOCAL 3u    /* device-to-device copies between engines of one process (RG_COMM_ALL_LOCAL) */
int rg_comm_info_get(rg_engine *h, rg_comm_info *out);
/* Several ranks in ONE process (`rg_create(cfg{..., n_devices, device_ids[]})`; the reference's own embedding
 * drives many RawNodes from one thread, src/raw_node.rs:284, examples/five_mem_node/main.rs:67-112). Two forms:
 *  (1) one THREAD per engine: each thread creates its engine and calls rg_comm_init / rg_publish_commit on it like a rank of
 *      its own process would (the library keeps no state across handles beyond a per-device engine count and the lazily
 *      loaded RCCL binding, both behind locks; rg_last_error is per thread) -- RCCL's ordinary multi-threaded use;
 *  (2) ONE thread for all of them: rg_comm_init_all makes engines[i] rank i of n, and every publication goes through
 *      rg_publish_commit_all(engines, n, flags) -- the n ranks' exchanges are issued together (RCCL: inside one
 *      ncclGroupStart / ncclGroupEnd, without which a single thread would block in rank 0's collective before rank 1's is
 *      issued). rg_publish_sync / rg_published_commit / rg_publish_stats_get / rg_comm_destroy stay per engine. Transport:
 *      RG_COMM_ALL_AUTO = RCCL when every engine has a device of its own (n > 1), else RG_COMM_ALL_LOCAL: device-to-device
 *      copies between the engines' buffers (same process, so every rank's slice is addressable: peer copies over xGMI between
 *      GPUs, plain copies inside one -- what several shards on ONE device use, which RCCL refuses). All engines hold the same
 *      number of groups. rg_publish_commit on such an engine is refused (RG_ERR_STATE). */
typedef struct {
    uint32_t ring_ticks;     /* as rg_comm_config */
    uint32_t overflow_slots; /* as rg_comm_config */
    uint32_t transport;      /* RG_COMM_ALL_* */
    uint32_t reserved;
} rg_comm_all_config;
#define RG_COMM_ALL_AUTO 0u
#define RG_COMM_ALL_RCCL 1u
#define RG_COMM_ALL_LOCAL 2u
int rg_comm_init_all(rg_engine *const *engines, uint32_t n, const rg_comm_all_config *cfg /* NULL = defaults */);
int rg_publish_commit_all(rg_engine *const *engines, uint32_t n, uint32_t flags);
/* Collective, asynchronous: publish this rank's commit advances since its previous publication (call it after every
 * tick for per-tick visibility, or less often: the ticks accumulate). The exchange runs on a side stream and overlaps
 * the ticks that follow. RG_PUBLISH_FULL forces the 8 B/group snapshot on ALL ranks (pass it on all of them). */
#define RG_PUBLISH_FULL 0x1u
int rg_publish_commit(rg_engine *h, uint32_t flags);
/* Wait for every publication issued so far and bring the replica up to date. */
int rg_publish_sync(rg_engine *h);
/* The replica in device memory: commit index of group g of rank r at [r * *stride + g]. Valid after rg_publish_sync. */
const uint64_t *rg_published_commit_ptr(rg_engine *h, uint64_t *stride);
/* rg_publish_sync, then groups [first, first + n) of `rank` to host memory. */
int rg_published_commit(rg_engine *h, uint32_t rank, uint64_t first, uint64_t n, uint64_t *host_commit);
typedef struct {
    uint64_t publications, full_publications, replica_updates;
    uint64_t bytes_per_rank_last;  /* what the last publication moved per rank */
    uint64_t bytes_per_rank_delta; /* size of a delta slice (header + list + 1 B/group) */
    uint64_t bytes_per_rank_full;  /* size of a full snapshot (8 B/group) */
    uint32_t overflow_slots, ring_ticks;
    double host_us_events, host_us_allgather, host_us_memset; /* host time spent inside rg_publish_commit, by part */
    uint64_t events_on_tick_packets; /* publications whose "slice complete" event rode on the dispatch packet of the dense tick they
                                        followed (rg_tick / rg_tick_device right before rg_publish_commit) instead of being recorded
                                        behind it: no packet of its own in the engine's queue (2.5 us of idle queue per tick saved) */
} rg_publish_stats;
int rg_publish_stats_get(rg_engine *h, rg_publish_stats *out);
/* Host twins of the encoding over caller-provided buffers (no GPU involved; CPU-only tests of the N > 1 exchange):
 * accumulate the advances old_commit -> new_commit into a slice of rg_pub_bytes_per_rank() bytes (zeroed by the caller
 * at the start of an interval), and add one gathered publication ([world] slices) to a replica [world][stride] with
 * stride = n_groups rounded up to 256; *lost_ranks = slices that ask for a full resynchronisation. */
uint64_t rg_pub_bytes_per_rank(uint64_t n_groups, uint32_t overflow_slots);
int rg_pub_accumulate_host(uint64_t n_groups, uint32_t overflow_slots, const uint64_t *old_commit,
                           const uint64_t *new_commit, uint8_t *slice);
int rg_pub_apply_host(uint64_t n_groups, uint32_t overflow_slots, uint32_t world, const uint8_t *gathered,
                      uint64_t *replica, uint32_t *lost_ranks);

/* ---- synthetic AppendResponse stream; same code on host and device ---- */
typedef struct {
    uint64_t seed;
    uint32_t workload; /* RG_WL_* */
    uint32_t reserved; /* RG_WL_MIXED only: bits 0-3 = fixed replica-set size (3/5/7) of a size-class shard, 0 = by group id % 3;
                          RG_WL_PLACE_SORTED (bit 4): the same groups, placed inside the shard by size class (all ids 0 mod 3
                          first, then 1 mod 3, then 2 mod 3) instead of interleaved -- contiguous classes let one launch skip the
                          peer slots a class does not have */
} rg_workload;
#define RG_WL_MAJORITY 2u /* BASELINE config 2 (and 1, 4): majority quorum over all P slots */
#define RG_WL_JOINT 3u    /* config 3: incoming {0,1,2} && outgoing {1,2,3}, slot 4.. learners */
#define RG_WL_MIXED 5u    /* config 5: P in {3,5,7} by group, 10% groups in post-election probe/reject */
#define RG_WL_PLACE_SORTED 0x10u /* rg_workload.reserved flag, see there */
#define RG_WL_GROUP_COMMIT 0x20u /* rg_workload.reserved flag: ProgressTracker.group_commit on in every group (RG_CFG_GROUP_COMMIT)
                                    and every peer in one of three commit groups (Progress.commit_group_id 1..3, by hash): each
                                    commit evaluation of the stream is the group-commit form (src/quorum/majority.rs:99-123) */
/* Initialise all engine state for the workload (device-side generator). */
int rg_workload_init(rg_engine *h, const rg_workload *w, uint64_t first_group_global);
/* Generate tick `tick`'s messages from the CURRENT device state into device message columns. */
int rg_workload_gen(rg_engine *h, const rg_workload *w, uint64_t first_group_global, uint64_t tick,
                    uint64_t *d_m_index, uint64_t *d_m_commit, uint64_t *d_m_hint, uint64_t *d_m_rs,
                    uint8_t *d_m_flags);
/* Host twins of the two calls above over caller-provided SoA arrays (no GPU involved): used by
 * CPU-only tests and to feed the CPU baseline the identical stream. */
typedef struct {
    uint64_t n_groups, stride;
    uint32_t n_slots, reserved;
    uint64_t *match, *next, *pr_commit, *pend_snap, *pend_rs, *gid;
    uint8_t *pflags;
    uint64_t *commit, *term_lo, *term_hi;
    uint32_t *cfg;
} rg_host_state;
int rg_workload_init_host(const rg_workload *w, uint64_t first_group_global, rg_host_state *s);
int rg_workload_gen_host(const rg_workload *w, uint64_t first_group_global, uint64_t tick,
                         const rg_host_state *s, uint64_t *m_index, uint64_t *m_commit,
                         uint64_t *m_hint, uint64_t *m_rs, uint8_t *m_flags);

#ifdef __cplusplus
}
#endif
#endif /* RAFTGROUPS_H */
