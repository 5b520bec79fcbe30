// GENERATED from include/raftgroups.h by tools/gen_rust_bindings.py -- do not edit.
// Link with `-lraftgroups` (build.rs: cargo:rustc-link-lib=dylib=raftgroups). Never compiled in the build image (no
// Rust toolchain there); tests/test_abi.py checks that it names every export of libraftgroups.so with the header's arity.
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_void};

pub const RG_MAX_SLOTS: u32 = 8;
pub const RG_ABI_VERSION: u32 = 7;
pub const RG_PF_STATE_MASK: u32 = 0x03;
pub const RG_STATE_PROBE: u32 = 0;
pub const RG_STATE_REPLICATE: u32 = 1;
pub const RG_STATE_SNAPSHOT: u32 = 2;
pub const RG_PF_PAUSED: u32 = 0x04;
pub const RG_PF_RECENT_ACTIVE: u32 = 0x08;
pub const RG_PF_PENDING_CONF: u32 = 0x20;
pub const RG_PF_INS_FULL: u32 = 0x10;
pub const RG_PF_PEND_SNAP: u32 = 0x40;
pub const RG_PF_PEND_RS: u32 = 0x80;
pub const RG_MF_VALID: u32 = 0x01;
pub const RG_MF_REJECT: u32 = 0x02;
pub const RG_MF_HAS_RS: u32 = 0x04;
pub const RG_MF_INS_FULL: u32 = 0x08;
pub const RG_MF_SENT: u32 = 0x10;
pub const RG_MF_APPEND: u32 = 0x20;
pub const RG_MF_HAS_LOGTERM: u32 = 0x80;
pub const RG_MF_BECOME_LEADER: u32 = 0x02;
pub const RG_MF_HEARTBEAT: u32 = 0x40;
pub const RG_CFG_GROUP_COMMIT: u32 = 0x00080000;
pub const RG_OUT_CHANGED: u32 = 0x1;
pub const RG_OUT_FAULT: u32 = 0x2;
pub const RG_OUT_TIMEOUT_NOW: u32 = 0x4;
pub const RG_OUT_APPENDED: u32 = 0x8;
pub const RG_OUT_BECAME_LEADER: u32 = 0x10;
pub const RG_OUT_HOST_HINT: u32 = 0x20;
pub const RG_TERM_RUNS: u32 = 8;
pub const RG_CACHE_AUTO: u32 = 0;
pub const RG_CACHE_PLAIN: u32 = 1;
pub const RG_CACHE_STREAM_MSGS: u32 = 2;
pub const RG_CACHE_STREAM_ALL: u32 = 3;
pub const RG_CACHE_RESIDENT: u32 = 4;
pub const RG_CFGF_NO_SIZE_CLASSES: u32 = 0x1;
pub const RG_CFGF_CLASS_BLOCK_ORDER: u32 = 0x2;
pub const RG_CFGF_IX64: u32 = 0x4;
pub const RG_VARIANT_DEFAULT: u32 = 0;
pub const RG_VARIANT_LANE: u32 = 1;
pub const RG_VARIANT_LDS: u32 = 2;
pub const RG_VARIANT_LDS_DMA: u32 = 4;
pub const RG_VARIANT_COMPACT: u32 = 5;
pub const RG_VARIANT_COOP: u32 = 3;
pub const RG_KERNEL_NONE: u32 = 0;
pub const RG_KERNEL_LANE: u32 = 1;
pub const RG_KERNEL_CLASSES: u32 = 2;
pub const RG_KERNEL_SPLIT: u32 = 3;
pub const RG_KERNEL_LDS: u32 = 4;
pub const RG_KERNEL_COMPACT: u32 = 5;
pub const RG_KERNEL_TICK_SEND: u32 = 6;
pub const RG_EV_UNREACHABLE: u32 = 1;
pub const RG_EV_SNAPSHOT_FINISH: u32 = 2;
pub const RG_EV_SNAPSHOT_FAILURE: u32 = 3;
pub const RG_MAX_FUSE: u32 = 8;
pub const RG_SEND_APPEND: u32 = 1;
pub const RG_SEND_SNAPSHOT: u32 = 2;
pub const RG_SEND_HOST: u32 = 3;
pub const RG_SEND_SKIP_BCAST_COMMIT: u32 = 0x1;
pub const RG_SEND_BYTES: u32 = 0x2;
pub const RG_SEND_LAST_IS_TAIL: u32 = 0x80000000;
pub const RG_SEND_LAST_IS_PREV: u32 = 0x40000000;
pub const RG_COMM_ID_BYTES: u32 = 128;
pub const RG_TRANSPORT_NONE: u32 = 0;
pub const RG_TRANSPORT_RCCL: u32 = 1;
pub const RG_TRANSPORT_CALLBACK: u32 = 2;
pub const RG_TRANSPORT_LOCAL: u32 = 3;
pub const RG_COMM_ALL_AUTO: u32 = 0;
pub const RG_COMM_ALL_RCCL: u32 = 1;
pub const RG_COMM_ALL_LOCAL: u32 = 2;
pub const RG_PUBLISH_FULL: u32 = 0x1;
pub const RG_WL_MAJORITY: u32 = 2;
pub const RG_WL_JOINT: u32 = 3;
pub const RG_WL_MIXED: u32 = 5;
pub const RG_WL_PLACE_SORTED: u32 = 0x10;
pub const RG_WL_GROUP_COMMIT: u32 = 0x20;

// rg_status
pub const RG_OK: i32 = 0;
pub const RG_ERR_INVALID_ARG: i32 = -1;
pub const RG_ERR_NO_DEVICE: i32 = -2;
pub const RG_ERR_OUT_OF_MEMORY: i32 = -3;
pub const RG_ERR_STEP_LOCAL_MSG: i32 = -4;
pub const RG_ERR_STEP_PEER_NOT_FOUND: i32 = -5;
pub const RG_ERR_SLOT_BUSY: i32 = -6;
pub const RG_ERR_HIGHER_TERM: i32 = -7;
pub const RG_ERR_STATE: i32 = -8;
pub const RG_ERR_NOT_ON_PATH: i32 = -9;
pub const RG_ERR_HOST_HINT: i32 = -10;

// rg_column
pub const RG_COL_MATCH: i32 = 0;
pub const RG_COL_NEXT: i32 = 1;
pub const RG_COL_PR_COMMIT: i32 = 2;
pub const RG_COL_PEND_SNAP: i32 = 3;
pub const RG_COL_PEND_RS: i32 = 4;
pub const RG_COL_GID: i32 = 5;
pub const RG_COL_PFLAGS: i32 = 6;
pub const RG_COL_COMMIT: i32 = 7;
pub const RG_COL_TERM_LO: i32 = 8;
pub const RG_COL_TERM_HI: i32 = 9;
pub const RG_COL_CFG: i32 = 10;
pub const RG_COL_OUT: i32 = 11;
pub const RG_COL_RUN_FIRST: i32 = 12;
pub const RG_COL_RUN_TERM: i32 = 13;
pub const RG_COL_DUMMY_INDEX: i32 = 14;
pub const RG_COL_DUMMY_TERM: i32 = 15;
pub const RG_COL_CUR_TERM: i32 = 16;
pub const RG_COL_HOST_HINT: i32 = 17;
pub const RG_COL_RUN_COUNT: i32 = 18;
pub const RG_COL_COUNT: i32 = 19;

pub enum RgEngine {} // opaque
pub type RgAllgatherFn = Option<unsafe extern "C" fn(*mut c_void, *const c_void, *mut c_void, u64, *mut c_void) -> i32>;

#[repr(C)]
pub struct RgMsgs {
    pub m_index: *const u64,
    pub m_commit: *const u64,
    pub m_hint: *const u64,
    pub m_rs: *const u64,
    pub m_flags: *const u8,
    pub m_logterm: *const u64,
}

#[repr(C)]
pub struct RgConfig {
    pub n_groups: u64,
    pub n_slots: u32,
    pub device: i32,
    pub variant: u32,
    pub max_inflight: u32,
    pub cache_policy: u32,
    pub flags: u32,
    pub cache_resident_groups: u64,
}

#[repr(C)]
pub struct RgDeviceInfo {
    pub arch: [c_char; 32],
    pub compute_units: u32,
    pub wavefront: u32,
    pub lds_per_workgroup: u64,
    pub hbm_bytes: u64,
    pub l2_bytes: u64,
    pub engine_bytes: u64,
    pub cache_policy: u32,
    pub engines_on_device: u32,
    pub resident_groups: u64,
    pub last_tick_kernel: u32,
    pub last_tick_streaming: u32,
    pub infinity_cache_bytes: u64,
    pub infinity_cache_queried: u32,
    pub last_tick_offset_bits: u32,
}

#[repr(C)]
pub struct RgGroupStatus {
    pub group: u64,
    pub commit: u64,
    pub term_lo: u64,
    pub last_index: u64,
    pub cfg: u32,
    pub out: u32,
    pub r#match: [u64; RG_MAX_SLOTS as usize],
    pub next: [u64; RG_MAX_SLOTS as usize],
    pub pr_commit: [u64; RG_MAX_SLOTS as usize],
    pub pend_snap: [u64; RG_MAX_SLOTS as usize],
    pub pend_rs: [u64; RG_MAX_SLOTS as usize],
    pub pflags: [u8; RG_MAX_SLOTS as usize],
    pub inflights: [u8; RG_MAX_SLOTS as usize],
}

#[repr(C)]
pub struct RgCellWrite {
    pub group: u64,
    pub slot: u32,
    pub field_mask: u32,
    pub r#match: u64,
    pub next: u64,
    pub pr_commit: u64,
    pub pend_snap: u64,
    pub pend_rs: u64,
    pub gid: u64,
    pub pflags: u8,
    pub pad: [u8; 7],
}

#[repr(C)]
pub struct RgProgressEvent {
    pub group: u64,
    pub slot: u32,
    pub kind: u32,
}

#[repr(C)]
pub struct RgSizeClass {
    pub first_group: u64,
    pub n_groups: u64,
    pub n_slots: u32,
    pub reserved: u32,
}

#[repr(C)]
pub struct RgHostHint {
    pub group: u64,
    pub slot_mask: u32,
    pub reserved: u32,
}

#[repr(C)]
pub struct RgResolvedHint {
    pub group: u64,
    pub index: u64,
    pub hint: u64,
    pub slot: u32,
    pub reserved: u32,
}

#[repr(C)]
pub struct RgAppendResponse {
    pub from: u64,
    pub term: u64,
    pub index: u64,
    pub commit: u64,
    pub reject_hint: u64,
    pub request_snapshot: u64,
    pub reject: u8,
    pub ins_full: u8,
    pub pad: [u8; 6],
    pub log_term: u64,
}

#[repr(C)]
pub struct RgDecodedMessage {
    pub msg_type: u32,
    pub reject: u32,
    pub to: u64,
    pub from: u64,
    pub term: u64,
    pub log_term: u64,
    pub index: u64,
    pub commit: u64,
    pub commit_term: u64,
    pub reject_hint: u64,
    pub request_snapshot: u64,
    pub priority: u64,
    pub n_entries: u64,
    pub has_snapshot: u32,
    pub context_len: u32,
}

#[repr(C)]
pub struct RgEntry {
    pub entry_type: u32,
    pub sync_log: u32,
    pub term: u64,
    pub index: u64,
    pub data: *const u8,
    pub data_len: u64,
    pub context: *const u8,
    pub context_len: u64,
}

#[repr(C)]
pub struct RgMessage {
    pub msg_type: u32,
    pub reject: u32,
    pub to: u64,
    pub from: u64,
    pub term: u64,
    pub log_term: u64,
    pub index: u64,
    pub commit: u64,
    pub commit_term: u64,
    pub reject_hint: u64,
    pub request_snapshot: u64,
    pub priority: u64,
    pub entries: *const RgEntry,
    pub n_entries: u64,
    pub snapshot: *const u8,
    pub snapshot_len: u64,
    pub context: *const u8,
    pub context_len: u64,
}

#[repr(C)]
pub struct RgSendItem {
    pub group: u64,
    pub prev_index: u64,
    pub last_index: u64,
    pub slot: u32,
    pub n_msgs: u16,
    pub kind: u16,
}

#[repr(C)]
pub struct RgLogSize {
    pub group: u64,
    pub index: u64,
    pub cum_bytes: u64,
}

#[repr(C)]
pub struct RgSentMsg {
    pub group: u64,
    pub last: u64,
    pub slot: u32,
    pub reserved: u32,
}

#[repr(C)]
pub struct RgWireMsg {
    pub group: u64,
    pub index: u64,
    pub commit: u64,
    pub hint: u64,
    pub rs: u64,
    pub log_term: u64,
    pub slot: u32,
    pub flags: u32,
    pub pad: u64,
}

#[repr(C)]
pub struct RgCommConfig {
    pub rank: u32,
    pub world: u32,
    pub unique_id: *const u8,
    pub transport: RgAllgatherFn,
    pub transport_user: *mut c_void,
    pub ring_ticks: u32,
    pub overflow_slots: u32,
}

#[repr(C)]
pub struct RgCommInfo {
    pub rank: u32,
    pub world: u32,
    pub transport: u32,
    pub in_process: u32,
    pub rccl_ranks: u32,
    pub rccl_rank: u32,
}

#[repr(C)]
pub struct RgCommAllConfig {
    pub ring_ticks: u32,
    pub overflow_slots: u32,
    pub transport: u32,
    pub reserved: u32,
}

#[repr(C)]
pub struct RgPublishStats {
    pub publications: u64,
    pub full_publications: u64,
    pub replica_updates: u64,
    pub bytes_per_rank_last: u64,
    pub bytes_per_rank_delta: u64,
    pub bytes_per_rank_full: u64,
    pub overflow_slots: u32,
    pub ring_ticks: u32,
    pub host_us_events: f64,
    pub host_us_allgather: f64,
    pub host_us_memset: f64,
    pub events_on_tick_packets: u64,
}

#[repr(C)]
pub struct RgWorkload {
    pub seed: u64,
    pub workload: u32,
    pub reserved: u32,
}

#[repr(C)]
pub struct RgHostState {
    pub n_groups: u64,
    pub stride: u64,
    pub n_slots: u32,
    pub reserved: u32,
    pub r#match: *mut u64,
    pub next: *mut u64,
    pub pr_commit: *mut u64,
    pub pend_snap: *mut u64,
    pub pend_rs: *mut u64,
    pub gid: *mut u64,
    pub pflags: *mut u8,
    pub commit: *mut u64,
    pub term_lo: *mut u64,
    pub term_hi: *mut u64,
    pub cfg: *mut u32,
}

extern "C" {
    pub fn rg_version() -> *const c_char;
    pub fn rg_abi_version() -> u32;
    pub fn rg_last_error() -> *const c_char;
    pub fn rg_device_count() -> i32;
    pub fn rg_create(cfg: *const RgConfig, out: *mut *mut RgEngine) -> i32;
    pub fn rg_destroy(h: *mut RgEngine);
    pub fn rg_stride(h: *const RgEngine) -> u64;
    pub fn rg_get_device_info(h: *const RgEngine, info: *mut RgDeviceInfo) -> i32;
    pub fn rg_set_stream(h: *mut RgEngine, hip_stream: *mut c_void) -> i32;
    pub fn rg_sync(h: *mut RgEngine) -> i32;
    pub fn rg_column_bytes(h: *const RgEngine, column: i32) -> u64;
    pub fn rg_load_column(h: *mut RgEngine, column: i32, host_src: *const c_void, bytes: u64) -> i32;
    pub fn rg_read_column(h: *mut RgEngine, column: i32, host_dst: *mut c_void, bytes: u64) -> i32;
    pub fn rg_column_ptr(h: *mut RgEngine, column: i32) -> *mut c_void;
    pub fn rg_checkpoint(h: *mut RgEngine) -> i32;
    pub fn rg_restore(h: *mut RgEngine) -> i32;
    pub fn rg_read_groups(h: *mut RgEngine, groups: *const u64, n: u64, host_out: *mut RgGroupStatus) -> i32;
    pub fn rg_write_cells(h: *mut RgEngine, cells: *const RgCellWrite, n: u64) -> i32;
    pub fn rg_set_config(h: *mut RgEngine, group: u64, cfg_word: u32) -> i32;
    pub fn rg_progress_events(h: *mut RgEngine, events: *const RgProgressEvent, n: u64) -> i32;
    pub fn rg_progress_event_dense(h: *mut RgEngine, kind: u32, host_slot_plus1: *const u8) -> i32;
    pub fn rg_size_classes(h: *mut RgEngine, out: *mut RgSizeClass, cap: u32, n: *mut u32) -> i32;
    pub fn rg_plan_placement(cfg_words: *const u32, n_groups: u64, n_slots: u32, perm: *mut u64, classes: *mut RgSizeClass, cap: u32, n_classes: *mut u32) -> i32;
    pub fn rg_permute_groups(h: *mut RgEngine, host_perm: *const u64) -> i32;
    pub fn rg_tick(h: *mut RgEngine, host_msgs: *const RgMsgs) -> i32;
    pub fn rg_tick_device(h: *mut RgEngine, dev_msgs: *const RgMsgs) -> i32;
    pub fn rg_tick_device_fused(h: *mut RgEngine, dev_msgs: *const RgMsgs, n_ticks: u32, dev_out_t: *mut u32, dev_commit_t: *mut u64) -> i32;
    pub fn rg_fused_ticks_done(h: *const RgEngine, n: *mut u32) -> i32;
    pub fn rg_recompute(h: *mut RgEngine) -> i32;
    pub fn rg_maximal_committed_index(h: *mut RgEngine, host_mci: *mut u64, host_used_gc: *mut u8) -> i32;
    pub fn rg_heartbeat_commits(h: *mut RgEngine, dev_hb_commit: *mut u64, host_hb_commit: *mut u64) -> i32;
    pub fn rg_results(h: *mut RgEngine, host_commit: *mut u64, host_out: *mut u32) -> i32;
    pub fn rg_result_counts(h: *mut RgEngine, n_changed: *mut u64, n_fault: *mut u64) -> i32;
    pub fn rg_host_hints(h: *mut RgEngine, host_items: *mut RgHostHint, cap: u64, n: *mut u64) -> i32;
    pub fn rg_resolve_host_hints(h: *mut RgEngine, items: *const RgResolvedHint, n: u64, host_applied: *mut u8) -> i32;
    pub fn rg_msg_stats(h: *mut RgEngine, dev_m_flags: *const u8, counts: *mut u64) -> i32;
    pub fn rg_vote_result(h: *mut RgEngine, host_yes: *const u8, host_no: *const u8, host_result: *mut u8) -> i32;
    pub fn rg_tally_votes(h: *mut RgEngine, host_yes: *const u8, host_no: *const u8, host_granted: *mut u8, host_rejected: *mut u8, host_result: *mut u8) -> i32;
    pub fn rg_quorum_recently_active(h: *mut RgEngine, host_result: *mut u8) -> i32;
    pub fn rg_set_peers(h: *mut RgEngine, group: u64, peer_ids: *const u64, n: u32, term: u64) -> i32;
    pub fn rg_step(h: *mut RgEngine, group: u64, m: *const RgAppendResponse) -> i32;
    pub fn rg_step_heartbeat_response(h: *mut RgEngine, group: u64, from: u64, term: u64, commit: u64, ins_full: u8) -> i32;
    pub fn rg_decode_message(bytes: *const u8, len: u64, out: *mut RgDecodedMessage) -> i32;
    pub fn rg_step_bytes(h: *mut RgEngine, group: u64, bytes: *const u8, len: u64, ins_full: u8) -> i32;
    pub fn rg_entry_size(e: *const RgEntry) -> u64;
    pub fn rg_limit_size(entries: *const RgEntry, n: u64, max_size: u64) -> u64;
    pub fn rg_message_size(m: *const RgMessage, len: *mut u64) -> i32;
    pub fn rg_encode_message(m: *const RgMessage, buf: *mut u8, cap: u64, len: *mut u64) -> i32;
    pub fn rg_local_append(h: *mut RgEngine, group: u64, new_last_index: u64) -> i32;
    pub fn rg_local_persisted(h: *mut RgEngine, group: u64, index: u64) -> i32;
    pub fn rg_mark_sent(h: *mut RgEngine, group: u64, peer_id: u64) -> i32;
    pub fn rg_local_become_leader(h: *mut RgEngine, group: u64, term: u64) -> i32;
    pub fn rg_report_unreachable(h: *mut RgEngine, group: u64, peer_id: u64) -> i32;
    pub fn rg_report_snapshot(h: *mut RgEngine, group: u64, peer_id: u64, failure: i32) -> i32;
    pub fn rg_flush(h: *mut RgEngine) -> i32;
    pub fn rg_send_appends(h: *mut RgEngine, max_entries_per_msg: u64, flags: u32) -> i32;
    pub fn rg_tick_send(h: *mut RgEngine, host_msgs: *const RgMsgs, max_entries_per_msg: u64, flags: u32) -> i32;
    pub fn rg_tick_device_send(h: *mut RgEngine, dev_msgs: *const RgMsgs, max_entries_per_msg: u64, flags: u32) -> i32;
    pub fn rg_log_sizes_enable(h: *mut RgEngine, window: u32) -> i32;
    pub fn rg_log_sizes_write(h: *mut RgEngine, recs: *const RgLogSize, n: u64) -> i32;
    pub fn rg_workload_sizes(h: *mut RgEngine, seed: u64, min_bytes: u32, spread: u32) -> i32;
    pub fn rg_update_state(h: *mut RgEngine, msgs: *const RgSentMsg, n: u64) -> i32;
    pub fn rg_flush_send(h: *mut RgEngine, max_entries_per_msg: u64, flags: u32) -> i32;
    pub fn rg_send_items(h: *mut RgEngine, host_items: *mut RgSendItem, cap: u64, n: *mut u64) -> i32;
    pub fn rg_send_items_ptr(h: *mut RgEngine) -> *const RgSendItem;
    pub fn rg_send_columns(h: *mut RgEngine, dev_prev_index: *mut *const u64, dev_last_index: *mut *const u64, dev_n_kind: *mut *const u32) -> i32;
    pub fn rg_send_tail_column(h: *mut RgEngine, dev_newest_inflight: *mut *const u64) -> i32;
    pub fn rg_inflights_bytes(h: *const RgEngine, ring: i32) -> u64;
    pub fn rg_read_inflights(h: *mut RgEngine, host_meta: *mut u32, host_ring: *mut u64) -> i32;
    pub fn rg_load_inflights(h: *mut RgEngine, host_meta: *const u32, host_ring: *const u64) -> i32;
    pub fn rg_mailbox_start(h: *mut RgEngine, idle_timeout_us: u32) -> i32;
    pub fn rg_mailbox_stop(h: *mut RgEngine) -> i32;
    pub fn rg_mailbox_stats(h: *const RgEngine, flushes_served: *mut u64, launches: *mut u64) -> i32;
    pub fn rg_ingest(h: *mut RgEngine, host_records: *const RgWireMsg, n: u64, n_duplicates: *mut u64) -> i32;
    pub fn rg_ingest_device(h: *mut RgEngine, dev_records: *const RgWireMsg, n: u64) -> i32;
    pub fn rg_ingested_duplicates(h: *mut RgEngine, n_duplicates: *mut u64) -> i32;
    pub fn rg_tick_ingested(h: *mut RgEngine, n_groups: *mut u64) -> i32;
    pub fn rg_ingest_tick(h: *mut RgEngine, host_records: *const RgWireMsg, n: u64, n_groups: *mut u64, n_duplicates: *mut u64) -> i32;
    pub fn rg_ingested_results(h: *mut RgEngine, groups: *mut u64, commit: *mut u64, out: *mut u32, cap: u64, n: *mut u64) -> i32;
    pub fn rg_comm_unique_id(id: *mut u8) -> i32;
    pub fn rg_comm_init(h: *mut RgEngine, cfg: *const RgCommConfig) -> i32;
    pub fn rg_comm_destroy(h: *mut RgEngine) -> i32;
    pub fn rg_comm_warmup() -> i32;
    pub fn rg_comm_info_get(h: *mut RgEngine, out: *mut RgCommInfo) -> i32;
    pub fn rg_comm_init_all(engines: *const *mut RgEngine, n: u32, cfg: *const RgCommAllConfig) -> i32;
    pub fn rg_publish_commit_all(engines: *const *mut RgEngine, n: u32, flags: u32) -> i32;
    pub fn rg_publish_commit(h: *mut RgEngine, flags: u32) -> i32;
    pub fn rg_publish_sync(h: *mut RgEngine) -> i32;
    pub fn rg_published_commit_ptr(h: *mut RgEngine, stride: *mut u64) -> *const u64;
    pub fn rg_published_commit(h: *mut RgEngine, rank: u32, first: u64, n: u64, host_commit: *mut u64) -> i32;
    pub fn rg_publish_stats_get(h: *mut RgEngine, out: *mut RgPublishStats) -> i32;
    pub fn rg_pub_bytes_per_rank(n_groups: u64, overflow_slots: u32) -> u64;
    pub fn rg_pub_accumulate_host(n_groups: u64, overflow_slots: u32, old_commit: *const u64, new_commit: *const u64, slice: *mut u8) -> i32;
    pub fn rg_pub_apply_host(n_groups: u64, overflow_slots: u32, world: u32, gathered: *const u8, replica: *mut u64, lost_ranks: *mut u32) -> i32;
    pub fn rg_workload_init(h: *mut RgEngine, w: *const RgWorkload, first_group_global: u64) -> i32;
    pub fn rg_workload_gen(h: *mut RgEngine, w: *const RgWorkload, first_group_global: u64, tick: u64, d_m_index: *mut u64, d_m_commit: *mut u64, d_m_hint: *mut u64, d_m_rs: *mut u64, d_m_flags: *mut u8) -> i32;
    pub fn rg_workload_init_host(w: *const RgWorkload, first_group_global: u64, s: *mut RgHostState) -> i32;
    pub fn rg_workload_gen_host(w: *const RgWorkload, first_group_global: u64, tick: u64, s: *const RgHostState, m_index: *mut u64, m_commit: *mut u64, m_hint: *mut u64, m_rs: *mut u64, m_flags: *mut u8) -> i32;
}

/// Error mapping back to raft::Error (src/errors.rs:6-50)
pub fn check(rc: i32) -> raft::Result<()> {
    match rc {
        0 => Ok(()),
        RG_ERR_STEP_LOCAL_MSG => Err(raft::Error::StepLocalMsg),
        RG_ERR_STEP_PEER_NOT_FOUND => Err(raft::Error::StepPeerNotFound), // checked BEFORE the term, as RawNode::step does
        _ => panic!("raftgroups: {}", unsafe { std::ffi::CStr::from_ptr(rg_last_error()) }.to_string_lossy()),
    }
}
