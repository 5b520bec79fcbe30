"""ctypes binding of the C ABI in include/raftgroups.h (raft_rs_amd/libraftgroups.so).

Names follow the reference's domain (groups, peers/slots, progress, commit), mirroring
ProgressTracker / RaftLog accessors where one exists (citations in include/raftgroups.h).
"""
import ctypes as C
import os

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RG_LIB_PATH") or os.path.join(PKG, "libraftgroups.so")  # RG_LIB_PATH: tuning builds


class EngineError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"raftgroups error {code}: {msg}")
        self.code = code


class COL:
    (MATCH, NEXT, PR_COMMIT, PEND_SNAP, PEND_RS, GID, PFLAGS, COMMIT, TERM_LO, TERM_HI, CFG, OUT, RUN_FIRST,
     RUN_TERM, DUMMY_INDEX, DUMMY_TERM, CUR_TERM, HOST_HINT, RUN_COUNT) = range(19)
    PER_SLOT = (0, 1, 2, 3, 4, 5)
    PER_RUN = (12, 13)
    NAMES = ("match", "next", "pr_commit", "pend_snap", "pend_rs", "gid", "pflags", "commit",
             "term_lo", "term_hi", "cfg", "out", "run_first", "run_term", "dummy_index", "dummy_term", "cur_term",
             "host_hint", "run_count")


TERM_RUNS = 8


class PF:
    STATE_MASK, PROBE, REPLICATE, SNAPSHOT, PAUSED, RECENT_ACTIVE, INS_FULL, PENDING_CONF = 0x3, 0, 1, 2, 0x4, 0x8, 0x10, 0x20


class MF:
    VALID, REJECT, HAS_RS, INS_FULL, SENT, APPEND, HEARTBEAT, HAS_LOGTERM = 0x01, 0x02, 0x04, 0x08, 0x10, 0x20, 0x40, 0x80
    BECOME_LEADER = 0x02  # on the leader's own slot (new term in m_hint)


class OUT:
    CHANGED, FAULT, TIMEOUT_NOW, APPENDED, BECAME_LEADER, HOST_HINT = 0x1, 0x2, 0x4, 0x8, 0x10, 0x20

    @staticmethod
    def send_append(o):
        return (int(o) >> 8) & 0xff

    @staticmethod
    def send_more(o):
        return (int(o) >> 16) & 0xff

    @staticmethod
    def free_to(o):
        return (int(o) >> 24) & 0xff


ERR = {"NOT_ON_PATH": -9, "INVALID_ARG": -1, "NO_DEVICE": -2, "OUT_OF_MEMORY": -3, "STEP_LOCAL_MSG": -4,
       "STEP_PEER_NOT_FOUND": -5, "SLOT_BUSY": -6, "HIGHER_TERM": -7, "STATE": -8, "HOST_HINT": -10}

WL_MAJORITY, WL_JOINT, WL_MIXED = 2, 3, 5
VARIANT_DEFAULT, VARIANT_LANE, VARIANT_LDS, VARIANT_COOP, VARIANT_LDS_DMA, VARIANT_COMPACT = 0, 1, 2, 3, 4, 5


def cfg_make(incoming, outgoing=0, self_slot=0, group_commit=False, transferee_plus1=0, present=None):
    """RG_CFG_MAKE of include/raftgroups.h."""
    if present is None:
        present = incoming | outgoing
    return ((incoming & 0xff) | ((outgoing & 0xff) << 8) | ((self_slot & 7) << 16) |
            (0x00080000 if group_commit else 0) | ((transferee_plus1 & 0xf) << 20) | ((present & 0xff) << 24))


class _Config(C.Structure):
    _fields_ = [("n_groups", C.c_uint64), ("n_slots", C.c_uint32), ("device", C.c_int32),
                ("variant", C.c_uint32), ("max_inflight", C.c_uint32),
                ("cache_policy", C.c_uint32), ("flags", C.c_uint32), ("cache_resident_groups", C.c_uint64)]


class CACHE:
    """rg_config.cache_policy (include/raftgroups.h: RG_CACHE_*)."""
    AUTO, PLAIN, STREAM_MSGS, STREAM_ALL, RESIDENT = range(5)
    NAMES = ("auto", "plain", "stream_msgs", "stream_all", "resident")


class CFGF:
    """rg_config.flags (RG_CFGF_*)."""
    NO_SIZE_CLASSES, CLASS_BLOCK_ORDER, IX64 = 0x1, 0x2, 0x4


class KERNEL:
    """rg_device_info.last_tick_kernel (RG_KERNEL_*)."""
    NONE, LANE, CLASSES, SPLIT, LDS, COMPACT, TICK_SEND = range(7)
    NAMES = ("none", "k_tick_lane", "k_tick_classes", "k_tick_split", "k_tick_lds", "k_tick_compact", "k_tick_send")


HOST_HINT_DTYPE = np.dtype([("group", "<u8"), ("slot_mask", "<u4"), ("reserved", "<u4")])
RESOLVED_HINT_DTYPE = np.dtype([("group", "<u8"), ("index", "<u8"), ("hint", "<u8"), ("slot", "<u4"), ("reserved", "<u4")])
GROUP_STATUS_DTYPE = np.dtype([("group", "<u8"), ("commit", "<u8"), ("term_lo", "<u8"), ("last_index", "<u8"),
                               ("cfg", "<u4"), ("out", "<u4"), ("match", "<u8", 8), ("next", "<u8", 8),
                               ("pr_commit", "<u8", 8), ("pend_snap", "<u8", 8), ("pend_rs", "<u8", 8),
                               ("pflags", "u1", 8), ("inflights", "u1", 8)])
assert GROUP_STATUS_DTYPE.itemsize == 376


class DeviceInfo(C.Structure):
    _fields_ = [("arch", C.c_char * 32), ("compute_units", C.c_uint32), ("wavefront", C.c_uint32),
                ("lds_per_workgroup", C.c_uint64), ("hbm_bytes", C.c_uint64), ("l2_bytes", C.c_uint64),
                ("engine_bytes", C.c_uint64), ("cache_policy", C.c_uint32), ("engines_on_device", C.c_uint32),
                ("resident_groups", C.c_uint64), ("last_tick_kernel", C.c_uint32), ("last_tick_streaming", C.c_uint32),
                ("infinity_cache_bytes", C.c_uint64), ("infinity_cache_queried", C.c_uint32), ("last_tick_offset_bits", C.c_uint32)]


import contextlib


@contextlib.contextmanager
def config_defaults(**kw):
    """Every Engine created inside the block gets these rg_config fields unless it names them itself."""
    old = dict(Engine._defaults)
    Engine._defaults.update(kw)
    try:
        yield
    finally:
        Engine._defaults.clear()
        Engine._defaults.update(old)


class _Msgs(C.Structure):
    _fields_ = [("m_index", C.c_void_p), ("m_commit", C.c_void_p), ("m_hint", C.c_void_p),
                ("m_rs", C.c_void_p), ("m_flags", C.c_void_p), ("m_logterm", C.c_void_p)]


class _Workload(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("workload", C.c_uint32), ("reserved", C.c_uint32)]


class _HostState(C.Structure):
    _fields_ = [("n_groups", C.c_uint64), ("stride", C.c_uint64), ("n_slots", C.c_uint32),
                ("reserved", C.c_uint32)] + \
               [(n, C.c_void_p) for n in ("match", "next", "pr_commit", "pend_snap", "pend_rs", "gid",
                                          "pflags", "commit", "term_lo", "term_hi", "cfg")]


class AppendResponse(C.Structure):
    _fields_ = [("from_", C.c_uint64), ("term", C.c_uint64), ("index", C.c_uint64), ("commit", C.c_uint64),
                ("reject_hint", C.c_uint64), ("request_snapshot", C.c_uint64), ("reject", C.c_uint8),
                ("ins_full", C.c_uint8), ("pad", C.c_uint8 * 6), ("log_term", C.c_uint64)]


class DecodedMessage(C.Structure):
    """rg_decoded_message: the fields of one protobuf-encoded eraftpb::Message (rg_decode_message)."""
    _fields_ = [("msg_type", C.c_uint32), ("reject", C.c_uint32), ("to", C.c_uint64), ("from_", C.c_uint64),
                ("term", C.c_uint64), ("log_term", C.c_uint64), ("index", C.c_uint64), ("commit", C.c_uint64),
                ("commit_term", C.c_uint64), ("reject_hint", C.c_uint64), ("request_snapshot", C.c_uint64),
                ("priority", C.c_uint64), ("n_entries", C.c_uint64), ("has_snapshot", C.c_uint32),
                ("context_len", C.c_uint32)]


class EntryC(C.Structure):
    """rg_entry: one eraftpb::Entry handed to the encoder (rg_entry_size / rg_limit_size / rg_encode_message)."""
    _fields_ = [("entry_type", C.c_uint32), ("sync_log", C.c_uint32), ("term", C.c_uint64), ("index", C.c_uint64),
                ("data", C.c_char_p), ("data_len", C.c_uint64), ("context", C.c_char_p), ("context_len", C.c_uint64)]


class MessageC(C.Structure):
    """rg_message: an outgoing eraftpb::Message (rg_message_size / rg_encode_message)."""
    _fields_ = [("msg_type", C.c_uint32), ("reject", C.c_uint32), ("to", C.c_uint64), ("from_", C.c_uint64),
                ("term", C.c_uint64), ("log_term", C.c_uint64), ("index", C.c_uint64), ("commit", C.c_uint64),
                ("commit_term", C.c_uint64), ("reject_hint", C.c_uint64), ("request_snapshot", C.c_uint64),
                ("priority", C.c_uint64), ("entries", C.POINTER(EntryC)), ("n_entries", C.c_uint64),
                ("snapshot", C.c_char_p), ("snapshot_len", C.c_uint64), ("context", C.c_char_p), ("context_len", C.c_uint64)]


class WireMsg(C.Structure):
    _fields_ = [("group", C.c_uint64), ("index", C.c_uint64), ("commit", C.c_uint64), ("hint", C.c_uint64),
                ("rs", C.c_uint64), ("log_term", C.c_uint64), ("slot", C.c_uint32), ("flags", C.c_uint32),
                ("pad", C.c_uint64)]


WIRE_DTYPE = np.dtype([("group", "<u8"), ("index", "<u8"), ("commit", "<u8"), ("hint", "<u8"), ("rs", "<u8"),
                       ("log_term", "<u8"), ("slot", "<u4"), ("flags", "<u4"), ("pad", "<u8")])
assert WIRE_DTYPE.itemsize == 64


class CellWrite(C.Structure):
    _fields_ = [("group", C.c_uint64), ("slot", C.c_uint32), ("field_mask", C.c_uint32),
                ("match", C.c_uint64), ("next", C.c_uint64), ("pr_commit", C.c_uint64),
                ("pend_snap", C.c_uint64), ("pend_rs", C.c_uint64), ("gid", C.c_uint64),
                ("pflags", C.c_uint8), ("pad", C.c_uint8 * 7)]


ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)


class CommConfig(C.Structure):
    _fields_ = [("rank", C.c_uint32), ("world", C.c_uint32), ("unique_id", C.c_void_p), ("transport", ALLGATHER_FN),
                ("transport_user", C.c_void_p), ("ring_ticks", C.c_uint32), ("overflow_slots", C.c_uint32)]


class PublishStats(C.Structure):
    _fields_ = [("publications", C.c_uint64), ("full_publications", C.c_uint64), ("replica_updates", C.c_uint64),
                ("bytes_per_rank_last", C.c_uint64), ("bytes_per_rank_delta", C.c_uint64),
                ("bytes_per_rank_full", C.c_uint64), ("overflow_slots", C.c_uint32), ("ring_ticks", C.c_uint32),
                ("host_us_events", C.c_double), ("host_us_allgather", C.c_double), ("host_us_memset", C.c_double),
                ("events_on_tick_packets", C.c_uint64)]


ABI_VERSION = 7  # RG_ABI_VERSION of include/raftgroups.h these ctypes layouts mirror (tests/test_abi.py compares)
class CommInfo(C.Structure):
    _fields_ = [("rank", C.c_uint32), ("world", C.c_uint32), ("transport", C.c_uint32), ("in_process", C.c_uint32),
                ("rccl_ranks", C.c_uint32), ("rccl_rank", C.c_uint32)]


TRANSPORT_NAMES = {0: "none", 1: "rccl", 2: "callback", 3: "local"}
COMM_ID_BYTES = 128
PUBLISH_FULL = 1

SEND_APPEND, SEND_SNAPSHOT, SEND_HOST = 1, 2, 3
SEND_SKIP_BCAST_COMMIT = 1
SEND_BYTES = 2
NO_LIMIT = (1 << 64) - 1
LOG_SIZE_DTYPE = np.dtype([("group", "<u8"), ("index", "<u8"), ("cum_bytes", "<u8")])
SENT_MSG_DTYPE = np.dtype([("group", "<u8"), ("last", "<u8"), ("slot", "<u4"), ("reserved", "<u4")])
PROGRESS_EVENT_DTYPE = np.dtype([("group", "<u8"), ("slot", "<u4"), ("kind", "<u4")])  # rg_progress_event
EV_UNREACHABLE, EV_SNAPSHOT_FINISH, EV_SNAPSHOT_FAILURE = 1, 2, 3
SEND_ITEM_DTYPE = np.dtype([("group", "<u8"), ("prev_index", "<u8"), ("last_index", "<u8"), ("slot", "<u4"),
                            ("n_msgs", "<u2"), ("kind", "<u2")])
assert SEND_ITEM_DTYPE.itemsize == 32

# every symbol include/raftgroups.h declares: (restype, argtypes)
_vp, _u64, _i = C.c_void_p, C.c_uint64, C.c_int
SYMBOLS = {
    "rg_version": (C.c_char_p, []),
    "rg_abi_version": (C.c_uint32, []),
    "rg_last_error": (C.c_char_p, []),
    "rg_device_count": (_i, []),
    "rg_create": (_i, [C.POINTER(_Config), C.POINTER(_vp)]),
    "rg_destroy": (None, [_vp]),
    "rg_stride": (_u64, [_vp]),
    "rg_get_device_info": (_i, [_vp, C.POINTER(DeviceInfo)]),
    "rg_set_stream": (_i, [_vp, _vp]),
    "rg_sync": (_i, [_vp]),
    "rg_column_bytes": (_u64, [_vp, _i]),
    "rg_load_column": (_i, [_vp, _i, _vp, _u64]),
    "rg_read_column": (_i, [_vp, _i, _vp, _u64]),
    "rg_column_ptr": (_vp, [_vp, _i]),
    "rg_checkpoint": (_i, [_vp]),
    "rg_restore": (_i, [_vp]),
    "rg_read_groups": (_i, [_vp, _vp, _u64, _vp]),
    "rg_write_cells": (_i, [_vp, C.POINTER(CellWrite), _u64]),
    "rg_set_config": (_i, [_vp, _u64, C.c_uint32]),
    "rg_tick": (_i, [_vp, C.POINTER(_Msgs)]),
    "rg_tick_device": (_i, [_vp, C.POINTER(_Msgs)]),
    "rg_tick_device_fused": (_i, [_vp, C.POINTER(_Msgs), C.c_uint32, _vp, _vp]),
    "rg_fused_ticks_done": (_i, [_vp, C.POINTER(C.c_uint32)]),
    "rg_recompute": (_i, [_vp]),
    "rg_maximal_committed_index": (_i, [_vp, _vp, _vp]),
    "rg_results": (_i, [_vp, _vp, _vp]),
    "rg_heartbeat_commits": (_i, [_vp, _vp, _vp]),
    "rg_step_heartbeat_response": (_i, [_vp, _u64, _u64, _u64, _u64, C.c_uint8]),
    "rg_result_counts": (_i, [_vp, C.POINTER(_u64), C.POINTER(_u64)]),
    "rg_plan_placement": (_i, [_vp, _u64, C.c_uint32, _vp, _vp, C.c_uint32, C.POINTER(C.c_uint32)]),
    "rg_permute_groups": (_i, [_vp, _vp]),
    "rg_size_classes": (_i, [_vp, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]),
    "rg_host_hints": (_i, [_vp, C.c_void_p, _u64, C.POINTER(_u64)]),
    "rg_resolve_host_hints": (_i, [_vp, C.c_void_p, _u64, C.c_void_p]),
    "rg_msg_stats": (_i, [_vp, _vp, C.POINTER(_u64 * 5)]),
    "rg_ingest": (_i, [_vp, _vp, _u64, C.POINTER(_u64)]),
    "rg_ingest_device": (_i, [_vp, _vp, _u64]),
    "rg_ingested_duplicates": (_i, [_vp, C.POINTER(_u64)]),
    "rg_tick_ingested": (_i, [_vp, C.POINTER(_u64)]),
    "rg_ingest_tick": (_i, [_vp, _vp, _u64, C.POINTER(_u64), C.POINTER(_u64)]),
    "rg_ingested_results": (_i, [_vp, _vp, _vp, _vp, _u64, C.POINTER(_u64)]),
    "rg_send_appends": (_i, [_vp, _u64, C.c_uint32]),
    "rg_tick_send": (_i, [_vp, _vp, _u64, C.c_uint32]),
    "rg_tick_device_send": (_i, [_vp, _vp, _u64, C.c_uint32]),
    "rg_flush_send": (_i, [_vp, _u64, C.c_uint32]),
    "rg_mailbox_start": (_i, [_vp, C.c_uint32]),
    "rg_mailbox_stop": (_i, [_vp]),
    "rg_mailbox_stats": (_i, [_vp, C.POINTER(_u64), C.POINTER(_u64)]),
    "rg_log_sizes_enable": (_i, [_vp, C.c_uint32]),
    "rg_log_sizes_write": (_i, [_vp, _vp, _u64]),
    "rg_workload_sizes": (_i, [_vp, _u64, C.c_uint32, C.c_uint32]),
    "rg_update_state": (_i, [_vp, _vp, _u64]),
    "rg_send_items": (_i, [_vp, _vp, _u64, C.POINTER(_u64)]),
    "rg_send_items_ptr": (_vp, [_vp]),
    "rg_send_columns": (_i, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    "rg_send_tail_column": (_i, [_vp, C.POINTER(_vp)]),
    "rg_inflights_bytes": (_u64, [_vp, _i]),
    "rg_read_inflights": (_i, [_vp, _vp, _vp]),
    "rg_load_inflights": (_i, [_vp, _vp, _vp]),
    "rg_vote_result": (_i, [_vp, _vp, _vp, _vp]),
    "rg_tally_votes": (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "rg_quorum_recently_active": (_i, [_vp, _vp]),
    "rg_set_peers": (_i, [_vp, _u64, C.POINTER(_u64), C.c_uint32, _u64]),
    "rg_step": (_i, [_vp, _u64, C.POINTER(AppendResponse)]),
    "rg_step_bytes": (_i, [_vp, _u64, C.c_char_p, _u64, C.c_uint8]),
    "rg_decode_message": (_i, [C.c_char_p, _u64, C.POINTER(DecodedMessage)]),
    "rg_progress_events": (_i, [_vp, C.c_void_p, _u64]),
    "rg_progress_event_dense": (_i, [_vp, C.c_uint32, C.c_void_p]),
    "rg_report_unreachable": (_i, [_vp, _u64, _u64]),
    "rg_report_snapshot": (_i, [_vp, _u64, _u64, _i]),
    "rg_entry_size": (_u64, [C.POINTER(EntryC)]),
    "rg_limit_size": (_u64, [C.POINTER(EntryC), _u64, _u64]),
    "rg_message_size": (_i, [C.POINTER(MessageC), C.POINTER(C.c_uint64)]),
    "rg_encode_message": (_i, [C.POINTER(MessageC), C.c_char_p, _u64, C.POINTER(C.c_uint64)]),
    "rg_local_append": (_i, [_vp, _u64, _u64]),
    "rg_local_persisted": (_i, [_vp, _u64, _u64]),
    "rg_mark_sent": (_i, [_vp, _u64, _u64]),
    "rg_local_become_leader": (_i, [_vp, _u64, _u64]),
    "rg_flush": (_i, [_vp]),
    "rg_comm_unique_id": (_i, [_vp]),
    "rg_comm_init": (_i, [_vp, C.POINTER(CommConfig)]),
    "rg_comm_destroy": (_i, [_vp]),
    "rg_comm_warmup": (_i, []),
    "rg_comm_info_get": (_i, [_vp, C.POINTER(CommInfo)]),
    "rg_publish_commit": (_i, [_vp, C.c_uint32]),
    "rg_comm_init_all": (_i, [_vp, C.c_uint32, _vp]),
    "rg_publish_commit_all": (_i, [_vp, C.c_uint32, C.c_uint32]),
    "rg_publish_sync": (_i, [_vp]),
    "rg_published_commit_ptr": (_vp, [_vp, C.POINTER(_u64)]),
    "rg_published_commit": (_i, [_vp, C.c_uint32, _u64, _u64, _vp]),
    "rg_publish_stats_get": (_i, [_vp, C.POINTER(PublishStats)]),
    "rg_pub_bytes_per_rank": (_u64, [_u64, C.c_uint32]),
    "rg_pub_accumulate_host": (_i, [_u64, C.c_uint32, _vp, _vp, _vp]),
    "rg_pub_apply_host": (_i, [_u64, C.c_uint32, C.c_uint32, _vp, _vp, C.POINTER(C.c_uint32)]),
    "rg_workload_init": (_i, [_vp, C.POINTER(_Workload), _u64]),
    "rg_workload_gen": (_i, [_vp, C.POINTER(_Workload), _u64, _u64, _vp, _vp, _vp, _vp, _vp]),
    "rg_workload_init_host": (_i, [C.POINTER(_Workload), _u64, C.POINTER(_HostState)]),
    "rg_workload_gen_host": (_i, [C.POINTER(_Workload), _u64, _u64, C.POINTER(_HostState), _vp, _vp, _vp, _vp, _vp]),
}

_lib = None


def load_library():
    """dlopen the in-tree HIP engine. Fails loudly when it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -m raft_rs_amd.build` "
                          "(hipcc, gfx950). The engine has no CPU fallback.")
    try:
        # torch ships its own HIP runtime; load it FIRST so the engine binds to the same runtime
        # instance (streams handed over with rg_set_stream must belong to it).
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if L.rg_abi_version() != ABI_VERSION:  # (the ABI is source-compatible only: struct layouts follow the header)
        raise ImportError(f"{LIB_PATH} was built with RG_ABI_VERSION {L.rg_abi_version()}, these bindings are written "
                          f"against {ABI_VERSION}: rebuild with `python -m raft_rs_amd.build --force`")
    _lib = L
    return L


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return a.ctypes.data
    return int(a)


class MsgBuffers:
    """Host-side message columns of one tick (layout of rg_msgs)."""

    def __init__(self, n_groups, n_slots, stride):
        self.n_groups, self.n_slots, self.stride = n_groups, n_slots, stride
        self.m_index = np.zeros((n_slots, stride), dtype=np.uint64)
        self.m_commit = np.zeros((n_slots, stride), dtype=np.uint64)
        self.m_hint = np.zeros((n_slots, stride), dtype=np.uint64)
        self.m_rs = np.zeros((n_slots, stride), dtype=np.uint64)
        self.m_logterm = np.zeros((n_slots, stride), dtype=np.uint64)
        self.m_flags = np.zeros((n_groups, 8), dtype=np.uint8)

    def clear(self):
        for a in (self.m_index, self.m_commit, self.m_hint, self.m_rs, self.m_logterm, self.m_flags):
            a[...] = 0

    def as_dict(self):
        return {"n_groups": self.n_groups, "n_slots": self.n_slots, "stride": self.stride,
                "m_index": self.m_index, "m_commit": self.m_commit, "m_hint": self.m_hint,
                "m_rs": self.m_rs, "m_logterm": self.m_logterm, "m_flags": self.m_flags}


class Engine:
    """One shard of raft groups resident on one MI355X (rg_engine)."""

    # what Engine() passes for the rg_config fields a caller leaves out (config_defaults: the tests run whole suites of
    # engines under one cache policy / offset width this way -- through the config struct, not through the environment)
    _defaults = {"cache_policy": CACHE.AUTO, "flags": 0, "cache_resident_groups": 0}

    def __init__(self, n_groups, n_slots, device=0, variant=VARIANT_DEFAULT, max_inflight=0, cache_policy=None,
                 flags=None, cache_resident_groups=None):
        """max_inflight > 0: Inflights rings of that capacity live on the device (send_appends after each tick).
        cache_policy / cache_resident_groups / flags: rg_config's (CACHE.*, CFGF.*); device_info() reports the decision."""
        self.L = load_library()
        self.h = _vp()
        d = Engine._defaults
        from_default = cache_policy is None
        cache_policy = d["cache_policy"] if cache_policy is None else cache_policy
        flags = d["flags"] if flags is None else flags
        cache_resident_groups = d["cache_resident_groups"] if cache_resident_groups is None else cache_resident_groups
        if max_inflight and cache_policy == CACHE.RESIDENT and from_default:
            cache_policy = CACHE.STREAM_ALL  # (a suite-wide default the send-stage engines cannot take: rg_create refuses it)
        cfg = _Config(n_groups, n_slots, device, variant, max_inflight, cache_policy, flags, cache_resident_groups)
        self._check(self.L.rg_create(C.byref(cfg), C.byref(self.h)))
        self.n_groups, self.n_slots, self.device, self.max_inflight = n_groups, n_slots, device, max_inflight
        self.stride = self.L.rg_stride(self.h)

    def close(self):
        if getattr(self, "h", None):
            self.L.rg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise EngineError(rc, self.L.rg_last_error().decode())
        return rc

    # ---- plumbing --------------------------------------------------------------------------
    def set_stream(self, stream_handle):
        self._check(self.L.rg_set_stream(self.h, _vp(stream_handle)))

    def sync(self):
        self._check(self.L.rg_sync(self.h))

    def device_info(self):
        """hipGetDeviceProperties of the engine's device + the device memory the engine holds."""
        d = DeviceInfo()
        self._check(self.L.rg_get_device_info(self.h, C.byref(d)))
        return {"arch": d.arch.decode(), "compute_units": d.compute_units, "wavefront": d.wavefront,
                "lds_per_workgroup": d.lds_per_workgroup, "hbm_bytes": d.hbm_bytes, "l2_bytes": d.l2_bytes,
                "engine_bytes": d.engine_bytes, "cache_policy": CACHE.NAMES[d.cache_policy],
                "engines_on_device": d.engines_on_device, "resident_groups": d.resident_groups,
                "last_tick_kernel": KERNEL.NAMES[d.last_tick_kernel], "last_tick_streaming": d.last_tick_streaming,
                "infinity_cache_bytes": d.infinity_cache_bytes, "infinity_cache_queried": bool(d.infinity_cache_queried),
                "last_tick_offset_bits": d.last_tick_offset_bits}

    def column_shape_dtype(self, col):
        if col in COL.PER_SLOT:
            return (self.n_slots, self.stride), np.uint64
        if col in COL.PER_RUN:
            return (TERM_RUNS, self.stride), np.uint64
        if col == COL.PFLAGS:
            return (self.n_groups, 8), np.uint8
        if col in (COL.CFG, COL.OUT):
            return (self.n_groups,), np.uint32
        if col in (COL.HOST_HINT, COL.RUN_COUNT):
            return (self.n_groups,), np.uint8
        return (self.n_groups,), np.uint64

    def load_column(self, col, arr):
        shape, dt = self.column_shape_dtype(col)
        a = np.ascontiguousarray(arr, dtype=dt)
        assert a.shape == shape, (COL.NAMES[col], a.shape, shape)
        self._check(self.L.rg_load_column(self.h, col, a.ctypes.data, a.nbytes))

    def read_column(self, col):
        shape, dt = self.column_shape_dtype(col)
        a = np.empty(shape, dtype=dt)
        self._check(self.L.rg_read_column(self.h, col, a.ctypes.data, a.nbytes))
        return a

    def column_ptr(self, col):
        return self.L.rg_column_ptr(self.h, col)

    def load_state(self, st):
        """st: dict with the column names of COL.NAMES[:11] (numpy arrays shaped as column_shape_dtype says)."""
        for col in range(COL.CFG + 1):
            self.load_column(col, st[COL.NAMES[col]])
        for col in (COL.RUN_FIRST, COL.RUN_TERM, COL.DUMMY_INDEX, COL.DUMMY_TERM, COL.CUR_TERM):  # optional term-run table
            if COL.NAMES[col] in st:
                self.load_column(col, st[COL.NAMES[col]])

    def read_state(self):
        st = {"n_groups": self.n_groups, "n_slots": self.n_slots, "stride": self.stride}
        for col in range(COL.OUT + 1):
            st[COL.NAMES[col]] = self.read_column(col)
        return st

    def read_groups(self, groups):
        """Status of whole groups (all Progress cells, commit, log range, cfg, last result word) as a
        GROUP_STATUS_DTYPE array -- the sparse counterpart of read_state()."""
        ids = np.ascontiguousarray(groups, dtype=np.uint64)
        out = np.zeros(len(ids), dtype=GROUP_STATUS_DTYPE)
        self._check(self.L.rg_read_groups(self.h, ids.ctypes.data, len(ids), out.ctypes.data))
        return out

    def checkpoint(self):
        self._check(self.L.rg_checkpoint(self.h))

    def restore(self):
        self._check(self.L.rg_restore(self.h))

    def write_cells(self, cells):
        """cells: list of dicts {group, slot, <column name>: value, ...}."""
        arr = (CellWrite * max(1, len(cells)))()
        for i, c in enumerate(cells):
            arr[i].group, arr[i].slot = c["group"], c["slot"]
            mask = 0
            for col in range(COL.PFLAGS + 1):
                name = COL.NAMES[col]
                if name in c:
                    mask |= 1 << col
                    setattr(arr[i], name, int(c[name]))
            arr[i].field_mask = mask
        self._check(self.L.rg_write_cells(self.h, arr, len(cells)))

    def set_config(self, group, cfg_word):
        """apply_conf for one group: rewrite its RG_CFG_* word."""
        self._check(self.L.rg_set_config(self.h, group, int(cfg_word)))

    # ---- hot path --------------------------------------------------------------------------
    def tick(self, msgs):
        """Host-buffer tick (H2D copy + kernel + sync)."""
        m = _Msgs(_ptr(msgs.m_index), _ptr(msgs.m_commit), _ptr(msgs.m_hint), _ptr(msgs.m_rs), _ptr(msgs.m_flags),
                  _ptr(getattr(msgs, "m_logterm", None)))
        self._check(self.L.rg_tick(self.h, C.byref(m)))

    def tick_device(self, m_index, m_commit, m_hint, m_rs, m_flags, m_logterm=None):
        """Device-pointer tick (asynchronous on the engine's stream)."""
        m = _Msgs(_ptr(m_index), _ptr(m_commit), _ptr(m_hint), _ptr(m_rs), _ptr(m_flags), _ptr(m_logterm))
        self._check(self.L.rg_tick_device(self.h, C.byref(m)))

    def ingest(self, records):
        """records: numpy structured array of WIRE_DTYPE (wire-order AppendResponse records). Returns the
        number of dropped duplicate / malformed records."""
        rec = np.ascontiguousarray(records, dtype=WIRE_DTYPE)
        dup = _u64(0)
        self._check(self.L.rg_ingest(self.h, rec.ctypes.data, len(rec), C.byref(dup)))
        return dup.value

    def ingest_device(self, dev_records_ptr, n):
        self._check(self.L.rg_ingest_device(self.h, _ptr(dev_records_ptr), n))

    def ingested_duplicates(self):
        d = _u64(0)
        self._check(self.L.rg_ingested_duplicates(self.h, C.byref(d)))
        return d.value

    def tick_ingested(self):
        n = _u64(0)
        self._check(self.L.rg_tick_ingested(self.h, C.byref(n)))
        return n.value

    def ingest_tick(self, records):
        """ingest + tick_ingested in one host<->device round trip. Returns (groups touched, dropped records)."""
        rec = np.ascontiguousarray(records, dtype=WIRE_DTYPE)
        n, dup = _u64(0), _u64(0)
        self._check(self.L.rg_ingest_tick(self.h, rec.ctypes.data, len(rec), C.byref(n), C.byref(dup)))
        return n.value, dup.value

    def ingested_results(self):
        n = _u64(0)
        self._check(self.L.rg_ingested_results(self.h, None, None, None, 0, C.byref(n)))
        k = n.value
        groups = np.empty(k, dtype=np.uint64)
        commit = np.empty(k, dtype=np.uint64)
        out = np.empty(k, dtype=np.uint32)
        if k:
            self._check(self.L.rg_ingested_results(self.h, groups.ctypes.data, commit.ctypes.data, out.ctypes.data, k,
                                                   C.byref(n)))
        return groups, commit, out

    def tick_device_fused(self, ticks, dev_out_t, dev_commit_t=None):
        """ticks: list (1..8) of (m_index, m_commit, m_hint, m_rs, m_flags) DEVICE pointers in tick order;
        dev_out_t: device u32 [T][G]; dev_commit_t: optional device u64 [T][G]. Asynchronous. Returns the number of ticks
        applied: len(ticks), or fewer when a log-term tick raised RG_OUT_HOST_HINT and the call stopped behind it
        (RG_ERR_HOST_HINT; answer the hints, then submit the rest)."""
        arr = (_Msgs * len(ticks))()
        for i, t in enumerate(ticks):
            arr[i] = _Msgs(*([_ptr(x) for x in t] + [None] * (6 - len(t))))
        rc = self.L.rg_tick_device_fused(self.h, arr, len(ticks), _ptr(dev_out_t), _ptr(dev_commit_t))
        if rc != ERR["HOST_HINT"]:
            self._check(rc)
        return self.fused_ticks_done()

    def fused_ticks_done(self):
        n = C.c_uint32(0)
        self._check(self.L.rg_fused_ticks_done(self.h, C.byref(n)))
        return int(n.value)

    # ---- send stage (device Inflights + maybe_send_append decisions) ------------------------------
    def send_appends(self, max_entries_per_msg=0, skip_bcast_commit=False, max_bytes=None):
        """Run the send stage for the tick that just ran (asynchronous). max_bytes: Config::max_size_per_msg in bytes over
        the entry sizes on the device (RG_SEND_BYTES; NO_LIMIT = no limit) instead of the entries-per-message model."""
        flags = SEND_SKIP_BCAST_COMMIT if skip_bcast_commit else 0
        if max_bytes is not None:
            max_entries_per_msg, flags = max_bytes, flags | SEND_BYTES
        self._check(self.L.rg_send_appends(self.h, max_entries_per_msg, flags))

    def tick_send(self, msgs, max_entries_per_msg=0, skip_bcast_commit=False, max_bytes=None):
        """tick(msgs) + send_appends(...) as ONE launch (k_tick_send); host buffers, synchronises like tick()."""
        flags = SEND_SKIP_BCAST_COMMIT if skip_bcast_commit else 0
        if max_bytes is not None:
            max_entries_per_msg, flags = max_bytes, flags | SEND_BYTES
        m = _Msgs(_ptr(msgs.m_index), _ptr(msgs.m_commit), _ptr(msgs.m_hint), _ptr(msgs.m_rs), _ptr(msgs.m_flags),
                  _ptr(getattr(msgs, "m_logterm", None)))
        self._check(self.L.rg_tick_send(self.h, C.byref(m), max_entries_per_msg, flags))

    def tick_device_send(self, m_index, m_commit, m_hint, m_rs, m_flags, m_logterm=None, max_entries_per_msg=0,
                         skip_bcast_commit=False, max_bytes=None):
        """tick_device(...) + send_appends(...) as ONE launch; device pointers, asynchronous."""
        flags = SEND_SKIP_BCAST_COMMIT if skip_bcast_commit else 0
        if max_bytes is not None:
            max_entries_per_msg, flags = max_bytes, flags | SEND_BYTES
        m = _Msgs(_ptr(m_index), _ptr(m_commit), _ptr(m_hint), _ptr(m_rs), _ptr(m_flags), _ptr(m_logterm))
        self._check(self.L.rg_tick_device_send(self.h, C.byref(m), max_entries_per_msg, flags))

    def flush_send(self, max_entries_per_msg=0, skip_bcast_commit=False, max_bytes=None):
        """flush() + send_appends(); one host<->device round trip for small batches."""
        flags = SEND_SKIP_BCAST_COMMIT if skip_bcast_commit else 0
        if max_bytes is not None:
            max_entries_per_msg, flags = max_bytes, flags | SEND_BYTES
        self._check(self.L.rg_flush_send(self.h, max_entries_per_msg, flags))

    def mailbox_start(self, idle_timeout_us=0):
        """Keep one workgroup resident for small flushes (no launch / synchronisation per flush); see raftgroups.h."""
        self._check(self.L.rg_mailbox_start(self.h, idle_timeout_us))

    def mailbox_stop(self):
        self._check(self.L.rg_mailbox_stop(self.h))

    def mailbox_stats(self):
        """(flushes served by the resident workgroup, times it was launched)"""
        a, b = _u64(0), _u64(0)
        self._check(self.L.rg_mailbox_stats(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def log_sizes_enable(self, window):
        """Byte-accurate max_size_per_msg: keep the cumulative sizes of every group's last `window` entries on the device."""
        self._check(self.L.rg_log_sizes_enable(self.h, window))

    def log_sizes_write(self, recs):
        """recs: LOG_SIZE_DTYPE array, one record per appended entry (cumulative bytes up to and including it)."""
        recs = np.ascontiguousarray(recs, dtype=LOG_SIZE_DTYPE)
        self._check(self.L.rg_log_sizes_write(self.h, recs.ctypes.data, len(recs)))

    def workload_sizes(self, seed, min_bytes, spread):
        self._check(self.L.rg_workload_sizes(self.h, seed, min_bytes, spread))

    def update_state(self, msgs):
        """msgs: SENT_MSG_DTYPE array of MsgAppends the HOST sent (RG_SEND_HOST items): Progress::update_state(last)."""
        msgs = np.ascontiguousarray(msgs, dtype=SENT_MSG_DTYPE)
        self._check(self.L.rg_update_state(self.h, msgs.ctypes.data, len(msgs)))

    def progress_events(self, events):
        """events: PROGRESS_EVENT_DTYPE array (or [(group, slot, kind)]): report_unreachable / report_snapshot in place."""
        events = np.ascontiguousarray(np.array(events, dtype=PROGRESS_EVENT_DTYPE))
        self._check(self.L.rg_progress_events(self.h, events.ctypes.data, len(events)))

    def progress_event_dense(self, kind, slot_plus1):
        """slot_plus1: u8 [G], 0 = nothing for the group, s + 1 = event `kind` goes to its slot s (rg_progress_event_dense)."""
        a = np.ascontiguousarray(slot_plus1, dtype=np.uint8)
        assert a.shape == (self.n_groups,)
        self._check(self.L.rg_progress_event_dense(self.h, kind, a.ctypes.data))

    def report_unreachable(self, group, peer_id):
        self._check(self.L.rg_report_unreachable(self.h, group, peer_id))

    def report_snapshot(self, group, peer_id, failure):
        self._check(self.L.rg_report_snapshot(self.h, group, peer_id, int(bool(failure))))

    def send_items(self):
        """Work items of the last send stage as a SEND_ITEM_DTYPE array (order unspecified)."""
        n = _u64(0)
        items = np.empty(4096, dtype=SEND_ITEM_DTYPE)  # small stages come back in one call
        self._check(self.L.rg_send_items(self.h, items.ctypes.data, len(items), C.byref(n)))
        if n.value > len(items):
            items = np.empty(n.value, dtype=SEND_ITEM_DTYPE)
            self._check(self.L.rg_send_items(self.h, items.ctypes.data, n.value, C.byref(n)))
        return items[:n.value]

    def send_columns(self):
        """Device pointers (prev_index u64 [P][stride], last_index u64 [P][stride], n_msgs | kind << 16 u32 [P][stride])
        of the last DENSE stage's work items."""
        a, b, c = _vp(), _vp(), _vp()
        self._check(self.L.rg_send_columns(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def send_tail_column(self):
        """Device pointer of the windows' newest-inflight column (u64 [P][stride]): where an item whose n / kind word has bit 31
        (SEND_LAST_IS_TAIL) keeps its last_index."""
        a = _vp()
        self._check(self.L.rg_send_tail_column(self.h, C.byref(a)))
        return a.value

    def read_inflights(self):
        """(meta u32 [P][stride] = start | count << 16, ring u64 [G][P][cap])."""
        meta = np.empty((self.n_slots, self.stride), dtype=np.uint32)
        ring = np.empty((self.n_groups, self.n_slots, self.max_inflight), dtype=np.uint64)
        self._check(self.L.rg_read_inflights(self.h, meta.ctypes.data, ring.ctypes.data))
        return meta, ring

    def load_inflights(self, meta, ring):
        meta = np.ascontiguousarray(meta, dtype=np.uint32)
        ring = np.ascontiguousarray(ring, dtype=np.uint64)
        assert meta.shape == (self.n_slots, self.stride) and ring.shape == (self.n_groups, self.n_slots, self.max_inflight)
        self._check(self.L.rg_load_inflights(self.h, meta.ctypes.data, ring.ctypes.data))

    def inflights(self, group, slot, meta=None, ring=None):
        """Logical contents (oldest first) of one Progress's Inflights."""
        if meta is None:
            meta, ring = self.read_inflights()
        m = int(meta[slot, group])
        start, count = m & 0xffff, m >> 16
        return [int(ring[group, slot, (start + i) % self.max_inflight]) for i in range(count)]

    def recompute(self):
        self._check(self.L.rg_recompute(self.h))

    def maximal_committed_index(self, with_flag=False):
        mci = np.empty(self.n_groups, dtype=np.uint64)
        gc = np.empty(self.n_groups, dtype=np.uint8) if with_flag else None
        self._check(self.L.rg_maximal_committed_index(self.h, mci.ctypes.data, _ptr(gc)))
        return (mci, gc) if with_flag else mci

    def heartbeat_commits(self):
        """bcast_heartbeat: min(matched, committed) per slot -> [P][stride] u64 (host copy)."""
        hb = np.empty((self.n_slots, self.stride), dtype=np.uint64)
        self._check(self.L.rg_heartbeat_commits(self.h, None, hb.ctypes.data))
        return hb

    def step_heartbeat_response(self, group, from_, term, commit=0, ins_full=False):
        self._check(self.L.rg_step_heartbeat_response(self.h, group, from_, term, commit, int(ins_full)))

    def results(self):
        commit = np.empty(self.n_groups, dtype=np.uint64)
        out = np.empty(self.n_groups, dtype=np.uint32)
        self._check(self.L.rg_results(self.h, commit.ctypes.data, out.ctypes.data))
        return commit, out

    def result_counts(self):
        a, b = _u64(0), _u64(0)
        self._check(self.L.rg_result_counts(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def size_classes(self):
        """rg_size_classes: the ranges of the shard the next dense tick treats as one size class each -> [(first_group, n_groups,
        n_slots)]; [] = the plain kernel runs."""
        dt = np.dtype([("first_group", "<u8"), ("n_groups", "<u8"), ("n_slots", "<u4"), ("reserved", "<u4")])
        out = np.zeros(64, dtype=dt)
        n = C.c_uint32(0)
        self._check(self.L.rg_size_classes(self.h, out.ctypes.data, len(out), C.byref(n)))
        if n.value > len(out):
            out = np.zeros(n.value, dtype=dt)
            self._check(self.L.rg_size_classes(self.h, out.ctypes.data, len(out), C.byref(n)))
        return [(int(r["first_group"]), int(r["n_groups"]), int(r["n_slots"])) for r in out[:n.value]]

    def permute_groups(self, perm):
        """rg_permute_groups: group i becomes the group that was at perm[i] -- every column, the Inflights, the mirror's tables."""
        perm = np.ascontiguousarray(perm, dtype=np.uint64)
        assert perm.shape == (self.n_groups,)
        self._check(self.L.rg_permute_groups(self.h, perm.ctypes.data))

    def place_by_size_class(self):
        """Plan from the engine's own cfg column (rg_plan_placement) and permute: a shard loaded with its replica-set sizes
        interleaved becomes the layout the one-launch class kernel runs. Returns (perm, the planned ranges)."""
        perm, classes = plan_placement(self.read_column(COL.CFG), self.n_slots)
        self.permute_groups(perm)
        return perm, classes

    def host_hints(self):
        """Groups whose last tick raised RG_OUT_HOST_HINT -> HOST_HINT_DTYPE array (group, slot_mask)."""
        n = _u64(0)
        items = np.zeros(256, dtype=HOST_HINT_DTYPE)
        self._check(self.L.rg_host_hints(self.h, items.ctypes.data, len(items), C.byref(n)))
        if n.value > len(items):
            items = np.zeros(n.value, dtype=HOST_HINT_DTYPE)
            self._check(self.L.rg_host_hints(self.h, items.ctypes.data, len(items), C.byref(n)))
        return items[:n.value]

    def resolve_host_hints(self, recs):
        """recs: RESOLVED_HINT_DTYPE array (or [(group, index, hint, slot, 0)]): the host's find_conflict_by_term answers.
        Returns u8[n]: maybe_decr_to returned true (send_append is due)."""
        recs = np.ascontiguousarray(np.array(recs, dtype=RESOLVED_HINT_DTYPE))
        applied = np.zeros(len(recs), dtype=np.uint8)
        self._check(self.L.rg_resolve_host_hints(self.h, recs.ctypes.data, len(recs), applied.ctypes.data))
        return applied

    def msg_stats(self, dev_m_flags):
        c = (_u64 * 5)()
        self._check(self.L.rg_msg_stats(self.h, _ptr(dev_m_flags), C.byref(c)))
        return {"valid": c[0], "rejects": c[1], "slots": c[2], "groups_with_events": c[3], "elections": c[4]}

    def vote_result(self, yes, no):
        yes = np.ascontiguousarray(yes, dtype=np.uint8)
        no = np.ascontiguousarray(no, dtype=np.uint8)
        res = np.empty(self.n_groups, dtype=np.uint8)
        self._check(self.L.rg_vote_result(self.h, yes.ctypes.data, no.ctypes.data, res.ctypes.data))
        return res

    def tally_votes(self, yes, no):
        """ProgressTracker::tally_votes: (granted, rejected, result) per group."""
        yes = np.ascontiguousarray(yes, dtype=np.uint8)
        no = np.ascontiguousarray(no, dtype=np.uint8)
        g, r, res = (np.empty(self.n_groups, dtype=np.uint8) for _ in range(3))
        self._check(self.L.rg_tally_votes(self.h, yes.ctypes.data, no.ctypes.data, g.ctypes.data, r.ctypes.data,
                                          res.ctypes.data))
        return g, r, res

    def quorum_recently_active(self):
        res = np.empty(self.n_groups, dtype=np.uint8)
        self._check(self.L.rg_quorum_recently_active(self.h, res.ctypes.data))
        return res

    # ---- message-at-a-time mirror of RawNode::step -----------------------------------------------
    def set_peers(self, group, peer_ids, term):
        arr = (_u64 * len(peer_ids))(*peer_ids)
        self._check(self.L.rg_set_peers(self.h, group, arr, len(peer_ids), term))

    def step(self, group, from_, term, index, commit=0, reject=False, reject_hint=0, request_snapshot=0,
             ins_full=False, log_term=0):
        m = AppendResponse(from_, term, index, commit, reject_hint, request_snapshot, int(reject), int(ins_full),
                           (C.c_uint8 * 6)(), log_term)
        self._check(self.L.rg_step(self.h, group, C.byref(m)))

    def step_bytes(self, group, data, ins_full=False):
        """RawNode::step on one protobuf-encoded eraftpb::Message (bytes); ins_full: the caller's Inflights::full() for the sender."""
        self._check(self.L.rg_step_bytes(self.h, group, bytes(data), len(data), int(ins_full)))

    def local_append(self, group, new_last_index):
        self._check(self.L.rg_local_append(self.h, group, new_last_index))

    def local_persisted(self, group, index):
        self._check(self.L.rg_local_persisted(self.h, group, index))

    def mark_sent(self, group, peer_id):
        self._check(self.L.rg_mark_sent(self.h, group, peer_id))

    def local_become_leader(self, group, term):
        """Raft::become_leader at `term` for one group, queued like the other leader-local events."""
        self._check(self.L.rg_local_become_leader(self.h, group, term))

    def flush(self):
        self._check(self.L.rg_flush(self.h))

    # ---- multi-GPU: publication of commit indices ---------------------------------------------------
    def comm_init(self, rank, world, unique_id=None, transport=None, ring_ticks=0, overflow_slots=0):
        """rg_comm_init (a collective). unique_id: bytes from comm_unique_id() (RCCL); transport: a Python callable
        (dev_send, dev_recv, bytes_per_rank, hip_stream) -> 0 standing in for the all-gather (tests, other fabrics)."""
        self._comm_keep = []
        cfg = CommConfig(rank, world, None, ALLGATHER_FN(0), None, ring_ticks, overflow_slots)
        if transport is not None:
            cb = ALLGATHER_FN(lambda user, s, r, n, st: int(transport(s, r, n, st) or 0))
            cfg.transport = cb
            self._comm_keep.append(cb)
        else:
            buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(bytes(unique_id))
            cfg.unique_id = C.addressof(buf)
            self._comm_keep.append(buf)
        self._check(self.L.rg_comm_init(self.h, C.byref(cfg)))
        self.comm_rank, self.comm_world = rank, world

    def comm_destroy(self):
        self._check(self.L.rg_comm_destroy(self.h))

    def comm_info(self):
        """rg_comm_info_get: rank / world as given, the transport, and -- over RCCL -- the communicator's own ncclCommCount /
        ncclCommUserRank (what proves how many ranks an exchange really spans)."""
        ci = CommInfo()
        self._check(self.L.rg_comm_info_get(self.h, C.byref(ci)))
        return {"rank": ci.rank, "world": ci.world, "transport": TRANSPORT_NAMES.get(ci.transport, str(ci.transport)),
                "in_process": bool(ci.in_process), "rccl_ranks": ci.rccl_ranks, "rccl_rank": ci.rccl_rank}

    def publish_commit(self, full=False):
        self._check(self.L.rg_publish_commit(self.h, PUBLISH_FULL if full else 0))

    def publish_sync(self):
        self._check(self.L.rg_publish_sync(self.h))

    def published_commit(self, rank=None):
        """The replica of `rank`'s commit indices (all ranks: [world][n_groups]) as a host array."""
        ranks = range(self.comm_world) if rank is None else [rank]
        out = np.empty((len(ranks), self.n_groups), dtype=np.uint64)
        for i, r in enumerate(ranks):
            self._check(self.L.rg_published_commit(self.h, r, 0, self.n_groups, out[i].ctypes.data))
        return out if rank is None else out[0]

    def published_commit_ptr(self):
        stride = _u64(0)
        return self.L.rg_published_commit_ptr(self.h, C.byref(stride)), stride.value

    def publish_stats(self):
        st = PublishStats()
        self._check(self.L.rg_publish_stats_get(self.h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in PublishStats._fields_}

    # ---- synthetic stream --------------------------------------------------------------------
    def workload_init(self, workload, seed=0x5EED5EED, first_group=0, fixed_peers=0, sorted_classes=False, group_commit=False):
        w = _Workload(seed, workload, fixed_peers | (WL_PLACE_SORTED if sorted_classes else 0) | (WL_GROUP_COMMIT if group_commit else 0))
        self._check(self.L.rg_workload_init(self.h, C.byref(w), first_group))

    def workload_gen(self, workload, tick, m_index, m_commit, m_hint, m_rs, m_flags, seed=0x5EED5EED, first_group=0,
                     fixed_peers=0, sorted_classes=False, group_commit=False):
        w = _Workload(seed, workload, fixed_peers | (WL_PLACE_SORTED if sorted_classes else 0) | (WL_GROUP_COMMIT if group_commit else 0))
        self._check(self.L.rg_workload_gen(self.h, C.byref(w), first_group, tick, _ptr(m_index), _ptr(m_commit),
                                           _ptr(m_hint), _ptr(m_rs), _ptr(m_flags)))


def _host_state_struct(st):
    return _HostState(st["n_groups"], st["stride"], st["n_slots"], 0,
                      *[st[k].ctypes.data for k in COL.NAMES[:11]])


WL_GROUP_COMMIT = 0x20  # rg_workload.reserved flag: group commit on, three commit groups (RG_WL_GROUP_COMMIT)
WL_PLACE_SORTED = 0x10  # rg_workload.reserved flag: the groups of a shard placed by replica-set size class (RG_WL_PLACE_SORTED)


def workload_init_host(st, workload, seed=0x5EED5EED, first_group=0, fixed_peers=0, sorted_classes=False, group_commit=False):
    """Host twin of Engine.workload_init over numpy columns (no GPU)."""
    L = load_library()
    w = _Workload(seed, workload, fixed_peers | (WL_PLACE_SORTED if sorted_classes else 0) | (WL_GROUP_COMMIT if group_commit else 0))
    s = _host_state_struct(st)
    rc = L.rg_workload_init_host(C.byref(w), first_group, C.byref(s))
    if rc:
        raise EngineError(rc, L.rg_last_error().decode())


def workload_gen_host(st, msgs, workload, tick, seed=0x5EED5EED, first_group=0, fixed_peers=0, sorted_classes=False, group_commit=False):
    """Host twin of Engine.workload_gen: messages of `tick` from the numpy state columns."""
    L = load_library()
    w = _Workload(seed, workload, fixed_peers | (WL_PLACE_SORTED if sorted_classes else 0) | (WL_GROUP_COMMIT if group_commit else 0))
    s = _host_state_struct(st)
    rc = L.rg_workload_gen_host(C.byref(w), first_group, tick, C.byref(s), msgs.m_index.ctypes.data,
                                msgs.m_commit.ctypes.data, msgs.m_hint.ctypes.data, msgs.m_rs.ctypes.data,
                                msgs.m_flags.ctypes.data)
    if rc:
        raise EngineError(rc, L.rg_last_error().decode())


class CommAllConfig(C.Structure):
    _fields_ = [("ring_ticks", C.c_uint32), ("overflow_slots", C.c_uint32), ("transport", C.c_uint32), ("reserved", C.c_uint32)]


COMM_ALL_AUTO, COMM_ALL_RCCL, COMM_ALL_LOCAL = 0, 1, 2
SEND_LAST_IS_TAIL, SEND_LAST_IS_PREV = 0x80000000, 0x40000000


def comm_init_all(engines, ring_ticks=0, overflow_slots=0, transport=COMM_ALL_AUTO):
    """rg_comm_init_all: the engines of ONE process become ranks 0..n-1 of one publication, driven by one thread."""
    L = engines[0].L
    arr = (C.c_void_p * len(engines))(*[e.h for e in engines])
    cfg = CommAllConfig(ring_ticks, overflow_slots, transport, 0)
    engines[0]._check(L.rg_comm_init_all(arr, len(engines), C.byref(cfg)))
    for i, e in enumerate(engines):
        e.comm_rank, e.comm_world = i, len(engines)


def publish_commit_all(engines, full=False):
    """rg_publish_commit_all: one publication of every rank (the n exchanges issued together)."""
    L = engines[0].L
    arr = (C.c_void_p * len(engines))(*[e.h for e in engines])
    engines[0]._check(L.rg_publish_commit_all(arr, len(engines), PUBLISH_FULL if full else 0))


def plan_placement(cfg_words, n_slots):
    """rg_plan_placement (pure host arithmetic): -> (perm u64[G] with perm[i] = current position of the group that goes to
    position i, [(first_group, n_groups, n_slots)] = the ranges the engine will derive from the permuted column)."""
    L = load_library()
    cfg = np.ascontiguousarray(cfg_words, dtype=np.uint32)
    perm = np.zeros(len(cfg), dtype=np.uint64)
    dt = np.dtype([("first_group", "<u8"), ("n_groups", "<u8"), ("n_slots", "<u4"), ("reserved", "<u4")])
    out = np.zeros(16, dtype=dt)
    n = C.c_uint32(0)
    rc = L.rg_plan_placement(cfg.ctypes.data, len(cfg), n_slots, perm.ctypes.data, out.ctypes.data, len(out), C.byref(n))
    if rc:
        raise EngineError(rc, L.rg_last_error().decode())
    return perm, [(int(r["first_group"]), int(r["n_groups"]), int(r["n_slots"])) for r in out[:min(n.value, len(out))]]


def comm_warmup():
    """rg_comm_warmup: load RCCL and run a one-rank communicator through its life on the current device (start-up cost, paid early)."""
    L = load_library()
    rc = L.rg_comm_warmup()
    if rc:
        raise EngineError(rc, L.rg_last_error().decode())


def comm_unique_id():
    """rg_comm_unique_id: the RCCL unique id rank 0 hands to the other ranks (128 bytes)."""
    L = load_library()
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    rc = L.rg_comm_unique_id(buf)
    if rc:
        raise EngineError(rc, L.rg_last_error().decode())
    return bytes(buf)


def pub_bytes_per_rank(n_groups, overflow_slots=0):
    return load_library().rg_pub_bytes_per_rank(n_groups, overflow_slots)


def pub_accumulate_host(old_commit, new_commit, slice_, overflow_slots=0):
    """Host twin of the ticks' publication byte: accumulate old -> new into `slice_` (uint8 array, in place)."""
    L = load_library()
    a, b = np.ascontiguousarray(old_commit, dtype=np.uint64), np.ascontiguousarray(new_commit, dtype=np.uint64)
    rc = L.rg_pub_accumulate_host(len(a), overflow_slots, a.ctypes.data, b.ctypes.data, slice_.ctypes.data)
    if rc:
        raise EngineError(rc, L.rg_last_error().decode())


def pub_apply_host(n_groups, world, gathered, replica, overflow_slots=0):
    """Host twin of the replica update: add one gathered publication to `replica` ([world][stride]). Returns the
    number of slices that ask for a full resynchronisation."""
    L = load_library()
    lost = C.c_uint32(0)
    rc = L.rg_pub_apply_host(n_groups, overflow_slots, world, gathered.ctypes.data, replica.ctypes.data, C.byref(lost))
    if rc:
        raise EngineError(rc, L.rg_last_error().decode())
    return lost.value


def decode_message(data):
    """rg_decode_message: the decoder behind rg_step_bytes alone (pure host code). Returns a dict of the fields."""
    m = DecodedMessage()
    L = load_library()
    rc = L.rg_decode_message(bytes(data), len(data), C.byref(m))
    if rc != 0:
        raise EngineError(rc, L.rg_last_error().decode())
    return {k.rstrip("_"): int(getattr(m, k)) for k, _ in DecodedMessage._fields_}


def _entries_c(entries):
    """[{entry_type, term, index, data, context, sync_log}] -> (EntryC array, keep-alive list)."""
    arr = (EntryC * max(1, len(entries)))()
    keep = []
    for a, e in zip(arr, entries):
        d, c = bytes(e.get("data", b"")), bytes(e.get("context", b""))
        keep += [d, c]
        a.entry_type, a.sync_log = int(e.get("entry_type", 0)), int(bool(e.get("sync_log", False)))
        a.term, a.index = int(e.get("term", 0)), int(e.get("index", 0))
        a.data, a.data_len, a.context, a.context_len = (d or None), len(d), (c or None), len(c)
    return arr, keep


def entry_size(entry):
    """rg_entry_size: Entry::compute_size()."""
    arr, _keep = _entries_c([entry])
    return int(load_library().rg_entry_size(arr))


def limit_size(entries, max_size):
    """rg_limit_size: how many of `entries` one message keeps under util::limit_size(max_size) (None = NO_LIMIT)."""
    arr, _keep = _entries_c(entries)
    return int(load_library().rg_limit_size(arr, len(entries), (1 << 64) - 1 if max_size is None else int(max_size)))


def encode_message(fields, entries=(), snapshot=None, context=b"", cap=None):
    """rg_encode_message: an outgoing eraftpb::Message -> bytes. `fields`: the scalar fields by their .proto names;
    `snapshot`: an already serialised eraftpb::Snapshot or None; `cap`: buffer size to offer (default: exactly enough)."""
    L = load_library()
    arr, _keep = _entries_c(list(entries))
    m = MessageC()
    for k, v in fields.items():
        setattr(m, "from_" if k == "from" else k, int(v))
    m.entries, m.n_entries = arr, len(entries)
    ctx = bytes(context)
    m.context, m.context_len = (ctx or None), len(ctx)
    if snapshot is not None:
        snap = bytes(snapshot)
        # (c_char_p(b"") is a valid non-NULL pointer: a present, empty Snapshot)
        m.snapshot, m.snapshot_len = snap, len(snap)
    n = C.c_uint64(0)
    rc = L.rg_message_size(C.byref(m), C.byref(n))
    if rc != 0:
        raise EngineError(rc, L.rg_last_error().decode())
    size = n.value if cap is None else int(cap)
    buf = C.create_string_buffer(max(1, size))
    rc = L.rg_encode_message(C.byref(m), buf, size, C.byref(n))
    if rc != 0:
        raise EngineError(rc, L.rg_last_error().decode())
    return buf.raw[:n.value]
