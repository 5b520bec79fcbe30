"""ctypes binding of oracle/libraft_oracle.so -- TEST INFRASTRUCTURE ONLY.

The oracle is the CPU restatement of the reference (see oracle/raft_oracle.h).
Nothing under raft_rs_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libraft_oracle.so")

PROBE, REPLICATE, SNAPSHOT = 0, 1, 2
VOTE_PENDING, VOTE_LOST, VOTE_WON = 0, 1, 2
U64_MAX = (1 << 64) - 1


def build(force=False):
    src = os.path.join(ORACLE_DIR, "raft_oracle.c")
    hdr = os.path.join(ORACLE_DIR, "raft_oracle.h")
    stale = (not os.path.exists(LIB_PATH)) or any(
        os.path.exists(p) and os.path.getmtime(p) > os.path.getmtime(LIB_PATH) for p in (src, hdr))
    if force or stale:
        if not os.path.exists(src):
            raise RuntimeError("oracle sources missing")
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s", "all"])
    return LIB_PATH


class Inflights(C.Structure):
    _fields_ = [("start", C.c_size_t), ("count", C.c_size_t), ("cap", C.c_size_t),
                ("len", C.c_size_t), ("buffer", C.POINTER(C.c_uint64))]


class Progress(C.Structure):
    _fields_ = [("matched", C.c_uint64), ("next_idx", C.c_uint64), ("state", C.c_uint8),
                ("paused", C.c_bool), ("pending_snapshot", C.c_uint64),
                ("pending_request_snapshot", C.c_uint64), ("recent_active", C.c_bool),
                ("ins", Inflights), ("commit_group_id", C.c_uint64),
                ("committed_index", C.c_uint64)]


class Index(C.Structure):
    _fields_ = [("index", C.c_uint64), ("group_id", C.c_uint64)]


class Msg(C.Structure):
    _fields_ = [("from_", C.c_uint64), ("index", C.c_uint64), ("commit", C.c_uint64),
                ("reject_hint", C.c_uint64), ("request_snapshot", C.c_uint64),
                ("reject", C.c_bool), ("ins_full", C.c_int8), ("log_term", C.c_uint64)]


class Out(C.Structure):
    _fields_ = [("handled", C.c_bool), ("send_append", C.c_bool), ("send_more", C.c_bool),
                ("commit_changed", C.c_bool), ("free_to", C.c_bool), ("timeout_now", C.c_bool)]


class SendMsg(C.Structure):
    _fields_ = [("group", C.c_uint64), ("to", C.c_uint64), ("index", C.c_uint64), ("n_entries", C.c_uint64),
                ("kind", C.c_uint32), ("pad", C.c_uint32)]


SEND_MSG_DTYPE = np.dtype([("group", "<u8"), ("to", "<u8"), ("index", "<u8"), ("n_entries", "<u8"), ("kind", "<u4"),
                           ("pad", "<u4")])
SEND_APPEND, SEND_SNAPSHOT, SEND_HOST = 1, 2, 3


class SoaState(C.Structure):
    _fields_ = [("n_groups", C.c_size_t), ("n_slots", C.c_size_t), ("stride", C.c_size_t)] + \
               [(n, C.c_void_p) for n in ("match", "next", "pr_commit", "pend_snap", "pend_rs",
                                          "gid", "pflags", "commit", "term_lo", "term_hi", "cfg",
                                          "run_first", "run_term", "dummy_index", "dummy_term", "cur_term")]


class SoaMsgs(C.Structure):
    _fields_ = [("n_groups", C.c_size_t), ("n_slots", C.c_size_t), ("stride", C.c_size_t)] + \
               [(n, C.c_void_p) for n in ("m_index", "m_commit", "m_hint", "m_rs", "m_flags", "m_logterm")]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(os.environ.get("RO_LIB_PATH") or build())  # RO_LIB_PATH: e.g. the ASan/UBSan build (make -C oracle asan)
    u64, sz, vp = C.c_uint64, C.c_size_t, C.c_void_p
    PP, PI = C.POINTER(Progress), C.POINTER(Inflights)
    sig = {
        "ro_ins_init": (None, [PI, sz]), "ro_ins_destroy": (None, [PI]),
        "ro_ins_full": (C.c_bool, [PI]), "ro_ins_add": (C.c_int, [PI, u64]),
        "ro_ins_free_to": (None, [PI, u64]), "ro_ins_free_first_one": (None, [PI]),
        "ro_ins_reset": (None, [PI]),
        "ro_progress_new": (None, [PP, u64, sz]), "ro_progress_destroy": (None, [PP]),
        "ro_progress_reset": (None, [PP, u64]), "ro_progress_become_probe": (None, [PP]),
        "ro_progress_become_replicate": (None, [PP]),
        "ro_progress_become_snapshot": (None, [PP, u64]),
        "ro_progress_maybe_snapshot_abort": (C.c_bool, [PP]),
        "ro_progress_maybe_update": (C.c_bool, [PP, u64]),
        "ro_progress_update_committed": (None, [PP, u64]),
        "ro_progress_maybe_decr_to": (C.c_bool, [PP, u64, u64, u64]),
        "ro_progress_is_paused": (C.c_bool, [PP]),
        "ro_progress_update_state": (C.c_int, [PP, u64]),
        "ro_majority": (sz, [sz]),
        "ro_majority_committed_index": (u64, [C.POINTER(Index), sz, C.c_bool, C.POINTER(C.c_bool)]),
        "ro_majority_vote_result": (C.c_int, [C.POINTER(C.c_uint8), sz]),
        "ro_joint_vote_result": (C.c_int, [C.c_int, C.c_int]),
        "ro_new": (vp, [sz]), "ro_free": (None, [vp]), "ro_n_groups": (sz, [vp]),
        "ro_group_config": (C.c_int, [vp, sz, u64, u64, C.POINTER(u64), sz, C.POINTER(u64), sz,
                                      C.POINTER(u64), sz, u64, sz]),
        "ro_group_set_group_commit": (None, [vp, sz, C.c_bool]),
        "ro_group_set_transferee": (None, [vp, sz, u64]),
        "ro_group_set_log": (C.c_int, [vp, sz, u64, u64, C.POINTER(u64), C.POINTER(u64), sz, u64, u64]),
        "ro_group_append": (None, [vp, sz, u64]),
        "ro_group_become_leader": (C.c_int, [vp, sz, u64]),
        "ro_group_progress": (PP, [vp, sz, u64]),
        "ro_group_committed": (u64, [vp, sz]), "ro_group_last_index": (u64, [vp, sz]),
        "ro_group_term": (u64, [vp, sz]),
        "ro_log_term": (u64, [vp, sz, u64]), "ro_log_commit_to": (C.c_int, [vp, sz, u64]),
        "ro_log_maybe_commit": (C.c_bool, [vp, sz, u64, u64]),
        "ro_log_find_conflict_by_term": (u64, [vp, sz, u64, u64]),
        "ro_maximal_committed_index": (u64, [vp, sz, C.POINTER(C.c_bool)]),
        "ro_maybe_commit": (C.c_bool, [vp, sz]),
        "ro_handle_append_response": (None, [vp, sz, C.POINTER(Msg), C.POINTER(Out)]),
        "ro_on_persist_entries": (C.c_bool, [vp, sz, u64]),
        "ro_handle_heartbeat_response": (None, [vp, sz, u64, u64, C.c_int8, C.POINTER(Out)]),
        "ro_heartbeat_commit": (u64, [vp, sz, u64]),
        "ro_handle_snapshot_status": (C.c_bool, [vp, sz, u64, C.c_bool]),
        "ro_handle_unreachable": (C.c_bool, [vp, sz, u64]),
        "ro_group_vote_result": (C.c_int, [vp, sz, C.POINTER(u64), C.POINTER(C.c_uint8), sz]),
        "ro_quorum_recently_active": (C.c_bool, [vp, sz, u64]),
        "ro_group_tally_votes": (C.c_int, [vp, sz, C.POINTER(u64), C.POINTER(C.c_uint8), sz, C.POINTER(sz), C.POINTER(sz)]),
        "ro_load_soa": (C.c_int, [vp, C.POINTER(SoaState), u64, sz]),
        "ro_store_soa": (C.c_int, [vp, C.POINTER(SoaState)]),
        "ro_tick_soa": (u64, [vp, C.POINTER(SoaMsgs), vp, sz, sz]),
        "ro_tick_soa_mt": (u64, [vp, C.POINTER(SoaMsgs), vp, sz]),
        "ro_maybe_send_append": (C.c_bool, [vp, sz, u64, C.c_bool, u64, C.POINTER(SendMsg)]),
        "ro_set_own_inflights": (None, [vp, C.c_bool]),
        "ro_send_stage_soa": (sz, [vp, vp, u64, C.c_bool, vp, sz, sz, sz]),
        "ro_group_set_pending_conf": (None, [vp, sz, C.c_bool]),
        "ro_set_limit_bytes": (None, [vp, C.c_bool]),
        "ro_group_append_entry_sizes": (None, [vp, sz, u64, vp, sz]),
        "ro_ins_contents": (sz, [vp, sz, u64, C.POINTER(u64), sz]),
        "ro_ins_export_soa": (sz, [vp, sz, sz, vp, vp, sz]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def u64arr(vals):
    return (C.c_uint64 * len(vals))(*vals)


def committed_index(pairs, use_group_commit=False):
    """majority committed index of [(index, gid), ...] in the given iteration order."""
    arr = (Index * max(1, len(pairs)))()
    for i, (ix, g) in enumerate(pairs):
        arr[i].index, arr[i].group_id = ix, g
    flag = C.c_bool(False)
    v = lib().ro_majority_committed_index(arr, len(pairs), use_group_commit, C.byref(flag))
    return v, bool(flag.value)


class Cluster:
    """Thin OO wrapper over ro_cluster."""

    def __init__(self, n_groups):
        self.L = lib()
        self.h = self.L.ro_new(n_groups)
        self.n = n_groups

    def __del__(self):
        if getattr(self, "h", None):
            self.L.ro_free(self.h)
            self.h = None

    def config(self, g, self_id, term, incoming, outgoing=(), learners=(), next_idx=1, max_inflight=256):
        r = self.L.ro_group_config(self.h, g, self_id, term, u64arr(list(incoming)), len(incoming),
                                   u64arr(list(outgoing)), len(outgoing), u64arr(list(learners)),
                                   len(learners), next_idx, max_inflight)
        assert r == 0
        return self

    def set_log(self, g, entries, committed=0, dummy=(0, 0)):
        """entries: [(term, index), ...] contiguous."""
        firsts, terms = [], []
        last = dummy[0]
        for term, idx in entries:
            if not terms or terms[-1] != term:
                firsts.append(idx)
                terms.append(term)
            last = idx
        r = self.L.ro_group_set_log(self.h, g, dummy[0], dummy[1], u64arr(firsts), u64arr(terms),
                                    len(firsts), last, committed)
        assert r == 0

    def pr(self, g, pid):
        p = self.L.ro_group_progress(self.h, g, pid)
        return p.contents if p else None

    def committed(self, g):
        return self.L.ro_group_committed(self.h, g)

    def last_index(self, g):
        return self.L.ro_group_last_index(self.h, g)

    def maybe_commit(self, g):
        return self.L.ro_maybe_commit(self.h, g)

    def mci(self, g):
        flag = C.c_bool(False)
        v = self.L.ro_maximal_committed_index(self.h, g, C.byref(flag))
        return v, bool(flag.value)

    def step(self, g, from_, index, commit=0, reject=False, reject_hint=0, request_snapshot=0, ins_full=-1,
             log_term=0):
        m = Msg(from_, index, commit, reject_hint, request_snapshot, reject, ins_full, log_term)
        o = Out()
        self.L.ro_handle_append_response(self.h, g, C.byref(m), C.byref(o))
        return o

    # ---- SoA adapters -------------------------------------------------
    def load_soa(self, st, term=2, max_inflight=0):
        s = _soa_state_struct(st)
        r = self.L.ro_load_soa(self.h, C.byref(s), term, max_inflight)
        assert r == 0, r

    def store_soa(self, st):
        s = _soa_state_struct(st)
        r = self.L.ro_store_soa(self.h, C.byref(s))
        assert r == 0, r

    def tick_soa(self, msgs, gout, g_begin=0, g_end=None):
        m = _soa_msgs_struct(msgs)
        return self.L.ro_tick_soa(self.h, C.byref(m), gout.ctypes.data, g_begin,
                                  self.n if g_end is None else g_end)

    def set_own_inflights(self, on=True):
        self.L.ro_set_own_inflights(self.h, on)

    def set_limit_bytes(self, on=True):
        """max_entries of maybe_send_append / send_stage_soa is Config::max_size_per_msg in bytes (literal limit_size)."""
        self.L.ro_set_limit_bytes(self.h, on)

    def append_entry_sizes(self, g, first_index, sizes):
        sizes = np.ascontiguousarray(sizes, dtype=np.uint32)
        self.L.ro_group_append_entry_sizes(self.h, g, first_index, sizes.ctypes.data, len(sizes))

    def maybe_send_append(self, g, to, allow_empty, max_entries=0):
        m = SendMsg()
        sent = self.L.ro_maybe_send_append(self.h, g, to, allow_empty, max_entries, C.byref(m))
        return sent, m

    def send_stage_soa(self, gout, max_entries=0, capacity=1 << 20, g_begin=0, g_end=None, skip_bcast_commit=False):
        """The reference's send decisions for the tick whose result words are gout -> SEND_MSG_DTYPE array
        (one record per message sent, in the order sent). Mutates the cluster: never re-run."""
        buf = np.zeros(capacity, dtype=SEND_MSG_DTYPE)
        n = self.L.ro_send_stage_soa(self.h, gout.ctypes.data, max_entries, skip_bcast_commit, buf.ctypes.data, len(buf), g_begin,
                                     self.n if g_end is None else g_end)
        assert n <= capacity, "send_stage_soa: capacity too small"
        return buf[:n]

    def ins_contents(self, g, pid, cap=65536):
        buf = (C.c_uint64 * cap)()
        n = self.L.ro_ins_contents(self.h, g, pid, buf, cap)
        return [int(buf[i]) for i in range(min(n, cap))]

    def ins_export(self, n_slots, stride, k=16):
        """Every window of the cluster at once: (counts u32 [P][stride], first_k u64 [n][P][k], largest count)."""
        counts = np.zeros((n_slots, stride), dtype=np.uint32)
        first_k = np.zeros((self.n, n_slots, k), dtype=np.uint64)
        mx = self.L.ro_ins_export_soa(self.h, n_slots, stride, counts.ctypes.data, first_k.ctypes.data, k)
        return counts, first_k, int(mx)

    def tick_soa_mt(self, msgs, gout, n_threads):
        m = _soa_msgs_struct(msgs)
        return self.L.ro_tick_soa_mt(self.h, C.byref(m), gout.ctypes.data, n_threads)


STATE_COLS = ("match", "next", "pr_commit", "pend_snap", "pend_rs", "gid", "pflags", "commit",
              "term_lo", "term_hi", "cfg")


TERM_RUNS = 8  # RG_TERM_RUNS / RO_TERM_RUNS (tests/test_abi.py compares the three)
TERM_TABLE_COLS = ("run_first", "run_term", "dummy_index", "dummy_term", "cur_term")


def _soa_state_struct(st):
    table = [st[k].ctypes.data if k in st else None for k in TERM_TABLE_COLS]
    if any(t is None for t in table):
        table = [None] * 5
    return SoaState(st["n_groups"], st["n_slots"], st["stride"], *[st[k].ctypes.data for k in STATE_COLS], *table)


def _soa_msgs_struct(msgs):
    lt = msgs.get("m_logterm")
    return SoaMsgs(msgs["n_groups"], msgs["n_slots"], msgs["stride"],
                   *[msgs[k].ctypes.data for k in ("m_index", "m_commit", "m_hint", "m_rs", "m_flags")],
                   lt.ctypes.data if lt is not None else None)


def alloc_state(n_groups, n_slots, stride=None):
    stride = stride or ((n_groups + 255) // 256) * 256
    st = {"n_groups": n_groups, "n_slots": n_slots, "stride": stride}
    for k in ("match", "next", "pr_commit", "pend_snap", "pend_rs", "gid"):
        st[k] = np.zeros((n_slots, stride), dtype=np.uint64)
    st["pflags"] = np.zeros((n_groups, 8), dtype=np.uint8)
    for k in ("commit", "term_lo", "term_hi"):
        st[k] = np.zeros(n_groups, dtype=np.uint64)
    st["cfg"] = np.zeros(n_groups, dtype=np.uint32)
    return st


def alloc_msgs(n_groups, n_slots, stride=None):
    stride = stride or ((n_groups + 255) // 256) * 256
    m = {"n_groups": n_groups, "n_slots": n_slots, "stride": stride}
    for k in ("m_index", "m_commit", "m_hint", "m_rs", "m_logterm"):
        m[k] = np.zeros((n_slots, stride), dtype=np.uint64)
    m["m_flags"] = np.zeros((n_groups, 8), dtype=np.uint8)
    return m


def add_term_table(st):
    """Attach an (empty) term-run table to a state dict: run_first/run_term [TERM_RUNS][stride], dummy [G]."""
    st["run_first"] = np.zeros((TERM_RUNS, st["stride"]), dtype=np.uint64)
    st["run_term"] = np.zeros((TERM_RUNS, st["stride"]), dtype=np.uint64)
    st["dummy_index"] = np.zeros(st["n_groups"], dtype=np.uint64)
    st["dummy_term"] = np.zeros(st["n_groups"], dtype=np.uint64)
    st["cur_term"] = np.zeros(st["n_groups"], dtype=np.uint64)
    return st
