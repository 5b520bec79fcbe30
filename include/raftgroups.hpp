// raftgroups.hpp -- the C++17 host side above the C ABI (include/raftgroups.h). Header-only; links against
// libraftgroups.so like any C caller. It mirrors the reference's own interface for this path -- same names, argument
// meaning and error behaviour -- so that a driver (or a test) reads like code written against raft-rs:
//
//   raftgroups::MultiRaft::step(group, Message)     RawNode::step           src/raw_node.rs:402-411 -> src/raft.rs:1280-1411
//   raftgroups::Message                             eraftpb::Message        proto/proto/eraftpb.proto:71-92 (the fields the path reads)
//   raftgroups::Progress / ProgressState            tracker::Progress       src/tracker/progress.rs:8-56, src/tracker/state.rs:22-29
//   raftgroups::Error (StepLocalMsg, ...)           raft::Error             src/errors.rs:6-50
//   MultiRaft::ready() -> LightReady per group      RawNode::ready          src/raw_node.rs:469-532, :643-651 (commit_index, messages)
//   MultiRaft::propose / on_persist_entries /       Raft::append_entry, on_persist_entries, become_leader
//     become_leader                                 src/raft.rs:976-1016, :1151-1202
//   MultiRaft::report_unreachable / report_snapshot RawNode::report_unreachable / report_snapshot   src/raw_node.rs:692-709
//   raftgroups::Storage                             raft::Storage           src/storage.rs:65-106 (one more argument: the group)
//   raftgroups::Entry / limit_size                  eraftpb::Entry, util::limit_size   eraftpb.proto:23-31, src/util.rs:52-76
//   build_messages / MultiRaft::messages            maybe_send_append -> prepare_send_entries / prepare_send_snapshot -> send
//                                                   src/raft.rs:773-819, :714-731, :664-712, :602-662
//   Message::write_to_bytes                         protobuf::Message::write_to_bytes (rg_encode_message)
//
// One MultiRaft holds the leader-side replication state of N raft groups on ONE GPU. Messages are queued by step() and
// applied -- all groups at once, on the device -- by ready(), which returns what the reference's Ready would carry for
// this path: the new commit index of every group that saw traffic and the MsgAppend sends the path asked for (as work
// items; MultiRaft::messages turns them into the reference's Messages -- entries, log_term, commit -- out of the host's
// Storage, and Message::write_to_bytes into the bytes a transport takes). There is no CPU path: constructing a MultiRaft
// without a gfx950 device throws Error{NoDevice}; only the message building (build_messages, Message::write_to_bytes,
// limit_size) is host code, because the log it reads is the host's.
#pragma once

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "raftgroups.h"

namespace raftgroups {

using u64 = std::uint64_t;
constexpr u64 INVALID_ID = 0;    // src/raft.rs:78
constexpr u64 INVALID_INDEX = 0; // src/raft.rs:81
constexpr u64 NO_LIMIT = ~0ULL;  // src/util.rs:14

// ---- raft::Error (src/errors.rs:6-50), plus what only a device back end can report ----
enum class ErrorKind {
    StepLocalMsg,     // Error::StepLocalMsg    (raw_node.rs:404-406)
    StepPeerNotFound, // Error::StepPeerNotFound (raw_node.rs:407-410)
    HigherTerm,       // m.term > term: Raft::step would become_follower (raft.rs:1284-1348) -- the host's to handle
    SlotBusy,         // a second message of one peer before ready(): call ready() first
    NotOnPath,        // step(bytes): a well-formed message of a type this path does not handle
    InvalidArgument,
    State,            // call sequence error
    HostHint,         // a fused launch stopped behind a tick that left a reject to the host (RG_ERR_HOST_HINT)
    NoDevice,         // no gfx950 device / HIP error (no CPU fallback)
    OutOfMemory,
};
class Error : public std::runtime_error {
  public:
    Error(ErrorKind k, int code, const std::string &what) : std::runtime_error(what), kind(k), code(code) {}
    ErrorKind kind;
    int code; // the rg_status the C ABI returned
};
inline void check(int rc) {
    if (rc == RG_OK) return;
    ErrorKind k = ErrorKind::InvalidArgument;
    switch (rc) {
    case RG_ERR_STEP_LOCAL_MSG: k = ErrorKind::StepLocalMsg; break;
    case RG_ERR_STEP_PEER_NOT_FOUND: k = ErrorKind::StepPeerNotFound; break;
    case RG_ERR_HIGHER_TERM: k = ErrorKind::HigherTerm; break;
    case RG_ERR_SLOT_BUSY: k = ErrorKind::SlotBusy; break;
    case RG_ERR_NOT_ON_PATH: k = ErrorKind::NotOnPath; break;
    case RG_ERR_STATE: k = ErrorKind::State; break;
    case RG_ERR_HOST_HINT: k = ErrorKind::HostHint; break;
    case RG_ERR_NO_DEVICE: k = ErrorKind::NoDevice; break;
    case RG_ERR_OUT_OF_MEMORY: k = ErrorKind::OutOfMemory; break;
    default: break;
    }
    throw Error(k, rc, rg_last_error());
}

// ---- eraftpb::EntryType / Entry (proto/proto/eraftpb.proto:17-31) ----
enum class EntryType : std::uint32_t { EntryNormal = 0, EntryConfChange = 1, EntryConfChangeV2 = 2 };
struct Entry {
    EntryType entry_type = EntryType::EntryNormal;
    u64 term = 0, index = 0;
    std::string data, context;
    bool sync_log = false;
    rg_entry c_entry() const { // (points into this Entry)
        rg_entry e;
        std::memset(&e, 0, sizeof e);
        e.entry_type = (std::uint32_t)entry_type;
        e.sync_log = sync_log;
        e.term = term;
        e.index = index;
        e.data = reinterpret_cast<const std::uint8_t *>(data.data());
        e.data_len = data.size();
        e.context = reinterpret_cast<const std::uint8_t *>(context.data());
        e.context_len = context.size();
        return e;
    }
    u64 compute_size() const { // Entry::compute_size(): what util::limit_size adds up
        const rg_entry e = c_entry();
        return rg_entry_size(&e);
    }
};
// util::limit_size (src/util.rs:52-76): truncate to what ONE message keeps under max_size_per_msg (NO_LIMIT: everything;
// the first entry always stays)
inline void limit_size(std::vector<Entry> &entries, u64 max) {
    std::vector<rg_entry> c;
    c.reserve(entries.size());
    for (const Entry &e : entries) c.push_back(e.c_entry());
    entries.resize((std::size_t)rg_limit_size(c.data(), c.size(), max));
}

// ---- eraftpb::MessageType / Message (proto/proto/eraftpb.proto:49-92): step() takes the two response types of the path,
//      messages() hands out what the path sends ----
enum class MessageType : std::uint32_t {
    MsgHup = 0, MsgBeat = 1, MsgPropose = 2, MsgAppend = 3, MsgAppendResponse = 4, MsgRequestVote = 5, MsgRequestVoteResponse = 6,
    MsgSnapshot = 7, MsgHeartbeat = 8, MsgHeartbeatResponse = 9, MsgUnreachable = 10, MsgSnapStatus = 11, MsgCheckQuorum = 12,
    MsgTransferLeader = 13, MsgTimeoutNow = 14, MsgReadIndex = 15, MsgReadIndexResp = 16, MsgRequestPreVote = 17,
    MsgRequestPreVoteResponse = 18
};
inline bool is_local_msg(MessageType t) { // raw_node.rs:57-66
    return t == MessageType::MsgHup || t == MessageType::MsgBeat || t == MessageType::MsgUnreachable ||
           t == MessageType::MsgSnapStatus || t == MessageType::MsgCheckQuorum;
}
struct Message {
    MessageType msg_type = MessageType::MsgAppendResponse;
    u64 to = INVALID_ID, from = INVALID_ID, term = 0;
    u64 log_term = 0; // MsgAppend: term(index). A rejecting response: > 0 makes find_conflict_by_term run on the device (raft.rs:1657-1660)
    u64 index = 0, commit = 0, commit_term = 0;
    bool reject = false;
    u64 reject_hint = 0;
    u64 request_snapshot = INVALID_INDEX;
    u64 priority = 0;
    std::vector<Entry> entries;
    bool has_snapshot = false;
    std::string snapshot; // a serialised eraftpb::Snapshot (the Storage's), when has_snapshot
    std::string context;

    // protobuf::Message::compute_size / write_to_bytes: canonical proto3 bytes (rg_encode_message)
    u64 compute_size() const {
        u64 n = 0;
        with_c([&](const rg_message &m) { check_(rg_message_size(&m, &n)); });
        return n;
    }
    std::string write_to_bytes() const {
        std::string out;
        with_c([&](const rg_message &m) {
            u64 n = 0;
            check_(rg_message_size(&m, &n));
            out.resize((std::size_t)n);
            check_(rg_encode_message(&m, reinterpret_cast<std::uint8_t *>(&out[0]), n, &n));
        });
        return out;
    }

  private:
    static void check_(int rc) {
        if (rc != RG_OK) throw Error(ErrorKind::InvalidArgument, rc, rg_last_error());
    }
    template <typename F> void with_c(F f) const {
        std::vector<rg_entry> c;
        c.reserve(entries.size());
        for (const Entry &e : entries) c.push_back(e.c_entry());
        rg_message m;
        std::memset(&m, 0, sizeof m);
        m.msg_type = (std::uint32_t)msg_type;
        m.reject = reject;
        m.to = to, m.from = from, m.term = term, m.log_term = log_term, m.index = index, m.commit = commit;
        m.commit_term = commit_term, m.reject_hint = reject_hint, m.request_snapshot = request_snapshot, m.priority = priority;
        m.entries = c.data();
        m.n_entries = c.size();
        static const std::uint8_t present_but_empty = 0;
        if (has_snapshot) {
            m.snapshot = snapshot.empty() ? &present_but_empty : reinterpret_cast<const std::uint8_t *>(snapshot.data());
            m.snapshot_len = snapshot.size();
        }
        m.context = reinterpret_cast<const std::uint8_t *>(context.data());
        m.context_len = context.size();
        f(m);
    }
};

// ---- raft::Storage (src/storage.rs:65-106) for many groups, and StorageError (src/errors.rs:54-70): what building a
//      message reads. The application implements it over whatever holds its logs; nothing here stores anything. ----
enum class StorageErrorKind { Compacted, Unavailable, SnapshotOutOfDate, SnapshotTemporarilyUnavailable, Other };
class StorageError : public std::runtime_error {
  public:
    StorageError(StorageErrorKind k, const std::string &what) : std::runtime_error(what), kind(k) {}
    StorageErrorKind kind;
};
struct Snapshot {
    std::string bytes;       // the serialised eraftpb::Snapshot
    u64 index = 0, term = 0; // its metadata (eraftpb.proto:33-37)
};
class Storage {
  public:
    virtual ~Storage() = default;
    // entries [low, high) of the group's log, cut by util::limit_size(max_size) -- at least one if any is in range
    // (storage.rs:72-80); StorageError{Compacted} below first_index
    virtual std::vector<Entry> entries(u64 group, u64 low, u64 high, u64 max_size) = 0;
    // term of entry idx in [first_index - 1, last_index] (storage.rs:82-86)
    virtual u64 term(u64 group, u64 idx) = 0;
    virtual u64 first_index(u64 group) = 0;
    virtual u64 last_index(u64 group) = 0;
    // the most recent snapshot, at or above request_index; StorageError{SnapshotTemporarilyUnavailable} while it is
    // being prepared (storage.rs:98-105)
    virtual Snapshot snapshot(u64 group, u64 request_index) = 0;
};

// RaftLog::term (src/raft_log.rs:122-140) over a Storage: 0 outside [first_index - 1, last_index]; *ok = false where the
// storage cannot say (the reference's Err: compacted / unavailable)
inline u64 log_term(Storage &st, u64 group, u64 idx, bool *ok) {
    *ok = true;
    const u64 dummy = st.first_index(group) - 1;
    if (idx < dummy || idx > st.last_index(group)) return 0;
    try {
        return st.term(group, idx);
    } catch (const StorageError &) {
        *ok = false;
        return 0;
    }
}
// RaftLog::find_conflict_by_term (src/raft_log.rs:209-235): the largest index <= `index` whose term is <= `term` -- the hint a
// leader derives from a rejection that carries the follower's log_term (src/raft.rs:1657-1660). The device answers it from its
// bounded term-run table; MultiRaft::ready(Storage &) answers with this walk where that table no longer reaches (RG_OUT_HOST_HINT).
inline u64 find_conflict_by_term(Storage &st, u64 group, u64 index, u64 term) {
    u64 conflict_index = index;
    if (index > st.last_index(group)) return index; // "index is out of range": returned as is (:214-223)
    for (;;) {
        bool ok;
        const u64 t = log_term(st, group, conflict_index, &ok);
        if (!ok || t <= term) return conflict_index;
        conflict_index -= 1;
    }
}

enum class SnapshotStatus { Finish, Failure }; // src/raw_node.rs:47-54

// ---- tracker::ProgressState / Progress ----
enum class ProgressState : std::uint8_t { Probe = RG_STATE_PROBE, Replicate = RG_STATE_REPLICATE, Snapshot = RG_STATE_SNAPSHOT };
struct Progress {
    u64 matched = 0, next_idx = 0;
    ProgressState state = ProgressState::Probe;
    bool paused = false, recent_active = false;
    u64 pending_snapshot = 0, pending_request_snapshot = 0;
    u64 committed_index = 0;
    unsigned inflights = 0; // ins.count() (0 unless the engine holds the Inflights)
    bool ins_full = false;  // ins.full()
    bool is_paused() const { // progress.rs:210-216
        return state == ProgressState::Probe ? paused : state == ProgressState::Replicate ? ins_full : true;
    }
};

// What a group looks like right after Raft::become_leader (raft.rs:1151-1202) -- or at any later point a host restores:
// the configuration (tracker.rs:37-49), the leader's log summary and one Progress per peer.
struct GroupSpec {
    u64 id = 1;                 // this node (the leader)
    u64 term = 1;               // Raft.term
    std::vector<u64> voters;    // voters.incoming
    std::vector<u64> voters_outgoing; // joint consensus: voters.outgoing
    std::vector<u64> learners;
    u64 first_index_of_term = 1; // first log index whose term == term (the leader's empty entry)
    u64 last_index = 1;          // RaftLog::last_index(); entries [first_index_of_term, last_index] carry `term`
    u64 committed = 0;           // RaftLog.committed
    u64 next_idx = 0;            // followers' Progress.next_idx (0 = last_index + 1, what Progress::reset leaves)
    u64 follower_matched = 0;    // followers' Progress.matched (0 right after an election)
    ProgressState follower_state = ProgressState::Probe;
};

// One MsgAppend / MsgSnapshot the path wants sent: the host builds it (prepare_send_entries, raft.rs:714-731)
struct SendItem {
    u64 to = INVALID_ID;
    u64 prev_index = 0, last_index = 0; // entries (prev_index, last_index]; prev_index == last_index: one empty MsgAppend
    unsigned n_msgs = 0;                // messages of max_entries_per_msg entries each
    bool snapshot = false;              // prepare_send_snapshot instead (last_index = the requested index, 0 = any)
    bool host = false;                  // RG_SEND_HOST: the host runs maybe_send_append for this peer itself
};
// What RawNode::ready / LightReady carry for this path, for ONE group that saw traffic
struct LightReady {
    u64 group = 0;
    u64 commit_index = 0;   // raft_log.committed after the batch (LightReady.commit_index, raw_node.rs:643-651)
    bool commit_changed = false; // some maybe_commit() returned true (raft.rs:1745)
    bool fault = false;          // where the reference would have panicked (fatal!): the group's input was malformed
    bool timeout_now = false;    // send_timeout_now(lead_transferee) (raft.rs:1764-1774)
    bool became_leader = false;
    std::vector<u64> send_append; // peers the path called send_append for (engines WITHOUT device Inflights: the host sends)
    std::vector<u64> send_more;   // peers in the `while maybe_send_append` loop (raft.rs:1761)
    std::vector<u64> free_to;     // peers whose ins.free_to(m.index) / free_first_one the host applies
    std::vector<SendItem> messages; // engines WITH device Inflights: the send decisions, already applied to the Progress
};

// ---- from a send decision to the reference's Messages (host code: the log is the host's) ----
// Who sends, and the limits its Config carries. `committed` is raft_log.committed after the batch (LightReady.commit_index).
struct SendContext {
    u64 group = 0;
    u64 id = INVALID_ID; // Raft.id: Message.from
    u64 term = 0;        // Raft.term: Message.term (Raft::send, raft.rs:649-651)
    u64 committed = 0;
    u64 max_size_per_msg = NO_LIMIT; // Config::max_size_per_msg in bytes (config.rs:58-63): engines run with RG_SEND_BYTES
    u64 max_entries_per_msg = 0;     // the equal-sized-entries stand-in (0 = no limit): engines run without RG_SEND_BYTES
};
// One SendItem -> the Messages Raft::maybe_send_append would have pushed for that peer, in order:
//  * an append item: n_msgs MsgAppend (prepare_send_entries, raft.rs:714-731: index = next_idx - 1, log_term =
//    term(index), entries = RaftLog::entries(next_idx, max_size) -- the storage's entries under util::limit_size --,
//    commit = raft_log.committed; Raft::send: from, term), each starting where the previous one ended (update_state,
//    already applied on the device); prev_index == last_index is the single empty MsgAppend of send_append (allow_empty);
//  * a snapshot item: one MsgSnapshot carrying Storage::snapshot(requested index) (prepare_send_snapshot, raft.rs:664-712),
//    or nothing while the storage answers SnapshotTemporarilyUnavailable (:677-686). The caller applies
//    Progress::become_snapshot(snapshot index) when a message came back (MultiRaft::messages does);
//  * a host item (RG_SEND_HOST): nothing -- the host's own maybe_send_append serves that peer.
// The device counted the messages from the sizes it was given (rg_log_sizes_write); a storage that cuts differently is a
// host bug and raises Error{State} rather than sending something the Progress does not reflect.
// *snapshot_index (optional) receives metadata.index of the snapshot a MsgSnapshot carries, 0 otherwise.
inline std::vector<Message> build_messages(const SendContext &c, const SendItem &s, Storage &st, u64 *snapshot_index = nullptr) {
    std::vector<Message> out;
    if (snapshot_index) *snapshot_index = 0;
    if (s.host) return out;
    Message m;
    m.to = s.to;
    m.from = c.id;
    m.term = c.term;
    if (s.snapshot) {
        Snapshot snap;
        try {
            snap = st.snapshot(c.group, s.last_index);
        } catch (const StorageError &e) {
            if (e.kind == StorageErrorKind::SnapshotTemporarilyUnavailable) return out;
            throw;
        }
        if (snap.index == 0) throw Error(ErrorKind::State, RG_ERR_STATE, "need non-empty snapshot (raft.rs:696-698)");
        m.msg_type = MessageType::MsgSnapshot;
        m.has_snapshot = true;
        m.snapshot = snap.bytes;
        if (snapshot_index) *snapshot_index = snap.index;
        out.push_back(m);
        return out;
    }
    m.msg_type = MessageType::MsgAppend;
    m.commit = c.committed;
    u64 next = s.prev_index + 1;
    for (unsigned k = 0; k < s.n_msgs; k++) {
        m.index = next - 1;
        m.log_term = st.term(c.group, next - 1);
        m.entries.clear();
        if (next <= s.last_index) {
            m.entries = st.entries(c.group, next, s.last_index + 1, c.max_entries_per_msg ? NO_LIMIT : c.max_size_per_msg);
            if (c.max_entries_per_msg && m.entries.size() > c.max_entries_per_msg) m.entries.resize((std::size_t)c.max_entries_per_msg);
            if (m.entries.empty() || m.entries.front().index != next)
                throw Error(ErrorKind::State, RG_ERR_STATE, "build_messages: the storage has no entries where the Progress points");
            next = m.entries.back().index + 1;
        } else if (s.n_msgs != 1) {
            throw Error(ErrorKind::State, RG_ERR_STATE, "build_messages: the storage cut the entries into fewer messages than the device counted");
        }
        out.push_back(m);
    }
    if (next != s.last_index + 1)
        throw Error(ErrorKind::State, RG_ERR_STATE, "build_messages: the storage cut the entries into more messages than the device counted");
    return out;
}

struct Config { // the subset of raft::Config (src/config.rs) the path depends on + the engine's shape
    u64 n_groups = 1;
    unsigned max_peers = 3;          // peer slots per group, 1..8
    int device = 0;
    unsigned max_inflight_msgs = 0;  // 0: Inflights stay with the host; else Config::max_inflight_msgs (config.rs:112), on the device
    u64 max_entries_per_msg = 0;     // stands in for max_size_per_msg with equal-sized entries (0 = NO_LIMIT)
    u64 max_size_per_msg = NO_LIMIT; // Config::max_size_per_msg in BYTES (config.rs:58-63): takes effect with log_size_window
    unsigned log_size_window = 0;    // > 0: the device keeps the sizes of each group's last N entries (power of two, 8..4096)
                                     //      and the send stage applies max_size_per_msg byte for byte (RG_SEND_BYTES)
    bool skip_bcast_commit = false;  // Config::skip_bcast_commit (config.rs:87)
};

class MultiRaft {
  public:
    explicit MultiRaft(const Config &c) : cfg_(c) {
        if (rg_abi_version() != RG_ABI_VERSION) // (the ABI is source-compatible only: this header's struct layouts must be the library's)
            throw Error(ErrorKind::State, RG_ERR_STATE, "libraftgroups was built with another RG_ABI_VERSION than this header");
        rg_config rc;
        std::memset(&rc, 0, sizeof rc);
        rc.n_groups = c.n_groups;
        rc.n_slots = c.max_peers;
        rc.device = c.device;
        rc.max_inflight = c.max_inflight_msgs;
        check(rg_create(&rc, &h_));
        stride_ = rg_stride(h_);
        const std::size_t cells = (std::size_t)c.max_peers * stride_;
        match_.assign(cells, 0);
        next_.assign(cells, 0);
        prc_.assign(cells, 0);
        pflags_.assign((std::size_t)c.n_groups * 8, 0);
        commit_.assign(c.n_groups, 0);
        lo_.assign(c.n_groups, 1);
        hi_.assign(c.n_groups, 0);
        term_.assign(c.n_groups, 0);
        cfgw_.assign(c.n_groups, 0);
        ids_.assign((std::size_t)c.n_groups * 8, 0);
        self_.assign(c.n_groups, 0);
    }
    MultiRaft(const MultiRaft &) = delete;
    MultiRaft &operator=(const MultiRaft &) = delete;
    ~MultiRaft() {
        if (h_) rg_destroy(h_);
    }
    rg_engine *handle() const { return h_; }
    const Config &config() const { return cfg_; }

    // ---- set-up: describe every group, then bootstrap() once (bulk column loads), then step away ----
    void init_group(u64 group, const GroupSpec &s) {
        if (group >= cfg_.n_groups) throw Error(ErrorKind::InvalidArgument, RG_ERR_INVALID_ARG, "init_group: no such group");
        std::vector<u64> ids;
        auto add = [&](const std::vector<u64> &v) {
            for (u64 id : v) {
                if (id == INVALID_ID) throw Error(ErrorKind::InvalidArgument, RG_ERR_INVALID_ARG, "peer id 0 is invalid (raw_node.rs:303)");
                bool seen = false;
                for (u64 x : ids) seen = seen || x == id;
                if (!seen) ids.push_back(id);
            }
        };
        add(s.voters);
        add(s.voters_outgoing);
        add(s.learners);
        if (ids.size() > cfg_.max_peers) throw Error(ErrorKind::InvalidArgument, RG_ERR_INVALID_ARG, "init_group: more peers than max_peers");
        auto mask = [&](const std::vector<u64> &v) {
            unsigned m = 0;
            for (u64 id : v)
                for (std::size_t i = 0; i < ids.size(); i++)
                    if (ids[i] == id) m |= 1u << i;
            return m;
        };
        int self = -1;
        for (std::size_t i = 0; i < ids.size(); i++)
            if (ids[i] == s.id) self = (int)i;
        if (self < 0) throw Error(ErrorKind::InvalidArgument, RG_ERR_INVALID_ARG, "init_group: the leader is not a member");
        for (unsigned i = 0; i < 8; i++) ids_[group * 8 + i] = i < ids.size() ? ids[i] : 0;
        self_[group] = (unsigned)self;
        cfgw_[group] = RG_CFG_MAKE(mask(s.voters), mask(s.voters_outgoing), self, 0, 0, (1u << ids.size()) - 1u);
        commit_[group] = s.committed;
        lo_[group] = s.first_index_of_term;
        hi_[group] = s.last_index;
        term_[group] = s.term;
        for (std::size_t i = 0; i < ids.size(); i++) {
            const std::size_t o = i * stride_ + group;
            const bool me = (int)i == self;
            // Progress::reset(last_index + 1) for everybody, then the leader's own: matched = persisted = last_index,
            // Replicate, committed_index = committed (raft.rs:960-970, :1163-1201)
            match_[o] = me ? s.last_index : s.follower_matched;
            next_[o] = me ? s.last_index + 1 : (s.next_idx ? s.next_idx : s.last_index + 1);
            prc_[o] = me ? s.committed : 0;
            pflags_[group * 8 + i] = (std::uint8_t)((me ? RG_STATE_REPLICATE : (unsigned)s.follower_state) | (me ? RG_PF_RECENT_ACTIVE : 0u));
        }
    }
    void bootstrap() {
        load(RG_COL_MATCH, match_.data());
        load(RG_COL_NEXT, next_.data());
        load(RG_COL_PR_COMMIT, prc_.data());
        load(RG_COL_PFLAGS, pflags_.data());
        load(RG_COL_COMMIT, commit_.data());
        load(RG_COL_TERM_LO, lo_.data());
        load(RG_COL_TERM_HI, hi_.data());
        load(RG_COL_CUR_TERM, term_.data());
        load(RG_COL_CFG, cfgw_.data());
        if (cfg_.log_size_window) {
            if (!cfg_.max_inflight_msgs)
                throw Error(ErrorKind::InvalidArgument, RG_ERR_INVALID_ARG, "log_size_window needs the Inflights on the device (max_inflight_msgs > 0)");
            check(rg_log_sizes_enable(h_, cfg_.log_size_window));
            cum_.assign(cfg_.n_groups, 0);
        }
        for (u64 g = 0; g < cfg_.n_groups; g++) {
            unsigned n = 0;
            while (n < 8 && ids_[g * 8 + n]) n++;
            if (n) check(rg_set_peers(h_, g, &ids_[g * 8], n, term_[g]));
        }
        booted_ = true;
    }

    // ---- RawNode::step (raw_node.rs:402-411): local message types and unknown peers are errors; Raft::step's term
    // gate drops a stale term silently and reports a higher one (the host steps down) ----
    // `ins_full`: with the Inflights on the HOST (max_inflight_msgs == 0) the caller's Inflights::full() for m.from -- the one
    // input of handle_append_response / handle_heartbeat_response that is neither in the message nor on the device.
    void step(u64 group, const Message &m, bool ins_full = false) {
        need_boot();
        switch (m.msg_type) {
        case MessageType::MsgAppendResponse: {
            rg_append_response r;
            std::memset(&r, 0, sizeof r);
            r.from = m.from;
            r.term = m.term;
            r.index = m.index;
            r.commit = m.commit;
            r.reject = m.reject;
            r.reject_hint = m.reject_hint;
            r.log_term = m.log_term;
            r.request_snapshot = m.request_snapshot;
            r.ins_full = ins_full ? 1 : 0;
            check(rg_step(h_, group, &r));
            // (a rejection that carries log_term may come back as RG_OUT_HOST_HINT: ready(Storage &) needs it again)
            if (m.reject && m.log_term) rejects_.push_back(PendingReject{group, m.from, m.index, m.reject_hint, m.log_term});
            return;
        }
        case MessageType::MsgHeartbeatResponse:
            check(rg_step_heartbeat_response(h_, group, m.from, m.term, m.commit, ins_full ? 1 : 0));
            return;
        default:
            if (is_local_msg(m.msg_type)) // raw_node.rs:404-406
                throw Error(ErrorKind::StepLocalMsg, RG_ERR_STEP_LOCAL_MSG, "raft: cannot step raft local message");
            throw Error(ErrorKind::NotOnPath, RG_ERR_NOT_ON_PATH, "a message type this path does not handle: the host's own Raft::step takes it");
        }
    }
    // ... and on the bytes a transport delivers: Message::parse_from_bytes + RawNode::step (rg_step_bytes). A message type
    // outside the path (MsgAppend, votes, ...) is Error{NotOnPath}: the host's own Raft::step takes it.
    // `ins_full`: the caller's Inflights::full() for the sender when the Inflights live on the host (not on the wire).
    void step(u64 group, const std::uint8_t *bytes, std::size_t len, bool ins_full = false) {
        need_boot();
        check(rg_step_bytes(h_, group, bytes, len, ins_full ? 1 : 0));
    }
    // ---- the leader's own events of a batch ----
    void propose(u64 group, u64 n_entries) { // Raft::append_entry (raft.rs:976-991): last_index += n
        need_boot();
        hi_[group] += n_entries;
        check(rg_local_append(h_, group, hi_[group]));
    }
    // ... with the entries themselves (byte-accurate max_size_per_msg, Config::log_size_window > 0): their sizes go to the
    // device before the stage that may send them. The host appends them to its own log (term = the leader's, indices
    // last_index + 1 ..) -- this class keeps no entries.
    void propose(u64 group, const std::vector<Entry> &entries) {
        need_boot();
        if (cfg_.log_size_window) {
            std::vector<rg_log_size> recs(entries.size());
            for (std::size_t i = 0; i < entries.size(); i++) {
                Entry e = entries[i]; // what Raft::append_entry stamps (raft.rs:980-984)
                e.term = term_[group];
                e.index = hi_[group] + 1 + i;
                cum_[group] += e.compute_size();
                recs[i].group = group, recs[i].index = e.index, recs[i].cum_bytes = cum_[group];
            }
            if (!recs.empty()) check(rg_log_sizes_write(h_, recs.data(), recs.size()));
        }
        propose(group, (u64)entries.size());
    }
    // The sizes of the entries a group's log already holds (after bootstrap(), before the first ready(); every index the
    // send stage may reach without RG_SEND_HOST: the last log_size_window - 1 entries)
    void load_log_sizes(u64 group, Storage &st) {
        need_boot();
        if (!cfg_.log_size_window) return;
        const u64 last = st.last_index(group), first = st.first_index(group);
        u64 lo = last + 1 > cfg_.log_size_window ? last + 1 - cfg_.log_size_window + 1 : 1;
        if (lo < first) lo = first;
        if (lo > last) return;
        const std::vector<Entry> ents = st.entries(group, lo, last + 1, NO_LIMIT);
        std::vector<rg_log_size> recs(ents.size() + 1);
        recs[0].group = group, recs[0].index = lo - 1, recs[0].cum_bytes = cum_[group]; // the base the first size is a difference to
        for (std::size_t i = 0; i < ents.size(); i++) {
            cum_[group] += ents[i].compute_size();
            recs[i + 1].group = group, recs[i + 1].index = ents[i].index, recs[i + 1].cum_bytes = cum_[group];
        }
        check(rg_log_sizes_write(h_, recs.data(), recs.size()));
    }
    void on_persist_entries(u64 group, u64 index) { check(rg_local_persisted(h_, group, index)); } // raft.rs:994-1016
    void become_leader(u64 group, u64 term) {                                                       // raft.rs:1151-1202
        need_boot();
        check(rg_local_become_leader(h_, group, term));
        term_[group] = term;
        hi_[group] += 1; // the new leader's empty entry
        if (cfg_.log_size_window) { // ... whose size record is the host's to write (raftgroups.h, "entry sizes")
            Entry e;
            e.term = term, e.index = hi_[group];
            cum_[group] += e.compute_size();
            const rg_log_size rec = {group, e.index, cum_[group]};
            check(rg_log_sizes_write(h_, &rec, 1));
        }
    }
    void mark_sent(u64 group, u64 to) { check(rg_mark_sent(h_, group, to)); } // host Inflights: a MsgAppend up to last_index went out
    // ---- RawNode::report_unreachable / report_snapshot (raw_node.rs:692-709): the two local messages that write a
    // Progress -- handle_unreachable (raft.rs:1931-1954: Replicate -> Probe), handle_snapshot_status (:1891-1929:
    // Snapshot -> Probe, paused). Applied on the device at once; call ready() first if the group has stepped messages
    // pending (Error{SlotBusy}: the reference applies local messages in call order). An unknown id is ignored. ----
    void report_unreachable(u64 group, u64 id) {
        need_boot();
        check(rg_report_unreachable(h_, group, id));
    }
    void report_snapshot(u64 group, u64 id, SnapshotStatus status) {
        need_boot();
        check(rg_report_snapshot(h_, group, id, status == SnapshotStatus::Failure));
    }

    // ---- RawNode::ready for every group with queued traffic: ONE launch for all of them ----
    // ready(): throws Error{State} if a rejection of the batch needs the host's log (RG_OUT_HOST_HINT: the group's log has seen more
    // term changes since its last snapshot than the device's table holds) -- use ready(Storage &), which answers from the Storage
    // (find_conflict_by_term) through rg_resolve_host_hints before it reports the batch.
    std::vector<LightReady> ready() { return ready_impl(nullptr); }
    std::vector<LightReady> ready(Storage &st) { return ready_impl(&st); }

  private:
    struct PendingReject {
        u64 group, from, index, reject_hint, log_term;
    };
    std::vector<LightReady> ready_impl(Storage *st) {
        need_boot();
        const bool dev_ins = cfg_.max_inflight_msgs != 0;
        if (dev_ins)
            check(rg_flush_send(h_, cfg_.log_size_window ? cfg_.max_size_per_msg : cfg_.max_entries_per_msg,
                                (cfg_.skip_bcast_commit ? RG_SEND_SKIP_BCAST_COMMIT : 0u) | (cfg_.log_size_window ? RG_SEND_BYTES : 0u)));
        else
            check(rg_flush(h_));
        u64 n = 0;
        check(rg_ingested_results(h_, nullptr, nullptr, nullptr, 0, &n));
        std::vector<u64> groups(n), commit(n);
        std::vector<std::uint32_t> out(n);
        if (n) check(rg_ingested_results(h_, groups.data(), commit.data(), out.data(), n, &n));
        // RG_OUT_HOST_HINT: maybe_decr_to of these rejections waits for find_conflict_by_term on the HOST's log
        // (src/raft.rs:1657-1660, src/raft_log.rs:209-235); with the Inflights on the device so do the groups' sends
        bool hinted = false;
        for (u64 i = 0; i < n; i++) hinted = hinted || (out[i] & RG_OUT_HOST_HINT);
        if (hinted) {
            if (!st)
                throw Error(ErrorKind::State, RG_ERR_STATE,
                            "a rejection of this batch needs the host's log (RG_OUT_HOST_HINT): call ready(Storage &)");
            u64 nh = 0;
            check(rg_host_hints(h_, nullptr, 0, &nh));
            std::vector<rg_host_hint> hh(nh);
            if (nh) check(rg_host_hints(h_, hh.data(), nh, &nh));
            std::vector<rg_resolved_hint> res;
            for (const rg_host_hint &x : hh)
                for (unsigned s = 0; s < cfg_.max_peers; s++) {
                    if (!((x.slot_mask >> s) & 1u)) continue;
                    const u64 from = ids_[x.group * 8 + s];
                    const PendingReject *pr = nullptr;
                    for (const PendingReject &c : rejects_)
                        if (c.group == x.group && c.from == from) pr = &c;
                    if (!pr) throw Error(ErrorKind::State, RG_ERR_STATE, "RG_OUT_HOST_HINT for a rejection this host did not step");
                    rg_resolved_hint r;
                    std::memset(&r, 0, sizeof r);
                    r.group = x.group, r.slot = s, r.index = pr->index;
                    r.hint = find_conflict_by_term(*st, x.group, pr->reject_hint, pr->log_term);
                    res.push_back(r);
                }
            if (!res.empty()) check(rg_resolve_host_hints(h_, res.data(), res.size(), nullptr));
            // the completed result words of those groups (the compact results above are not rewritten)
            std::vector<u64> hg;
            for (const rg_host_hint &x : hh) hg.push_back(x.group);
            std::vector<rg_group_status> gs(hg.size());
            if (!hg.empty()) check(rg_read_groups(h_, hg.data(), hg.size(), gs.data()));
            for (u64 i = 0; i < n; i++)
                for (std::size_t k = 0; k < hg.size(); k++)
                    if (groups[i] == hg[k]) out[i] = gs[k].out;
        }
        rejects_.clear();
        std::vector<LightReady> rd(n);
        for (u64 i = 0; i < n; i++) {
            LightReady &r = rd[i];
            const u64 g = groups[i];
            r.group = g;
            r.commit_index = commit[i];
            r.commit_changed = (out[i] & RG_OUT_CHANGED) != 0;
            r.fault = (out[i] & RG_OUT_FAULT) != 0;
            r.timeout_now = (out[i] & RG_OUT_TIMEOUT_NOW) != 0;
            r.became_leader = (out[i] & RG_OUT_BECAME_LEADER) != 0;
            for (unsigned s = 0; s < cfg_.max_peers; s++) {
                if ((RG_OUT_SEND_APPEND(out[i]) >> s) & 1u) r.send_append.push_back(ids_[g * 8 + s]);
                if ((RG_OUT_SEND_MORE(out[i]) >> s) & 1u) r.send_more.push_back(ids_[g * 8 + s]);
                if ((RG_OUT_FREE_TO(out[i]) >> s) & 1u) r.free_to.push_back(ids_[g * 8 + s]);
            }
        }
        if (dev_ins && n) {
            u64 k = 0;
            std::vector<rg_send_item> items(n * cfg_.max_peers); // at most one item per peer of a touched group
            check(rg_send_items(h_, items.data(), items.size(), &k));
            items.resize(k < items.size() ? k : items.size());
            for (const rg_send_item &it : items) {
                for (LightReady &r : rd) {
                    if (r.group != it.group) continue;
                    SendItem s;
                    s.to = ids_[it.group * 8 + it.slot];
                    s.prev_index = it.prev_index;
                    s.last_index = it.last_index;
                    s.n_msgs = it.n_msgs;
                    s.snapshot = it.kind == RG_SEND_SNAPSHOT;
                    s.host = it.kind == RG_SEND_HOST;
                    r.messages.push_back(s);
                    break;
                }
            }
        }
        return rd;
    }

  public:
    // ---- Ready.messages for one group of ready(): the send decisions as the reference's Messages, built out of the host's
    // Storage (build_messages); a snapshot that was actually fetched is followed by Progress::become_snapshot on the
    // device (progress.rs:117-121), as prepare_send_snapshot does (raft.rs:699-711) ----
    std::vector<Message> messages(const LightReady &rd, Storage &st) {
        need_boot();
        SendContext c;
        c.group = rd.group;
        c.id = ids_[rd.group * 8 + self_[rd.group]];
        c.term = term_[rd.group];
        c.committed = rd.commit_index;
        if (cfg_.log_size_window) c.max_size_per_msg = cfg_.max_size_per_msg;
        else c.max_entries_per_msg = cfg_.max_entries_per_msg;
        std::vector<Message> out;
        for (const SendItem &s : rd.messages) {
            u64 sindex = 0; // metadata.index of the snapshot that goes out
            const std::vector<Message> ms = build_messages(c, s, st, &sindex);
            if (s.snapshot && !ms.empty()) become_snapshot(rd.group, s.to, sindex);
            out.insert(out.end(), ms.begin(), ms.end());
        }
        return out;
    }
    // ---- Raft::bcast_heartbeat (raft.rs:885-891 -> send_heartbeat :822-844) for one group: a MsgHeartbeat per peer with a
    // Progress, the leader excepted, carrying commit = min(pr.matched, raft_log.committed) -- "the leader MUST NOT forward
    // the follower's commit to an unmatched index" -- and the read-index context, if any ----
    std::vector<Message> bcast_heartbeat(u64 group, const std::string &ctx = std::string()) {
        need_boot();
        rg_group_status st;
        check(rg_read_groups(h_, &group, 1, &st));
        std::vector<Message> out;
        for (unsigned s = 0; s < cfg_.max_peers; s++) {
            if (!((RG_CFG_PRESENT(st.cfg) >> s) & 1u) || s == RG_CFG_SELF(st.cfg)) continue;
            Message m;
            m.msg_type = MessageType::MsgHeartbeat;
            m.to = ids_[group * 8 + s];
            m.from = ids_[group * 8 + RG_CFG_SELF(st.cfg)];
            m.term = term_[group];
            m.commit = st.match[s] < st.commit ? st.match[s] : st.commit;
            m.context = ctx;
            out.push_back(m);
        }
        return out;
    }
    // Progress::become_snapshot(snapshot_idx) (progress.rs:117-121): reset_state(Snapshot) + pending_snapshot
    void become_snapshot(u64 group, u64 id, u64 snapshot_idx) {
        rg_group_status st;
        check(rg_read_groups(h_, &group, 1, &st));
        const int s = slot_of(group, id);
        rg_cell_write c;
        std::memset(&c, 0, sizeof c);
        c.group = group;
        c.slot = (std::uint32_t)s;
        c.field_mask = (1u << RG_COL_PEND_SNAP) | (1u << RG_COL_PFLAGS);
        c.pend_snap = snapshot_idx;
        c.pflags = (std::uint8_t)((st.pflags[s] & ~(RG_PF_STATE_MASK | RG_PF_PAUSED)) | RG_STATE_SNAPSHOT);
        check(rg_write_cells(h_, &c, 1));
    }

    // ---- ProgressTracker::get / Status (tracker.rs:261-287, status.rs:25-52) ----
    Progress progress(u64 group, u64 id) {
        rg_group_status st;
        check(rg_read_groups(h_, &group, 1, &st));
        const int s = slot_of(group, id);
        Progress p;
        p.matched = st.match[s];
        p.next_idx = st.next[s];
        p.state = (ProgressState)(st.pflags[s] & RG_PF_STATE_MASK);
        p.paused = (st.pflags[s] & RG_PF_PAUSED) != 0;
        p.recent_active = (st.pflags[s] & RG_PF_RECENT_ACTIVE) != 0;
        p.pending_snapshot = st.pend_snap[s];
        p.pending_request_snapshot = st.pend_rs[s];
        p.committed_index = st.pr_commit[s];
        p.inflights = st.inflights[s];
        p.ins_full = (st.pflags[s] & RG_PF_INS_FULL) != 0;
        return p;
    }
    u64 committed(u64 group) { // raft_log.committed
        rg_group_status st;
        check(rg_read_groups(h_, &group, 1, &st));
        return st.commit;
    }
    u64 last_index(u64 group) {
        rg_group_status st;
        check(rg_read_groups(h_, &group, 1, &st));
        return st.last_index;
    }
    // Overwrite fields of one Progress (what the send path and the other host-side writers do between batches:
    // update_state / become_snapshot / become_probe ...)
    void set_progress(u64 group, u64 id, const Progress &p) {
        rg_cell_write c;
        std::memset(&c, 0, sizeof c);
        c.group = group;
        c.slot = (std::uint32_t)slot_of(group, id);
        c.field_mask = (1u << RG_COL_MATCH) | (1u << RG_COL_NEXT) | (1u << RG_COL_PR_COMMIT) | (1u << RG_COL_PEND_SNAP) |
                       (1u << RG_COL_PEND_RS) | (1u << RG_COL_PFLAGS);
        c.match = p.matched;
        c.next = p.next_idx;
        c.pr_commit = p.committed_index;
        c.pend_snap = p.pending_snapshot;
        c.pend_rs = p.pending_request_snapshot;
        c.pflags = (std::uint8_t)((unsigned)p.state | (p.paused ? RG_PF_PAUSED : 0u) | (p.recent_active ? RG_PF_RECENT_ACTIVE : 0u));
        check(rg_write_cells(h_, &c, 1));
    }

  private:
    void need_boot() const {
        if (!booted_) throw Error(ErrorKind::State, RG_ERR_STATE, "MultiRaft: bootstrap() has not run");
    }
    int slot_of(u64 group, u64 id) const {
        if (group < cfg_.n_groups && id != INVALID_ID)
            for (unsigned s = 0; s < cfg_.max_peers; s++)
                if (ids_[group * 8 + s] == id) return (int)s;
        throw Error(ErrorKind::StepPeerNotFound, RG_ERR_STEP_PEER_NOT_FOUND, "raft: no such peer in this group");
    }
    void load(int col, const void *src) { check(rg_load_column(h_, col, src, rg_column_bytes(h_, col))); }

    Config cfg_;
    rg_engine *h_ = nullptr;
    u64 stride_ = 0;
    bool booted_ = false;
    std::vector<PendingReject> rejects_; // rejections with log_term stepped since the last ready()
    std::vector<u64> match_, next_, prc_, commit_, lo_, hi_, term_, ids_, cum_;
    std::vector<std::uint8_t> pflags_;
    std::vector<std::uint32_t> cfgw_;
    std::vector<unsigned> self_;
};

} // namespace raftgroups
