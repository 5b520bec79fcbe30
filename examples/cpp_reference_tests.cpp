// cpp_reference_tests.cpp -- the reference's own tests for this path, restated against include/raftgroups.hpp (the C++
// host side above the C ABI) and run on the GPU. Each function names the reference test it follows; the expectations are
// the reference's rows, verbatim.
//   g++ -std=c++17 -Iinclude examples/cpp_reference_tests.cpp -o cpp_reference_tests -Lraft_rs_amd -lraftgroups
//   (plus -Wl,-rpath,$PWD/raft_rs_amd), then ./cpp_reference_tests on an MI355X
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>

#include "log_fixture.hpp"

using namespace raftgroups;

#define EXPECT(cond, ...)                                   \
    do {                                                    \
        if (!(cond)) {                                      \
            std::fprintf(stderr, "%s:%d: ", __FILE__, __LINE__); \
            std::fprintf(stderr, __VA_ARGS__);              \
            std::fprintf(stderr, "\n");                     \
            std::exit(1);                                   \
        }                                                   \
    } while (0)

static Message new_message(u64 from, MessageType t, u64 term) {
    Message m;
    m.from = from;
    m.msg_type = t;
    m.term = term;
    return m;
}
static std::vector<u64> range(u64 n) {
    std::vector<u64> v;
    for (u64 i = 1; i <= n; i++) v.push_back(i);
    return v;
}
static size_t count_messages(const std::vector<LightReady> &rd) {
    size_t n = 0;
    for (const LightReady &r : rd)
        for (const SendItem &m : r.messages) n += m.n_msgs;
    return n;
}

// harness/tests/integration_cases/test_raft.rs:2611-2675
static void test_leader_append_response() {
    struct Row { u64 index; bool reject; u64 wmatch, wnext; size_t wmsg_num; u64 windex, wcommitted; };
    const Row tests[] = {
        {3, true, 0, 3, 0, 0, 0},  // stale resp; no replies
        {2, true, 0, 2, 1, 1, 0},  // denied resp; decrease next and send probing message
        {2, false, 2, 4, 2, 2, 2}, // accepted resp; leader commits; broadcast with committed index
        {0, false, 0, 3, 0, 0, 0},
    };
    int i = 0;
    for (const Row &t : tests) {
        Config c;
        c.n_groups = 1;
        c.max_peers = 3;
        c.max_inflight_msgs = 256;
        MultiRaft sm(c);
        // initial raft logs: entries (term 0, 1) (term 1, 2); become_candidate takes the node to term 1, become_leader
        // appends its empty entry (term 1, 3); followers: match = 0, next = 3 (Progress::reset ran before the append)
        GroupSpec s;
        s.id = 1;
        s.term = 1;
        s.voters = {1, 2, 3};
        s.first_index_of_term = 2;
        s.last_index = 3;
        s.committed = 0;
        s.next_idx = 3;
        sm.init_group(0, s);
        sm.bootstrap();

        Message m = new_message(2, MessageType::MsgAppendResponse, 1);
        m.index = t.index;
        m.reject = t.reject;
        m.reject_hint = t.index;
        sm.step(0, m);
        const std::vector<LightReady> rd = sm.ready();

        const Progress pr = sm.progress(0, 2);
        EXPECT(pr.matched == t.wmatch, "#%d: match = %llu, want %llu", i, (unsigned long long)pr.matched, (unsigned long long)t.wmatch);
        EXPECT(pr.next_idx == t.wnext, "#%d: next = %llu, want %llu", i, (unsigned long long)pr.next_idx, (unsigned long long)t.wnext);
        EXPECT(count_messages(rd) == t.wmsg_num, "#%d msg_num = %zu, want %zu", i, count_messages(rd), t.wmsg_num);
        for (const LightReady &r : rd)
            for (const SendItem &msg : r.messages) {
                EXPECT(msg.prev_index == t.windex, "#%d index = %llu, want %llu", i, (unsigned long long)msg.prev_index, (unsigned long long)t.windex);
                EXPECT(r.commit_index == t.wcommitted, "#%d commit = %llu, want %llu", i, (unsigned long long)r.commit_index, (unsigned long long)t.wcommitted);
            }
        EXPECT(sm.committed(0) == t.wcommitted, "#%d committed = %llu", i, (unsigned long long)sm.committed(0));
        i++;
    }
}

// harness/tests/integration_cases/test_raft_paper.rs:499-534: an entry is committed once the leader that created it has
// replicated it on a majority of the servers
static void test_leader_acknowledge_commit() {
    struct Row { u64 size; std::set<u64> acceptors; bool wack; };
    const Row tests[] = {
        {1, {}, true},      {3, {}, false},       {3, {2}, true},          {3, {2, 3}, true},          {5, {}, false},
        {5, {2}, false},    {5, {2, 3}, true},    {5, {2, 3, 4}, true},    {5, {2, 3, 4, 5}, true},
    };
    int i = 0;
    for (const Row &t : tests) {
        Config c;
        c.n_groups = 1;
        c.max_peers = (unsigned)t.size;
        c.max_inflight_msgs = 256;
        MultiRaft r(c);
        // become_candidate + become_leader at term 1, commit_noop_entry: everybody holds and has acknowledged entry 1
        GroupSpec s;
        s.id = 1;
        s.term = 1;
        s.voters = range(t.size);
        s.first_index_of_term = 1;
        s.last_index = 1;
        s.committed = 1;
        s.next_idx = 2;
        s.follower_matched = 1;
        s.follower_state = ProgressState::Replicate;
        r.init_group(0, s);
        r.bootstrap();
        const u64 li = r.last_index(0);
        r.propose(0, 1); // MsgPropose with one entry
        r.on_persist_entries(0, li + 1);
        const std::vector<LightReady> rd = r.ready();
        // the MsgAppend of the proposal went to every follower: entries (1, 2]
        std::map<u64, SendItem> sent;
        for (const LightReady &x : rd)
            for (const SendItem &m : x.messages) sent[m.to] = m;
        EXPECT(sent.size() == t.size - 1, "#%d: %zu MsgAppend, want %llu", i, sent.size(), (unsigned long long)(t.size - 1));
        for (const auto &kv : sent) {
            if (!t.acceptors.count(kv.first)) continue;
            Message m = new_message(kv.first, MessageType::MsgAppendResponse, 1); // accept_and_reply
            m.index = kv.second.last_index;
            r.step(0, m);
        }
        r.ready();
        const bool g = r.committed(0) > li;
        EXPECT(g == t.wack, "#%d: ack commit = %d, want %d", i, (int)g, (int)t.wack);
        i++;
    }
}

// harness/tests/integration_cases/test_raft_flow_control.rs:24-58: a full Inflights window stops the leader from sending
static void test_msg_app_flow_control_full() {
    Config c;
    c.n_groups = 1;
    c.max_peers = 2;
    c.max_inflight_msgs = 256;
    MultiRaft r(c);
    GroupSpec s;
    s.id = 1;
    s.term = 1;
    s.voters = {1, 2};
    s.first_index_of_term = 1;
    s.last_index = 1;
    s.committed = 0;
    s.next_idx = 1; // force the progress to be in replicate state: become_replicate -> next = matched + 1
    s.follower_state = ProgressState::Replicate;
    r.init_group(0, s);
    r.bootstrap();
    for (unsigned i = 0; i < c.max_inflight_msgs; i++) { // fill in the inflights window
        r.propose(0, 1);
        const size_t ms = count_messages(r.ready());
        EXPECT(ms == 1, "#%u: ms count = %zu, want 1", i, ms);
    }
    EXPECT(r.progress(0, 2).ins_full, "ensure 1: the window must be full");
    for (unsigned i = 0; i < 10; i++) { // ensure 2
        r.propose(0, 1);
        const size_t ms = count_messages(r.ready());
        EXPECT(ms == 0, "#%u: ms count = %zu, want 0", i, ms);
    }
}

// src/raw_node.rs:402-411 (RawNode::step) and the term gate of Raft::step (src/raft.rs:1282-1411)
static void test_raw_node_step() {
    Config c;
    c.n_groups = 2;
    c.max_peers = 3;
    MultiRaft rn(c);
    for (u64 g = 0; g < 2; g++) {
        GroupSpec s;
        s.id = 1;
        s.term = 3;
        s.voters = {1, 2, 3};
        s.first_index_of_term = 5;
        s.last_index = 9;
        s.committed = 4;
        s.next_idx = 5;
        rn.init_group(g, s);
    }
    rn.bootstrap();
    try { // raw_node.rs:404-406: local messages are refused
        rn.step(0, new_message(1, MessageType::MsgHup, 0));
        EXPECT(false, "a local message must be refused");
    } catch (const Error &e) {
        EXPECT(e.kind == ErrorKind::StepLocalMsg, "want StepLocalMsg, got %d", e.code);
    }
    try { // raw_node.rs:407-410: a response from a peer without a Progress
        rn.step(0, new_message(9, MessageType::MsgAppendResponse, 3));
        EXPECT(false, "an unknown peer must be refused");
    } catch (const Error &e) {
        EXPECT(e.kind == ErrorKind::StepPeerNotFound, "want StepPeerNotFound, got %d", e.code);
    }
    try { // raft.rs:1284-1348: a higher term deposes the leader -- the host's business
        rn.step(0, new_message(2, MessageType::MsgAppendResponse, 4));
        EXPECT(false, "a higher term must be reported");
    } catch (const Error &e) {
        EXPECT(e.kind == ErrorKind::HigherTerm, "want HigherTerm, got %d", e.code);
    }
    { // the same gate on the wire bytes: msg_type = MsgRequestVote (5), from = 2, term = 3 -> not this path's
        const std::uint8_t vote[] = {0x08, 0x05, 0x18, 0x02, 0x20, 0x03};
        try {
            rn.step(0, vote, sizeof vote);
            EXPECT(false, "a vote request is not on this path");
        } catch (const Error &e) {
            EXPECT(e.kind == ErrorKind::NotOnPath, "want NotOnPath, got %d", e.code);
        }
    }
    Message stale = new_message(2, MessageType::MsgAppendResponse, 2); // raft.rs:1349-1411: a lower term is dropped
    stale.index = 9;
    rn.step(0, stale);
    // MsgAppendResponse { from: 2, term: 3, index: 9 } as protobuf bytes (field 1 = 4, 3 = 2, 4 = 3, 6 = 9)
    const std::uint8_t ack[] = {0x08, 0x04, 0x18, 0x02, 0x20, 0x03, 0x30, 0x09};
    rn.step(1, ack, sizeof ack);
    const std::vector<LightReady> rd = rn.ready();
    EXPECT(rn.committed(0) == 4 && rn.progress(0, 2).matched == 0, "the stale response must change nothing");
    EXPECT(rn.committed(1) == 9 && rn.progress(1, 2).matched == 9, "group 1 commits 9 with {1, 2}");
    bool seen = false;
    for (const LightReady &r : rd)
        if (r.group == 1) {
            seen = true;
            EXPECT(r.commit_changed && r.commit_index == 9, "LightReady.commit_index of group 1");
        }
    EXPECT(seen, "group 1 must have a Ready");
}

// harness/tests/integration_cases/test_raft_paper.rs:425-457: upon receiving a proposal the leader appends it and issues
// AppendEntries to every follower -- checked on the MESSAGES (MultiRaft::messages builds them from the send stage's work
// items and the host's Storage) and on their bytes
static void test_leader_start_replication() {
    Config c;
    c.n_groups = 1;
    c.max_peers = 3;
    c.max_inflight_msgs = 256;
    MultiRaft r(c);
    LogFixture st;
    st.append(0, 1); // become_leader's empty entry (term 1, index 1), committed by commit_noop_entry
    GroupSpec s;
    s.id = 1;
    s.term = 1;
    s.voters = {1, 2, 3};
    s.first_index_of_term = 1;
    s.last_index = 1;
    s.committed = 1;
    s.next_idx = 2;
    s.follower_matched = 1;
    s.follower_state = ProgressState::Replicate;
    r.init_group(0, s);
    r.bootstrap();
    const u64 li = r.last_index(0);
    st.append(0, 1, "somedata"); // MsgPropose: the host's log takes the entry ...
    r.propose(0, 1);             // ... and the path hears about it
    const std::vector<LightReady> rd = r.ready();
    EXPECT(r.last_index(0) == li + 1, "last_index = %llu, want %llu", (unsigned long long)r.last_index(0), (unsigned long long)(li + 1));
    EXPECT(r.committed(0) == li, "committed = %llu, want %llu", (unsigned long long)r.committed(0), (unsigned long long)li);
    EXPECT(rd.size() == 1, "one group saw traffic");
    std::map<u64, Message> msgs;
    for (const Message &m : r.messages(rd[0], st)) msgs[m.to] = m;
    EXPECT(msgs.size() == 2 && msgs.count(2) && msgs.count(3), "a MsgAppend for each follower");
    for (const auto &kv : msgs) {
        const Message &m = kv.second;
        EXPECT(m.msg_type == MessageType::MsgAppend && m.from == 1 && m.term == 1 && m.index == li && m.log_term == 1 && m.commit == li,
               "MsgAppend to %llu: index %llu log_term %llu commit %llu", (unsigned long long)kv.first, (unsigned long long)m.index,
               (unsigned long long)m.log_term, (unsigned long long)m.commit);
        EXPECT(m.entries.size() == 1 && m.entries[0].term == 1 && m.entries[0].index == li + 1 && m.entries[0].data == "somedata",
               "wents = [(1, li + 1, SOME_DATA)]");
        const std::string bytes = m.write_to_bytes();
        rg_decoded_message d;
        check(rg_decode_message(reinterpret_cast<const std::uint8_t *>(bytes.data()), bytes.size(), &d));
        EXPECT(d.msg_type == 3 && d.to == kv.first && d.from == 1 && d.index == li && d.commit == li && d.n_entries == 1, "the bytes read back");
    }
}

// harness/tests/integration_cases/test_raft.rs:369-435 with its literal configuration: max_inflight_msgs = 3,
// max_size_per_msg = 2048 BYTES, ten 1000-byte proposals. The device counts the messages from the entry sizes
// (RG_SEND_BYTES); the host cuts the same entries out of its Storage -- the two must agree message by message.
static void test_progress_flow_control() {
    Config c;
    c.n_groups = 1;
    c.max_peers = 2;
    c.max_inflight_msgs = 3;
    c.max_size_per_msg = 2048;
    c.log_size_window = 64;
    MultiRaft r(c);
    LogFixture st;
    st.append(0, 1); // the empty entry that confirms the election
    GroupSpec s;
    s.id = 1;
    s.term = 1;
    s.voters = {1, 2};
    s.first_index_of_term = 1;
    s.last_index = 1;
    s.committed = 0;
    s.next_idx = 1; // Progress::reset(last_index + 1) ran before the empty entry was appended
    s.follower_state = ProgressState::Probe;
    r.init_group(0, s);
    r.bootstrap();
    r.load_log_sizes(0, st);
    { // nothing is persisted in this test (no r.persist()): the leader's own matched stays 0, the commit index cannot move
        Progress me = r.progress(0, 1);
        me.matched = 0;
        r.set_progress(0, 1, me);
    }
    // While node 2 is in probe state, propose a bunch of entries.
    const std::string data(1000, 'a');
    std::vector<Message> ms;
    for (int i = 0; i < 10; i++) {
        st.append(0, 1, data);
        Entry e;
        e.data = data;
        r.propose(0, std::vector<Entry>{e});
        for (const LightReady &x : r.ready())
            for (const Message &m : r.messages(x, st)) ms.push_back(m);
    }
    // First append has two entries: the empty entry to confirm the election, and the first proposal (only one proposal
    // gets sent because we're in probe state).
    EXPECT(ms.size() == 1, "%zu messages in Probe, want 1", ms.size());
    EXPECT(ms[0].msg_type == MessageType::MsgAppend && ms[0].entries.size() == 2 && ms[0].entries[0].data.empty() &&
               ms[0].entries[1].data.size() == 1000,
           "the first append: the empty entry and the first proposal");
    EXPECT(ms[0].entries[0].compute_size() == 4 && ms[0].entries[1].compute_size() == 1007, "entry sizes 4 and 1007");
    // When this append is acked, we change to replicate state and can send multiple messages at once.
    Message ack = new_message(2, MessageType::MsgAppendResponse, 1);
    ack.index = ms[0].entries[1].index;
    r.step(0, ack);
    ms.clear();
    for (const LightReady &x : r.ready())
        for (const Message &m : r.messages(x, st)) ms.push_back(m);
    EXPECT(ms.size() == 3, "%zu messages after the ack, want 3", ms.size());
    for (std::size_t i = 0; i < ms.size(); i++)
        EXPECT(ms[i].msg_type == MessageType::MsgAppend && ms[i].entries.size() == 2, "%zu: expected 2 entries, got %zu", i, ms[i].entries.size());
    EXPECT(r.progress(0, 2).ins_full, "three messages in flight: the window is full");
    // Ack all three of those messages together and get the last two messages (containing three entries).
    ack.index = ms[2].entries[1].index;
    r.step(0, ack);
    ms.clear();
    for (const LightReady &x : r.ready())
        for (const Message &m : r.messages(x, st)) ms.push_back(m);
    EXPECT(ms.size() == 2, "%zu messages after the second ack, want 2", ms.size());
    EXPECT(ms[0].entries.size() == 2 && ms[1].entries.size() == 1, "2 + 1 entries");
    EXPECT(ms[1].entries[0].index == 11 && r.progress(0, 2).next_idx == 12, "everything up to entry 11 is on its way");
    for (const Message &m : ms) { // every message obeys the byte limit the way util::limit_size defines it
        u64 total = 0;
        for (const Entry &e : m.entries) total += e.compute_size();
        EXPECT(total <= 2048 && m.write_to_bytes().size() == m.compute_size(), "a message of %llu entry bytes", (unsigned long long)total);
    }
}

// harness/tests/integration_cases/test_raft.rs:2913-2933: MsgUnreachable puts a Replicate peer back to Probe
static void test_recv_msg_unreachable() {
    Config c;
    c.n_groups = 1;
    c.max_peers = 2;
    MultiRaft r(c);
    GroupSpec s; // three previous entries (term 1), become_leader's empty entry at 4
    s.id = 1;
    s.term = 1;
    s.voters = {1, 2};
    s.first_index_of_term = 1;
    s.last_index = 4;
    s.committed = 0;
    r.init_group(0, s);
    r.bootstrap();
    Progress p2 = r.progress(0, 2); // set node 2 to state replicate
    p2.matched = 3;
    p2.state = ProgressState::Replicate;
    p2.next_idx = 6; // become_replicate (next = 4), optimistic_update(5)
    r.set_progress(0, 2, p2);
    r.report_unreachable(0, 2);
    const Progress peer_2 = r.progress(0, 2);
    EXPECT(peer_2.state == ProgressState::Probe, "state = %d, want Probe", (int)peer_2.state);
    EXPECT(peer_2.matched + 1 == peer_2.next_idx, "matched %llu next_idx %llu", (unsigned long long)peer_2.matched,
           (unsigned long long)peer_2.next_idx);
    r.report_unreachable(0, 7); // no progress available: ignored
}

// harness/tests/integration_cases/test_raft_snap.rs:68-87 (test_snapshot_failure) and :89-109 (test_snapshot_succeed)
static void test_snapshot_failure_and_succeed() {
    struct Row { SnapshotStatus status; u64 wnext; };
    const Row tests[] = {{SnapshotStatus::Failure, 1}, {SnapshotStatus::Finish, 12}};
    for (const Row &t : tests) {
        Config c;
        c.n_groups = 1;
        c.max_peers = 2;
        MultiRaft sm(c);
        GroupSpec s; // sm.restore(testing_snap()): snapshot (index 11, term 11); become_leader's empty entry at 12
        s.id = 1;
        s.term = 1;
        s.voters = {1, 2};
        s.first_index_of_term = 12;
        s.last_index = 12;
        s.committed = 11;
        sm.init_group(0, s);
        sm.bootstrap();
        Progress p2 = sm.progress(0, 2);
        p2.next_idx = 1;
        p2.state = ProgressState::Snapshot; // become_snapshot(11)
        p2.pending_snapshot = 11;
        sm.set_progress(0, 2, p2);
        sm.report_snapshot(0, 2, t.status);
        const Progress voter_2 = sm.progress(0, 2);
        EXPECT(voter_2.pending_snapshot == 0, "pending_snapshot = %llu", (unsigned long long)voter_2.pending_snapshot);
        EXPECT(voter_2.next_idx == t.wnext, "next_idx = %llu, want %llu", (unsigned long long)voter_2.next_idx, (unsigned long long)t.wnext);
        EXPECT(voter_2.paused && voter_2.state == ProgressState::Probe, "paused Probe");
    }
}

// harness/tests/integration_cases/test_raft.rs:2680-2752: MsgBeat makes the leader send a MsgHeartbeat to every follower,
// with commit = min(matched, committed), no entries, index 0, log_term 0
static void test_bcast_beat() {
    const u64 offset = 1000;
    struct Row { u64 committed, want2, want3; };
    // the reference's row (nothing is acknowledged: committed = the snapshot's index), and one where a follower's matched
    // index is what bounds its heartbeat's commit
    const Row tests[] = {{offset, offset, offset}, {offset + 8, offset + 5, offset + 8}};
    for (const Row &t : tests) {
        Config c;
        c.n_groups = 1;
        c.max_peers = 3;
        MultiRaft sm(c);
        GroupSpec s; // log.offset = 1000, term 2 after become_candidate, the empty entry at 1001 and ten more entries
        s.id = 1;
        s.term = 2;
        s.voters = {1, 2, 3};
        s.first_index_of_term = offset + 1;
        s.last_index = offset + 11;
        s.committed = t.committed;
        sm.init_group(0, s);
        sm.bootstrap();
        const u64 last_index = sm.last_index(0);
        Progress p = sm.progress(0, 2); // slow follower
        p.matched = offset + 5, p.next_idx = offset + 6;
        sm.set_progress(0, 2, p);
        p = sm.progress(0, 3); // normal follower
        p.matched = last_index, p.next_idx = last_index + 1;
        sm.set_progress(0, 3, p);
        const std::vector<Message> msgs = sm.bcast_heartbeat(0);
        EXPECT(msgs.size() == 2, "%zu heartbeats, want 2", msgs.size());
        std::map<u64, u64> want_commit_map = {{2, t.want2}, {3, t.want3}};
        for (const Message &m : msgs) {
            EXPECT(m.msg_type == MessageType::MsgHeartbeat && m.from == 1 && m.term == 2, "type = %u", (unsigned)m.msg_type);
            EXPECT(m.index == 0 && m.log_term == 0 && m.entries.empty(), "prev_index / prev_term / entries");
            EXPECT(want_commit_map.count(m.to) && m.commit == want_commit_map[m.to], "to %llu: commit = %llu", (unsigned long long)m.to,
                   (unsigned long long)m.commit);
            want_commit_map.erase(m.to);
            const std::string bytes = m.write_to_bytes();
            rg_decoded_message d;
            check(rg_decode_message(reinterpret_cast<const std::uint8_t *>(bytes.data()), bytes.size(), &d));
            EXPECT(d.msg_type == 8 && d.to == m.to && d.commit == m.commit && d.term == 2, "the heartbeat's bytes");
        }
    }
}

// harness/tests/integration_cases/test_raft.rs:5573-5839 test_fast_log_rejection, leader side, three of its rows: the follower's
// rejection carries (reject_hint, log_term) and the leader turns it into the next probe through find_conflict_by_term
// (raft_log.rs:209-235, raft.rs:1657-1660). This host loads no term-run table, so the device cannot answer the walk below the
// leader's own term: the tick hands the reject back (RG_OUT_HOST_HINT) and ready(Storage &) answers from the host's log -- with
// the Inflights on the device too, where the group's sends wait for that answer. ready() without a Storage must refuse such a
// batch loudly instead of reporting it half-applied.
static void test_fast_log_rejection() {
    struct Row {
        std::vector<std::pair<u64, u64>> leader_log; // (term, index)
        u64 reject_hint_term, reject_hint_index, next_append_term, next_append_index;
    };
    const Row rows[] = {
        {{{1, 1}, {2, 2}, {2, 3}, {4, 4}, {4, 5}, {4, 6}, {4, 7}}, 3, 7, 2, 3},
        {{{1, 1}, {2, 2}, {2, 3}, {3, 4}, {4, 5}, {4, 6}, {4, 7}, {5, 8}}, 3, 8, 3, 4},
        {{{1, 1}, {1, 2}, {1, 3}, {4, 4}, {5, 5}}, 4, 4, 4, 4},
    };
    for (unsigned inflights : {0u, 4u}) {
        int i = 0;
        for (const Row &row : rows) {
            Config c;
            c.n_groups = 1;
            c.max_peers = 3;
            c.max_inflight_msgs = inflights;
            MultiRaft ld(c);
            LogFixture st;
            for (const auto &e : row.leader_log) st.append(0, e.first);
            const u64 last = row.leader_log.size();
            // become_leader's noop at last + 1 (raft.rs:1163-1194). (The reference's fixture campaigns from term 0, i.e. leads at
            // term 1 over entries of terms up to 5; a log's terms never decrease, so this host leads at term 6 -- the walk and its
            // answer do not depend on the leader's term.)
            st.append(0, 6);
            GroupSpec s;
            s.id = 1;
            s.term = 6;
            s.voters = {1, 2, 3};
            s.first_index_of_term = last + 1;
            s.last_index = last + 1;
            s.committed = 0;
            s.next_idx = last + 1;
            s.follower_state = ProgressState::Probe;
            ld.init_group(0, s);
            ld.bootstrap();
            Progress p = ld.progress(0, 2);
            p.paused = true; // the probe MsgAppend(index = next - 1 = last) went out
            ld.set_progress(0, 2, p);
            Message m;
            m.msg_type = MessageType::MsgAppendResponse;
            m.from = 2, m.to = 1, m.term = 6, m.index = last;
            m.reject = true, m.reject_hint = row.reject_hint_index, m.log_term = row.reject_hint_term;
            ld.step(0, m);
            bool refused = false;
            if (i == 0 && inflights == 0) { // (once: the batch is consumed by the flush, so this engine is done afterwards)
                try {
                    ld.ready();
                } catch (const Error &e) {
                    refused = e.kind == ErrorKind::State;
                }
                EXPECT(refused, "ready() without a Storage reports a batch that needs the host's log");
                i++;
                continue;
            }
            const std::vector<LightReady> rd = ld.ready(st);
            EXPECT(rd.size() == 1, "one group saw traffic");
            bool sa = false;
            for (u64 id : rd[0].send_append) sa = sa || id == 2;
            EXPECT(sa || inflights, "send_append(2) after the rejection (row %d)", i);
            const u64 next = ld.progress(0, 2).next_idx;
            EXPECT(next - 1 == row.next_append_index || (inflights && next - 1 >= row.next_append_index),
                   "row %d (inflights %u): next append index %llu, want %llu", i, inflights, (unsigned long long)(next - 1),
                   (unsigned long long)row.next_append_index);
            if (!inflights) {
                EXPECT(st.term(0, next - 1) == row.next_append_term, "row %d: next append term", i);
            } else {
                // the device made the send decision as well: the probe that follows the rejection starts at next_append_index
                bool found = false;
                for (const SendItem &it : rd[0].messages)
                    if (it.to == 2) found = it.prev_index == row.next_append_index;
                EXPECT(found, "row %d: a MsgAppend to 2 with index = %llu", i, (unsigned long long)row.next_append_index);
            }
            i++;
        }
    }
}

int main() {
    try {
        Config probe;
        MultiRaft m(probe);
    } catch (const Error &e) {
        if (e.kind == ErrorKind::NoDevice) {
            std::printf("no gfx950 device: %s (there is no CPU fallback)\n", e.what());
            return 2;
        }
        throw;
    }
    test_leader_append_response();
    test_leader_acknowledge_commit();
    test_msg_app_flow_control_full();
    test_raw_node_step();
    test_leader_start_replication();
    test_progress_flow_control();
    test_recv_msg_unreachable();
    test_snapshot_failure_and_succeed();
    test_bcast_beat();
    test_fast_log_rejection();
    std::printf("CPP_REFERENCE_TESTS_OK\n");
    return 0;
}
