// cpp_message_builder.cpp -- the host half of the send path, on the CPU: from the send stage's work items to the reference's
// Messages and their bytes (include/raftgroups.hpp: Storage, Entry, limit_size, build_messages, Message::write_to_bytes).
// No device is touched: everything here is the host code around the engine, checked against reference tests restated
// with their own rows. (The same builder behind a MultiRaft on the GPU: examples/cpp_reference_tests.cpp.)
//   g++ -std=c++17 -Iinclude examples/cpp_message_builder.cpp -o cpp_message_builder -Lraft_rs_amd -lraftgroups
#include <cstdio>
#include <cstdlib>

#include "log_fixture.hpp"

using namespace raftgroups;

#define EXPECT(cond, ...)                                        \
    do {                                                         \
        if (!(cond)) {                                           \
            std::fprintf(stderr, "%s:%d: ", __FILE__, __LINE__); \
            std::fprintf(stderr, __VA_ARGS__);                   \
            std::fprintf(stderr, "\n");                          \
            std::exit(1);                                        \
        }                                                        \
    } while (0)

static Entry new_entry(u64 index, u64 term) {
    Entry e;
    e.index = index;
    e.term = term;
    return e;
}
static bool same(const std::vector<Entry> &a, const std::vector<std::pair<u64, u64>> &w) {
    if (a.size() != w.size()) return false;
    for (std::size_t i = 0; i < a.size(); i++)
        if (a[i].index != w[i].first || a[i].term != w[i].second) return false;
    return true;
}

// src/storage.rs:508-569 test_storage_entries: Storage::entries under max_size (util::limit_size over compute_size())
static void test_storage_entries() {
    const u64 s4 = new_entry(4, 4).compute_size(), s5 = new_entry(5, 5).compute_size(), s6 = new_entry(6, 6).compute_size();
    EXPECT(s4 == 4 && s5 == 4 && s6 == 4, "Entry{term, index} below 128 is two 2-byte varint fields");
    struct Row { u64 lo, hi, maxsize; bool compacted; std::vector<std::pair<u64, u64>> w; };
    const Row tests[] = {
        {2, 6, NO_LIMIT, true, {}},
        {3, 4, NO_LIMIT, false, {{3, 3}}},
        {4, 5, NO_LIMIT, false, {{4, 4}}},
        {4, 6, NO_LIMIT, false, {{4, 4}, {5, 5}}},
        {4, 7, NO_LIMIT, false, {{4, 4}, {5, 5}, {6, 6}}},
        {4, 7, 0, false, {{4, 4}}}, // even if maxsize is zero, the first entry should be returned
        {4, 7, s4 + s5, false, {{4, 4}, {5, 5}}}, // limit to 2
        {4, 7, s4 + s5 + s6 / 2, false, {{4, 4}, {5, 5}}},
        {4, 7, s4 + s5 + s6 - 1, false, {{4, 4}, {5, 5}}},
        {4, 7, s4 + s5 + s6, false, {{4, 4}, {5, 5}, {6, 6}}}, // all
    };
    int i = 0;
    for (const Row &t : tests) {
        LogFixture st;
        st.log(0).snap_index = 2; // entries[0] = (3, 3): first_index = 3
        for (u64 k = 3; k <= 6; k++) st.append(0, k);
        bool compacted = false;
        std::vector<Entry> e;
        try {
            e = st.entries(0, t.lo, t.hi, t.maxsize);
        } catch (const StorageError &err) {
            compacted = err.kind == StorageErrorKind::Compacted;
        }
        EXPECT(compacted == t.compacted && same(e, t.w), "#%d: entries(%llu, %llu, %llu)", i, (unsigned long long)t.lo,
               (unsigned long long)t.hi, (unsigned long long)t.maxsize);
        i++;
    }
}

// src/raft_log.rs:1140-1260 test_slice, the rows with a limit: entries of (index, term) = (i, i) for i in 101..199
static void test_slice_limits() {
    const u64 offset = 100, num = 100, half = offset + num / 2;
    const u64 halfe_size = new_entry(half, half).compute_size();
    LogFixture st;
    st.log(0).snap_index = offset;
    for (u64 i = 1; i < num; i++) st.append(0, offset + i);
    struct Row { u64 from, to, limit; std::vector<std::pair<u64, u64>> w; };
    const Row tests[] = {
        {half - 1, half + 1, NO_LIMIT, {{half - 1, half - 1}, {half, half}}},
        {half - 1, half + 1, 0, {{half - 1, half - 1}}},
        {half - 1, half + 1, halfe_size + 1, {{half - 1, half - 1}}},
        {half - 2, half + 1, halfe_size + 1, {{half - 2, half - 2}}},
        {half - 1, half + 1, halfe_size * 2, {{half - 1, half - 1}, {half, half}}},
        {half - 1, half + 2, halfe_size * 3, {{half - 1, half - 1}, {half, half}, {half + 1, half + 1}}},
        {half, half + 2, halfe_size, {{half, half}}},
        {half, half + 2, halfe_size * 2, {{half, half}, {half + 1, half + 1}}},
    };
    int i = 0;
    for (const Row &t : tests) {
        EXPECT(same(st.entries(0, t.from, t.to, t.limit), t.w), "#%d: slice(%llu, %llu, %llu)", i, (unsigned long long)t.from,
               (unsigned long long)t.to, (unsigned long long)t.limit);
        i++;
    }
}

// The decoder of the input side reads a built message back (rg_decode_message)
static rg_decoded_message parse(const std::string &bytes) {
    rg_decoded_message d;
    check(rg_decode_message(reinterpret_cast<const std::uint8_t *>(bytes.data()), bytes.size(), &d));
    return d;
}

// harness/tests/integration_cases/test_raft_paper.rs:425-457 test_leader_start_replication, the message half: after the
// proposal the send stage reports, per follower, entries (li, li + 1]; the Messages built from that are the test's
// expect_msgs: MsgAppend { from 1, to, term 1, index li, log_term 1, commit li, entries [(1, li + 1, SOME_DATA)] }
static void test_leader_start_replication_messages() {
    LogFixture st;
    st.append(0, 1);              // the leader's empty entry (become_leader), committed by commit_noop_entry
    st.append(0, 1, "somedata");  // the proposal
    const u64 li = 1;
    SendContext c;
    c.group = 0, c.id = 1, c.term = 1, c.committed = li;
    for (u64 to : {2ULL, 3ULL}) {
        SendItem s;
        s.to = to, s.prev_index = li, s.last_index = li + 1, s.n_msgs = 1;
        const std::vector<Message> ms = build_messages(c, s, st);
        EXPECT(ms.size() == 1, "one MsgAppend per follower");
        const Message &m = ms[0];
        EXPECT(m.msg_type == MessageType::MsgAppend && m.from == 1 && m.to == to && m.term == 1 && m.index == li && m.log_term == 1 &&
                   m.commit == li && m.entries.size() == 1 && m.entries[0].term == 1 && m.entries[0].index == li + 1 &&
                   m.entries[0].data == "somedata" && !m.reject && !m.has_snapshot,
               "the MsgAppend of test_leader_start_replication");
        const std::string bytes = m.write_to_bytes();
        EXPECT(bytes.size() == m.compute_size(), "compute_size");
        const rg_decoded_message d = parse(bytes);
        EXPECT(d.msg_type == 3 && d.to == to && d.from == 1 && d.term == 1 && d.log_term == 1 && d.index == li && d.commit == li &&
                   d.n_entries == 1 && !d.has_snapshot,
               "the bytes read back");
        // msg_type 3, to, from 1, term 1, log_term 1, index 1, entries [ {term 1, index 2, data "somedata"} ], commit 1
        const unsigned char want[] = {0x08, 0x03, 0x10, (unsigned char)to, 0x18, 0x01, 0x20, 0x01, 0x28, 0x01, 0x30, 0x01, 0x3a, 0x0e,
                                      0x10, 0x01, 0x18, 0x02, 0x22, 0x08, 's', 'o', 'm', 'e', 'd', 'a', 't', 'a', 0x40, 0x01};
        EXPECT(bytes == std::string(reinterpret_cast<const char *>(want), sizeof want), "canonical proto3 bytes");
    }
}

// Raft::send_append with nothing to send (allow_empty, raft.rs:773-819): prev_index == last_index is one MsgAppend without entries
static void test_empty_append() {
    LogFixture st;
    for (int i = 0; i < 5; i++) st.append(0, 2);
    SendContext c;
    c.id = 1, c.term = 2, c.committed = 4;
    SendItem s;
    s.to = 3, s.prev_index = 5, s.last_index = 5, s.n_msgs = 1;
    const std::vector<Message> ms = build_messages(c, s, st);
    EXPECT(ms.size() == 1 && ms[0].entries.empty() && ms[0].index == 5 && ms[0].log_term == 2 && ms[0].commit == 4, "empty MsgAppend");
}

// What the device's RG_SEND_BYTES stage counts (`while maybe_send_append`: entries(next, max_size) under util::limit_size,
// next = last + 1, raft.rs:1761) restated as a loop, against the builder: same number of messages, same cuts.
static unsigned count_like_the_stage(const std::vector<Entry> &log, u64 snap, u64 prev, u64 last, u64 max_size, std::vector<u64> *cuts) {
    unsigned n = 0;
    u64 next = prev + 1;
    while (next <= last) {
        std::vector<rg_entry> c;
        for (u64 i = next; i <= last; i++) c.push_back(log[i - snap - 1].c_entry());
        const u64 keep = rg_limit_size(c.data(), c.size(), max_size);
        next += keep;
        cuts->push_back(next - 1);
        n++;
    }
    return n;
}
static void test_byte_limited_split() {
    unsigned long long x = 88172645463325252ULL;
    auto rnd = [&]() {
        x ^= x << 13, x ^= x >> 7, x ^= x << 17;
        return x;
    };
    unsigned n_multi = 0;
    for (int round = 0; round < 300; round++) {
        LogFixture st;
        st.log(7).snap_index = rnd() % 50;
        st.log(7).snap_term = 1;
        const u64 snap = st.log(7).snap_index, n = 1 + rnd() % 40;
        for (u64 i = 0; i < n; i++) {
            if (rnd() % 9 == 0) { // Entry::default()-sized payloads are rare in a log, a zero-size ENTRY is impossible (index > 0)
                st.append(7, 1 + i / 10);
            } else {
                st.append(7, 1 + i / 10, std::string((std::size_t)(rnd() % 120), 'x'));
            }
        }
        const u64 last = snap + n, prev = snap + rnd() % n;
        const u64 max_size = rnd() % 4 == 0 ? 0 : rnd() % 400;
        std::vector<u64> cuts;
        SendItem s;
        s.to = 2, s.prev_index = prev, s.last_index = last;
        s.n_msgs = count_like_the_stage(st.log(7).entries, snap, prev, last, max_size, &cuts);
        SendContext c;
        c.group = 7, c.id = 1, c.term = 9, c.committed = prev, c.max_size_per_msg = max_size;
        const std::vector<Message> ms = build_messages(c, s, st);
        EXPECT(ms.size() == s.n_msgs, "round %d: %zu messages, the stage counted %u", round, ms.size(), s.n_msgs);
        u64 at = prev;
        for (std::size_t k = 0; k < ms.size(); k++) {
            EXPECT(ms[k].index == at && ms[k].log_term == st.term(7, at), "round %d: message %zu starts at %llu", round, k, (unsigned long long)at);
            EXPECT(!ms[k].entries.empty() && ms[k].entries.front().index == at + 1 && ms[k].entries.back().index == cuts[k], "round %d: cut %zu", round, k);
            u64 total = 0;
            for (const Entry &e : ms[k].entries) total += e.compute_size();
            EXPECT(ms[k].entries.size() == 1 || total <= max_size, "round %d: message %zu holds %llu bytes of entries, max %llu", round, k,
                   (unsigned long long)total, (unsigned long long)max_size);
            EXPECT(parse(ms[k].write_to_bytes()).n_entries == ms[k].entries.size(), "round %d: bytes of message %zu", round, k);
            at = cuts[k];
        }
        n_multi += ms.size() > 1;
        // a storage that cuts differently from what the device counted is refused, not papered over
        if (s.n_msgs > 1) {
            SendItem bad = s;
            bad.n_msgs = s.n_msgs - 1;
            bool threw = false;
            try {
                build_messages(c, bad, st);
            } catch (const Error &e) {
                threw = e.kind == ErrorKind::State;
            }
            EXPECT(threw, "round %d: one message too few must raise Error{State}", round);
            bad.n_msgs = s.n_msgs + 1;
            threw = false;
            try {
                build_messages(c, bad, st);
            } catch (const Error &e) {
                threw = e.kind == ErrorKind::State;
            }
            EXPECT(threw, "round %d: one message too many must raise Error{State}", round);
        }
    }
    EXPECT(n_multi > 100, "the byte limit must have split most sends (%u)", n_multi);
}

// The equal-sized stand-in (engines without the size table): max_entries_per_msg entries per message
static void test_entry_count_split() {
    LogFixture st;
    for (int i = 0; i < 11; i++) st.append(0, 3, "payload");
    SendContext c;
    c.id = 1, c.term = 3, c.committed = 2, c.max_entries_per_msg = 4;
    SendItem s;
    s.to = 2, s.prev_index = 2, s.last_index = 11, s.n_msgs = 3; // 9 entries: 4 + 4 + 1
    const std::vector<Message> ms = build_messages(c, s, st);
    EXPECT(ms.size() == 3 && ms[0].entries.size() == 4 && ms[1].entries.size() == 4 && ms[2].entries.size() == 1, "4 + 4 + 1");
    EXPECT(ms[0].index == 2 && ms[1].index == 6 && ms[2].index == 10 && ms[2].entries[0].index == 11, "each message starts where the last ended");
}

// prepare_send_snapshot (raft.rs:664-712) and harness/tests/integration_cases/test_raft.rs:4903-4965
// (test_request_snapshot_unavailable): nothing goes out while the storage answers SnapshotTemporarilyUnavailable
static void test_snapshot_items() {
    LogFixture st;
    st.log(0).snap_index = 11;
    st.log(0).snap_term = 11;
    st.append(0, 11);
    SendContext c;
    c.id = 1, c.term = 11, c.committed = 12;
    SendItem s;
    s.to = 2, s.snapshot = true, s.last_index = 0, s.n_msgs = 1; // "any snapshot"
    st.log(0).snap_unavailable_once = true;
    u64 sindex = 99;
    EXPECT(build_messages(c, s, st, &sindex).empty() && sindex == 0, "temporarily unavailable: no message, no become_snapshot");
    std::vector<Message> ms = build_messages(c, s, st, &sindex);
    EXPECT(ms.size() == 1 && ms[0].msg_type == MessageType::MsgSnapshot && ms[0].has_snapshot && ms[0].to == 2 && ms[0].from == 1 &&
               ms[0].term == 11 && sindex == 11,
           "MsgSnapshot with the storage's snapshot (index 11)");
    const rg_decoded_message d = parse(ms[0].write_to_bytes());
    EXPECT(d.msg_type == 7 && d.has_snapshot && d.n_entries == 0, "MsgSnapshot bytes");
    s.last_index = 14; // a follower's request_snapshot: the snapshot must not be older (storage.rs:104)
    ms = build_messages(c, s, st, &sindex);
    EXPECT(ms.size() == 1 && sindex == 14, "requested index 14");
    LogFixture empty;
    SendItem none;
    none.to = 2, none.snapshot = true, none.n_msgs = 1;
    bool threw = false;
    try {
        build_messages(c, none, empty); // snapshot index 0: "need non-empty snapshot" (raft.rs:696-698 panics)
    } catch (const Error &e) {
        threw = e.kind == ErrorKind::State;
    }
    EXPECT(threw, "an empty snapshot is refused");
    SendItem host;
    host.to = 2, host.host = true, host.prev_index = 3, host.last_index = 9, host.n_msgs = 0;
    EXPECT(build_messages(c, host, st).empty(), "RG_SEND_HOST items are the host's own maybe_send_append");
}

// A log the host compacted under the device's feet is an error, not a silent gap
static void test_compacted_under_the_stage() {
    LogFixture st;
    st.log(0).snap_index = 20;
    st.log(0).snap_term = 2;
    for (int i = 0; i < 5; i++) st.append(0, 2);
    SendContext c;
    c.id = 1, c.term = 2;
    SendItem s;
    s.to = 2, s.prev_index = 10, s.last_index = 25, s.n_msgs = 1;
    bool threw = false;
    try {
        build_messages(c, s, st);
    } catch (const StorageError &e) {
        threw = e.kind == StorageErrorKind::Compacted;
    }
    EXPECT(threw, "StorageError{Compacted} reaches the caller");
}

int main() {
    test_storage_entries();
    test_slice_limits();
    test_leader_start_replication_messages();
    test_empty_append();
    test_byte_limited_split();
    test_entry_count_split();
    test_snapshot_items();
    test_compacted_under_the_stage();
    std::printf("CPP_MESSAGE_BUILDER_OK\n");
    return 0;
}
