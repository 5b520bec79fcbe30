// message_bench.cpp -- what the HOST half of the send path costs: send item -> Storage::entries / term -> Message ->
// write_to_bytes, one thread, no device (include/raftgroups.hpp: build_messages, rg_encode_message). The reference spends its
// CPU time exactly here (Storage::entries + protobuf Message construction, not the Progress arithmetic).
//   g++ -O2 -std=c++17 -Iinclude examples/message_bench.cpp -o message_bench -Lraft_rs_amd -lraftgroups
#include <chrono>
#include <cstdio>

#include "log_fixture.hpp"

using namespace raftgroups;

// a Storage over ONE shared log shape (every group has the same 64 entries): the bench measures the builder, not a map lookup
struct FlatLog : Storage {
    std::vector<Entry> log; // indices 1..n
    std::vector<Entry> entries(u64, u64 low, u64 high, u64 max_size) override {
        std::vector<Entry> out(log.begin() + (low - 1), log.begin() + (high - 1));
        limit_size(out, max_size);
        return out;
    }
    u64 term(u64, u64 idx) override { return idx ? log[idx - 1].term : 0; }
    u64 first_index(u64) override { return 1; }
    u64 last_index(u64) override { return log.size(); }
    Snapshot snapshot(u64, u64) override { throw StorageError(StorageErrorKind::SnapshotTemporarilyUnavailable, "none"); }
};

int main(int argc, char **argv) {
    const unsigned payload = argc > 1 ? (unsigned)atoi(argv[1]) : 100, per_msg = argc > 2 ? (unsigned)atoi(argv[2]) : 1;
    const u64 n_items = argc > 3 ? (u64)atoll(argv[3]) : 2000000;
    FlatLog st;
    for (u64 i = 1; i <= 64; i++) {
        Entry e;
        e.term = 3, e.index = i, e.data = std::string(payload, 'x');
        st.log.push_back(e);
    }
    SendContext c;
    c.id = 1, c.term = 3, c.committed = 40;
    u64 bytes = 0, msgs = 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (u64 i = 0; i < n_items; i++) {
        SendItem s;
        s.to = 2 + i % 4;
        s.prev_index = 20 + i % 8;
        s.last_index = s.prev_index + per_msg;
        s.n_msgs = 1;
        c.group = i;
        for (const Message &m : build_messages(c, s, st)) {
            bytes += m.write_to_bytes().size();
            msgs++;
        }
    }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("%llu MsgAppend x %u entries x %u B payload: %.2f M messages/s, %.1f MB/s of wire bytes, one thread (%.0f ns per message)\n",
                (unsigned long long)msgs, per_msg, payload, msgs / dt / 1e6, bytes / dt / 1e6, dt / msgs * 1e9);
    return 0;
}
