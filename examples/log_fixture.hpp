// log_fixture.hpp -- TEST FIXTURE for the example programs: the logs of a few raft groups in plain vectors, behind
// raftgroups::Storage. It behaves the way the reference's tests expect a Storage to behave (the contract of
// src/storage.rs:65-106 as MemStorage honours it, :387-453: Compacted below first_index, Unavailable above last_index,
// the entry before first_index still has a term, a snapshot request can be temporarily unavailable once). Not part of the
// library: an application puts its own log store behind raftgroups::Storage.
#pragma once

#include <map>

#include "raftgroups.hpp"

struct LogFixture : raftgroups::Storage {
    using u64 = raftgroups::u64;
    struct Log {
        u64 snap_index = 0, snap_term = 0;      // the entry before first_index (SnapshotMetadata)
        std::vector<raftgroups::Entry> entries; // indices snap_index + 1 ..
        bool snap_unavailable_once = false;     // MemStorageCore::trigger_snap_unavailable
    };
    std::map<u64, Log> logs;

    Log &log(u64 group) { return logs[group]; }
    void append(u64 group, u64 term, const std::string &data = std::string()) {
        Log &l = logs[group];
        raftgroups::Entry e;
        e.term = term;
        e.index = last_index(group) + 1;
        e.data = data;
        l.entries.push_back(e);
    }

    u64 first_index(u64 group) override { return logs[group].snap_index + 1; }
    u64 last_index(u64 group) override {
        const Log &l = logs[group];
        return l.snap_index + l.entries.size();
    }
    std::vector<raftgroups::Entry> entries(u64 group, u64 low, u64 high, u64 max_size) override {
        Log &l = logs[group];
        if (low < first_index(group)) throw raftgroups::StorageError(raftgroups::StorageErrorKind::Compacted, "log compacted");
        if (high > last_index(group) + 1) throw std::out_of_range("index out of bound"); // (the reference panics)
        std::vector<raftgroups::Entry> out(l.entries.begin() + (low - l.snap_index - 1), l.entries.begin() + (high - l.snap_index - 1));
        raftgroups::limit_size(out, max_size);
        return out;
    }
    u64 term(u64 group, u64 idx) override {
        Log &l = logs[group];
        if (idx == l.snap_index) return l.snap_term;
        if (idx < first_index(group)) throw raftgroups::StorageError(raftgroups::StorageErrorKind::Compacted, "log compacted");
        if (idx > last_index(group)) throw raftgroups::StorageError(raftgroups::StorageErrorKind::Unavailable, "log unavailable");
        return l.entries[idx - l.snap_index - 1].term;
    }
    raftgroups::Snapshot snapshot(u64 group, u64 request_index) override {
        Log &l = logs[group];
        if (l.snap_unavailable_once) {
            l.snap_unavailable_once = false;
            throw raftgroups::StorageError(raftgroups::StorageErrorKind::SnapshotTemporarilyUnavailable, "snapshot is temporarily unavailable");
        }
        raftgroups::Snapshot s;
        s.index = l.snap_index < request_index ? request_index : l.snap_index;
        s.term = l.snap_term;
        // eraftpb::Snapshot { metadata = 2: SnapshotMetadata { index = 2, term = 3 } } -- written through the library's own
        // encoder conventions by hand: two varint fields inside a length-delimited one
        std::string meta;
        auto put = [&](unsigned field, u64 v) {
            if (!v) return;
            meta.push_back((char)(field << 3));
            while (v >= 0x80) {
                meta.push_back((char)(v | 0x80));
                v >>= 7;
            }
            meta.push_back((char)v);
        };
        put(2, s.index);
        put(3, s.term);
        if (!meta.empty()) {
            s.bytes.push_back((char)(2 << 3 | 2));
            s.bytes.push_back((char)meta.size());
            s.bytes += meta;
        }
        return s;
    }
};
