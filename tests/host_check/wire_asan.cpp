// wire_asan.cpp -- TEST INFRASTRUCTURE: raft_rs_amd/csrc/rg_wire.h (the protobuf decoder behind rg_decode_message /
// rg_step_bytes) compiled with -fsanitize=address,undefined and driven over byte strings that sit in heap buffers of
// EXACTLY their length (so a read one byte past the end is an ASan error): every line of stdin is one hex string, followed
// by `n_mut` seeded mutations of it (byte flips, truncations, insertions). Prints the number of accepted / refused inputs.
// Then the ENCODER (rg_wire_message_size / rg_wire_encode / rg_wire_limit_size, behind rg_encode_message): `n_enc` seeded
// random messages whose payloads sit in exact-size heap buffers, written into an output buffer of exactly the computed
// size (one byte more or fewer written is an ASan error or a mismatch) and read back by the decoder.
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all wire_asan.cpp -o wire_asan
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../raft_rs_amd/csrc/rg_wire.h"

static unsigned long long rng_state = 0x9E3779B97F4A7C15ULL;
static unsigned long long rnd() {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}

static unsigned long long n_ok = 0, n_bad = 0;
static void feed(const std::vector<uint8_t> &v) {
    uint8_t *buf = (uint8_t *)malloc(v.size() ? v.size() : 1); // exact size: ASan guards the byte behind it
    if (!v.empty()) memcpy(buf, v.data(), v.size());
    rg_decoded_message m;
    rg_wire_u64 bad = 0;
    const bool ok = rg_wire_decode(buf, v.size(), &m, &bad);
    if (bad > v.size()) {
        std::fprintf(stderr, "walk stopped %llu bytes in, input has %zu\n", (unsigned long long)bad, v.size());
        std::exit(3);
    }
    (ok ? n_ok : n_bad)++;
    free(buf);
}

static uint8_t *heap_bytes(size_t n) { // exact size; NULL for 0 bytes (the API allows it)
    if (!n) return nullptr;
    uint8_t *b = (uint8_t *)malloc(n);
    for (size_t i = 0; i < n; i++) b[i] = (uint8_t)rnd();
    return b;
}
static rg_wire_u64 rnd_u64() {
    switch (rnd() % 5) {
    case 0: return 0;
    case 1: return rnd() % 200;
    case 2: return rnd() & 0xffffffffULL;
    case 3: return rnd();
    default: return ~(rg_wire_u64)0;
    }
}
static void encode_round(int n_enc) {
    for (int it = 0; it < n_enc; it++) {
        rg_message m;
        memset(&m, 0, sizeof m);
        m.msg_type = (uint32_t)(rnd() % 19);
        m.reject = (uint32_t)(rnd() % 3 == 0);
        m.to = rnd_u64(), m.from = rnd_u64(), m.term = rnd_u64(), m.log_term = rnd_u64(), m.index = rnd_u64();
        m.commit = rnd_u64(), m.commit_term = rnd_u64(), m.reject_hint = rnd_u64(), m.request_snapshot = rnd_u64();
        m.priority = rnd_u64();
        const size_t ne = rnd() % 2 ? rnd() % 6 : 0;
        rg_entry *ents = ne ? (rg_entry *)malloc(ne * sizeof(rg_entry)) : nullptr;
        for (size_t i = 0; i < ne; i++) {
            rg_entry &e = ents[i];
            memset(&e, 0, sizeof e);
            if (rnd() % 8 == 0) continue; // Entry::default()
            e.entry_type = (uint32_t)(rnd() % 3);
            e.sync_log = (uint32_t)(rnd() % 4 == 0) * 7u; // any non-zero value is `true`
            e.term = rnd_u64(), e.index = rnd_u64();
            e.data_len = rnd() % 3 ? rnd() % 200 : 0;
            e.data = heap_bytes(e.data_len);
            e.context_len = rnd() % 5 ? 0 : rnd() % 20;
            e.context = heap_bytes(e.context_len);
        }
        m.entries = ents, m.n_entries = ne;
        static uint8_t present_but_empty;
        if (rnd() % 6 == 0) {
            m.snapshot_len = rnd() % 2 ? rnd() % 40 : 0;
            m.snapshot = m.snapshot_len ? heap_bytes(m.snapshot_len) : &present_but_empty;
            // (random bytes are not a Snapshot: the read-back below skips messages that carry one)
        }
        m.context_len = rnd() % 4 ? 0 : rnd() % 24;
        m.context = heap_bytes(m.context_len);
        rg_wire_u64 len = 0;
        if (!rg_wire_message_size(&m, &len)) std::exit(4);
        uint8_t *out = (uint8_t *)malloc(len ? len : 1);
        uint8_t *end = rg_wire_encode(&m, out);
        if ((rg_wire_u64)(end - out) != len) {
            std::fprintf(stderr, "encoder wrote %lld bytes, rg_wire_message_size said %llu\n", (long long)(end - out), (unsigned long long)len);
            std::exit(5);
        }
        const rg_wire_u64 keep = rg_wire_limit_size(ents, ne, rnd() % 600);
        if (keep > ne || (ne && !keep)) std::exit(6);
        if (!m.snapshot_len) {
            rg_decoded_message d;
            rg_wire_u64 bad = 0;
            if (!rg_wire_decode(out, len, &d, &bad) || d.msg_type != m.msg_type || d.reject != m.reject || d.to != m.to ||
                d.from != m.from || d.term != m.term || d.log_term != m.log_term || d.index != m.index || d.commit != m.commit ||
                d.commit_term != m.commit_term || d.reject_hint != m.reject_hint || d.request_snapshot != m.request_snapshot ||
                d.priority != m.priority || d.n_entries != ne || d.has_snapshot != (m.snapshot ? 1u : 0u) ||
                d.context_len != m.context_len) {
                std::fprintf(stderr, "round trip mismatch in message %d (stopped at byte %llu of %llu)\n", it, (unsigned long long)bad,
                             (unsigned long long)len);
                std::exit(7);
            }
        }
        free(out);
        for (size_t i = 0; i < ne; i++) {
            free((void *)ents[i].data);
            free((void *)ents[i].context);
        }
        free(ents);
        if (m.snapshot_len) free((void *)m.snapshot);
        free((void *)m.context);
    }
}

int main(int argc, char **argv) {
    const int n_mut = argc > 1 ? atoi(argv[1]) : 200;
    const int n_enc = argc > 2 ? atoi(argv[2]) : 20000;
    char line[1 << 16];
    while (std::fgets(line, sizeof line, stdin)) {
        std::vector<uint8_t> base;
        for (const char *c = line; c[0] && c[1] && c[0] != '\n'; c += 2) {
            unsigned x;
            if (std::sscanf(c, "%2x", &x) != 1) break;
            base.push_back((uint8_t)x);
        }
        feed(base);
        for (int k = 0; k < n_mut; k++) {
            std::vector<uint8_t> v = base;
            switch (rnd() % 4) {
            case 0:
                for (int j = 0; j < 3 && !v.empty(); j++) v[rnd() % v.size()] = (uint8_t)rnd();
                break;
            case 1:
                if (!v.empty()) v.resize(rnd() % v.size());
                break;
            case 2: {
                const size_t at = rnd() % (v.size() + 1);
                for (int j = 0; j < 1 + (int)(rnd() % 4); j++) v.insert(v.begin() + at, (uint8_t)rnd());
                break;
            }
            default:
                v.resize(rnd() % 24);
                for (auto &b : v) b = (uint8_t)rnd();
            }
            feed(v);
        }
    }
    encode_round(n_enc);
    std::printf("WIRE_ASAN_OK accepted %llu refused %llu encoded %d\n", n_ok, n_bad, n_enc);
    return 0;
}
