// wire_asan.cpp -- TEST INFRASTRUCTURE: raft_rs_amd/csrc/rg_wire.h (the protobuf decoder behind rg_decode_message /
// rg_step_bytes) compiled with -fsanitize=address,undefined and driven over byte strings that sit in heap buffers of
// EXACTLY their length (so a read one byte past the end is an ASan error): every line of stdin is one hex string, followed
// by `n_mut` seeded mutations of it (byte flips, truncations, insertions). Prints the number of accepted / refused inputs.
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

int main(int argc, char **argv) {
    const int n_mut = argc > 1 ? atoi(argv[1]) : 200;
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
    std::printf("WIRE_ASAN_OK accepted %llu refused %llu\n", n_ok, n_bad);
    return 0;
}
