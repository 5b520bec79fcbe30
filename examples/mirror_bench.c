/* mirror_bench.c -- throughput of the message-at-a-time host mirror (rg_step / rg_local_* / rg_flush) from plain C.
 * One "round" = every group appends and persists 2 entries, every follower acks them, then one rg_flush.
 *   gcc -O2 -std=c99 -Iinclude examples/mirror_bench.c -o mirror_bench -Lraft_rs_amd -lraftgroups -Wl,-rpath,$PWD/raft_rs_amd */
#define _POSIX_C_SOURCE 199309L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "raftgroups.h"

#define CHECK(call)                                                          \
    do {                                                                     \
        int rc__ = (call);                                                   \
        if (rc__ != RG_OK) {                                                 \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc__, rg_last_error()); \
            return 2;                                                        \
        }                                                                    \
    } while (0)

static double now(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}

int main(int argc, char **argv) {
    const uint64_t G = argc > 1 ? strtoull(argv[1], 0, 10) : 200000;
    const double touched = argc > 2 ? atof(argv[2]) : 1.0; /* fraction of groups with traffic per round */
    enum { P = 5, TERM = 5, ROUNDS = 5 };
    rg_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.n_groups = G;
    cfg.n_slots = P;
    rg_engine *h;
    CHECK(rg_create(&cfg, &h));
    rg_workload w = {0x5EED5EEDull, RG_WL_MAJORITY, 0};
    CHECK(rg_workload_init(h, &w, 0));
    uint64_t ids[P] = {11, 12, 13, 14, 15};
    for (uint64_t g = 0; g < G; g++) CHECK(rg_set_peers(h, g, ids, P, TERM));
    uint64_t *hi = malloc(G * 8), *commit = malloc(G * 8), *groups = malloc(G * 8);
    uint32_t *out = malloc(G * 4);
    CHECK(rg_read_column(h, RG_COL_TERM_HI, hi, G * 8));
    const uint64_t step = touched >= 1.0 ? 1 : (uint64_t)(1.0 / touched);
    for (int r = 0; r < ROUNDS; r++) {
        uint64_t calls = 0, n = 0;
        const double t0 = now();
        for (uint64_t g = (uint64_t)r % step; g < G; g += step) {
            hi[g] += 2;
            CHECK(rg_local_append(h, g, hi[g]));
            CHECK(rg_local_persisted(h, g, hi[g]));
            for (int p = 1; p < P; p++) {
                rg_append_response m;
                memset(&m, 0, sizeof m);
                m.from = ids[p];
                m.term = TERM;
                m.index = hi[g];
                m.commit = hi[g] - 2;
                CHECK(rg_mark_sent(h, g, ids[p]));
                CHECK(rg_step(h, g, &m));
            }
            calls += 2 + 2 * (P - 1);
        }
        const double t1 = now();
        CHECK(rg_flush(h));
        CHECK(rg_ingested_results(h, groups, commit, out, G, &n));
        const double t2 = now();
        uint64_t changed = 0;
        for (uint64_t i = 0; i < n; i++) changed += out[i] & RG_OUT_CHANGED;
        printf("round %d: %llu groups, %llu mirror calls in %.1f ms (%.1f M calls/s); flush+results %.2f ms; %llu commits moved\n",
               r, (unsigned long long)n, (unsigned long long)calls, (t1 - t0) * 1e3, calls / (t1 - t0) / 1e6,
               (t2 - t1) * 1e3, (unsigned long long)changed);
    }
    rg_destroy(h);
    return 0;
}
