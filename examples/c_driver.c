/* c_driver.c -- the C ABI used directly from plain C (no Python, no torch): a 3-peer group commits an
 * entry once a majority has acknowledged it. Build + run (on an MI355X):
 *   gcc -std=c99 -Iinclude examples/c_driver.c -o examples/c_driver -Lraft_rs_amd -lraftgroups \
 *       -Wl,-rpath,$PWD/raft_rs_amd && ./examples/c_driver
 * Mirrors harness/tests/integration_cases/test_raft_paper.rs:499-534 (test_leader_acknowledge_commit). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "raftgroups.h"

#define CHECK(call)                                                                  \
    do {                                                                             \
        int rc__ = (call);                                                           \
        if (rc__ != RG_OK) {                                                         \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc__, rg_last_error());         \
            return 2;                                                                \
        }                                                                            \
    } while (0)

int main(void) {
    enum { G = 4, P = 3 };
    rg_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.n_groups = G;
    cfg.n_slots = P;
    rg_engine *h = NULL;
    CHECK(rg_create(&cfg, &h));
    const uint64_t stride = rg_stride(h);

    /* every group: leader = slot 0 with entries 1..2 of its term persisted; followers at match 1; commit 1 */
    uint64_t *col = calloc(P * stride, 8);
    uint64_t per_group[G];
    uint32_t cfgw[G];
    uint8_t pflags[G][8];
    memset(pflags, 0, sizeof pflags);
    for (int g = 0; g < G; g++) {
        for (int p = 0; p < P; p++) {
            col[p * stride + g] = p == 0 ? 2 : 1;
            pflags[g][p] = RG_STATE_REPLICATE;
        }
        cfgw[g] = RG_CFG_MAKE(0x7, 0, 0, 0, 0, 0x7);
    }
    CHECK(rg_load_column(h, RG_COL_MATCH, col, rg_column_bytes(h, RG_COL_MATCH)));
    for (int g = 0; g < G; g++)
        for (int p = 0; p < P; p++) col[p * stride + g] = (p == 0 ? 2 : 1) + 1;
    CHECK(rg_load_column(h, RG_COL_NEXT, col, rg_column_bytes(h, RG_COL_NEXT)));
    CHECK(rg_load_column(h, RG_COL_PFLAGS, pflags, rg_column_bytes(h, RG_COL_PFLAGS)));
    CHECK(rg_load_column(h, RG_COL_CFG, cfgw, rg_column_bytes(h, RG_COL_CFG)));
    for (int g = 0; g < G; g++) per_group[g] = 1;
    CHECK(rg_load_column(h, RG_COL_COMMIT, per_group, rg_column_bytes(h, RG_COL_COMMIT)));
    CHECK(rg_load_column(h, RG_COL_TERM_LO, per_group, rg_column_bytes(h, RG_COL_TERM_LO)));
    for (int g = 0; g < G; g++) per_group[g] = 2;
    CHECK(rg_load_column(h, RG_COL_TERM_HI, per_group, rg_column_bytes(h, RG_COL_TERM_HI)));

    /* group g receives acks for index 2 from g followers (0, 1 or 2 of them) */
    rg_wire_msg recs[G * P];
    uint64_t n = 0, dup = 0, touched = 0;
    for (int g = 0; g < G; g++) {
        for (int k = 0; k < (g < 3 ? g : 2); k++) {
            memset(&recs[n], 0, sizeof recs[n]);
            recs[n].group = (uint64_t)g;
            recs[n].slot = 1u + (uint32_t)k;
            recs[n].index = 2;
            recs[n].commit = 1;
            recs[n].flags = RG_MF_VALID;
            n++;
        }
    }
    CHECK(rg_ingest(h, recs, n, &dup));
    CHECK(rg_tick_ingested(h, &touched));
    uint64_t commit[G];
    uint32_t out[G];
    CHECK(rg_results(h, commit, out));
    int ok = dup == 0 && touched == 3;
    for (int g = 0; g < G; g++) {
        const int acks = g < 3 ? g : 2;
        const uint64_t want = acks >= 1 ? 2 : 1; /* leader + one follower = majority of 3 */
        printf("group %d: %d acks -> commit %llu (want %llu) out %#x\n", g, acks, (unsigned long long)commit[g],
               (unsigned long long)want, out[g]);
        ok = ok && commit[g] == want && ((out[g] & RG_OUT_CHANGED) != 0) == (want == 2);
    }
    /* ---- the resident small-batch path: one workgroup stays on the device and answers small flushes out of pinned
     *      memory (no launch, no synchronisation per flush) -- what a latency-bound RawNode::step loop turns on ---- */
    {
        uint64_t served = 0, launches = 0;
        CHECK(rg_mailbox_start(h, 0));
        for (int round = 0; round < 3 && ok; round++) { /* follower 2 of group 3 acknowledges index 2, thrice (stale twice) */
            memset(&recs[0], 0, sizeof recs[0]);
            recs[0].group = 3;
            recs[0].slot = 2;
            recs[0].index = 2;
            recs[0].commit = 2;
            recs[0].flags = RG_MF_VALID;
            CHECK(rg_ingest_tick(h, recs, 1, &touched, &dup));
            ok = ok && touched == 1 && dup == 0;
        }
        CHECK(rg_mailbox_stats(h, &served, &launches));
        CHECK(rg_mailbox_stop(h));
        printf("mailbox: %llu flushes served by the resident workgroup (%llu launch)\n", (unsigned long long)served,
               (unsigned long long)launches);
        ok = ok && served == 3 && launches == 1;
    }
    rg_destroy(h);

    /* ---- the optional send stage: Inflights (window of 2 messages) on the device, decisions by the engine ---- */
    cfg.max_inflight = 2;
    CHECK(rg_create(&cfg, &h));
    for (int g = 0; g < G; g++)
        for (int p = 0; p < P; p++) col[p * stride + g] = 2;
    CHECK(rg_load_column(h, RG_COL_MATCH, col, rg_column_bytes(h, RG_COL_MATCH)));
    for (int g = 0; g < G; g++)
        for (int p = 0; p < P; p++) col[p * stride + g] = 3;
    CHECK(rg_load_column(h, RG_COL_NEXT, col, rg_column_bytes(h, RG_COL_NEXT)));
    CHECK(rg_load_column(h, RG_COL_PFLAGS, pflags, rg_column_bytes(h, RG_COL_PFLAGS)));
    CHECK(rg_load_column(h, RG_COL_CFG, cfgw, rg_column_bytes(h, RG_COL_CFG)));
    for (int g = 0; g < G; g++) per_group[g] = 2;
    CHECK(rg_load_column(h, RG_COL_COMMIT, per_group, rg_column_bytes(h, RG_COL_COMMIT)));
    CHECK(rg_load_column(h, RG_COL_TERM_HI, per_group, rg_column_bytes(h, RG_COL_TERM_HI)));
    for (int g = 0; g < G; g++) per_group[g] = 1;
    CHECK(rg_load_column(h, RG_COL_TERM_LO, per_group, rg_column_bytes(h, RG_COL_TERM_LO)));
    rg_send_item items[3 * G * P];
    for (int round = 1; round <= 3 && ok; round++) { /* three proposals of one entry: the third finds the windows full */
        for (int g = 0; g < G; g++) {
            memset(&recs[g], 0, sizeof recs[g]);
            recs[g].group = (uint64_t)g;
            recs[g].slot = 0;                    /* the leader's own slot */
            recs[g].commit = 2u + (uint64_t)round; /* RG_MF_APPEND: new last_index */
            recs[g].flags = RG_MF_APPEND;
        }
        CHECK(rg_ingest_tick(h, recs, G, &touched, &dup));
        CHECK(rg_send_appends(h, 0, 0));
        CHECK(rg_send_items(h, items, sizeof items / sizeof items[0], &n));
        printf("proposal %d: %llu groups ticked, %llu MsgAppend work items\n", round, (unsigned long long)touched,
               (unsigned long long)n);
        ok = ok && touched == G && dup == 0 && n == (round <= 2 ? (uint64_t)G * (P - 1) : 0);
        for (uint64_t i = 0; i < n && ok; i++)
            ok = items[i].kind == RG_SEND_APPEND && items[i].n_msgs == 1 && items[i].prev_index == 1u + (uint64_t)round &&
                 items[i].last_index == 2u + (uint64_t)round && items[i].slot >= 1 && items[i].slot < P;
    }
    /* ---- Config::max_size_per_msg in bytes: the host keeps the cumulative entry sizes on the device, the stage applies
     *      util::limit_size. Both followers acknowledge index 4 (their windows empty), entry 5 is sent: one message ---- */
    if (ok) {
        rg_log_size sz[G * 4];
        uint64_t k = 0;
        CHECK(rg_log_sizes_enable(h, 8));
        for (int g = 0; g < G; g++)
            for (uint64_t idx = 2; idx <= 5; idx++) { /* entries 2..5 of 100 bytes each: cumulative sums */
                sz[k].group = (uint64_t)g;
                sz[k].index = idx;
                sz[k].cum_bytes = 100 * (idx - 1);
                k++;
            }
        CHECK(rg_log_sizes_write(h, sz, k));
        n = 0;
        for (int g = 0; g < G; g++)
            for (int p = 1; p < P; p++) {
                memset(&recs[n], 0, sizeof recs[n]);
                recs[n].group = (uint64_t)g;
                recs[n].slot = (uint32_t)p;
                recs[n].index = 4;
                recs[n].commit = 2;
                recs[n].flags = RG_MF_VALID;
                n++;
            }
        CHECK(rg_ingest_tick(h, recs, n, &touched, &dup));
        CHECK(rg_send_appends(h, 150, RG_SEND_BYTES)); /* 150 bytes: one 100-byte entry per MsgAppend */
        CHECK(rg_send_items(h, items, sizeof items / sizeof items[0], &n));
        printf("byte-limited stage: %llu MsgAppend work items\n", (unsigned long long)n);
        ok = ok && n == (uint64_t)G * (P - 1);
        for (uint64_t i = 0; i < n && ok; i++)
            ok = items[i].kind == RG_SEND_APPEND && items[i].n_msgs == 1 && items[i].prev_index == 4 && items[i].last_index == 5;
    }
    /* ---- multi-GPU entry points from plain C: this rank is the whole world (ncclAllGather at world size 1); with more
     *      ranks only `rank` / `world` change and rank 0's unique id travels over the host's own channel ---- */
    if (ok) {
        uint8_t id[RG_COMM_ID_BYTES];
        rg_comm_config cc;
        uint64_t column[G], published[G];
        uint32_t out2[G];
        rg_publish_stats ps;
        memset(&cc, 0, sizeof cc);
        CHECK(rg_comm_unique_id(id));
        cc.rank = 0;
        cc.world = 1;
        cc.unique_id = id;
        CHECK(rg_comm_init(h, &cc)); /* collective: communicator + one full publication */
        for (int g = 0; g < G; g++) { /* every leader persists what it appended; both followers acknowledge index 5 */
            for (int p = 0; p < P; p++) {
                rg_wire_msg *r = &recs[g * P + p];
                memset(r, 0, sizeof *r);
                r->group = (uint64_t)g;
                r->slot = (uint32_t)p;
                r->index = 5;
                r->commit = 2;
                r->flags = RG_MF_VALID;
            }
        }
        CHECK(rg_ingest_tick(h, recs, (uint64_t)G * P, &touched, &dup));
        CHECK(rg_publish_commit(h, 0)); /* asynchronous: the tick's commit advances, one byte per group */
        CHECK(rg_results(h, column, out2));
        CHECK(rg_published_commit(h, 0, 0, G, published));
        CHECK(rg_publish_stats_get(h, &ps));
        for (int g = 0; g < G; g++) ok = ok && column[g] == 5 && published[g] == column[g];
        ok = ok && ps.publications == 2 && ps.full_publications == 1 && ps.bytes_per_rank_delta < ps.bytes_per_rank_full;
        printf("published commit indices: %llu %llu %llu %llu (delta slice %llu B, full column %llu B)\n",
               (unsigned long long)published[0], (unsigned long long)published[1], (unsigned long long)published[2],
               (unsigned long long)published[3], (unsigned long long)ps.bytes_per_rank_delta,
               (unsigned long long)ps.bytes_per_rank_full);
        CHECK(rg_comm_destroy(h));
    }
    /* ---- one process, ONE thread, several engines (raftgroups.h: "several ranks in ONE process"): two more shards of
     *      the same shape become ranks 0 and 1 of one publication; both live on this box's one device, so the exchange is
     *      the in-process transport (with a device each it is RCCL inside one ncclGroupStart / ncclGroupEnd) ---- */
    if (ok) {
        rg_engine *e2[2] = {NULL, NULL};
        rg_workload w = {0x5EED5EEDull, RG_WL_MAJORITY, 0};
        uint64_t own[2][G], seen[G];
        uint32_t o2[G];
        rg_device_info di;
        for (int r = 0; r < 2; r++) {
            CHECK(rg_create(&cfg, &e2[r]));
            CHECK(rg_workload_init(e2[r], &w, (uint64_t)r * G));
        }
        CHECK(rg_get_device_info(e2[1], &di));
        ok = ok && di.engines_on_device == 3; /* h and the two shards: counted per DEVICE, at rg_create */
        CHECK(rg_comm_init_all(e2, 2, NULL)); /* NULL = defaults; AUTO picks the in-process transport for a shared device */
        ok = ok && rg_publish_commit(e2[0], 0) == RG_ERR_STATE; /* such ranks are published together */
        for (int r = 0; r < 2; r++) CHECK(rg_recompute(e2[r])); /* (any commit-changing call accumulates into the slice) */
        CHECK(rg_publish_commit_all(e2, 2, 0));
        for (int r = 0; r < 2; r++) CHECK(rg_results(e2[r], own[r], o2));
        for (int reader = 0; reader < 2; reader++)
            for (int r = 0; r < 2; r++) {
                CHECK(rg_published_commit(e2[reader], (uint32_t)r, 0, G, seen));
                for (int g = 0; g < G; g++) ok = ok && seen[g] == own[r][g];
            }
        printf("two engines, one thread: rank 1's first commit index %llu seen by rank 0\n", (unsigned long long)seen[0]);
        for (int r = 0; r < 2; r++) rg_destroy(e2[r]);
    }
    rg_device_info info;
    CHECK(rg_get_device_info(h, &info));
    printf("device %s, %u CUs, wave%u, engine holds %llu bytes\n", info.arch, info.compute_units, info.wavefront,
           (unsigned long long)info.engine_bytes);
    rg_destroy(h);
    free(col);
    puts(ok ? "C_DRIVER_OK" : "C_DRIVER_FAILED");
    return ok ? 0 : 1;
}
