/*
 * raft_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE). See raft_oracle.h.
 *
 * Message-at-a-time restatement of pingcap/raft-rs v0.6.0's leader-side
 * progress/commit path with the reference's data model. Citations are
 * file:line relative to /root/reference. Nothing in raft_rs_amd/ links this.
 */
#include "raft_oracle.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* Inflights -- src/tracker/inflights.rs:42-125                         */
/* ------------------------------------------------------------------ */

void ro_ins_init(ro_inflights *ins, size_t cap) { /* inflights.rs:44-50 */
    ins->start = 0;
    ins->count = 0;
    ins->cap = cap;
    ins->len = 0;
    ins->buffer = cap ? (uint64_t *)calloc(cap, sizeof(uint64_t)) : NULL;
}

void ro_ins_destroy(ro_inflights *ins) {
    free(ins->buffer);
    ins->buffer = NULL;
}

bool ro_ins_full(const ro_inflights *ins) { return ins->count == ins->cap; } /* :54-56 */

int ro_ins_add(ro_inflights *ins, uint64_t inflight) { /* :65-81 */
    if (ro_ins_full(ins)) return -1; /* panic!("cannot add into a full inflights") */
    size_t next = ins->start + ins->count;
    if (next >= ins->cap) next -= ins->cap;
    if (next > ins->len) return -1; /* assert!(next <= self.buffer.len()) */
    if (next == ins->len) ins->len++;
    ins->buffer[next] = inflight;
    ins->count++;
    return 0;
}

void ro_ins_free_to(ro_inflights *ins, uint64_t to) { /* :84-110 */
    if (ins->count == 0 || to < ins->buffer[ins->start]) return; /* out of the left side of the window */
    size_t i = 0, idx = ins->start;
    while (i < ins->count) {
        if (to < ins->buffer[idx]) break; /* found the first large inflight */
        idx++;
        if (idx >= ins->cap) idx -= ins->cap;
        i++;
    }
    ins->count -= i;
    ins->start = idx;
}

void ro_ins_free_first_one(ro_inflights *ins) { /* :114-117 */
    ro_ins_free_to(ins, ins->buffer[ins->start]);
}

void ro_ins_reset(ro_inflights *ins) { /* :121-124 */
    ins->count = 0;
    ins->start = 0;
}

/* ------------------------------------------------------------------ */
/* Progress -- src/tracker/progress.rs:58-244                           */
/* ------------------------------------------------------------------ */

void ro_progress_new(ro_progress *p, uint64_t next_idx, size_t ins_size) { /* :60-73 */
    p->matched = 0;
    p->next_idx = next_idx;
    p->state = RO_PROBE;
    p->paused = false;
    p->pending_snapshot = 0;
    p->pending_request_snapshot = 0;
    p->recent_active = false;
    ro_ins_init(&p->ins, ins_size);
    p->commit_group_id = 0;
    p->committed_index = 0;
}

void ro_progress_destroy(ro_progress *p) { ro_ins_destroy(&p->ins); }

static void reset_state(ro_progress *p, uint8_t state) { /* :75-80 */
    p->paused = false;
    p->pending_snapshot = 0;
    p->state = state;
    ro_ins_reset(&p->ins);
}

void ro_progress_reset(ro_progress *p, uint64_t next_idx) { /* :82-92 */
    p->matched = 0;
    p->next_idx = next_idx;
    p->state = RO_PROBE;
    p->paused = false;
    p->pending_snapshot = 0;
    p->pending_request_snapshot = RO_INVALID_INDEX;
    p->recent_active = false;
    ro_ins_reset(&p->ins);
}

void ro_progress_become_probe(ro_progress *p) { /* :95-107 */
    if (p->state == RO_SNAPSHOT) {
        uint64_t pending_snapshot = p->pending_snapshot;
        reset_state(p, RO_PROBE);
        uint64_t a = p->matched + 1, b = pending_snapshot + 1;
        p->next_idx = a > b ? a : b;
    } else {
        reset_state(p, RO_PROBE);
        p->next_idx = p->matched + 1;
    }
}

void ro_progress_become_replicate(ro_progress *p) { /* :111-114 */
    reset_state(p, RO_REPLICATE);
    p->next_idx = p->matched + 1;
}

void ro_progress_become_snapshot(ro_progress *p, uint64_t snapshot_idx) { /* :118-121 */
    reset_state(p, RO_SNAPSHOT);
    p->pending_snapshot = snapshot_idx;
}

bool ro_progress_maybe_snapshot_abort(const ro_progress *p) { /* :132-134 */
    return p->state == RO_SNAPSHOT && p->matched >= p->pending_snapshot;
}

bool ro_progress_maybe_update(ro_progress *p, uint64_t n) { /* :138-150 */
    bool need_update = p->matched < n;
    if (need_update) {
        p->matched = n;
        p->paused = false; /* resume() */
    }
    if (p->next_idx < n + 1) p->next_idx = n + 1;
    return need_update;
}

void ro_progress_update_committed(ro_progress *p, uint64_t committed_index) { /* :153-157 */
    if (committed_index > p->committed_index) p->committed_index = committed_index;
}

bool ro_progress_maybe_decr_to(ro_progress *p, uint64_t rejected, uint64_t match_hint,
                               uint64_t request_snapshot) { /* :168-206 */
    if (p->state == RO_REPLICATE) {
        /* stale if "rejected" is smaller than "match", or equal with no snapshot request */
        if (rejected < p->matched ||
            (rejected == p->matched && request_snapshot == RO_INVALID_INDEX)) {
            return false;
        }
        if (request_snapshot == RO_INVALID_INDEX) {
            p->next_idx = p->matched + 1;
        } else {
            p->pending_request_snapshot = request_snapshot;
        }
        return true;
    }
    /* stale if "rejected" does not match next - 1, unless it requests a snapshot */
    if ((p->next_idx == 0 || p->next_idx - 1 != rejected) &&
        request_snapshot == RO_INVALID_INDEX) {
        return false;
    }
    if (request_snapshot == RO_INVALID_INDEX) {
        uint64_t h = match_hint + 1;
        p->next_idx = rejected < h ? rejected : h;
        if (p->next_idx < 1) p->next_idx = 1;
    } else if (p->pending_request_snapshot == RO_INVALID_INDEX) {
        p->pending_request_snapshot = request_snapshot;
    }
    p->paused = false; /* resume() */
    return true;
}

bool ro_progress_is_paused(const ro_progress *p) { /* :210-216 */
    switch (p->state) {
    case RO_PROBE: return p->paused;
    case RO_REPLICATE: return ro_ins_full(&p->ins);
    default: return true;
    }
}

int ro_progress_update_state(ro_progress *p, uint64_t last) { /* :231-243 */
    switch (p->state) {
    case RO_REPLICATE:
        p->next_idx = last + 1; /* optimistic_update :161 */
        /* self.ins.add(last): panics on a full window (inflights.rs:66-68) -- reported, like every panic of this
         * restatement, as -1. (cap == 0: a Progress whose Inflights the caller does not model.) */
        if (p->ins.cap && ro_ins_add(&p->ins, last) != 0) return -1;
        return 0;
    case RO_PROBE:
        p->paused = true;
        return 0;
    default:
        return -1; /* panic!("updating progress state in unhandled state") */
    }
}

/* ------------------------------------------------------------------ */
/* Quorum -- src/util.rs:118-120, src/quorum/majority.rs, joint.rs      */
/* ------------------------------------------------------------------ */

size_t ro_majority(size_t total) { return total / 2 + 1; }

uint64_t ro_majority_committed_index(const ro_index *acked, size_t n, bool use_group_commit,
                                     bool *used_group_commit) { /* majority.rs:70-124 */
    if (n == 0) { /* :71-75 */
        if (used_group_commit) *used_group_commit = true;
        return UINT64_MAX;
    }
    ro_index stack_arr[16];
    ro_index *matched = n <= 16 ? stack_arr : (ro_index *)malloc(n * sizeof(ro_index));
    memcpy(matched, acked, n * sizeof(ro_index));
    /* "Reverse sort": matched.sort_by(|a, b| b.index.cmp(&a.index)) -- stable (:95) */
    for (size_t i = 1; i < n; i++) {
        ro_index x = matched[i];
        size_t j = i;
        while (j > 0 && matched[j - 1].index < x.index) {
            matched[j] = matched[j - 1];
            j--;
        }
        matched[j] = x;
    }
    size_t quorum = ro_majority(n);
    ro_index quorum_index = matched[quorum - 1];
    uint64_t result;
    bool flag = false;
    if (!use_group_commit) { /* :99-101 */
        result = quorum_index.index;
    } else { /* :102-123 */
        uint64_t quorum_commit_index = quorum_index.index;
        uint64_t checked_group_id = quorum_index.group_id;
        bool single_group = true;
        bool returned = false;
        result = 0;
        for (size_t i = 0; i < n; i++) {
            const ro_index *m = &matched[i];
            if (m->group_id == 0) {
                single_group = false;
                continue;
            }
            if (checked_group_id == 0) {
                checked_group_id = m->group_id;
                continue;
            }
            if (checked_group_id == m->group_id) continue;
            result = m->index < quorum_commit_index ? m->index : quorum_commit_index;
            flag = true;
            returned = true;
            break;
        }
        if (!returned) {
            result = single_group ? quorum_commit_index : matched[n - 1].index;
            flag = false;
        }
    }
    if (matched != stack_arr) free(matched);
    if (used_group_commit) *used_group_commit = flag;
    return result;
}

int ro_majority_vote_result(const uint8_t *votes, size_t n) { /* majority.rs:130-154 */
    if (n == 0) return RO_VOTE_WON;
    size_t yes = 0, missing = 0;
    for (size_t i = 0; i < n; i++) {
        if (votes[i] == 2) yes++;
        else if (votes[i] == 0) missing++;
    }
    size_t q = ro_majority(n);
    if (yes >= q) return RO_VOTE_WON;
    if (yes + missing >= q) return RO_VOTE_PENDING;
    return RO_VOTE_LOST;
}

int ro_joint_vote_result(int i, int o) { /* joint.rs:56-67 */
    if (i == RO_VOTE_WON && o == RO_VOTE_WON) return RO_VOTE_WON;
    if (i == RO_VOTE_LOST || o == RO_VOTE_LOST) return RO_VOTE_LOST;
    return RO_VOTE_PENDING;
}

/* ------------------------------------------------------------------ */
/* Per-group state with the reference's data model                      */
/* ------------------------------------------------------------------ */

#define RO_MAP_CAP 16 /* open-addressing table; ids per group <= 8 in every test */

typedef struct { /* HashMap<u64, Progress>, src/tracker.rs:181 */
    uint64_t keys[RO_MAP_CAP]; /* 0 = empty (INVALID_ID is never a peer id, raw_node.rs:303) */
    ro_progress vals[RO_MAP_CAP];
    uint64_t order[RO_MAP_CAP]; /* insertion order, the iteration order used here */
    size_t len;
} ro_pmap;

typedef struct { /* HashSet<u64> */
    uint64_t ids[RO_MAP_CAP];
    size_t len;
} ro_idset;

typedef struct {
    uint64_t first;
    uint64_t term;
} ro_run;

typedef struct {
    /* ProgressTracker (src/tracker.rs:195-209) */
    ro_pmap progress;
    ro_idset incoming, outgoing, learners;
    bool group_commit;
    size_t max_inflight;
    /* Raft (src/raft.rs:164-260): the fields the path reads */
    uint64_t id, term, lead_transferee;
    bool pending_conf; /* has_pending_conf(): pending_conf_index > raft_log.applied (raft.rs:2679-2682) */
    /* RaftLog (src/raft_log.rs:33-59) reduced to what term()/commit_to() read */
    uint64_t dummy_index, dummy_term, last_index, committed;
    ro_run *runs;
    size_t n_runs;
    bool configured;
    /* Entry::compute_size() of the entries [ent_first, ent_first + ent_n) -- what util::limit_size adds up
     * (src/util.rs:52-76); only present when the caller models Config::max_size_per_msg in bytes */
    uint32_t *ent_size;
    uint64_t ent_first;
    size_t ent_n, ent_cap;
} ro_group;

struct ro_cluster {
    size_t n;
    ro_group *g;
    bool own_inflights; /* SoA ticks use the Progress's own Inflights instead of the RG_MF_INS_FULL bit */
    bool limit_bytes;   /* ro_maybe_send_append's limit is Config::max_size_per_msg in BYTES (ro_set_limit_bytes) */
};

static inline size_t fx_slot(uint64_t id) { /* FxHash: multiply by the Fx seed (src/lib.rs:602-604) */
    return (size_t)((id * 0x517cc1b727220a95ULL) >> 60) & (RO_MAP_CAP - 1);
}

static ro_progress *pmap_get(ro_pmap *m, uint64_t id) {
    if (id == 0) return NULL;
    size_t s = fx_slot(id);
    for (size_t i = 0; i < RO_MAP_CAP; i++) {
        size_t k = (s + i) & (RO_MAP_CAP - 1);
        if (m->keys[k] == id) return &m->vals[k];
        if (m->keys[k] == 0) return NULL;
    }
    return NULL;
}

static ro_progress *pmap_insert(ro_pmap *m, uint64_t id) {
    if (id == 0 || m->len >= RO_MAP_CAP - 1) return NULL;
    size_t s = fx_slot(id);
    for (size_t i = 0; i < RO_MAP_CAP; i++) {
        size_t k = (s + i) & (RO_MAP_CAP - 1);
        if (m->keys[k] == id) return &m->vals[k];
        if (m->keys[k] == 0) {
            m->keys[k] = id;
            m->order[m->len++] = id;
            return &m->vals[k];
        }
    }
    return NULL;
}

static bool idset_contains(const ro_idset *s, uint64_t id) {
    for (size_t i = 0; i < s->len; i++)
        if (s->ids[i] == id) return true;
    return false;
}

static void idset_insert(ro_idset *s, uint64_t id) {
    if (!idset_contains(s, id) && s->len < RO_MAP_CAP) s->ids[s->len++] = id;
}

static void group_clear(ro_group *gr) {
    for (size_t k = 0; k < RO_MAP_CAP; k++)
        if (gr->progress.keys[k]) ro_progress_destroy(&gr->progress.vals[k]);
    free(gr->runs);
    free(gr->ent_size);
    memset(gr, 0, sizeof(*gr));
}

ro_cluster *ro_new(size_t n_groups) {
    ro_cluster *c = (ro_cluster *)calloc(1, sizeof(*c));
    if (!c) return NULL;
    c->n = n_groups;
    c->g = (ro_group *)calloc(n_groups ? n_groups : 1, sizeof(ro_group));
    if (!c->g) {
        free(c);
        return NULL;
    }
    return c;
}

void ro_free(ro_cluster *c) {
    if (!c) return;
    for (size_t i = 0; i < c->n; i++) group_clear(&c->g[i]);
    free(c->g);
    free(c);
}

size_t ro_n_groups(const ro_cluster *c) { return c->n; }

int ro_group_config(ro_cluster *c, size_t g, uint64_t self_id, uint64_t term,
                    const uint64_t *incoming, size_t n_in, const uint64_t *outgoing, size_t n_out,
                    const uint64_t *learners, size_t n_learners, uint64_t next_idx,
                    size_t max_inflight) {
    if (g >= c->n) return -1;
    ro_group *gr = &c->g[g];
    group_clear(gr);
    gr->id = self_id;
    gr->term = term;
    gr->max_inflight = max_inflight;
    const uint64_t *lists[3] = {incoming, outgoing, learners};
    size_t lens[3] = {n_in, n_out, n_learners};
    ro_idset *sets[3] = {&gr->incoming, &gr->outgoing, &gr->learners};
    for (int k = 0; k < 3; k++) {
        for (size_t i = 0; i < lens[k]; i++) {
            uint64_t id = lists[k][i];
            idset_insert(sets[k], id);
            if (!pmap_get(&gr->progress, id)) { /* apply_conf Add, tracker.rs:384-391 */
                ro_progress *p = pmap_insert(&gr->progress, id);
                if (!p) return -1;
                ro_progress_new(p, next_idx, max_inflight);
            }
        }
    }
    gr->configured = true;
    return 0;
}

void ro_group_set_group_commit(ro_cluster *c, size_t g, bool enable) { c->g[g].group_commit = enable; }
void ro_group_set_transferee(ro_cluster *c, size_t g, uint64_t id) { c->g[g].lead_transferee = id; }

int ro_group_set_log(ro_cluster *c, size_t g, uint64_t dummy_index, uint64_t dummy_term,
                     const uint64_t *run_first, const uint64_t *run_term, size_t n_runs,
                     uint64_t last_index, uint64_t committed) {
    if (g >= c->n) return -1;
    ro_group *gr = &c->g[g];
    free(gr->runs);
    gr->runs = n_runs ? (ro_run *)malloc(n_runs * sizeof(ro_run)) : NULL;
    gr->n_runs = n_runs;
    for (size_t i = 0; i < n_runs; i++) {
        gr->runs[i].first = run_first[i];
        gr->runs[i].term = run_term[i];
    }
    gr->dummy_index = dummy_index;
    gr->dummy_term = dummy_term;
    gr->last_index = last_index;
    gr->committed = committed;
    return 0;
}

void ro_group_append(ro_cluster *c, size_t g, uint64_t n) { /* raft.rs:976-991: entries get self.term */
    ro_group *gr = &c->g[g];
    if (n == 0) return;
    if (gr->n_runs == 0 || gr->runs[gr->n_runs - 1].term != gr->term) {
        gr->runs = (ro_run *)realloc(gr->runs, (gr->n_runs + 1) * sizeof(ro_run));
        gr->runs[gr->n_runs].first = gr->last_index + 1;
        gr->runs[gr->n_runs].term = gr->term;
        gr->n_runs++;
    }
    gr->last_index += n;
}

/* Raft::reset(term) (src/raft.rs:942-971) followed by Raft::become_leader (src/raft.rs:1151-1202), reduced to the
 * fields this model holds. Returns -1 where the reference asserts (last_index != persisted, :1170) -- the state is
 * changed all the same, like every other "fault" of this restatement. `persisted` is the leader's own matched
 * (on_persist_entries keeps the two equal, :994-1016). */
int ro_group_become_leader(ro_cluster *c, size_t g, uint64_t term) {
    ro_group *gr = &c->g[g];
    int rc = 0;
    /* reset(term) */
    if (gr->term != term) gr->term = term; /* :943-946 (vote is not modelled) */
    gr->lead_transferee = RO_INVALID_ID;   /* abort_leader_transfer, :953 */
    /* pending_conf_index / applied stay with the host (has_pending_conf is an input bit of this model) */
    uint64_t last_index = gr->last_index, committed = gr->committed;
    ro_progress *self = pmap_get(&gr->progress, gr->id);
    uint64_t persisted = self ? self->matched : 0;
    for (size_t i = 0; i < gr->progress.len; i++) { /* :964-970 */
        uint64_t id = gr->progress.order[i];
        ro_progress *pr = pmap_get(&gr->progress, id);
        if (!pr) continue;
        ro_progress_reset(pr, last_index + 1);
        if (id == gr->id) {
            pr->matched = persisted;
            pr->committed_index = committed;
        }
    }
    /* become_leader */
    if (last_index != persisted) rc = -1; /* assert_eq!(last_index, self.raft_log.persisted), :1170 */
    if (self) ro_progress_become_replicate(self); /* :1176-1181 */
    ro_group_append(c, g, 1); /* append_entry(&mut [Entry::default()]), :1191-1194 */
    return rc;
}

ro_progress *ro_group_progress(ro_cluster *c, size_t g, uint64_t id) {
    return pmap_get(&c->g[g].progress, id);
}
uint64_t ro_group_committed(const ro_cluster *c, size_t g) { return c->g[g].committed; }
uint64_t ro_group_last_index(const ro_cluster *c, size_t g) { return c->g[g].last_index; }
uint64_t ro_group_term(const ro_cluster *c, size_t g) { return c->g[g].term; }
/* the whole log as runs of equal-term entries (dummy entry excluded): first index / term of run k -> first[k], term[k];
 * returns the number of runs (only `cap` are written) */
size_t ro_group_log_runs(const ro_cluster *c, size_t g, uint64_t *first, uint64_t *term, size_t cap,
                         uint64_t *dummy_index, uint64_t *dummy_term) {
    const ro_group *gr = &c->g[g];
    for (size_t k = 0; k < gr->n_runs && k < cap; k++) {
        first[k] = gr->runs[k].first;
        term[k] = gr->runs[k].term;
    }
    if (dummy_index) *dummy_index = gr->dummy_index;
    if (dummy_term) *dummy_term = gr->dummy_term;
    return gr->n_runs;
}

/* ------------------------------------------------------------------ */
/* RaftLog::{term, commit_to, maybe_commit, find_conflict_by_term}      */
/* ------------------------------------------------------------------ */

uint64_t ro_log_term(const ro_cluster *c, size_t g, uint64_t idx) { /* raft_log.rs:122-140 */
    const ro_group *gr = &c->g[g];
    /* the valid term range is [index of dummy entry, last index] */
    if (idx < gr->dummy_index || idx > gr->last_index) return 0;
    if (idx == gr->dummy_index) return gr->dummy_term;
    uint64_t t = 0;
    for (size_t i = 0; i < gr->n_runs; i++) {
        if (gr->runs[i].first <= idx) t = gr->runs[i].term;
        else break;
    }
    return t;
}

int ro_log_commit_to(ro_cluster *c, size_t g, uint64_t to_commit) { /* raft_log.rs:286-300 */
    ro_group *gr = &c->g[g];
    if (gr->committed >= to_commit) return 0; /* never decrease commit */
    if (gr->last_index < to_commit) return -1; /* fatal!("to_commit {} is out of range") */
    gr->committed = to_commit;
    return 0;
}

bool ro_log_maybe_commit(ro_cluster *c, size_t g, uint64_t max_index, uint64_t term) { /* :487-499 */
    ro_group *gr = &c->g[g];
    if (max_index > gr->committed && ro_log_term(c, g, max_index) == term) {
        ro_log_commit_to(c, g, max_index);
        return true;
    }
    return false;
}

uint64_t ro_log_find_conflict_by_term(const ro_cluster *c, size_t g, uint64_t index,
                                      uint64_t term) { /* raft_log.rs:209-235 */
    const ro_group *gr = &c->g[g];
    uint64_t conflict_index = index;
    if (index > gr->last_index) return index;
    for (;;) {
        /* term() below the dummy index returns Ok(0), which is <= term: the loop ends there */
        uint64_t t = ro_log_term(c, g, conflict_index);
        if (t > term) conflict_index -= 1;
        else return conflict_index;
    }
}

/* ------------------------------------------------------------------ */
/* maximal_committed_index / maybe_commit / handle_append_response       */
/* ------------------------------------------------------------------ */

static uint64_t majority_ci(ro_group *gr, const ro_idset *voters, bool use_gc, bool *flag) {
    ro_index acked[RO_MAP_CAP];
    for (size_t i = 0; i < voters->len; i++) { /* majority.rs:80-82 + tracker.rs:183-190 */
        ro_progress *p = pmap_get(&gr->progress, voters->ids[i]);
        if (p) {
            acked[i].index = p->matched;
            acked[i].group_id = p->commit_group_id;
        } else { /* unwrap_or_default() */
            acked[i].index = 0;
            acked[i].group_id = 0;
        }
    }
    return ro_majority_committed_index(acked, voters->len, use_gc, flag);
}

uint64_t ro_maximal_committed_index(ro_cluster *c, size_t g, bool *used_group_commit) {
    /* tracker.rs:294-298 -> joint.rs:47-51 */
    ro_group *gr = &c->g[g];
    bool fi = false, fo = false;
    uint64_t i_idx = majority_ci(gr, &gr->incoming, gr->group_commit, &fi);
    uint64_t o_idx = majority_ci(gr, &gr->outgoing, gr->group_commit, &fo);
    if (used_group_commit) *used_group_commit = fi && fo;
    return i_idx < o_idx ? i_idx : o_idx;
}

bool ro_maybe_commit(ro_cluster *c, size_t g) { /* raft.rs:893-904 */
    ro_group *gr = &c->g[g];
    uint64_t mci = ro_maximal_committed_index(c, g, NULL);
    if (ro_log_maybe_commit(c, g, mci, gr->term)) {
        ro_progress *self = pmap_get(&gr->progress, gr->id);
        if (self) ro_progress_update_committed(self, gr->committed);
        return true;
    }
    return false;
}

void ro_handle_append_response(ro_cluster *c, size_t g, const ro_msg *m, ro_out *out) {
    /* raft.rs:1559-1775 */
    ro_group *gr = &c->g[g];
    memset(out, 0, sizeof(*out));
    uint64_t next_probe_index = m->reject_hint; /* :1560 */
    if (m->reject && m->log_term > 0)           /* :1562, :1657-1660 */
        next_probe_index = ro_log_find_conflict_by_term(c, g, m->reject_hint, m->log_term);
    ro_progress *pr = pmap_get(&gr->progress, m->from); /* :1663-1673 */
    if (!pr) return;
    out->handled = true;
    pr->recent_active = true;                      /* :1674 */
    ro_progress_update_committed(pr, m->commit);   /* :1677 */

    if (m->reject) { /* :1679-1722 */
        if (ro_progress_maybe_decr_to(pr, m->index, next_probe_index, m->request_snapshot)) {
            if (pr->state == RO_REPLICATE) ro_progress_become_probe(pr);
            out->send_append = true; /* self.send_append(m.from) */
        }
        return;
    }

    bool old_paused = m->ins_full < 0 ? ro_progress_is_paused(pr) /* :1724 */
                      : (pr->state == RO_PROBE       ? pr->paused
                         : pr->state == RO_REPLICATE ? (m->ins_full != 0)
                                                     : true);
    if (!ro_progress_maybe_update(pr, m->index)) return; /* :1725-1727 */

    switch (pr->state) { /* :1729-1743 */
    case RO_PROBE: ro_progress_become_replicate(pr); break;
    case RO_SNAPSHOT:
        if (ro_progress_maybe_snapshot_abort(pr)) ro_progress_become_probe(pr);
        break;
    default:
        ro_ins_free_to(&pr->ins, m->index);
        out->free_to = true;
        break;
    }

    if (ro_maybe_commit(c, g)) { /* :1745-1748 */
        out->commit_changed = true; /* host: if should_bcast_commit() { bcast_append() } */
    } else if (old_paused) {        /* :1749-1751 */
        out->send_append = true;
    }
    out->send_more = true; /* :1761 */

    if (gr->lead_transferee != 0 && m->from == gr->lead_transferee) { /* :1764-1774 */
        if (pr->matched == gr->last_index) out->timeout_now = true;
    }
}

void ro_handle_heartbeat_response(ro_cluster *c, size_t g, uint64_t from, uint64_t commit, int8_t ins_full,
                                  ro_out *out) { /* raft.rs:1777-1803 (the read-index part is host-side) */
    ro_group *gr = &c->g[g];
    memset(out, 0, sizeof(*out));
    ro_progress *pr = pmap_get(&gr->progress, from);
    if (!pr) return;
    out->handled = true;
    ro_progress_update_committed(pr, commit); /* :1791 */
    pr->recent_active = true;                 /* :1792 */
    pr->paused = false;                       /* :1793 resume() */
    bool full = ins_full < 0 ? ro_ins_full(&pr->ins) : (ins_full != 0);
    if (pr->state == RO_REPLICATE && full) { /* :1796-1798 */
        if (ins_full < 0) ro_ins_free_first_one(&pr->ins);
        out->free_to = true;
    }
    if (pr->matched < gr->last_index || pr->pending_request_snapshot != RO_INVALID_INDEX) /* :1800-1803 */
        out->send_append = true;
}

/* MsgSnapStatus (RawNode::report_snapshot, raw_node.rs:701-709): raft.rs:1891-1929. Returns whether a Progress was
 * touched. */
bool ro_handle_snapshot_status(ro_cluster *c, size_t g, uint64_t from, bool reject) {
    ro_progress *pr = pmap_get(&c->g[g].progress, from);
    if (!pr) return false;                      /* "no progress available" */
    if (pr->state != RO_SNAPSHOT) return false; /* :1903-1905 */
    if (reject) {
        pr->pending_snapshot = 0; /* snapshot_failure(), progress.rs:124-127 */
        ro_progress_become_probe(pr);
    } else {
        ro_progress_become_probe(pr);
    }
    /* If snapshot finish, wait for the msgAppResp from the remote node before sending out the next msgAppend.
     * If snapshot failure, wait for a heartbeat interval before next try */
    pr->paused = true;                                 /* pause() */
    pr->pending_request_snapshot = RO_INVALID_INDEX;   /* :1928 */
    return true;
}

/* MsgUnreachable (RawNode::report_unreachable, raw_node.rs:692-698): raft.rs:1931-1954 */
bool ro_handle_unreachable(ro_cluster *c, size_t g, uint64_t from) {
    ro_progress *pr = pmap_get(&c->g[g].progress, from);
    if (!pr) return false;
    /* During optimistic replication, if the remote becomes unreachable, there is huge probability that a MsgAppend is lost. */
    if (pr->state == RO_REPLICATE) ro_progress_become_probe(pr);
    return true;
}

uint64_t ro_heartbeat_commit(ro_cluster *c, size_t g, uint64_t to) { /* raft.rs:830-838 */
    ro_group *gr = &c->g[g];
    ro_progress *pr = pmap_get(&gr->progress, to);
    if (!pr) return 0;
    return pr->matched < gr->committed ? pr->matched : gr->committed;
}

bool ro_on_persist_entries(ro_cluster *c, size_t g, uint64_t index) { /* raft.rs:994-1016 */
    ro_group *gr = &c->g[g];
    ro_progress *pr = pmap_get(&gr->progress, gr->id);
    if (!pr) return false;
    /* if pr.maybe_update(index) && self.maybe_commit() && should_bcast_commit() { bcast_append() } */
    return ro_progress_maybe_update(pr, index) && ro_maybe_commit(c, g);
}

int ro_group_vote_result(ro_cluster *c, size_t g, const uint64_t *ids, const uint8_t *votes,
                         size_t n) { /* tracker.rs:339-341 -> joint.rs:56-67 */
    ro_group *gr = &c->g[g];
    const ro_idset *sets[2] = {&gr->incoming, &gr->outgoing};
    int r[2];
    for (int k = 0; k < 2; k++) {
        uint8_t v[RO_MAP_CAP];
        for (size_t i = 0; i < sets[k]->len; i++) {
            v[i] = 0;
            for (size_t j = 0; j < n; j++)
                if (ids[j] == sets[k]->ids[i]) v[i] = votes[j];
        }
        r[k] = ro_majority_vote_result(v, sets[k]->len);
    }
    return ro_joint_vote_result(r[0], r[1]);
}

int ro_group_tally_votes(ro_cluster *c, size_t g, const uint64_t *ids, const uint8_t *votes, size_t n,
                         size_t *granted, size_t *rejected) { /* tracker.rs:313-333 */
    ro_group *gr = &c->g[g];
    *granted = *rejected = 0;
    for (size_t j = 0; j < n; j++) {
        if (votes[j] == 0) continue; /* no vote recorded for this id */
        bool voter = false;          /* self.conf.voters.contains(id): incoming or outgoing (joint.rs:69-72) */
        for (size_t i = 0; i < gr->incoming.len; i++) voter = voter || gr->incoming.ids[i] == ids[j];
        for (size_t i = 0; i < gr->outgoing.len; i++) voter = voter || gr->outgoing.ids[i] == ids[j];
        if (!voter) continue;
        if (votes[j] == 2) (*granted)++;
        else (*rejected)++;
    }
    return ro_group_vote_result(c, g, ids, votes, n);
}

bool ro_quorum_recently_active(ro_cluster *c, size_t g, uint64_t perspective_of) {
    /* tracker.rs:346-361 + has_quorum :367-372 */
    ro_group *gr = &c->g[g];
    uint64_t ids[RO_MAP_CAP];
    uint8_t votes[RO_MAP_CAP];
    size_t n = 0;
    for (size_t i = 0; i < gr->progress.len; i++) {
        uint64_t id = gr->progress.order[i];
        ro_progress *pr = pmap_get(&gr->progress, id);
        if (id == perspective_of) {
            pr->recent_active = true;
            ids[n] = id;
            votes[n++] = 2;
        } else if (pr->recent_active) {
            ids[n] = id;
            votes[n++] = 2;
            pr->recent_active = false;
        }
    }
    return ro_group_vote_result(c, g, ids, votes, n) == RO_VOTE_WON;
}

/* ------------------------------------------------------------------ */
/* SoA adapters. Bit layouts = include/raftgroups.h (asserted by tests). */
/* ------------------------------------------------------------------ */

#define RO_PF_STATE_MASK 0x03u
#define RO_PF_PAUSED 0x04u
#define RO_PF_RECENT_ACTIVE 0x08u
#define RO_PF_INS_FULL 0x10u
#define RO_PF_PENDING_CONF 0x20u
#define RO_PF_PEND_SNAP 0x40u /* the engine's summary bits: pending_snapshot != 0 / pending_request_snapshot != 0 */
#define RO_PF_PEND_RS 0x80u   /* (derived on store, ignored on load) */
#define RO_MF_VALID 0x01u
#define RO_MF_REJECT 0x02u
#define RO_MF_HAS_RS 0x04u
#define RO_MF_INS_FULL 0x08u
#define RO_MF_SENT 0x10u
#define RO_MF_APPEND 0x20u
#define RO_MF_BECOME_LEADER 0x02u /* on the leader's own slot */
#define RO_MF_HEARTBEAT 0x40u
#define RO_MF_HAS_LOGTERM 0x80u
#define RO_OUT_CHANGED 0x1u
#define RO_OUT_FAULT 0x2u
#define RO_OUT_TIMEOUT_NOW 0x4u
#define RO_OUT_APPENDED 0x8u
#define RO_OUT_BECAME_LEADER 0x10u

int ro_load_soa(ro_cluster *c, const ro_soa_state *s, uint64_t term, size_t max_inflight) {
    if (s->n_groups > c->n || s->n_slots > 8) return -1;
    for (size_t g = 0; g < s->n_groups; g++) {
        uint32_t cfg = s->cfg[g];
        uint32_t inc = cfg & 0xff, outg = (cfg >> 8) & 0xff, self = (cfg >> 16) & 7;
        uint32_t present = (cfg >> 24) & 0xff, xfer = (cfg >> 20) & 0xf;
        uint64_t in_ids[8], out_ids[8], l_ids[8];
        size_t ni = 0, no = 0, nl = 0;
        for (uint32_t p = 0; p < s->n_slots; p++) {
            if (inc & (1u << p)) in_ids[ni++] = p + 1;
            if (outg & (1u << p)) out_ids[no++] = p + 1;
            if ((present & (1u << p)) && !((inc | outg) & (1u << p))) l_ids[nl++] = p + 1;
        }
        /* voters WITHOUT a Progress (present bit clear) stay in the sets but get no map entry */
        if (ro_group_config(c, g, self + 1, term, NULL, 0, NULL, 0, l_ids, nl, 1, max_inflight))
            return -1;
        ro_group *gr = &c->g[g];
        for (size_t i = 0; i < ni; i++) {
            idset_insert(&gr->incoming, in_ids[i]);
            if (present & (1u << (in_ids[i] - 1)) && !pmap_get(&gr->progress, in_ids[i]))
                ro_progress_new(pmap_insert(&gr->progress, in_ids[i]), 1, max_inflight);
        }
        for (size_t i = 0; i < no; i++) {
            idset_insert(&gr->outgoing, out_ids[i]);
            if (present & (1u << (out_ids[i] - 1)) && !pmap_get(&gr->progress, out_ids[i]))
                ro_progress_new(pmap_insert(&gr->progress, out_ids[i]), 1, max_inflight);
        }
        gr->group_commit = (cfg & 0x00080000u) != 0;
        gr->lead_transferee = xfer; /* slot+1 == id */
        gr->pending_conf = (s->pflags[g * 8 + self] & RO_PF_PENDING_CONF) != 0;
        for (uint32_t p = 0; p < s->n_slots; p++) {
            ro_progress *pr = pmap_get(&gr->progress, p + 1);
            if (!pr) continue;
            size_t o = (size_t)p * s->stride + g;
            uint8_t f = s->pflags[g * 8 + p];
            pr->matched = s->match[o];
            pr->next_idx = s->next[o];
            pr->committed_index = s->pr_commit[o];
            pr->pending_snapshot = s->pend_snap[o];
            pr->pending_request_snapshot = s->pend_rs[o];
            pr->commit_group_id = s->gid[o];
            pr->state = f & RO_PF_STATE_MASK;
            pr->paused = (f & RO_PF_PAUSED) != 0;
            pr->recent_active = (f & RO_PF_RECENT_ACTIVE) != 0;
        }
        /* log: everything in [term_lo, term_hi] has the current term; everything before it an
         * older one (dummy at term_lo-1 with term-1, or an older-term run when term_lo > 1). An
         * empty range (term_lo > term_hi) = no entry of the current term yet. */
        uint64_t lo = s->term_lo[g], hi = s->term_hi[g];
        if (s->run_first) { /* explicit table: dummy entry + older runs + the leader's own run [lo, hi] */
            uint64_t rf[RO_TERM_RUNS + 1], rt[RO_TERM_RUNS + 1];
            size_t nr = 0;
            for (int k = 0; k < RO_TERM_RUNS; k++) {
                uint64_t f = s->run_first[(size_t)k * s->stride + g];
                if (f == 0) continue;
                rf[nr] = f;
                rt[nr] = s->run_term[(size_t)k * s->stride + g];
                nr++;
            }
            gr->term = s->cur_term[g];
            if (lo <= hi) {
                rf[nr] = lo;
                rt[nr] = gr->term;
                nr++;
            }
            ro_group_set_log(c, g, s->dummy_index[g], s->dummy_term[g], rf, rt, nr, hi, s->commit[g]);
        } else if (lo <= hi) {
            uint64_t rf[1] = {lo}, rt[1] = {term};
            ro_group_set_log(c, g, lo - 1, term - 1, rf, rt, 1, hi, s->commit[g]);
        } else {
            /* no entry of the current term yet: must be encoded as term_lo == term_hi + 1
             * (become_leader appends its noop at last_index + 1, raft.rs:1163-1194) */
            if (lo != hi + 1) return -2;
            ro_group_set_log(c, g, hi, term - 1, NULL, NULL, 0, hi, s->commit[g]);
        }
    }
    return 0;
}

int ro_store_soa(ro_cluster *c, ro_soa_state *s) {
    if (s->n_groups > c->n) return -1;
    for (size_t g = 0; g < s->n_groups; g++) {
        ro_group *gr = &c->g[g];
        for (uint32_t p = 0; p < s->n_slots; p++) {
            ro_progress *pr = pmap_get(&gr->progress, p + 1);
            if (!pr) continue;
            size_t o = (size_t)p * s->stride + g;
            s->match[o] = pr->matched;
            s->next[o] = pr->next_idx;
            s->pr_commit[o] = pr->committed_index;
            s->pend_snap[o] = pr->pending_snapshot;
            s->pend_rs[o] = pr->pending_request_snapshot;
            s->gid[o] = pr->commit_group_id;
            s->pflags[g * 8 + p] = (uint8_t)(pr->state | (pr->paused ? RO_PF_PAUSED : 0) |
                                             (pr->recent_active ? RO_PF_RECENT_ACTIVE : 0));
            if (c->own_inflights && pr->state == RO_REPLICATE && pr->ins.cap && ro_ins_full(&pr->ins))
                s->pflags[g * 8 + p] |= RO_PF_INS_FULL;
            if (gr->pending_conf && p + 1 == gr->id) s->pflags[g * 8 + p] |= RO_PF_PENDING_CONF;
            if (pr->pending_snapshot) s->pflags[g * 8 + p] |= RO_PF_PEND_SNAP;
            if (pr->pending_request_snapshot) s->pflags[g * 8 + p] |= RO_PF_PEND_RS;
        }
        s->commit[g] = gr->committed;
        s->term_hi[g] = gr->last_index;
        /* the index range of the leader's term: the log's last run when it carries gr->term, else empty */
        bool own = gr->n_runs && gr->runs[gr->n_runs - 1].term == gr->term;
        s->term_lo[g] = own ? gr->runs[gr->n_runs - 1].first : gr->last_index + 1;
        /* cfg: only the transferee field is state (abort_leader_transfer); the rest is an input */
        s->cfg[g] = (s->cfg[g] & ~(0xfu << 20)) | (((uint32_t)gr->lead_transferee & 0xfu) << 20);
        if (s->run_first) {
            /* the SoA table is a bounded VIEW of the log: the newest RO_TERM_RUNS runs of older terms (what the engine's
             * table holds by contract, include/raftgroups.h: RG_COL_RUN_FIRST). The log itself is not shortened. */
            size_t older = gr->n_runs - (own ? 1 : 0);
            size_t skip = older > RO_TERM_RUNS ? older - RO_TERM_RUNS : 0;
            for (size_t k = 0; k < RO_TERM_RUNS; k++) {
                s->run_first[k * s->stride + g] = skip + k < older ? gr->runs[skip + k].first : 0;
                s->run_term[k * s->stride + g] = skip + k < older ? gr->runs[skip + k].term : 0;
            }
            s->cur_term[g] = gr->term;
        }
    }
    return 0;
}

uint64_t ro_tick_soa(ro_cluster *c, const ro_soa_msgs *m, uint32_t *gout, size_t g_begin,
                     size_t g_end) {
    uint64_t stepped = 0;
    for (size_t g = g_begin; g < g_end; g++) {
        ro_group *gr = &c->g[g];
        uint32_t out = 0;
        uint32_t self_slot = (uint32_t)(gr->id - 1);
        /* RG_MF_BECOME_LEADER on the leader's own slot: the election lands before every message of the tick
         * (they answer the new leader). Engine contract for malformed input: a term that is not above the current
         * one raises the fault bit and the event is ignored. */
        if (self_slot < m->n_slots && (m->m_flags[g * 8 + self_slot] & RO_MF_BECOME_LEADER) &&
            pmap_get(&gr->progress, gr->id)) {
            uint64_t new_term = m->m_hint[(size_t)self_slot * m->stride + g];
            if (new_term > gr->term) {
                if (ro_group_become_leader(c, g, new_term) != 0) out |= RO_OUT_FAULT;
                out |= RO_OUT_BECAME_LEADER | RO_OUT_APPENDED; /* bcast_append follows become_leader, raft.rs:2190-2191 */
                /* (the log keeps EVERY run, like RaftLog: nothing here knows how deep the engine's table is) */
            } else {
                out |= RO_OUT_FAULT;
            }
        }
        uint64_t last0 = gr->last_index; /* what the host's send path saw before this tick */
        for (uint32_t p = 0; p < m->n_slots; p++) {
            uint8_t f = m->m_flags[g * 8 + p];
            if (!(f & (RO_MF_VALID | RO_MF_SENT | RO_MF_APPEND | RO_MF_HEARTBEAT))) continue;
            size_t o = (size_t)p * m->stride + g;
            ro_progress *pr = pmap_get(&gr->progress, p + 1);
            if (!pr) continue; /* "no progress available" raft.rs:1663-1673 */
            if (p == self_slot) {
                if (f & RO_MF_APPEND) { /* append_entry, raft.rs:976-991 */
                    uint64_t nl = m->m_commit[o];
                    if (nl > gr->last_index) {
                        ro_group_append(c, g, nl - gr->last_index);
                        out |= RO_OUT_APPENDED; /* step_leader MsgPropose: bcast_append follows, raft.rs:2049-2053 */
                    }
                }
                if (f & RO_MF_VALID) { /* on_persist_entries, raft.rs:994-1016 */
                    uint64_t idx = m->m_index[o];
                    if ((idx >> 63) || idx > gr->last_index) out |= RO_OUT_FAULT;
                    if (ro_on_persist_entries(c, g, idx)) out |= RO_OUT_CHANGED;
                    stepped++;
                }
                continue;
            }
            if (f & RO_MF_SENT) { /* prepare_send_entries -> update_state(last), raft.rs:726-729 */
                if (ro_progress_update_state(pr, last0) != 0) out |= RO_OUT_FAULT;
            }
            if (f & RO_MF_HEARTBEAT) { /* MsgHeartbeatResponse, raft.rs:1777-1803 */
                ro_out ho;
                ro_handle_heartbeat_response(c, g, p + 1, m->m_commit[o],
                                             c->own_inflights ? -1 : ((f & RO_MF_INS_FULL) ? 1 : 0), &ho);
                stepped++;
                if (ho.send_append) out |= 1u << (8 + p);
                if (ho.free_to) out |= 1u << (24 + p);
                continue;
            }
            if (!(f & RO_MF_VALID)) continue;
            ro_msg msg;
            msg.from = p + 1;
            msg.index = m->m_index[o];
            msg.commit = m->m_commit[o];
            msg.reject = (f & RO_MF_REJECT) != 0;
            msg.reject_hint = msg.reject ? m->m_hint[o] : 0;
            msg.request_snapshot = (f & RO_MF_HAS_RS) ? m->m_rs[o] : 0;
            msg.log_term = (msg.reject && (f & RO_MF_HAS_LOGTERM) && m->m_logterm) ? m->m_logterm[o] : 0;
            msg.ins_full = c->own_inflights ? -1 : ((f & RO_MF_INS_FULL) ? 1 : 0);
            if ((msg.index >> 63) || (!msg.reject && msg.index > gr->last_index))
                out |= RO_OUT_FAULT;
            ro_out ro;
            ro_handle_append_response(c, g, &msg, &ro);
            stepped++;
            if (ro.commit_changed) out |= RO_OUT_CHANGED;
            if (ro.timeout_now) out |= RO_OUT_TIMEOUT_NOW;
            if (ro.send_append) out |= 1u << (8 + p);
            if (ro.send_more) out |= 1u << (16 + p);
            if (ro.free_to) out |= 1u << (24 + p);
        }
        if (gout) gout[g] = out;
    }
    return stepped;
}

/* ------------------------------------------------------------------ */
/* send decisions -- src/raft.rs:664-731, 773-819, 857-864              */
/* ------------------------------------------------------------------ */

void ro_set_own_inflights(ro_cluster *c, bool on) { c->own_inflights = on; }

size_t ro_ins_contents(ro_cluster *c, size_t g, uint64_t id, uint64_t *buf, size_t cap) {
    ro_progress *pr = pmap_get(&c->g[g].progress, id);
    if (!pr) return 0;
    size_t idx = pr->ins.start;
    for (size_t i = 0; i < pr->ins.count; i++) {
        if (i < cap) buf[i] = pr->ins.buffer[idx];
        if (++idx >= pr->ins.cap) idx -= pr->ins.cap;
    }
    return pr->ins.count;
}

size_t ro_ins_export_soa(ro_cluster *c, size_t n_slots, size_t stride, uint32_t *counts, uint64_t *first_k, size_t k) {
    size_t max_count = 0;
    for (size_t g = 0; g < c->n; g++) {
        for (size_t p = 0; p < n_slots; p++) {
            ro_progress *pr = pmap_get(&c->g[g].progress, p + 1);
            uint64_t *dst = first_k + (g * n_slots + p) * k;
            for (size_t i = 0; i < k; i++) dst[i] = 0;
            counts[p * stride + g] = 0;
            if (!pr) continue;
            counts[p * stride + g] = (uint32_t)pr->ins.count;
            if (pr->ins.count > max_count) max_count = pr->ins.count;
            size_t idx = pr->ins.start;
            for (size_t i = 0; i < pr->ins.count && i < k; i++) {
                dst[i] = pr->ins.buffer[idx];
                if (++idx >= pr->ins.cap) idx -= pr->ins.cap;
            }
        }
    }
    return max_count;
}

/* prepare_send_snapshot (raft.rs:664-712) up to the point where the snapshot is fetched */
static bool ro_decide_send_snapshot(const ro_progress *pr) {
    return pr->recent_active; /* :665-672 "ignore sending snapshot ... not recently active" */
}

void ro_set_limit_bytes(ro_cluster *c, bool on) { c->limit_bytes = on; }

void ro_group_append_entry_sizes(ro_cluster *c, size_t g, uint64_t first_index, const uint32_t *sizes, size_t n) {
    ro_group *gr = &c->g[g];
    if (gr->ent_n == 0) gr->ent_first = first_index;
    if (first_index < gr->ent_first || first_index > gr->ent_first + gr->ent_n) {
        fprintf(stderr, "ro_group_append_entry_sizes: group %zu holds [%llu, %llu), got %llu\n", g,
                (unsigned long long)gr->ent_first, (unsigned long long)(gr->ent_first + gr->ent_n),
                (unsigned long long)first_index);
        abort();
    }
    size_t at = (size_t)(first_index - gr->ent_first); /* overwriting a suffix = a truncated-and-rewritten log tail */
    if (at + n > gr->ent_cap) {
        size_t cap = gr->ent_cap ? gr->ent_cap : 64;
        while (cap < at + n) cap *= 2;
        gr->ent_size = (uint32_t *)realloc(gr->ent_size, cap * sizeof(uint32_t));
        gr->ent_cap = cap;
    }
    memcpy(gr->ent_size + at, sizes, n * sizeof(uint32_t));
    gr->ent_n = at + n;
}

static uint64_t ro_entry_size(const ro_group *gr, uint64_t index) {
    if (index < gr->ent_first || index >= gr->ent_first + gr->ent_n) {
        fprintf(stderr, "ro_entry_size: no size for entry %llu (have [%llu, %llu))\n", (unsigned long long)index,
                (unsigned long long)gr->ent_first, (unsigned long long)(gr->ent_first + gr->ent_n));
        abort();
    }
    return gr->ent_size[index - gr->ent_first];
}

/* util::limit_size (src/util.rs:52-76) over the entries [next, next + n): how many survive `max` bytes. Literal,
 * including its `size == 0` test: the first entry is always kept -- and so is every entry that follows a prefix of
 * zero-size entries (Entry::default().compute_size() == 0), whatever `max` says. RaftLog::slice applies it to the
 * stable part first and to the whole again (raft_log.rs:583-608); a prefix rule, so one pass gives the same count. */
static uint64_t ro_limit_size(const ro_group *gr, uint64_t next, uint64_t n, uint64_t max) {
    if (n <= 1) return n;                 /* if entries.len() <= 1 { return; } */
    if (max == UINT64_MAX) return n;      /* None | Some(NO_LIMIT) => return */
    uint64_t size = 0, limit = 0;
    for (uint64_t k = 0; k < n; k++) {    /* take_while */
        uint64_t e = ro_entry_size(gr, next + k);
        if (size == 0) {
            size += e;
            limit++;
            continue;
        }
        size += e;
        if (size <= max) limit++;
        else break;
    }
    return limit;
}

bool ro_maybe_send_append(ro_cluster *c, size_t g, uint64_t to, bool allow_empty, uint64_t max_entries,
                          ro_send_msg *m) {
    ro_group *gr = &c->g[g];
    ro_progress *pr = pmap_get(&gr->progress, to);
    memset(m, 0, sizeof(*m));
    if (!pr) return false;
    if (ro_progress_is_paused(pr)) return false; /* :780-788 */
    m->group = g;
    m->to = to;
    if (pr->pending_request_snapshot != RO_INVALID_INDEX) { /* :791-795 */
        if (!ro_decide_send_snapshot(pr)) return false;
        m->kind = RO_SEND_SNAPSHOT;
        m->index = pr->next_idx - 1;
        return true;
    }
    /* raft_log.entries(pr.next_idx, max_msg_size) (raft_log.rs:382-389): Ok(empty) above last_index, else
     * slice(): Err(Compacted) below first_index (:463-470), else the entries up to the size limit */
    uint64_t first_index = gr->dummy_index + 1;
    bool ents_err = false;
    uint64_t n = 0;
    if (pr->next_idx <= gr->last_index) {
        if (pr->next_idx < first_index) {
            ents_err = true;
        } else {
            n = gr->last_index - pr->next_idx + 1;
            if (c->limit_bytes) n = ro_limit_size(gr, pr->next_idx, n, max_entries); /* util::limit_size, in bytes */
            else if (max_entries && n > max_entries) n = max_entries;                /* ... for equal-sized entries */
        }
    }
    if (!allow_empty && (ents_err || n == 0)) return false; /* :797-799 */
    /* raft_log.term(next_idx - 1) (raft_log.rs:122-140) cannot fail inside [dummy, last]; outside it is Ok(0) */
    if (ents_err) { /* :808-813 send snapshot if we failed to get term or entries */
        if (!ro_decide_send_snapshot(pr)) return false;
        m->kind = RO_SEND_SNAPSHOT;
        m->index = pr->next_idx - 1;
        return true;
    }
    /* prepare_send_entries (:714-731) */
    m->kind = RO_SEND_APPEND;
    m->index = pr->next_idx - 1;
    m->n_entries = n;
    if (n) ro_progress_update_state(pr, pr->next_idx + n - 1);
    return true;
}

void ro_group_set_pending_conf(ro_cluster *c, size_t g, bool pending) { c->g[g].pending_conf = pending; }

size_t ro_send_stage_soa(ro_cluster *c, const uint32_t *gout, uint64_t max_entries, bool skip_bcast_commit,
                         ro_send_msg *msgs, size_t cap, size_t g_begin, size_t g_end) {
    size_t k = 0;
    ro_send_msg m;
    for (size_t g = g_begin; g < g_end && g < c->n; g++) {
        ro_group *gr = &c->g[g];
        uint32_t out = gout[g];
        if (!out) continue;
        /* should_bcast_commit() (raft.rs:2684-2686) gates the broadcast of a commit advance only */
        bool should_bcast_commit = !skip_bcast_commit || gr->pending_conf;
        bool bcast = (out & RO_OUT_APPENDED) != 0 || ((out & RO_OUT_CHANGED) != 0 && should_bcast_commit);
        for (uint32_t p = 0; p < RO_MAP_CAP && p < 8; p++) {
            uint64_t id = p + 1;
            if (id == gr->id || !pmap_get(&gr->progress, id)) continue; /* bcast_append skips self (:859-863) */
            bool sa = bcast || ((out >> (8 + p)) & 1u), sm = (out >> (16 + p)) & 1u;
            bool snap = false;
            if (sa && ro_maybe_send_append(c, g, id, true, max_entries, &m)) {
                if (k < cap) msgs[k] = m;
                k++;
                snap = m.kind == RO_SEND_SNAPSHOT;
            }
            /* become_snapshot (applied by the caller) pauses the Progress: the loop would stop there */
            while (sm && !snap && ro_maybe_send_append(c, g, id, false, max_entries, &m)) {
                if (k < cap) msgs[k] = m;
                k++;
                snap = m.kind == RO_SEND_SNAPSHOT;
            }
        }
    }
    return k;
}

/* Groups are independent, so the CPU baseline may split a tick over threads by group range. */
typedef struct {
    ro_cluster *c;
    const ro_soa_msgs *m;
    uint32_t *gout;
    size_t a, b;
    uint64_t stepped;
} ro_mt_job;

static void *ro_mt_worker(void *arg) {
    ro_mt_job *j = (ro_mt_job *)arg;
    j->stepped = ro_tick_soa(j->c, j->m, j->gout, j->a, j->b);
    return NULL;
}

uint64_t ro_tick_soa_mt(ro_cluster *c, const ro_soa_msgs *m, uint32_t *gout, size_t n_threads) {
    size_t n = m->n_groups < c->n ? m->n_groups : c->n;
    if (n_threads <= 1) return ro_tick_soa(c, m, gout, 0, n);
    if (n_threads > 1024) n_threads = 1024;
    pthread_t *th = (pthread_t *)malloc(n_threads * sizeof(pthread_t));
    ro_mt_job *jobs = (ro_mt_job *)malloc(n_threads * sizeof(ro_mt_job));
    uint64_t total = 0;
    for (size_t i = 0; i < n_threads; i++) {
        jobs[i].c = c;
        jobs[i].m = m;
        jobs[i].gout = gout;
        jobs[i].a = i * n / n_threads;
        jobs[i].b = (i + 1) * n / n_threads;
        jobs[i].stepped = 0;
        pthread_create(&th[i], NULL, ro_mt_worker, &jobs[i]);
    }
    for (size_t i = 0; i < n_threads; i++) {
        pthread_join(th[i], NULL);
        total += jobs[i].stepped;
    }
    free(th);
    free(jobs);
    return total;
}
