"""Known-answer tables of the reference's table-driven unit tests.

The rows are extracted MECHANICALLY from the reference sources by tests/golden/make_golden.py into
reference_tables.json (each table records the file:line range it came from; only the input/expected tuples are
data -- the reference's test code is not copied). This module only shapes them the way the consumers
(tests/test_oracle_golden.py, tests/scenarios.py) read them, and documents what the columns mean.
"""
import json
import os

PROBE, REPLICATE, SNAPSHOT = 0, 1, 2
_STATE = {"Probe": PROBE, "Replicate": REPLICATE, "Snapshot": SNAPSHOT}

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_tables.json"), encoding="utf-8") as _f:
    _T = json.load(_f)
SOURCES = {k: v["source"] for k, v in _T.items()}


def _rows(name):
    return _T[name]["rows"]


# src/tracker/progress.rs test_progress_is_paused: (state, paused, want)
PROGRESS_IS_PAUSED = [(_STATE[s], p, w) for s, p, w in _rows("PROGRESS_IS_PAUSED")]

# src/tracker/progress.rs test_progress_become_probe: ((state, matched, next, pending_snapshot, ins_size), want_next)
PROGRESS_BECOME_PROBE = [((_STATE[p[0]], p[1], p[2], p[3], p[4]), w) for p, w in _rows("PROGRESS_BECOME_PROBE")]

# src/tracker/progress.rs test_progress_update: prev (match, next); rows (update, want_match, want_next, want_ok)
PROGRESS_UPDATE_PREV = (_T["PROGRESS_UPDATE"]["constants"]["prev_m"], _T["PROGRESS_UPDATE"]["constants"]["prev_n"])
PROGRESS_UPDATE = [tuple(r) for r in _rows("PROGRESS_UPDATE")]

# src/tracker/progress.rs test_progress_maybe_decr: (state, match, next, rejected, last(hint), want_ok, want_next)
PROGRESS_MAYBE_DECR = [(_STATE[r[0]],) + tuple(r[1:]) for r in _rows("PROGRESS_MAYBE_DECR")]

# src/raft_log.rs test_commit_to: log (1,1),(2,2),(3,3), committed 2; (commit, want, panics)
COMMIT_TO = [tuple(r) for r in _rows("COMMIT_TO")]

# harness/tests/integration_cases/test_raft.rs test_commit: (matches, log [(term, index)], sm_term, want_commit).
# Peer 1 is the leader; its matched is the persisted log length (not overwritten by the test) and equals
# matches[0] in every row.
TEST_COMMIT = [(m, [tuple(e) for e in log], t, w) for m, log, t, w in _rows("TEST_COMMIT")]

# test_raft.rs test_group_commit: (matches, group_ids, want_with_group_commit, want_with_plain_quorum); log = term-1
# entries min(matches)..=max(matches), leader term 1, commit starts at 0.
TEST_GROUP_COMMIT = [tuple(r) for r in _rows("TEST_GROUP_COMMIT")]

# test_raft.rs test_group_commit_consistent, the LEADER rows (check_group_commit_consistent = use_group_commit &&
# mci == committed, src/raft.rs:557-576, needs role == Leader and applied >= first index of the term): log terms
# 1 x5 (1..5), 2 x3 (6..8), term 2. (matches, group_ids, committed, applied, want), want None when
# apply_to_current_term fails.
TEST_GROUP_COMMIT_CONSISTENT = [(m, g, c, a, w) for m, g, c, a, role, w in _rows("TEST_GROUP_COMMIT_CONSISTENT")
                                if role == "Leader"]

# test_raft.rs test_leader_append_response. Setup derived from the test body: storage entries (term 0, idx 1),
# (term 1, idx 2); become_candidate (term 1) + become_leader appends the noop at idx 3 (term 1) which is NOT yet
# persisted, so the leader's own matched = 2; followers reset to match 0 / next 3 / Probe (Raft::reset runs before
# the noop append, src/raft.rs:960-970,1163-1194). Rows: (index, reject, want_match, want_next_after_send,
# want_msg_num, want_msg_index, want_msg_commit). want_next includes the send path's update_state.
TEST_LEADER_APPEND_RESPONSE = [tuple(r) for r in _rows("TEST_LEADER_APPEND_RESPONSE")]

# test_raft_paper.rs test_leader_only_commits_log_from_current_term: log (1,1),(2,2) + term-3 noop at 3 + proposal
# at 4, all persisted (leader matched 4), voters {1,2}; (ack index from peer 2, want_commit)
TEST_LEADER_ONLY_COMMITS_CURRENT_TERM = [tuple(r) for r in _rows("TEST_LEADER_ONLY_COMMITS_CURRENT_TERM")]

# test_raft_paper.rs test_leader_acknowledge_commit: (cluster size, acceptor ids, want_committed). After
# commit_noop_entry every peer has matched 1 and commit is 1; the proposal is index 2 (term 1), persisted by the
# leader; acceptors ack index 2.
TEST_LEADER_ACKNOWLEDGE_COMMIT = [(n, sorted(int(k) for k, v in acc.items() if v), w)
                                  for n, acc, w in _rows("TEST_LEADER_ACKNOWLEDGE_COMMIT")]
