"""Known-answer tables transcribed BY HAND from the reference's table-driven unit tests.

Each table cites the reference file:line (relative to /root/reference) it was read from. Only the
input/expected tuples are reproduced (they are data, the reference's test *code* is not copied);
tests/test_oracle_golden.py and tests/test_scenarios.py re-run them against the oracle and the engine.
"""
PROBE, REPLICATE, SNAPSHOT = 0, 1, 2

# src/tracker/progress.rs:265-281 test_progress_is_paused: (state, paused, want)
PROGRESS_IS_PAUSED = [
    (PROBE, False, False), (PROBE, True, True), (REPLICATE, False, False), (REPLICATE, True, False),
    (SNAPSHOT, False, True), (SNAPSHOT, True, True),
]

# src/tracker/progress.rs:297-329 test_progress_become_probe:
# ((state, matched, next, pending_snapshot, ins_size), want_next)
PROGRESS_BECOME_PROBE = [
    ((REPLICATE, 1, 5, 0, 256), 2),
    ((SNAPSHOT, 1, 5, 10, 256), 11),  # snapshot finish
    ((SNAPSHOT, 1, 5, 0, 256), 2),    # snapshot failure
]

# src/tracker/progress.rs:351-373 test_progress_update: prev (match, next) = (3, 5);
# (update, want_match, want_next, want_ok)
PROGRESS_UPDATE_PREV = (3, 5)
PROGRESS_UPDATE = [(2, 3, 5, False), (3, 3, 5, False), (4, 4, 5, True), (5, 5, 6, True)]

# src/tracker/progress.rs:376-412 test_progress_maybe_decr:
# (state, match, next, rejected, last(hint), want_ok, want_next)
PROGRESS_MAYBE_DECR = [
    (REPLICATE, 5, 10, 5, 5, False, 10),
    (REPLICATE, 5, 10, 4, 4, False, 10),
    (REPLICATE, 5, 10, 9, 9, True, 6),
    (PROBE, 0, 0, 0, 0, False, 0),
    (PROBE, 0, 10, 5, 5, False, 10),
    (PROBE, 0, 10, 9, 9, True, 9),
    (PROBE, 0, 2, 1, 1, True, 1),
    (PROBE, 0, 1, 0, 0, True, 1),
    (PROBE, 0, 10, 9, 2, True, 3),
    (PROBE, 0, 10, 9, 0, True, 1),
]

# src/raft_log.rs:1498-1522 test_commit_to: log (1,1),(2,2),(3,3), committed 2; (commit, want, panics)
COMMIT_TO = [(3, 3, False), (1, 2, False), (4, 0, True)]

# harness/tests/integration_cases/test_raft.rs:1145-1240 test_commit:
# (matches, log [(term, index)], sm_term, want_commit). Peer 1 is the leader; its matched is the
# persisted log length (not overwritten by the test, :1222-1230) and equals matches[0] in every row.
TEST_COMMIT = [
    ([1], [(1, 1)], 1, 1),
    ([1], [(1, 1)], 2, 0),
    ([2], [(1, 1), (2, 2)], 2, 2),
    ([1], [(2, 1)], 2, 1),
    ([2, 1, 1], [(1, 1), (2, 2)], 1, 1),
    ([2, 1, 1], [(1, 1), (1, 2)], 2, 0),
    ([2, 1, 2], [(1, 1), (2, 2)], 2, 2),
    ([2, 1, 2], [(1, 1), (1, 2)], 2, 0),
    ([2, 1, 1, 1], [(1, 1), (2, 2)], 1, 1),
    ([2, 1, 1, 1], [(1, 1), (1, 2)], 2, 0),
    ([2, 1, 1, 2], [(1, 1), (2, 2)], 1, 1),
    ([2, 1, 1, 2], [(1, 1), (1, 2)], 2, 0),
    ([2, 1, 2, 2], [(1, 1), (2, 2)], 2, 2),
    ([2, 1, 2, 2], [(1, 1), (1, 2)], 2, 0),
]

# harness/tests/integration_cases/test_raft.rs:5092-5163 test_group_commit:
# (matches, group_ids, want_with_group_commit, want_with_plain_quorum); log = term-1 entries
# min(matches)..=max(matches), leader term 1, commit starts at 0.
TEST_GROUP_COMMIT = [
    ([1], [0], 1, 1),
    ([1], [1], 1, 1),
    ([2, 2, 1], [1, 2, 1], 2, 2),
    ([2, 2, 1], [1, 1, 2], 1, 2),
    ([2, 2, 1], [1, 0, 1], 1, 2),
    ([2, 2, 1], [0, 0, 0], 1, 2),
    ([4, 2, 1, 3], [0, 0, 0, 0], 1, 2),
    ([4, 2, 1, 3], [1, 0, 0, 0], 1, 2),
    ([4, 2, 1, 3], [0, 1, 0, 2], 2, 2),
    ([4, 2, 1, 3], [0, 2, 1, 0], 1, 2),
    ([4, 2, 1, 3], [1, 1, 1, 1], 2, 2),
    ([4, 2, 1, 3], [1, 1, 2, 1], 1, 2),
    ([4, 2, 1, 3], [1, 2, 1, 1], 2, 2),
    ([4, 2, 1, 3], [4, 3, 2, 1], 2, 2),
]

# harness/tests/integration_cases/test_raft.rs:5166-5287 test_group_commit_consistent (leader rows
# only: check_group_commit_consistent = use_group_commit && mci == committed, src/raft.rs:557-576,
# needs role == Leader and applied >= first index of the term): log terms 1 x5 (1..5), 2 x3 (6..8),
# term 2. (matches, group_ids, committed, applied, want) with want None when apply_to_current_term fails.
TEST_GROUP_COMMIT_CONSISTENT = [
    ([8], [0], 8, 6, False),
    ([8], [1], 8, 5, None),
    ([8, 2, 0], [1, 2, 1], 2, 2, None),
    ([8, 2, 6], [1, 1, 2], 6, 6, True),
    ([8, 2, 6], [1, 1, 2], 6, 5, None),
    ([8, 6, 6], [0, 0, 0], 6, 6, False),
    ([8, 6, 6], [1, 1, 1], 6, 6, False),
    ([8, 6, 6], [1, 1, 0], 6, 6, False),
]

# harness/tests/integration_cases/test_raft.rs:2611-2675 test_leader_append_response.
# Setup derived from the test body: storage entries (term 0, idx 1), (term 1, idx 2); become_candidate
# (term 1) + become_leader appends the noop at idx 3 (term 1) which is NOT yet persisted, so the
# leader's own matched = 2; followers reset to match 0 / next 3 / Probe (Raft::reset runs before the
# noop append, src/raft.rs:960-970,1163-1194). Rows: (index, reject, want_match, want_next_after_send,
# want_msg_num, want_msg_index, want_msg_commit). want_next includes the send path's update_state.
TEST_LEADER_APPEND_RESPONSE = [
    (3, True, 0, 3, 0, 0, 0),   # stale resp; no replies
    (2, True, 0, 2, 1, 1, 0),   # denied resp; decrease next and send probing message
    (2, False, 2, 4, 2, 2, 2),  # accepted resp; leader commits; broadcast with committed index
    (0, False, 0, 3, 0, 0, 0),
]

# harness/tests/integration_cases/test_raft_paper.rs:1012-1052
# test_leader_only_commits_log_from_current_term: log (1,1),(2,2) + term-3 noop at 3 + proposal at 4,
# all persisted (leader matched 4), voters {1,2}; (ack index from peer 2, want_commit)
TEST_LEADER_ONLY_COMMITS_CURRENT_TERM = [(1, 0), (2, 0), (3, 3)]

# harness/tests/integration_cases/test_raft_paper.rs:499-534 test_leader_acknowledge_commit:
# (cluster size, acceptor ids, want_committed). After commit_noop_entry every peer has matched 1 and
# commit is 1; the proposal is index 2 (term 1), persisted by the leader; acceptors ack index 2.
TEST_LEADER_ACKNOWLEDGE_COMMIT = [
    (1, [], True), (3, [], False), (3, [2], True), (3, [2, 3], True), (5, [], False), (5, [2], False),
    (5, [2, 3], True), (5, [2, 3, 4], True), (5, [2, 3, 4, 5], True),
]
