"""Scenario tests shared by the oracle (CPU) and the engine (GPU): each function takes a backend
class from tests/backends.py and replays one of the reference's integration tests
(harness/tests/integration_cases/*.rs, cited per function) at the boundary of the hot path."""
from golden import reference_tables as T

PROBE, REPLICATE, SNAPSHOT = 0, 1, 2


def scenario_test_commit(B):
    """test_raft.rs:1145-1240 test_commit."""
    for i, (matches, log, sm_term, want) in enumerate(T.TEST_COMMIT):
        n = len(matches)
        ld = B(1, sm_term, list(range(1, n + 1)), log=log)
        for j, m in enumerate(matches):
            ld.set_progress(j + 1, match=m, next=m + 1)
        ld.maybe_commit()
        assert ld.committed() == want, f"#{i}: committed = {ld.committed()}, want {want}"


def scenario_test_group_commit(B):
    """test_raft.rs:5092-5163 test_group_commit."""
    for i, (matches, gids, g_w, q_w) in enumerate(T.TEST_GROUP_COMMIT):
        n = len(matches)
        log = [(1, k) for k in range(min(matches), max(matches) + 1)]
        ld = B(1, 1, list(range(1, n + 1)), log=log, dummy=(min(matches) - 1, 0))
        for j, (m, g) in enumerate(zip(matches, gids)):
            ld.set_progress(j + 1, match=m, next=m + 1, commit_group_id=g)
        ld.enable_group_commit(True)
        ld.maybe_commit()  # assign_commit_groups as leader (src/raft.rs:541-543)
        assert ld.committed() == g_w, f"#{i}: leader group committed {ld.committed()}, want {g_w}"
        ld.enable_group_commit(False)
        ld.maybe_commit()  # enable_group_commit(false) re-evaluates (src/raft.rs:513-518)
        assert ld.committed() == q_w, f"#{i}: quorum committed {ld.committed()}, want {q_w}"


def scenario_test_group_commit_consistent(B):
    """test_raft.rs:5166-5287 (leader rows): check_group_commit_consistent, src/raft.rs:557-576."""
    log = [(1, k) for k in range(1, 6)] + [(2, k) for k in range(6, 9)]
    for i, (matches, gids, committed, applied, want) in enumerate(T.TEST_GROUP_COMMIT_CONSISTENT):
        n = len(matches)
        ld = B(1, 2, list(range(1, n + 1)), log=log, committed=committed)
        for j, (m, g) in enumerate(zip(matches, gids)):
            ld.set_progress(j + 1, match=m, next=m + 1, commit_group_id=g)
        ld.enable_group_commit(True)
        apply_to_current_term = applied >= 6  # term(applied) == self.term (src/raft.rs:580-586)
        if not apply_to_current_term:
            got = None
        else:
            idx, used = ld.mci()
            got = used and idx == ld.committed()
        assert got == want, f"#{i}: consistency = {got}, want {want}"


def scenario_test_leader_append_response(B):
    """test_raft.rs:2611-2675 test_leader_append_response (setup derivation in reference_tables.py)."""
    for i, (index, reject, wmatch, wnext, wmsg_num, windex, wcommitted) in enumerate(T.TEST_LEADER_APPEND_RESPONSE):
        ld = B(1, 1, [1, 2, 3], log=[(0, 1), (1, 2), (1, 3)], committed=0, next_idx=3)
        ld.set_progress(1, match=2, next=3, state=REPLICATE)
        out = ld.step(2, index, reject=reject, reject_hint=index)
        # the send path the hot path asked for, modelled as the reference's test observes it
        msgs = []
        if out["changed"]:  # bcast_append to 2 and 3 (should_bcast_commit: skip_bcast_commit = false)
            targets = [2, 3]
        elif out["send_append"] or out["send_more"]:
            targets = [2]
        else:
            targets = []
        for to in targets:
            pr = ld.progress(to)
            if pr["state"] == PROBE and pr["paused"]:
                continue
            has_entries = pr["next"] <= 3
            if not has_entries and not (out["changed"] or out["send_append"]):
                continue  # maybe_send_append(.., allow_empty=false) sends nothing (raft.rs:797-800)
            msgs.append((pr["next"] - 1, ld.committed()))
            if has_entries:
                ld.sent(to)
        pr2 = ld.progress(2)
        assert pr2["match"] == wmatch, f"#{i}: match = {pr2['match']}, want {wmatch}"
        assert pr2["next"] == wnext, f"#{i}: next = {pr2['next']}, want {wnext}"
        assert len(msgs) == wmsg_num, f"#{i}: msg_num = {len(msgs)}, want {wmsg_num}"
        for j, (mi, mc) in enumerate(msgs):
            assert mi == windex, f"#{i}.{j}: index = {mi}, want {windex}"
            assert mc == wcommitted, f"#{i}.{j}: commit = {mc}, want {wcommitted}"


def scenario_leader_only_commits_log_from_current_term(B):
    """test_raft_paper.rs:1012-1052."""
    for i, (index, wcommit) in enumerate(T.TEST_LEADER_ONLY_COMMITS_CURRENT_TERM):
        ld = B(1, 3, [1, 2], log=[(1, 1), (2, 2), (3, 3), (3, 4)], committed=0, next_idx=3)
        ld.set_progress(1, match=4, next=5, state=REPLICATE)
        ld.step(2, index)
        assert ld.committed() == wcommit, f"#{i}: commit = {ld.committed()}, want {wcommit}"


def scenario_leader_acknowledge_commit(B):
    """test_raft_paper.rs:499-534."""
    for i, (size, acceptors, wack) in enumerate(T.TEST_LEADER_ACKNOWLEDGE_COMMIT):
        ld = B(1, 1, list(range(1, size + 1)), log=[(1, 1), (1, 2)], committed=1, next_idx=2)
        for pid in range(1, size + 1):
            ld.set_progress(pid, match=1, next=2, state=REPLICATE)
        ld.persisted(2)
        for pid in acceptors:
            ld.step(pid, 2)
        assert (ld.committed() > 1) == wack, f"#{i}: ack commit = {ld.committed() > 1}, want {wack}"


def scenario_snapshot_abort(B):
    """test_raft_snap.rs:112-131 test_snapshot_abort: snapshot (index 11, term 11) restored, leader at
    term 1 with an unpersisted noop at 12; peer 2 in Snapshot(pending 11), next 1."""
    ld = B(1, 1, [1, 2], log=[(1, 12)], committed=11, dummy=(11, 11), next_idx=12)
    ld.set_progress(1, match=11, next=12, state=REPLICATE)
    ld.set_progress(2, next=1, state=SNAPSHOT, pending_snapshot=11)
    ld.step(2, 11)
    pr = ld.progress(2)
    assert pr["pending_snapshot"] == 0 and pr["next"] == 12 and pr["state"] == PROBE


def scenario_request_snapshot(B):
    """test_raft_snap.rs:155-233 test_request_snapshot, the handle_append_response parts."""
    ld = B(1, 1, [1, 2], log=[(1, 12)], committed=11, dummy=(11, 11), next_idx=12)
    ld.set_progress(1, match=11, next=12, state=REPLICATE)
    ld.step(2, 11)  # advance matched: Probe -> Replicate
    assert ld.progress(2)["state"] == REPLICATE
    rs = ld.committed()
    # out-of-order request snapshot (index 9 < matched 11) is ignored
    out = ld.step(2, 9, reject=True, reject_hint=0, request_snapshot=rs)
    assert ld.progress(2)["state"] == REPLICATE and not out["send_append"]
    # in-order one: maybe_decr_to accepts, Replicate -> Probe, send_append (the send path then turns
    # pending_request_snapshot into become_snapshot, raft.rs:791-795 -- host side)
    out = ld.step(2, 11, reject=True, reject_hint=0, request_snapshot=rs)
    pr = ld.progress(2)
    assert out["send_append"] and pr["state"] == PROBE and pr["pending_request_snapshot"] == rs and pr["next"] == 12
    ld.set_progress(2, state=SNAPSHOT, pending_snapshot=11, paused=False)  # become_snapshot(11) by the send path
    # a repeated ack at the same index does not leave Snapshot: maybe_update returns false first (:217-224)
    ld.step(2, 11)
    pr = ld.progress(2)
    assert pr["state"] == SNAPSHOT and pr["pending_snapshot"] == 11 and pr["next"] == 12


def scenario_unconditional_next_bump(B):
    """progress.rs:145-147: next_idx is raised even when the ack is stale (SURVEY.md 8d known-answer cell)."""
    ld = B(1, 1, [1, 2, 3], log=[(1, k) for k in range(1, 10)], committed=0)
    ld.set_progress(1, match=9, next=10, state=REPLICATE)
    ld.set_progress(2, match=5, next=3, state=REPLICATE)
    out = ld.step(2, 4, commit=3)
    pr = ld.progress(2)
    assert pr["match"] == 5 and pr["next"] == 5 and not out["send_more"]
    assert pr["recent_active"] and pr["committed_index"] == 3, "stale acks still mark activity (raft.rs:1674-1677)"


def scenario_old_paused_resend_and_transfer(B):
    """raft.rs:1749-1751 (old_paused -> send_append when commit did not move) and :1764-1774
    (transferee caught up -> timeout_now); cf. test_msg_append_response_wait_reset test_raft.rs:1484-1529."""
    ld = B(1, 1, [1, 2, 3], log=[(1, k) for k in range(1, 6)], committed=5)
    for pid in (1, 2, 3):
        ld.set_progress(pid, match=5 if pid == 1 else 3, next=6 if pid == 1 else 4, state=REPLICATE)
    ld.set_progress(2, match=5)
    out = ld.step(3, 4, ins_full=True)  # commit already 5: no change; window was full -> resend
    assert out["send_append"] and out["send_more"] and out["free_to"] and not out["changed"]
    ld.set_transferee(3)
    out = ld.step(3, 5)
    assert out["timeout_now"]
    # paused probe gets resumed by an accepting ack (Probe -> Replicate)
    ld.set_progress(2, match=1, next=2, state=PROBE, paused=True)
    out = ld.step(2, 2)
    pr = ld.progress(2)
    assert pr["state"] == REPLICATE and not pr["paused"] and pr["next"] == 3 and out["send_append"]


def scenario_learners_never_count(B):
    """test_raft.rs:3891-3943 test_learner_log_replication: a learner's acks update its Progress but
    never move the commit index (tracker.rs:43-49)."""
    ld = B(1, 1, [1, 2, 3], learners=[4], log=[(1, k) for k in range(1, 6)], committed=1)
    ld.set_progress(1, match=5, next=6, state=REPLICATE)
    for pid in (2, 3, 4):
        ld.set_progress(pid, match=1, next=2, state=REPLICATE)
    out = ld.step(4, 5)
    assert ld.progress(4)["match"] == 5 and ld.committed() == 1 and not out["changed"]
    out = ld.step(2, 4)
    assert ld.committed() == 4 and out["changed"]
    assert ld.progress(1)["committed_index"] == 4, "leader's own Progress.committed_index follows (raft.rs:896-900)"


def scenario_joint_needs_both_majorities(B):
    """joint.rs:47-51 through maybe_commit: incoming {1,2,3} && outgoing {3,4,5} (cf. joint_commit.txt)."""
    ld = B(1, 1, [1, 2, 3], outgoing=[3, 4, 5], log=[(1, k) for k in range(1, 11)], committed=0)
    ld.set_progress(1, match=10, next=11, state=REPLICATE)
    for pid in (2, 3, 4, 5):
        ld.set_progress(pid, match=0, next=1, state=REPLICATE)
    ld.step(2, 8)
    assert ld.committed() == 0, "incoming {10,8,0} -> 8 but outgoing {0,0,0} -> 0"
    ld.step(4, 6)
    assert ld.committed() == 0, "outgoing {0,6,0} -> 0"
    out = ld.step(5, 7)
    assert ld.committed() == 6 and out["changed"], "outgoing {0,6,7} -> 6, incoming -> 8, min = 6"
    assert ld.mci() == (6, False)
    ld.step(3, 9)
    assert ld.committed() == 7, "incoming {10,8,9} -> 9, outgoing {9,6,7} -> 7"


def scenario_handle_heartbeat_resp(B):
    """test_raft.rs:1398-1440 test_handle_heartbeat_resp: a heartbeat response from a peer that is behind
    re-sends MsgAppend until an MsgAppResp catches it up; + test_progress_resume_by_heartbeat_resp
    (test_raft.rs:331-346) and the heartbeat commit rule min(matched, committed) (raft.rs:830-838)."""
    ld = B(1, 1, [1, 2], log=[(1, 1), (2, 2), (3, 3), (1, 4)], committed=4, next_idx=4)
    # (log terms are irrelevant here; leader at last_index 4, peer 2 just reset to Probe/match 0)
    ld.set_progress(1, match=4, next=5, state=REPLICATE)
    assert ld.heartbeat_commit(2) == 0, "MUST NOT forward the follower's commit to an unmatched index"
    out = ld.step_heartbeat_response(2)
    assert out["send_append"]
    out = ld.step_heartbeat_response(2)
    assert out["send_append"], "a second heartbeat response generates another MsgApp re-send"
    ld.step(2, 4)  # MsgAppResp catches the peer up
    assert ld.heartbeat_commit(2) == 4
    out = ld.step_heartbeat_response(2, commit=4)
    assert not out["send_append"], "once caught up, heartbeats no longer send MsgApp"
    assert ld.progress(2)["committed_index"] == 4 and ld.progress(2)["recent_active"]
    # resume by heartbeat response
    ld.set_progress(2, paused=True, state=PROBE)
    ld.step_heartbeat_response(2)
    assert not ld.progress(2)["paused"]
    # a full inflight window gets one slot freed (raft.rs:1796-1798)
    ld.set_progress(2, state=REPLICATE)
    assert ld.step_heartbeat_response(2, ins_full=True)["free_first_one"]
    assert not ld.step_heartbeat_response(2, ins_full=False)["free_first_one"]
    # a pending snapshot request also triggers send_append (raft.rs:1800)
    ld.set_progress(2, pending_request_snapshot=3)
    assert ld.step_heartbeat_response(2)["send_append"]


def scenario_commit_after_remove_node(B):
    """test_raft.rs:3291-3340 test_commit_after_remove_node: a pending entry becomes committed when a conf
    change reduces the quorum (apply_conf + post_conf_change's maybe_commit, raft.rs:2630)."""
    # leader 1 of {1,2} at term 1: noop 1, conf-change entry 2, "hello" 3 -- all persisted by the leader
    ld = B(1, 1, [1, 2], log=[(1, 1), (1, 2), (1, 3)], committed=0, next_idx=1)
    ld.set_progress(1, match=3, next=4, state=REPLICATE)
    out = ld.step(2, 2)  # node 2 acknowledges the config change, committing entries 1..2
    assert ld.committed() == 2 and out["changed"]
    ld.remove_node(2)     # applying it leaves {1}
    assert ld.maybe_commit(), "post_conf_change: the pending command can now commit"
    assert ld.committed() == 3


def scenario_fast_log_rejection(B):
    """test_raft.rs:5573-5839 test_fast_log_rejection, leader side: the follower's rejection carries
    (reject_hint, log_term); find_conflict_by_term (raft_log.rs:209-235) turns it into the next probe."""
    import json
    import os
    rows = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                                       "fast_log_rejection.json")))["rows"]
    assert len(rows) == 8
    for i, r in enumerate(rows):
        leader_log = [(t, idx) for t, idx in r["leader_log"]]
        last = len(leader_log)
        # become_candidate from term 0 -> term 1; become_leader appends its noop at last+1 (raft.rs:1163-1194)
        ld = B(1, 1, [1, 2, 3], log=leader_log + [(1, last + 1)], committed=0, next_idx=last + 1)
        ld.set_progress(1, match=last, next=last + 1, state=REPLICATE)
        ld.sent(2)  # the probe MsgAppend(index = next - 1 = last) went out: Probe is now paused
        out = ld.step(2, last, reject=True, reject_hint=r["reject_hint_index"], log_term=r["reject_hint_term"])
        assert out["send_append"], i
        nxt = ld.progress(2)["next"]
        assert nxt - 1 == r["next_append_index"], f"#{i}: next append index {nxt - 1}, want {r['next_append_index']}"
        assert ld.log_term(nxt - 1) == r["next_append_term"], f"#{i}: next append term"


def scenario_progress_committed_index(B):
    """test_raft.rs:116-299 test_progress_committed_index, the leader-side steps whose messages the test
    spells out: #3 rejections carry commit 4 and must not move Progress.committed_index; the re-sent append
    is acknowledged at 7; #4 delayed responses with a smaller commit must not lower it (progress.rs:153-157)."""
    def committed_indexes(ld):
        return tuple(ld.progress(i)["committed_index"] for i in (1, 2, 3))
    # --- #3: leader 2, term 2, log 1..7 (4 = its noop; 5,6 proposed while isolated, 7 after), commit 4
    ld = B(2, 2, [1, 2, 3], log=[(1, 1), (1, 2), (1, 3), (2, 4), (2, 5), (2, 6), (2, 7)], committed=4)
    ld.set_progress(2, match=7, next=8, state=REPLICATE, committed_index=4)
    for pid in (1, 3):  # followers acked the noop (match 4); the leader optimistically streamed up to 7
        ld.set_progress(pid, match=4, next=7, state=REPLICATE, committed_index=4)
    for pid in (1, 3):  # MsgAppendResponse index: 6 commit: 4 reject: true reject_hint: 4
        out = ld.step(pid, 6, reject=True, reject_hint=4, commit=4)
        pr = ld.progress(pid)
        assert out["send_append"] and pr["state"] == PROBE and pr["next"] == 5, pid  # MsgAppend index 4, entries 5..7
    assert committed_indexes(ld) == (4, 4, 4) and ld.committed() == 4
    for pid in (3, 1):  # the re-sent append is accepted: index 7, the follower's commit is still 4
        ld.sent(pid)
        ld.step(pid, 7, commit=4)
    assert ld.committed() == 7
    assert committed_indexes(ld) == (4, 7, 4), "the leader's own entry follows commit; followers report theirs later"
    for pid in (1, 3):  # the next round of responses carries commit 7
        ld.step(pid, 7, commit=7)
    assert committed_indexes(ld) == (7, 7, 7)
    # --- #4: leader 1, term 3, log up to 10, everything at 8; responses for 9/10 arrive out of order
    ld = B(1, 3, [1, 2, 3], log=[(1, k) for k in range(1, 8)] + [(3, 8), (3, 9), (3, 10)], committed=8)
    ld.set_progress(1, match=10, next=11, state=REPLICATE, committed_index=8)
    for pid in (2, 3):
        ld.set_progress(pid, match=8, next=11, state=REPLICATE, committed_index=8)
    for pid in (3, 2):  # m1, m2: index 10 commit 10 (the newer responses overtake)
        ld.step(pid, 10, commit=10)
    assert ld.committed() == 10 and committed_indexes(ld) == (10, 10, 10)
    for pid in (2, 3):  # the delayed ones: index 10 commit 9
        out = ld.step(pid, 10, commit=9)
        assert not out["changed"] and not out["send_more"], "stale ack: maybe_update returns false"
    assert committed_indexes(ld) == (10, 10, 10)


def scenario_send_path_update_state(B):
    """Progress::update_state through the SENT event (progress.rs:231-243; test_raft.rs:2793-2910
    test_leader_increase_next / test_send_append_for_progress_{probe,replicate,snapshot}): Replicate =>
    optimistic next = last + 1; Probe => one send, then paused; Snapshot => the reference panics (fault)."""
    ld = B(1, 1, [1, 2, 3], log=[(1, k) for k in range(1, 7)], committed=0)
    ld.set_progress(1, match=6, next=7, state=REPLICATE)
    ld.set_progress(2, match=2, next=3, state=REPLICATE)
    assert ld.sent(2) == 0
    assert ld.progress(2)["next"] == 7, "Replicate: optimistically increase next to last_index + 1"
    ld.set_progress(3, match=0, next=4, state=PROBE, paused=False)
    assert ld.sent(3) == 0
    pr = ld.progress(3)
    assert pr["paused"] and pr["next"] == 4, "Probe: pause after one message, next unchanged"
    ld.set_progress(3, state=SNAPSHOT, pending_snapshot=5)
    assert ld.sent(3) != 0, "Snapshot: update_state panics in the reference -> reported as a fault"


def scenario_leader_commit_preceding_entries(B):
    """test_raft_paper.rs:541-581 test_leader_commit_preceding_entries: committing an entry of the leader's own
    term (3) commits every preceding entry, including those created by previous leaders."""
    for i, prev in enumerate(([], [(2, 1)], [(1, 1), (2, 2)], [(1, 1)])):
        li = len(prev)
        log = list(prev) + [(3, li + 1), (3, li + 2)]  # become_leader's noop, then the proposal
        ld = B(1, 3, [1, 2, 3], log=log, committed=0, next_idx=li + 1)
        ld.set_progress(1, match=0, next=li + 3, state=REPLICATE)
        ld.persisted(li + 2)
        assert ld.committed() == 0, f"#{i}: nothing is committed by the leader alone"
        out = ld.step(2, li + 2)  # accept_and_reply of the MsgAppend
        assert out["changed"] and ld.committed() == li + 2, f"#{i}: committed = {ld.committed()}, want {li + 2}"
        ld.step(3, li + 2)
        assert ld.committed() == li + 2


def scenario_leader_increase_next(B):
    """test_raft.rs:2793-2827 test_leader_increase_next: a proposal optimistically moves next to last_index + 1 for a
    peer in Replicate (3 previous entries + noop + proposal + 1 = 6) and leaves it alone for a peer in Probe."""
    for state, next_idx, wnext in ((REPLICATE, 2, 6), (PROBE, 2, 2)):
        ld = B(1, 2, [1, 2], log=[(1, 1), (1, 2), (1, 3), (2, 4)], committed=0, next_idx=4, max_inflight=256)
        ld.set_progress(1, match=4, next=5, state=REPLICATE)
        ld.set_progress(2, match=0, next=next_idx, state=state, paused=False)
        ms = ld.propose()
        assert len(ms) == 1 and ms[0][2] == next_idx - 1 and ms[0][3] == 4, ms
        assert ld.progress(2)["next"] == wnext, f"{state}: next = {ld.progress(2)['next']}, want {wnext}"


def scenario_sending_snapshot_set_pending_snapshot(B):
    """test_raft_snap.rs:27-48 test_sending_snapshot_set_pending_snapshot + :51-65
    test_pending_snapshot_pause_replication: a reject sends next_idx below first_index, the entries are compacted
    away, so the append turns into a snapshot (decided on the device, fetched by the host, which applies
    become_snapshot(11)); replication to that peer then pauses."""
    ld = B(1, 1, [1, 2], log=[(1, 12)], committed=11, dummy=(11, 11), next_idx=12, max_inflight=256)
    ld.set_progress(1, match=12, next=13, state=REPLICATE)
    ld.set_progress(2, match=0, next=12, state=PROBE, paused=False)  # next = first_index: node 2 needs a snapshot
    ms = ld.reject(2, 11)
    assert ms == [(2, 2, 0, 0)], f"want one snapshot decision for peer 2, got {ms}"
    assert ld.progress(2)["next"] == 1 and ld.progress(2)["recent_active"]
    ld.become_snapshot(2, 11)
    pr = ld.progress(2)
    assert pr["pending_snapshot"] == 11 and pr["state"] == SNAPSHOT
    assert ld.propose() == [], "a pending snapshot pauses replication"


def scenario_request_snapshot_through_send_path(B):
    """test_raft_snap.rs:155-233 test_request_snapshot with the send decisions made by the stage: the in-order
    request flips Replicate -> Probe, the re-send becomes a snapshot decision, the host applies become_snapshot;
    acks and heartbeats do not leave Snapshot."""
    ld = B(1, 1, [1, 2], log=[(1, 12)], committed=11, dummy=(11, 11), next_idx=12, max_inflight=256)
    ld.set_progress(1, match=11, next=13, state=REPLICATE)  # snapshot at 11 persisted, the noop (12) not yet
    ld.set_progress(2, match=0, next=12, state=PROBE, paused=False)
    assert ld.ack(2, 11) == [(2, 1, 11, 1)] and ld.progress(2)["state"] == REPLICATE  # advance matched
    assert ld.reject(2, 9, request_snapshot=11) == [], "out of order request snapshot messages are ignored"
    assert ld.progress(2)["state"] == REPLICATE
    ms = ld.reject(2, 11, request_snapshot=11)
    assert ms == [(2, 2, 11, 0)], f"want the snapshot decision, got {ms}"
    ld.become_snapshot(2, 11)  # the host fetched the snapshot (index 11 = request_snapshot_idx)
    pr = ld.progress(2)
    assert (pr["state"], pr["pending_snapshot"], pr["next"]) == (SNAPSHOT, 11, 12)
    assert ld.ack(2, 11) == [], "append responses do not set the state from snapshot to probe"
    pr = ld.progress(2)
    assert (pr["state"], pr["pending_snapshot"], pr["next"]) == (SNAPSHOT, 11, 12)
    assert ld.heartbeat_response(2) == [] and ld.progress(2)["state"] == SNAPSHOT


def scenario_request_snapshot_unavailable(B):
    """test_raft.rs:4903-4965 test_request_snapshot_unavailable: while the storage cannot produce the snapshot
    (SnapshotTemporarilyUnavailable) the peer stays in Probe and every repeated request -- never stale, even though
    reject != next - 1 -- yields the snapshot decision again; once the host has the snapshot it applies
    become_snapshot."""
    ld = B(1, 1, [1, 2], log=[(1, i) for i in range(1, 15)], committed=14, next_idx=15, max_inflight=256)
    ld.set_progress(1, match=14, next=15, state=REPLICATE)
    ld.set_progress(2, match=14, next=15, state=REPLICATE, recent_active=True)
    for attempt in range(3):
        ms = ld.reject(2, 14, request_snapshot=14)
        assert ms == [(2, 2, 14, 0)], (attempt, ms)
        pr = ld.progress(2)
        assert pr["state"] == PROBE and pr["pending_request_snapshot"] == 14 and pr["next"] == 15, (attempt, pr)
        # attempts 0 and 1: the host's storage answers SnapshotTemporarilyUnavailable -> nothing is applied
    ld.become_snapshot(2, 14)
    pr = ld.progress(2)
    assert pr["state"] == SNAPSHOT and pr["pending_snapshot"] == 14


def _flow_leader(B, cap):
    """new_test_raft(1, [1, 2], ..) after become_candidate + become_leader (noop at index 1), peer 2 forced
    into Replicate (test_raft_flow_control.rs:24-31)."""
    ld = B(1, 1, [1, 2], log=[(1, 1)], committed=0, next_idx=1, max_inflight=cap)
    ld.set_progress(1, match=1, next=2, state=REPLICATE)
    ld.set_progress(2, match=0, next=1, state=REPLICATE)
    return ld


def scenario_msg_app_flow_control_full(B, cap=16):
    """test_raft_flow_control.rs:24-56 test_msg_app_flow_control_full: one MsgAppend per proposal until the
    inflight window is full, then none."""
    ld = _flow_leader(B, cap)
    for i in range(cap):
        ms = ld.propose()
        assert len(ms) == 1 and ms[0][0] == 2, f"#{i}: ms = {ms}, want 1 message"
    assert ld.ins_full(2)
    for i in range(10):
        assert ld.propose() == [], f"#{i}: want no message on a full window"


def scenario_msg_app_flow_control_move_forward(B, cap=12):
    """test_raft_flow_control.rs:63-108 test_msg_app_flow_control_move_forward: an ack of index tt frees the
    window up to tt; stale acks below it free nothing."""
    ld = _flow_leader(B, cap)
    for _ in range(cap):
        ld.propose()
    for tt in range(2, cap):  # 1 is the noop, 2 the first proposal
        ld.ack(2, tt)  # move the window forward
        ms = ld.propose()  # fill it again
        assert len(ms) == 1, f"#{tt}: ms = {ms}, want 1 message"
        assert ld.ins_full(2), f"#{tt}: the window must be full again"
        for i in range(tt):
            ld.ack(2, i)
            assert ld.ins_full(2), f"#{tt}.{i}: a stale ack must not free the window"


def scenario_msg_app_flow_control_recv_heartbeat(B, cap=8):
    """test_raft_flow_control.rs:115-177 test_msg_app_flow_control_recv_heartbeat: a heartbeat response on a
    full window frees exactly one slot."""
    ld = _flow_leader(B, cap)
    for _ in range(cap):
        ld.propose()
    for tt in range(1, 5):
        assert ld.ins_full(2), f"#{tt}: window must be full"
        for i in range(tt):  # the first response frees a slot, the others find the window not full
            ld.heartbeat_response(2)
            assert not ld.ins_full(2), f"#{tt}.{i}: want a free slot"
        ms = ld.propose()  # one slot
        assert len(ms) == 1, f"#{tt}: free slot = 0, want 1"
        for i in range(10):  # and just one slot
            assert ld.propose() == [], f"#{tt}.{i}: ms should be empty"
        ld.heartbeat_response(2)  # clear all pending messages (frees one slot and re-sends the backlog)


def scenario_send_append_for_progress(B):
    """test_raft.rs:2830-2910 test_send_append_for_progress_{probe,replicate,snapshot} at the decision level:
    Probe sends once and pauses, a heartbeat response resumes it; Replicate keeps sending optimistically;
    Snapshot sends nothing."""
    ld = B(1, 1, [1, 2], log=[(1, 1)], committed=0, next_idx=2, max_inflight=256)
    ld.set_progress(1, match=1, next=2, state=REPLICATE)
    ld.set_progress(2, match=0, next=2, state=PROBE, paused=False)
    ms = ld.propose()
    assert [m[0] for m in ms] == [2] and ld.progress(2)["paused"], "probe: the first proposal is sent, then paused"
    for _ in range(3):
        assert ld.propose() == [], "probe: paused, nothing is sent"
    pr = ld.progress(2)
    assert pr["next"] == 2 and pr["paused"]
    ms = ld.heartbeat_response(2)  # resume() + send_append: the whole backlog in one message
    assert len(ms) == 1 and ms[0][2] == 1 and ms[0][3] == 4 and ld.progress(2)["paused"]
    # replicate: every proposal goes out, next runs ahead of matched
    ld.set_progress(2, match=1, next=2, state=REPLICATE)
    for i in range(10):
        ms = ld.propose()
        assert len(ms) == 1, (i, ms)
    assert ld.progress(2)["next"] == 16, "optimistic next = last_index + 1 (1 noop + 4 + 10 proposals)"
    # snapshot: nothing is sent
    ld.set_progress(2, state=SNAPSHOT, pending_snapshot=10)
    for _ in range(3):
        assert ld.propose() == []


def scenario_progress_paused(B):
    """test_raft.rs:349-366 test_progress_paused: a fresh leader of {1, 2} (become_leader: the follower is in Probe at
    next = last_index + 1, its own noop appended) takes three MsgPropose in a row -- read_messages() holds ONE MsgAppend: the
    first proposal's bcast_append probes, Progress::update_state pauses the peer (progress.rs:238), the other two find it paused."""
    ld = B(1, 1, [1, 2], log=[(1, 1)], committed=0, next_idx=1, max_inflight=256)
    ld.set_progress(1, match=1, next=2, state=REPLICATE)
    ld.set_progress(2, match=0, next=1, state=PROBE, paused=False)
    ms = []
    for _ in range(3):
        ms += ld.propose()
    assert len(ms) == 1 and ms[0][0] == 2, ms
    assert ms[0][2] == 0 and ms[0][3] == 2, "prev_index 0, the noop and the first proposal in one message"
    pr = ld.progress(2)
    assert pr["paused"] and pr["next"] == 1


def scenario_progress_flow_control(B):
    """test_raft.rs:369-435 test_progress_flow_control: max_inflight_msgs = 3, max_size_per_msg = 2048 with
    1000-byte proposals = two entries per MsgAppend. Probe sends one message; its ack switches to Replicate and
    releases three messages of two entries; acking those releases the last two (2 + 1 entries)."""
    ld = B(1, 1, [1, 2], log=[(1, 1)], committed=0, next_idx=1, max_inflight=3, max_entries=2)
    ld.set_progress(1, match=0, next=2, state=REPLICATE)  # nothing persisted: the commit index cannot move
    ld.set_progress(2, match=0, next=1, state=PROBE, paused=False)
    ms = []
    for _ in range(10):
        ms += ld.propose()
    assert ms == [(2, 1, 0, 2)], "one MsgAppend in Probe: the noop and the first proposal"
    ms = ld.ack(2, 2)  # -> Replicate, and multiple messages at once
    assert ms == [(2, 1, 2, 2), (2, 1, 4, 2), (2, 1, 6, 2)], ms
    assert ld.ins_full(2) and ld.inflights(2) == [4, 6, 8]
    ms = ld.ack(2, 8)  # ack all three: the last two messages carry three entries
    assert ms == [(2, 1, 8, 2), (2, 1, 10, 1)], ms
    assert ld.inflights(2) == [10, 11] and ld.progress(2)["next"] == 12


def scenario_progress_flow_control_bytes(B):
    """test_raft.rs:369-435 test_progress_flow_control with its literal configuration: max_size_per_msg = 2048 BYTES
    over the real entry sizes (util::limit_size) instead of the two-entries-per-message stand-in: a proposal's entry
    {term 1, index < 128, 1000-byte data} is 2 + 2 + (1 + 2 + 1000) = 1007 bytes of protobuf, the leader's empty entry
    {term 1, index 1} is 4 -- so the first message carries the noop AND the first proposal (1011 bytes), later ones
    two proposals (2014 <= 2048 < 3021)."""
    size = lambda idx: 4 if idx == 1 else 1007
    ld = B(1, 1, [1, 2], log=[(1, 1)], committed=0, next_idx=1, max_inflight=3, max_bytes=2048, entry_bytes=size)
    ld.set_progress(1, match=0, next=2, state=REPLICATE)
    ld.set_progress(2, match=0, next=1, state=PROBE, paused=False)
    ms = []
    for _ in range(10):
        ms += ld.propose()
    assert ms == [(2, 1, 0, 2)], "one MsgAppend in Probe: the noop and the first proposal"
    ms = ld.ack(2, 2)
    assert ms == [(2, 1, 2, 2), (2, 1, 4, 2), (2, 1, 6, 2)], ms
    assert ld.ins_full(2) and ld.inflights(2) == [4, 6, 8]
    ms = ld.ack(2, 8)
    assert ms == [(2, 1, 8, 2), (2, 1, 10, 1)], ms
    assert ld.inflights(2) == [10, 11] and ld.progress(2)["next"] == 12


def scenario_msg_append_response_wait_reset(B):
    """test_raft.rs:1484-1529 test_msg_append_response_wait_reset: an ack releases a peer from the probe wait;
    a proposal is broadcast only to peers that are not waiting."""
    ld = B(1, 1, [1, 2, 3], log=[(1, 1)], committed=0, next_idx=1, max_inflight=256)
    ld.set_progress(1, match=1, next=2, state=REPLICATE)  # the noop is persisted
    for pid in (2, 3):  # bcast_append after the election: one probe message each, now waiting
        ld.set_progress(pid, match=0, next=1, state=PROBE, paused=True)
    ms = ld.ack(2, 1)  # node 2 acks the first entry, making it committed
    assert ld.committed() == 1
    assert ms == [(2, 1, 1, 0)], "the commit is broadcast; node 3 is still waiting"
    ms = ld.propose()
    ld.persisted(2)
    assert ms == [(2, 1, 1, 1)], "node 2 left the wait state due to its MsgAppResp, node 3 is still waiting"
    ms = ld.ack(3, 1)  # releases the wait: entry 2 is sent
    assert ms == [(3, 1, 1, 1)], ms


def scenario_skip_bcast_commit(B):
    """test_raw_node.rs:714-779 test_skip_bcast_commit, leader side: with Config::skip_bcast_commit the empty
    MsgAppend that only carries a new commit index is not sent -- unless a conf change is pending
    (should_bcast_commit, raft.rs:2684-2686); the switch can be flipped at run time."""
    ld = B(1, 1, [1, 2, 3], log=[(1, 1)], committed=1, next_idx=2, max_inflight=256)
    for pid in (1, 2, 3):
        ld.set_progress(pid, match=1, next=2, state=REPLICATE)
    ld.skip_bcast_commit = True
    assert ld.propose() == [(2, 1, 1, 1), (3, 1, 1, 1)]
    ld.persisted(2)
    assert ld.ack(2, 2) == [] and ld.committed() == 2, "the commit moved, nobody is told until the next message"
    assert ld.ack(3, 2) == []
    ld.skip_bcast_commit = False  # adjustable at run time
    assert ld.propose() == [(2, 1, 2, 1), (3, 1, 2, 1)]
    ld.persisted(3)
    assert ld.ack(2, 3) == [(2, 1, 3, 0), (3, 1, 3, 0)] and ld.committed() == 3, "empty appends broadcast the commit"
    ld.ack(3, 3)
    ld.skip_bcast_commit = True
    ld.set_pending_conf(True)  # when committing a conf change the leader always broadcasts the commit
    assert ld.propose() == [(2, 1, 3, 1), (3, 1, 3, 1)]
    ld.persisted(4)
    assert ld.ack(3, 4) == [(2, 1, 4, 0), (3, 1, 4, 0)] and ld.committed() == 4
    ld.set_pending_conf(False)
    assert ld.propose() == [(2, 1, 4, 1), (3, 1, 4, 1)]
    ld.persisted(5)
    assert ld.ack(2, 5) == [] and ld.committed() == 5


def scenario_progress_leader(B):
    """test_raft.rs:302-328 test_progress_leader: become_candidate + become_leader on an empty log, persist the
    no-op entry; through five proposals the leader's own Progress stays Replicate with matched = the persisted index
    and next = matched + 1."""
    ld = B(1, 0, [1, 2], log=[], committed=0, next_idx=1)
    ld.become_leader(1)
    ld.persisted(1)  # raft.persist(): for the no-op entry
    ld.set_progress(2, state=REPLICATE)
    for i in range(5):
        pr = ld.progress(1)
        assert pr["state"] == REPLICATE
        assert pr["match"] == i + 1 and pr["next"] == pr["match"] + 1, (i, pr)
        ld.append(1)         # step(MsgPropose)
        ld.persisted(i + 2)  # raft.persist()


def scenario_become_leader_resets_every_progress(B):
    """Raft::reset (raft.rs:942-971) through become_leader (:1151-1202), as test_raft.rs:116-299
    test_progress_committed_index observes it across its elections: every Progress is Progress::reset(last + 1)
    (progress.rs:82-92) -- matched 0, Probe, not paused, no pending snapshot / request, not recently active -- and keeps
    its committed_index; the leader's own keeps matched = persisted, takes committed_index = committed and is
    Replicate; the new term's range starts with the leader's empty entry, so nothing older commits by counting
    (raft_log.rs:487-499)."""
    log = [(1, 1), (1, 2), (1, 3), (2, 4), (2, 5)]
    ld = B(2, 2, [1, 2, 3], learners=[4], log=log, committed=4)
    ld.set_progress(2, match=5, next=6, state=REPLICATE, committed_index=4, recent_active=True)
    ld.set_progress(1, match=5, next=6, state=REPLICATE, committed_index=4, recent_active=True)
    ld.set_progress(3, match=3, next=6, state=SNAPSHOT, pending_snapshot=5, paused=True, committed_index=3)
    ld.set_progress(4, match=2, next=3, state=PROBE, paused=True, pending_request_snapshot=7, committed_index=2)
    ld.set_transferee(3)
    ld.become_leader(5)
    for pid, cidx in ((1, 4), (3, 3), (4, 2)):
        assert ld.progress(pid) == {"match": 0, "next": 6, "state": PROBE, "paused": False, "pending_snapshot": 0,
                                    "pending_request_snapshot": 0, "recent_active": False, "committed_index": cidx}, pid
    me = ld.progress(2)
    assert (me["match"], me["next"], me["state"], me["paused"], me["committed_index"]) == (5, 6, REPLICATE, False, 4)
    assert ld.committed() == 4
    ld.persisted(6)  # the leader's own empty entry at index 6, term 5
    # followers at 5 hold a quorum of the OLD term's entry 5: not committed by counting replicas (raft.rs:893-904)
    for pid in (1, 3):
        out = ld.step(pid, 5)
        assert not out["changed"] and not out["timeout_now"], "the aborted transfer sends no MsgTimeoutNow"
        assert ld.progress(pid)["state"] == REPLICATE
    assert ld.committed() == 4
    assert ld.step(1, 6)["changed"] and ld.committed() == 6, "an entry of the new term commits everything before it"


def scenario_leader_start_replication(B):
    """test_raft_paper.rs:425-465 test_leader_start_replication: after become_leader the no-op goes to both
    followers (commit_noop_entry :24-46: one MsgAppend each, one empty entry, prev index 0); once it is committed a
    proposal is sent to both with index = li, one entry, commit = li."""
    ld = B(1, 0, [1, 2, 3], log=[], committed=0, next_idx=1, max_inflight=256)
    assert ld.become_leader_and_bcast(1) == [(2, 1, 0, 1), (3, 1, 0, 1)]
    ld.ack(2, 1)
    ld.ack(3, 1)
    ld.persisted(1)
    li = 1
    assert ld.committed() == li
    assert ld.propose() == [(2, 1, li, 1), (3, 1, li, 1)]
    assert ld.committed() == li


def scenario_recv_msg_unreachable(B):
    """test_raft.rs:2913-2933 test_recv_msg_unreachable: three previous entries at term 1, become_leader's empty entry at 4;
    node 2 matched 3, Replicate, optimistic_update(5); MsgUnreachable puts it back to Probe with next = matched + 1.
    A second report finds a Probe peer and changes nothing; a peer in Snapshot is left alone (raft.rs:1947-1949)."""
    ld = B(1, 1, [1, 2, 3], log=[(1, 1), (1, 2), (1, 3), (1, 4)], committed=0, next_idx=4)
    ld.set_progress(1, match=4, next=5, state=REPLICATE)
    ld.set_progress(2, match=3, next=6, state=REPLICATE)
    ld.set_progress(3, match=0, next=1, state=SNAPSHOT, pending_snapshot=4)
    ld.report_unreachable(2)
    pr = ld.progress(2)
    assert pr["state"] == PROBE and pr["match"] + 1 == pr["next"] == 4 and not pr["paused"]
    ld.set_progress(2, paused=True)  # a Probe peer that was sent to
    ld.report_unreachable(2)
    assert ld.progress(2) == dict(pr, paused=True), "only a Replicate peer reacts"
    before = ld.progress(3)
    ld.report_unreachable(3)
    assert ld.progress(3) == before and before["state"] == SNAPSHOT
    ld.report_unreachable(9)  # "no progress available": ignored


def scenario_snapshot_failure_and_succeed(B):
    """test_raft_snap.rs:68-87 test_snapshot_failure and :89-109 test_snapshot_succeed: snapshot (index 11, term 11)
    restored, peer 2 at next 1 in Snapshot(pending 11). MsgSnapStatus{reject} -> pending_snapshot 0, next 1, paused;
    MsgSnapStatus{ok} -> pending_snapshot 0, next 12, paused. Both clear pending_request_snapshot (raft.rs:1928); a peer
    that is not in Snapshot ignores the message (:1903-1905)."""
    for failure, wnext in ((True, 1), (False, 12)):
        ld = B(1, 1, [1, 2], log=[(1, 12)], committed=11, dummy=(11, 11), next_idx=12)
        ld.set_progress(1, match=11, next=12, state=REPLICATE)
        ld.set_progress(2, next=1, state=SNAPSHOT, pending_snapshot=11, pending_request_snapshot=11)
        ld.report_snapshot(2, failure)
        pr = ld.progress(2)
        assert pr["pending_snapshot"] == 0 and pr["next"] == wnext and pr["paused"], (failure, pr)
        assert pr["state"] == PROBE and pr["pending_request_snapshot"] == 0, (failure, pr)
        ld.report_snapshot(2, not failure)  # no longer in Snapshot
        assert ld.progress(2) == pr
        me = ld.progress(1)
        ld.report_snapshot(1, failure)  # the leader's own Progress is Replicate
        assert ld.progress(1) == me


def scenario_unreachable_resets_the_window(B, cap=4):
    """handle_unreachable's become_probe is Progress::reset_state: ins.reset() (progress.rs:75-80). A Replicate peer with
    messages in flight loses them; the next proposal is sent from matched + 1 again, once (Probe pauses)."""
    ld = _flow_leader(B, cap)
    for _ in range(3):
        assert len(ld.propose()) == 1
    assert ld.inflights(2) == [2, 3, 4] and ld.progress(2)["next"] == 5
    ld.report_unreachable(2)
    pr = ld.progress(2)
    assert pr["state"] == PROBE and pr["next"] == 1 and ld.inflights(2) == [] and not ld.ins_full(2)
    ms = ld.propose()
    assert ms == [(2, 1, 0, 5)], ms  # one MsgAppend with everything from index 1 (no limit), then paused
    assert ld.propose() == []


FLOW = [scenario_unreachable_resets_the_window, scenario_leader_start_replication, scenario_request_snapshot_unavailable, scenario_request_snapshot_through_send_path, scenario_sending_snapshot_set_pending_snapshot, scenario_leader_increase_next, scenario_skip_bcast_commit, scenario_progress_flow_control, scenario_progress_flow_control_bytes, scenario_msg_append_response_wait_reset, scenario_msg_app_flow_control_full, scenario_msg_app_flow_control_move_forward,
        scenario_msg_app_flow_control_recv_heartbeat, scenario_send_append_for_progress, scenario_progress_paused]

ALL = [scenario_test_commit, scenario_test_group_commit, scenario_test_group_commit_consistent,
       scenario_test_leader_append_response, scenario_leader_only_commits_log_from_current_term,
       scenario_leader_acknowledge_commit, scenario_snapshot_abort, scenario_request_snapshot,
       scenario_unconditional_next_bump, scenario_old_paused_resend_and_transfer,
       scenario_learners_never_count, scenario_joint_needs_both_majorities, scenario_handle_heartbeat_resp,
       scenario_commit_after_remove_node, scenario_fast_log_rejection, scenario_progress_committed_index,
       scenario_send_path_update_state, scenario_leader_commit_preceding_entries, scenario_progress_leader,
       scenario_become_leader_resets_every_progress, scenario_recv_msg_unreachable, scenario_snapshot_failure_and_succeed]
