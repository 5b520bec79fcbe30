"""Pin the CPU oracle to the reference's own golden vectors (CPU only).

Sources (file:line relative to /root/reference):
  * src/quorum/testdata/*.txt  -> tests/golden/quorum_vectors.json (tests/golden/make_golden.py),
    harness semantics of src/quorum/datadriven_test.rs:5-306 restated in `build_case`;
  * table-driven unit tests of src/tracker/progress.rs:246-413, src/tracker/inflights.rs:127-256,
    src/raft_log.rs:858-918,1498-1522 transcribed in tests/golden/reference_tables.py.
"""
import itertools
import json
import os
import random

import pytest

import oracle_lib as O
from golden import reference_tables as T

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "quorum_vectors.json"), encoding="utf-8") as f:
    VEC = json.load(f)


def parse_result(s):
    s = s.strip()
    if s.endswith("∞"):
        return O.U64_MAX
    return int(s)


def build_case(args, key="idx"):
    """datadriven_test.rs:99-127: ids get values in (cfg, cfgj) order without repetition; `_` = none."""
    ids = [int(x) for x in args.get("cfg", [])]
    joint = "cfgj" in args
    idsj = [] if (not joint or args["cfgj"] == ["zero"]) else [int(x) for x in args["cfgj"]]
    vals = args.get(key, [])
    gids = args.get("gid", [])
    order = []
    for i in ids + idsj:
        if i not in order:
            order.append(i)
    assert len(order) == len(vals), "mismatched input for voters"
    if gids:
        assert len(gids) == len(order)
    look = {}
    for pos, i in enumerate(order):
        v = vals[pos]
        g = int(gids[pos]) if gids and gids[pos] != "_" else 0
        if key == "idx":
            if v != "_":
                look[i] = (int(v), g)  # entries with index 0 are dropped: "no entry" (:121-126)
        else:
            look[i] = {"y": 2, "n": 1, "_": 0}[v]
    return ids, idsj, joint, look


def majority_ci(ids, look, gc=False):
    return O.committed_index([look.get(i, (0, 0)) for i in ids], gc)


def joint_ci(ids, idsj, look, gc=False):
    a, fa = majority_ci(ids, look, gc)
    b, fb = majority_ci(idsj, look, gc)
    return min(a, b), fa and fb


@pytest.mark.parametrize("fname", ["majority_commit.txt", "joint_commit.txt"])
def test_committed_golden(fname):
    n = 0
    for case in VEC[fname]:
        assert case["cmd"] == "committed"
        ids, idsj, joint, look = build_case(case["args"])
        want = parse_result(case["result"])
        if joint:
            got = joint_ci(ids, idsj, look)[0]
            assert joint_ci(idsj, ids, look)[0] == got, "symmetry (datadriven_test.rs:176-181)"
        else:
            got = majority_ci(ids, look)[0]
            assert joint_ci(ids, [], look)[0] == got, "zero-joint (datadriven_test.rs:186-192)"
            assert joint_ci(ids, ids, look)[0] == got, "self-joint (:194-199)"
            for i in ids:  # overlaying (:201-245)
                if i in look and got > look[i][0]:
                    for lower in (look[i][0] - 1, 0):
                        l2 = dict(look)
                        l2[i] = (lower, look[i][1])
                        assert majority_ci(ids, l2)[0] == got, f"overlay {i}->{lower}"
        assert got == want, f"{fname}:{case['line']}: got {got}, want {want}"
        # iteration-order independence (the reference iterates an FxHashSet)
        if len(ids) <= 6:
            for perm in itertools.permutations(ids):
                assert majority_ci(list(perm), look)[0] == majority_ci(ids, look)[0]
        n += 1
    assert n == {"majority_commit.txt": 16, "joint_commit.txt": 50}[fname]


def test_group_committed_golden():
    cases = VEC["joint_group_commit.txt"]
    assert len(cases) == 14
    for case in cases:
        assert case["cmd"] == "group_committed"
        ids, idsj, joint, look = build_case(case["args"])
        assert joint
        want = parse_result(case["result"])
        got = joint_ci(ids, idsj, look, gc=True)
        assert got[0] == want, f"joint_group_commit.txt:{case['line']}: got {got}, want {want}"
        assert joint_ci(idsj, ids, look, gc=True) == got
        # every voter iteration order gives the same answer
        for pa in itertools.permutations(ids):
            for pb in (itertools.permutations(idsj) if len(idsj) <= 4 else [tuple(idsj)]):
                assert joint_ci(list(pa), list(pb), look, gc=True) == got


def test_group_commit_is_order_independent_randomised():
    rnd = random.Random(7)
    for _ in range(3000):
        n = rnd.randint(1, 7)
        vals = [(rnd.randint(0, 4), rnd.randint(0, 3)) for _ in range(n)]
        base = O.committed_index(vals, True)
        perms = list(itertools.permutations(vals)) if n <= 5 else [tuple(rnd.sample(vals, n)) for _ in range(60)]
        for p in perms:
            assert O.committed_index(list(p), True) == base, (vals, p)


@pytest.mark.parametrize("fname", ["majority_vote.txt", "joint_vote.txt"])
def test_vote_golden(fname):
    import ctypes as C
    names = {"VotePending": O.VOTE_PENDING, "VoteLost": O.VOTE_LOST, "VoteWon": O.VOTE_WON}

    def maj(ids, look):
        arr = (C.c_uint8 * max(1, len(ids)))(*[look.get(i, 0) for i in ids])
        return O.lib().ro_majority_vote_result(arr, len(ids))

    n = 0
    for case in VEC[fname]:
        assert case["cmd"] == "vote"
        ids, idsj, joint, look = build_case(case["args"], key="votes")
        if joint:
            got = O.lib().ro_joint_vote_result(maj(ids, look), maj(idsj, look))
            assert got == O.lib().ro_joint_vote_result(maj(idsj, look), maj(ids, look))
        else:
            got = maj(ids, look)
        assert got == names[case["result"]], f"{fname}:{case['line']}"
        n += 1
    assert n == {"majority_vote.txt": 22, "joint_vote.txt": 39}[fname]


# ---- src/tracker/progress.rs tables ------------------------------------------------------------
def new_progress(state, matched, next_idx, pending_snapshot, ins_size):
    p = O.Progress()
    O.lib().ro_progress_new(p, next_idx, ins_size)
    p.state, p.matched, p.pending_snapshot = state, matched, pending_snapshot
    return p


def test_progress_is_paused():
    for i, (state, paused, want) in enumerate(T.PROGRESS_IS_PAUSED):
        p = new_progress(state, 0, 0, 0, 256)
        p.paused = paused
        assert O.lib().ro_progress_is_paused(p) == want, i


def test_progress_resume():
    L = O.lib()
    p = O.Progress()
    L.ro_progress_new(p, 2, 256)
    p.paused = True
    L.ro_progress_maybe_decr_to(p, 1, 1, 0)
    assert not p.paused
    p.paused = True
    L.ro_progress_maybe_update(p, 2)
    assert not p.paused


def test_progress_become_probe():
    for i, ((state, matched, nxt, pend, ins), wnext) in enumerate(T.PROGRESS_BECOME_PROBE):
        p = new_progress(state, matched, nxt, pend, ins)
        O.lib().ro_progress_become_probe(p)
        assert (p.state, p.matched, p.next_idx) == (O.PROBE, matched, wnext), i


def test_progress_become_replicate_and_snapshot():
    L = O.lib()
    p = new_progress(O.PROBE, 1, 5, 0, 256)
    L.ro_progress_become_replicate(p)
    assert (p.state, p.matched, p.next_idx) == (O.REPLICATE, 1, 2)
    p = new_progress(O.PROBE, 1, 5, 0, 256)
    L.ro_progress_become_snapshot(p, 10)
    assert (p.state, p.matched, p.pending_snapshot) == (O.SNAPSHOT, 1, 10)


def test_progress_update():
    for i, (update, wm, wn, wok) in enumerate(T.PROGRESS_UPDATE):
        p = O.Progress()
        O.lib().ro_progress_new(p, T.PROGRESS_UPDATE_PREV[1], 256)
        p.matched = T.PROGRESS_UPDATE_PREV[0]
        assert O.lib().ro_progress_maybe_update(p, update) == wok, i
        assert (p.matched, p.next_idx) == (wm, wn), i


def test_progress_maybe_decr():
    for i, (state, m, n, rejected, last, w, wn) in enumerate(T.PROGRESS_MAYBE_DECR):
        p = new_progress(state, m, n, 0, 0)
        assert O.lib().ro_progress_maybe_decr_to(p, rejected, last, 0) == w, i
        assert (p.matched, p.next_idx) == (m, wn), i


# ---- src/tracker/inflights.rs tables -------------------------------------------------------------
def ins_view(ins):
    return ins.start, ins.count, [ins.buffer[i] for i in range(ins.len)]


def test_inflights_tables():
    L = O.lib()
    ins = O.Inflights()
    L.ro_ins_init(ins, 10)
    for i in range(5):
        L.ro_ins_add(ins, i)
    assert ins_view(ins) == (0, 5, [0, 1, 2, 3, 4])
    for i in range(5, 10):
        L.ro_ins_add(ins, i)
    assert ins_view(ins) == (0, 10, list(range(10)))
    assert L.ro_ins_add(ins, 11) == -1, "cannot add into a full inflights (panics in the reference)"
    ins2 = O.Inflights()
    L.ro_ins_init(ins2, 10)
    ins2.start, ins2.len = 5, 5  # inflight2.start = 5; buffer.extend_from_slice(&[0; 5])
    for i in range(5):
        L.ro_ins_add(ins2, i)
    assert ins_view(ins2) == (5, 5, [0, 0, 0, 0, 0, 0, 1, 2, 3, 4])
    for i in range(5, 10):
        L.ro_ins_add(ins2, i)
    assert ins_view(ins2) == (5, 10, [5, 6, 7, 8, 9, 0, 1, 2, 3, 4])
    # test_inflight_free_to (inflights.rs:186-236)
    ins = O.Inflights()
    L.ro_ins_init(ins, 10)
    for i in range(10):
        L.ro_ins_add(ins, i)
    L.ro_ins_free_to(ins, 4)
    assert ins_view(ins) == (5, 5, list(range(10)))
    L.ro_ins_free_to(ins, 8)
    assert ins_view(ins) == (9, 1, list(range(10)))
    for i in range(10, 15):
        L.ro_ins_add(ins, i)
    L.ro_ins_free_to(ins, 12)
    assert ins_view(ins) == (3, 2, [10, 11, 12, 13, 14, 5, 6, 7, 8, 9])
    L.ro_ins_free_to(ins, 14)
    assert ins_view(ins) == (5, 0, [10, 11, 12, 13, 14, 5, 6, 7, 8, 9])
    # test_inflight_free_first_one (:238-255)
    ins = O.Inflights()
    L.ro_ins_init(ins, 10)
    for i in range(10):
        L.ro_ins_add(ins, i)
    L.ro_ins_free_first_one(ins)
    assert ins_view(ins) == (1, 9, list(range(10)))


# ---- src/raft_log.rs tables ---------------------------------------------------------------------
def test_log_term_tables():
    L = O.lib()
    cl = O.Cluster(1).config(0, 1, 1, [1])
    offset, num = 100, 100  # test_term (raft_log.rs:890-918): snapshot (100, term 1), entries (100+i, term i)
    cl.set_log(0, [(i, offset + i) for i in range(1, num)], dummy=(offset, 1))
    for idx, want in [(offset - 1, 0), (offset, 1), (offset + num // 2, num // 2), (offset + num - 1, num - 1),
                      (offset + num, 0)]:
        assert L.ro_log_term(cl.h, 0, idx) == want, idx
    # test_term_with_unstable_snapshot (:858-887): only the unstable snapshot index answers
    cl.set_log(0, [], dummy=(10069, 1))
    for idx, want in [(10064, 0), (10065, 0), (10068, 0), (10069, 1)]:
        assert L.ro_log_term(cl.h, 0, idx) == want, idx


def test_commit_to_table():
    L = O.lib()
    for commit, wcommit, wpanic in T.COMMIT_TO:  # raft_log.rs:1498-1522
        cl = O.Cluster(1).config(0, 1, 3, [1])
        cl.set_log(0, [(1, 1), (2, 2), (3, 3)], committed=2)
        rc = L.ro_log_commit_to(cl.h, 0, commit)
        assert (rc != 0) == wpanic
        if not wpanic:
            assert cl.committed(0) == wcommit


def test_find_conflict_by_term_example():
    # the worked example in the reference's comments (src/raft.rs:1566-1583): leader terms
    # idx 1..9 = 1 3 3 3 5 5 5 5 5; rejection hint (index 6, term 2) => probe at index 1
    L = O.lib()
    cl = O.Cluster(1).config(0, 1, 5, [1])
    cl.set_log(0, [(1, 1), (3, 2), (3, 3), (3, 4), (5, 5), (5, 6), (5, 7), (5, 8), (5, 9)])
    assert L.ro_log_find_conflict_by_term(cl.h, 0, 6, 2) == 1
    assert L.ro_log_find_conflict_by_term(cl.h, 0, 9, 5) == 9
    assert L.ro_log_find_conflict_by_term(cl.h, 0, 12, 5) == 12  # out of range: returned as is (:214-223)


def test_committed_golden_files_are_what_the_extractor_produces():
    """tests/golden/*.json are mechanical extracts of the reference's own vectors: where the reference tree is present
    (the build container) re-extract and compare byte for byte; elsewhere (the GPU box) there is nothing to check."""
    import importlib.util
    if not os.path.isdir("/root/reference/src/quorum/testdata"):
        pytest.skip("reference tree not present")
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    want = {name: mg.extract_table(path, fn, consts) for name, path, fn, consts in mg.TABLES}
    with open(os.path.join(HERE, "golden", "reference_tables.json"), encoding="utf-8") as f:
        assert json.load(f) == want
    quorum = {name: mg.parse_file(os.path.join(mg.REF, name)) for name in mg.FILES}
    assert quorum == VEC
    assert sum(len(v) for v in VEC.values()) == 141 and sum(len(v["rows"]) for v in want.values()) == 82


def test_update_state_on_a_full_window_panics_like_the_reference():
    """Progress::update_state in Replicate calls ins.add(last) (progress.rs:231-243), which panics on a full window
    ("cannot add into a full inflights", inflights.rs:66-68): the restatement reports it instead of skipping the add."""
    import ctypes as C
    L = O.lib()
    p = O.Progress()
    L.ro_progress_new(C.byref(p), 1, 2)
    L.ro_progress_become_replicate(C.byref(p))
    assert L.ro_progress_update_state(C.byref(p), 5) == 0 and L.ro_progress_update_state(C.byref(p), 6) == 0
    assert L.ro_ins_full(C.byref(p.ins))
    assert L.ro_progress_update_state(C.byref(p), 7) == -1 and p.next_idx == 8, "the optimistic next is applied first"
    assert L.ro_ins_add(C.byref(p.ins), 9) == -1
    L.ro_progress_destroy(C.byref(p))
