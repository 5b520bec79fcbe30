#!/usr/bin/env python3
"""Extract the reference's golden vectors into JSON: the data-driven quorum files, test_fast_log_rejection, and the
rows of the table-driven unit tests (reference_tables.json; tests/golden/reference_tables.py shapes them).

Run in the build container (where /root/reference exists):

    python tests/golden/make_golden.py

Reads  /root/reference/src/quorum/testdata/{majority_commit,joint_commit,
joint_group_commit,majority_vote,joint_vote}.txt  (grammar:
/root/reference/datadriven/src/test_data_reader.rs:37-206 and
line_sparser.rs:50-74: `#` comments, a directive line `cmd key=(v1,v2) key=v`,
a `----` separator, then the expected output up to the first blank line) and
writes tests/golden/quorum_vectors.json. Only the *result line* of each
expected block is kept (the last line: the committed index, or the vote
result); the ASCII-art `describe` output above it is presentation, not
semantics, and is not reproduced.

The harness semantics restated by the consumers of this file are those of
/root/reference/src/quorum/datadriven_test.rs:5-306 (ids are assigned to the
idx/gid/votes lists in (cfg, cfgj) order without repetition; `_` = no entry;
`cfgj=zero` = joint with an empty outgoing config).
"""
import json
import os
import re
import sys

REF = "/root/reference/src/quorum/testdata"
FILES = ["majority_commit.txt", "joint_commit.txt", "joint_group_commit.txt",
         "majority_vote.txt", "joint_vote.txt"]
ARG_RE = re.compile(r"(\w+)=(\([^)]*\)|\S+)")


def parse_file(path):
    cases = []
    with open(path, encoding="utf-8") as f:
        lines = f.read().split("\n")
    i = 0
    while i < len(lines):
        line = lines[i].strip()
        if not line or line.startswith("#"):
            i += 1
            continue
        lineno = i + 1
        parts = line.split(None, 1)
        cmd = parts[0]
        args = {}
        if len(parts) > 1:
            for key, val in ARG_RE.findall(parts[1]):
                if val.startswith("("):
                    vals = [v.strip() for v in val[1:-1].split(",") if v.strip()]
                else:
                    vals = [val]
                args.setdefault(key, []).extend(vals)
        i += 1
        assert lines[i].strip() == "----", (path, i)
        i += 1
        expected = []
        while i < len(lines) and lines[i].strip() != "":
            expected.append(lines[i])
            i += 1
        cases.append({"line": lineno, "cmd": cmd, "args": args, "result": expected[-1].strip(),
                      "n_expected_lines": len(expected)})
    return cases


def extract_fast_log_rejection():
    """harness/tests/integration_cases/test_raft.rs:5573-5775 test_fast_log_rejection: 8 rows of
    (leader_log, follower_log, reject_hint_term, reject_hint_index, next_append_term, next_append_index);
    logs are lists of empty_entry(term, index). Extracted as data (no test code is copied)."""
    path = "/root/reference/harness/tests/integration_cases/test_raft.rs"
    src = open(path, encoding="utf-8").read()
    a = src.index("fn test_fast_log_rejection()")
    b = src.index("for (", a)
    body = src[a:b]
    rows = []
    # each row: "(" vec![..], vec![..], n, n, n, n ")"
    for m in re.finditer(r"\(\s*vec!\[(.*?)\],\s*vec!\[(.*?)\],\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+),\s*\)", body, flags=re.S):
        logs = []
        for k in (1, 2):
            logs.append([[int(t), int(i)] for t, i in re.findall(r"empty_entry\((\d+),\s*(\d+)\)", m.group(k))])
        rows.append({"leader_log": logs[0], "follower_log": logs[1], "reject_hint_term": int(m.group(3)),
                     "reject_hint_index": int(m.group(4)), "next_append_term": int(m.group(5)),
                     "next_append_index": int(m.group(6))})
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fast_log_rejection.json")
    with open(dst, "w", encoding="utf-8") as f:
        json.dump({"source": "harness/tests/integration_cases/test_raft.rs:5573-5775", "rows": rows}, f, indent=1)
    print(len(rows), "rows ->", dst)


# ---------------------------------------------------------------------------------------------------------------
# Table-driven unit tests of the reference: the rows of `let (mut) tests = vec![ ... ];` as data.
# A row is a Rust tuple literal; the few constructors the tables use are mapped to plain values:
#   vec![..] -> list, empty_entry(t, i) / new_entry(t, i) -> [t, i], new_progress(state, matched, next,
#   pending_snapshot, ins_size) -> [..], map!(k => v, ..) -> {k: v}, ProgressState::X / StateRole::X -> "X",
#   Some(x) -> x, None -> null, 3u64 -> 3, and the `let` constants a table refers to (prev_m, prev_n, matched).
# Only the rows are read -- no test CODE is copied.
# ---------------------------------------------------------------------------------------------------------------
TABLES = [
    # (name, file relative to /root/reference, test fn, constants the rows refer to)
    ("PROGRESS_IS_PAUSED", "src/tracker/progress.rs", "test_progress_is_paused", {}),
    ("PROGRESS_BECOME_PROBE", "src/tracker/progress.rs", "test_progress_become_probe", {"matched": None}),
    ("PROGRESS_UPDATE", "src/tracker/progress.rs", "test_progress_update", {"prev_m": None, "prev_n": None}),
    ("PROGRESS_MAYBE_DECR", "src/tracker/progress.rs", "test_progress_maybe_decr", {}),
    ("COMMIT_TO", "src/raft_log.rs", "test_commit_to", {}),
    ("TEST_COMMIT", "harness/tests/integration_cases/test_raft.rs", "test_commit", {}),
    ("TEST_GROUP_COMMIT", "harness/tests/integration_cases/test_raft.rs", "test_group_commit", {}),
    ("TEST_GROUP_COMMIT_CONSISTENT", "harness/tests/integration_cases/test_raft.rs", "test_group_commit_consistent", {}),
    ("TEST_LEADER_APPEND_RESPONSE", "harness/tests/integration_cases/test_raft.rs", "test_leader_append_response", {}),
    ("TEST_LEADER_ONLY_COMMITS_CURRENT_TERM", "harness/tests/integration_cases/test_raft_paper.rs",
     "test_leader_only_commits_log_from_current_term", {}),
    ("TEST_LEADER_ACKNOWLEDGE_COMMIT", "harness/tests/integration_cases/test_raft_paper.rs",
     "test_leader_acknowledge_commit", {}),
]


def _matching(src, start, open_ch, close_ch):
    depth = 0
    for i in range(start, len(src)):
        if src[i] == open_ch:
            depth += 1
        elif src[i] == close_ch:
            depth -= 1
            if depth == 0:
                return i
    raise ValueError("unbalanced")


def extract_table(path, fn, consts):
    src = open(os.path.join("/root/reference", path), encoding="utf-8").read()
    a = src.index("fn " + fn + "()")
    line_a = src.count("\n", 0, a) + 1
    m = re.compile(r"let\s+(?:mut\s+)?tests\s*=\s*vec!\[").search(src, a)
    lb = m.end() - 1
    rb = _matching(src, lb, "[", "]")
    line_b = src.count("\n", 0, rb) + 1
    # constants defined between the fn header and the table: `let matched = 1u64;` / `let (a, b) = (3u64, 5u64);`
    env = {}
    head = src[a:lb]
    for name in consts:
        m1 = re.search(r"let\s+" + name + r"\s*=\s*(\d+)", head)
        if m1:
            env[name] = int(m1.group(1))
    m2 = re.search(r"let\s+\(([\w\s,]+)\)\s*=\s*\(([^)]*)\)", head)
    if m2:
        for k, v in zip([x.strip() for x in m2.group(1).split(",")], m2.group(2).split(",")):
            env[k] = int(re.match(r"\s*(\d+)", v).group(1))
    body = re.sub(r"//[^\n]*", "", src[lb:rb + 1])
    body = re.sub(r"(\d+)(?:u64|usize|u32|i64)", r"\1", body)
    body = body.replace("vec![", "[")
    body = re.sub(r"\b(?:empty_entry|new_entry|new_progress)\(", "L(", body)
    body = re.sub(r"\bmap!\(", "M(", body).replace("=>", ",")
    body = re.sub(r"\b(?:ProgressState|StateRole)::(\w+)", r"'\1'", body)
    body = re.sub(r"\bSome\(", "S(", body)
    body = re.sub(r"\btrue\b", "True", body)
    body = re.sub(r"\bfalse\b", "False", body)

    def M(*kv):
        return {str(kv[i]): kv[i + 1] for i in range(0, len(kv), 2)}

    scope = {"L": lambda *x: list(x), "M": M, "S": lambda x: x, "None": None, "True": True, "False": False, "__builtins__": {}}
    scope.update(env)
    rows = eval(body, scope)  # noqa: S307 -- a literal of numbers, strings and the four constructors above

    def plain(x):
        if isinstance(x, (tuple, list)):
            return [plain(y) for y in x]
        return x
    return {"source": f"{path}:{line_a}-{line_b} {fn}", "constants": env, "rows": [plain(r) for r in rows]}


def extract_tables():
    out = {name: extract_table(path, fn, consts) for name, path, fn, consts in TABLES}
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_tables.json")
    with open(dst, "w", encoding="utf-8") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print({k: len(v["rows"]) for k, v in out.items()}, "->", dst)


def main():
    extract_fast_log_rejection()
    extract_tables()
    out = {}
    for name in FILES:
        out[name] = parse_file(os.path.join(REF, name))
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "quorum_vectors.json")
    with open(dst, "w", encoding="utf-8") as f:
        json.dump(out, f, indent=1, ensure_ascii=False, sort_keys=True)
    print({k: len(v) for k, v in out.items()}, "->", dst)


if __name__ == "__main__":
    sys.exit(main())
