#!/usr/bin/env python3
"""Extract the reference's data-driven quorum golden vectors into JSON.

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


def main():
    extract_fast_log_rejection()
    out = {}
    for name in FILES:
        out[name] = parse_file(os.path.join(REF, name))
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "quorum_vectors.json")
    with open(dst, "w", encoding="utf-8") as f:
        json.dump(out, f, indent=1, ensure_ascii=False, sort_keys=True)
    print({k: len(v) for k, v in out.items()}, "->", dst)


if __name__ == "__main__":
    sys.exit(main())
