#!/usr/bin/env python3
"""Golden vectors for the wire format of the path's OUTPUT: the eraftpb::Message a leader sends (MsgAppend with its
entries, MsgSnapshot, MsgHeartbeat, MsgTimeoutNow), as bytes.

Run in the build container (where /root/reference exists):

    python tests/golden/make_eraftpb_outgoing.py          # writes tests/golden/eraftpb_outgoing.json

Like make_eraftpb_vectors.py (whose .proto parser and descriptor builder it imports): the message types come out of
/root/reference/proto/proto/eraftpb.proto and the protobuf RUNTIME serialises seeded random messages, so nothing about the
format is typed in here. Every vector carries the complete content (`message`: scalar fields, `entries` with hex payloads,
the serialised `snapshot`, `context`) and what the runtime made of it (`hex`, `entry_sizes` = Entry.ByteSize() = rust-protobuf's
Entry::compute_size()). `limit_size` cases restate util::limit_size (src/util.rs:52-76) literally over those sizes.
Consumers: tests/test_wire_format.py (rg_encode_message / rg_entry_size / rg_limit_size).
"""
import json
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_eraftpb_vectors as M  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "eraftpb_outgoing.json")
SCALARS = ("to", "from", "term", "log_term", "index", "commit", "commit_term", "reject_hint", "request_snapshot", "priority")


def limit_size(sizes, max_size):
    """util::limit_size (src/util.rs:52-76), literally, over Entry::compute_size() values; None = NO_LIMIT."""
    if len(sizes) <= 1 or max_size is None:
        return len(sizes)
    size, n = 0, 0
    for s in sizes:
        if size == 0:
            size += s
            n += 1
            continue
        size += s
        if size > max_size:
            break
        n += 1
    return n


def rand_bytes(rng, hi):
    return bytes(rng.randrange(256) for _ in range(rng.randrange(0, hi)))


def rand_entry(rng, Entry, index, term):
    e = Entry()
    k = rng.random()
    if k < 0.12:
        return e, {}  # Entry::default(): 0 bytes on the wire, the case limit_size special-cases
    e.entry_type = rng.choice([0, 0, 0, 1, 2])
    e.term, e.index = term, index
    if rng.random() < 0.85:
        e.data = rand_bytes(rng, 300 if rng.random() < 0.1 else 40)
    if rng.random() < 0.2:
        e.context = rand_bytes(rng, 12)
    e.sync_log = rng.random() < 0.15
    return e, {"entry_type": int(e.entry_type), "term": int(e.term), "index": int(e.index), "data": e.data.hex(),
               "context": e.context.hex(), "sync_log": bool(e.sync_log)}


def build(cls, enums):
    Message, Entry, Snapshot = cls["Message"], cls["Entry"], cls["Snapshot"]
    types = dict(enums["MessageType"])
    rng = random.Random(0x0E7AF2)
    vectors, limits = [], []
    kinds = ["MsgAppend"] * 8 + ["MsgHeartbeat"] * 2 + ["MsgSnapshot"] * 2 + ["MsgTimeoutNow"] + sorted(types)
    for i in range(220):
        tname = kinds[i % len(kinds)]
        m = Message()
        m.msg_type = types[tname]
        fields = {"msg_type": int(m.msg_type)}
        for f in SCALARS:
            if rng.random() < 0.75:
                setattr(m, f, M.rand_u64(rng))
            fields[f] = int(getattr(m, f))
        m.reject = rng.random() < 0.2
        fields["reject"] = int(m.reject)
        entries, sizes = [], []
        if tname in ("MsgAppend", "MsgPropose") or rng.random() < 0.08:
            first = M.rand_u64(rng) % (1 << 62)
            term = M.rand_u64(rng)
            for k in range(rng.randrange(0, 7)):
                e, d = rand_entry(rng, Entry, first + k, term)
                m.entries.add().CopyFrom(e)
                entries.append(d)
                sizes.append(e.ByteSize())
        snapshot = None
        if tname == "MsgSnapshot" or rng.random() < 0.05:
            s = Snapshot()
            if rng.random() < 0.8:
                s.data = rand_bytes(rng, 30)
                s.metadata.index, s.metadata.term = M.rand_u64(rng), M.rand_u64(rng)
                s.metadata.conf_state.voters.extend([1, 2, 3])
            m.snapshot.CopyFrom(s)  # (an EMPTY snapshot is still a present field: tag 0x4a, length 0)
            snapshot = s.SerializeToString(deterministic=True).hex()
        context = b""
        if rng.random() < 0.2:
            context = rand_bytes(rng, 20)
            m.context = context
        data = m.SerializeToString(deterministic=True)
        assert Message.FromString(data) == m
        vectors.append({"type": tname, "message": fields, "entries": entries, "entry_sizes": sizes, "snapshot": snapshot,
                        "context": context.hex(), "hex": data.hex()})
        if len(sizes) >= 1:
            total = sum(sizes)
            for mx in sorted({0, 1, sizes[0], sizes[0] + 1, total // 2, max(0, total - 1), total, total + 1}):
                limits.append({"vector": len(vectors) - 1, "max": mx, "keep": limit_size(sizes, mx)})
            limits.append({"vector": len(vectors) - 1, "max": None, "keep": len(sizes)})
    # the doc example of util::limit_size (src/util.rs:32-50): five equal entries of 100 bytes, Some(220) keeps 2, Some(0) keeps 1
    e = Entry()
    e.data = b"*" * 100
    doc_case = {"entry": {"data": e.data.hex()}, "entry_size": e.ByteSize(), "n": 5,
                "cases": [{"max": 220, "keep": limit_size([e.ByteSize()] * 5, 220)}, {"max": 0, "keep": limit_size([e.ByteSize()] * 5, 0)}]}
    return types, vectors, limits, doc_case


def main():
    package, enums, messages = M.parse_proto(open(M.PROTO, encoding="utf-8").read())
    cls = M.build_classes(package, enums, messages)
    types, vectors, limits, doc_case = build(cls, enums)
    head = {"source": "proto/proto/eraftpb.proto (Entry :23-31, Snapshot :33-44, Message :71-92), serialised by the protobuf runtime "
                      "from descriptors parsed out of that file (tests/golden/make_eraftpb_outgoing.py); limit_size: "
                      "src/util.rs:52-76 restated over Entry.ByteSize()",
            "message_types": types, "limit_size_doc_example": doc_case}
    text = (json.dumps(head, sort_keys=True)[:-1] + ', "vectors": [\n' +
            ",\n".join(json.dumps(v, sort_keys=True) for v in vectors) + '\n], "limit_size": [\n' +
            ",\n".join(json.dumps(v, sort_keys=True) for v in limits) + "\n]}\n")
    if "--check" in sys.argv:
        sys.exit(0 if os.path.exists(OUT) and open(OUT).read() == text else 1)
    open(OUT, "w").write(text)
    print(f"wrote {OUT}: {len(vectors)} vectors, {len(limits)} limit_size cases")


if __name__ == "__main__":
    main()
