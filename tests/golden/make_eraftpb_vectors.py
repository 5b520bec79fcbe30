#!/usr/bin/env python3
"""Golden vectors for the wire format of the path's input: protobuf-encoded eraftpb::Message.

Run in the build container (where /root/reference exists):

    python tests/golden/make_eraftpb_vectors.py          # writes tests/golden/eraftpb_messages.json

The message types are NOT typed in here: the script parses /root/reference/proto/proto/eraftpb.proto (proto3; enums,
messages, scalar / message / repeated fields -- the whole grammar that file uses), builds the descriptors with the
protobuf runtime (no protoc in the image) and lets THAT runtime serialise seeded random messages. Every vector is
`{"hex": <serialised bytes>, "fields": {...}}`; consumers (tests/test_wire_format.py) check rg_decode_message -- the
decoder behind rg_step_bytes -- against them. A few hand-made malformed byte strings (truncated varint, a length that
runs past the end, a group wire type) carry `"error": true`.
"""
import json
import os
import random
import re
import sys

PROTO = "/root/reference/proto/proto/eraftpb.proto"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "eraftpb_messages.json")

SCALARS = {"uint64": 4, "uint32": 13, "int64": 3, "int32": 5, "bool": 8, "bytes": 12, "string": 9}  # FieldDescriptorProto.Type


def parse_proto(text):
    """-> (package, {enum: [(name, number)]}, {message: [(label, type, name, number)]})"""
    text = re.sub(r"//[^\n]*", "", text)
    package = re.search(r"package\s+([\w.]+)\s*;", text).group(1)
    enums, messages = {}, {}
    for kind, name, body in re.findall(r"(enum|message)\s+(\w+)\s*\{([^{}]*)\}", text):
        if kind == "enum":
            enums[name] = [(n, int(v)) for n, v in re.findall(r"(\w+)\s*=\s*(\d+)\s*;", body)]
        else:
            messages[name] = [(lab or "", typ, fname, int(num))
                              for lab, typ, fname, num in re.findall(r"(repeated\s+)?([\w.]+)\s+(\w+)\s*=\s*(\d+)\s*;", body)]
    return package, enums, messages


def build_classes(package, enums, messages):
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    fd = descriptor_pb2.FileDescriptorProto(name="eraftpb.proto", package=package, syntax="proto3")
    for name, values in enums.items():
        e = fd.enum_type.add(name=name)
        for n, v in values:
            e.value.add(name=n, number=v)
    for name, fields in messages.items():
        m = fd.message_type.add(name=name)
        for lab, typ, fname, num in fields:
            f = m.field.add(name=fname, number=num)
            f.label = 3 if lab.strip() == "repeated" else 1
            if typ in SCALARS:
                f.type = SCALARS[typ]
            elif typ in enums:
                f.type, f.type_name = 14, f".{package}.{typ}"
            elif typ in messages:
                f.type, f.type_name = 11, f".{package}.{typ}"
            else:
                raise ValueError(f"unknown type {typ} in {name}.{fname}")
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return {name: message_factory.GetMessageClass(pool.FindMessageTypeByName(f"{package}.{name}")) for name in messages}


def mistyped_known_field(data, messages, enums, name="Message"):
    """Does `data` -- bytes the protobuf runtime ACCEPTS as message `name` -- carry a field of the schema with a wire type
    other than the declared one (varint for integers / bools / enums, length-delimited for bytes and messages, either for a
    repeated uint64)? The Python runtime files such a field under the unknown fields and carries on; both codecs the reference
    can be built with refuse the whole message (rust-protobuf 2: WireError::UnexpectedWireType out of the generated merge_from;
    prost: check_wire_type), so Message::parse_from_bytes fails and RawNode::step never sees it. Schema-driven (the parsed
    .proto), recursing into nested messages as a real parser does; inside an unknown group everything is opaque."""
    decl = {num: (lab.strip() == "repeated", typ) for lab, typ, _, num in messages[name]}

    def rd_varint(b, i):
        v = s = 0
        while True:
            c = b[i]
            i += 1
            v |= (c & 0x7f) << s
            s += 7
            if not c & 0x80:
                return v, i

    def skip_group(b, i, field):
        while True:
            key, i = rd_varint(b, i)
            f, wt = key >> 3, key & 7
            if wt == 4:
                assert f == field
                return i
            if wt == 0:
                _, i = rd_varint(b, i)
            elif wt == 1:
                i += 8
            elif wt == 2:
                n, i = rd_varint(b, i)
                i += n
            elif wt == 3:
                i = skip_group(b, i, f)
            elif wt == 5:
                i += 4

    i = 0
    while i < len(data):
        key, i = rd_varint(data, i)
        f, wt = key >> 3, key & 7
        if f in decl:
            rep, typ = decl[f]
            is_len = typ in ("bytes", "string") or typ in messages
            ok = (wt == 2) if is_len else (wt in (0, 2) if rep else wt == 0)
            if not ok:
                return True
        if wt == 0:
            _, i = rd_varint(data, i)
        elif wt == 1:
            i += 8
        elif wt == 2:
            n, i = rd_varint(data, i)
            if f in decl and decl[f][1] in messages and mistyped_known_field(data[i:i + n], messages, enums, decl[f][1]):
                return True
            i += n
        elif wt == 3:
            i = skip_group(data, i, f)
        elif wt == 5:
            i += 4
    return False


def rand_u64(rng):
    k = rng.random()
    if k < 0.15:
        return 0
    if k < 0.5:
        return rng.randrange(1, 200)
    if k < 0.8:
        return rng.randrange(1, 1 << 32)
    if k < 0.95:
        return rng.randrange(1 << 32, 1 << 64)
    return (1 << 64) - 1


def main():
    package, enums, messages = parse_proto(open(PROTO, encoding="utf-8").read())
    cls = build_classes(package, enums, messages)
    Message, Entry, Snapshot = cls["Message"], cls["Entry"], cls["Snapshot"]
    types = dict(enums["MessageType"])
    rng = random.Random(0xE7AF)
    vectors = []
    order = ["MsgAppendResponse"] * 6 + ["MsgHeartbeatResponse"] * 3 + sorted(types)
    for i in range(240):
        tname = order[i % len(order)]
        m = Message()
        m.msg_type = types[tname]
        for f in ("to", "from", "term", "log_term", "index", "commit", "commit_term", "reject_hint", "request_snapshot",
                  "priority"):
            if rng.random() < 0.7:
                setattr(m, f, rand_u64(rng))
        m.reject = rng.random() < 0.4
        if tname in ("MsgAppend", "MsgPropose") or rng.random() < 0.1:
            for _ in range(rng.randrange(0, 4)):
                e = m.entries.add()
                e.term, e.index = rand_u64(rng), rand_u64(rng)
                e.data = bytes(rng.randrange(256) for _ in range(rng.randrange(0, 40)))
        if tname == "MsgSnapshot" or rng.random() < 0.05:
            m.snapshot.data = bytes(rng.randrange(256) for _ in range(rng.randrange(0, 30)))
            m.snapshot.metadata.index = rand_u64(rng)
            m.snapshot.metadata.term = rand_u64(rng)
        if rng.random() < 0.2:
            m.context = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 20)))
        data = m.SerializeToString(deterministic=True)
        vectors.append({"hex": data.hex(), "type": tname, "fields": {
            "msg_type": int(m.msg_type), "to": int(m.to), "from": int(getattr(m, "from")), "term": int(m.term),
            "log_term": int(m.log_term), "index": int(m.index), "commit": int(m.commit), "commit_term": int(m.commit_term),
            "reject": int(m.reject), "reject_hint": int(m.reject_hint), "request_snapshot": int(m.request_snapshot),
            "priority": int(m.priority), "n_entries": len(m.entries), "has_snapshot": int(m.HasField("snapshot")),
            "context_len": len(m.context)}})
    # an unknown field (number 99, varint and length-delimited) in front of a real message: protobuf skips it
    base = bytes.fromhex(vectors[0]["hex"])
    vectors.append({"hex": (bytes([0x98, 0x06, 0x2A]) + bytes([0x9A, 0x06, 0x03, 1, 2, 3]) + base).hex(), "type": "unknown-fields",
                    "fields": vectors[0]["fields"]})
    # a well-formed unknown GROUP (field 20: START 0xa3 0x01, a varint field inside, END 0xa4 0x01) is skipped as well
    vectors.append({"hex": (bytes([0xA3, 0x01, 0x08, 0x05, 0xA4, 0x01]) + base).hex(), "type": "unknown-group", "fields": vectors[0]["fields"]})
    bad_cases = [(b"\x08", "truncated varint"), (b"\x3a\x7f\x00", "length past the end"), (b"\x0b\x00", "group that never ends"),
                 (b"\x00\x01", "field number 0"), (b"\x19\x01\x02", "truncated fixed64"),
                 (b"\x08" + b"\xff" * 11, "varint longer than 10 bytes"),
                 (b"\x3a\x02\x10\x80", "an entry whose own bytes are malformed (truncated varint inside field 7)"),
                 (b"\x4a\x04\x12\x02\x10\x80", "a snapshot whose metadata is malformed"),
                 (b"\x80\x80\x80\x80\x10\x01", "a tag that does not fit 32 bits"),
                 (b"\xa3\x01\x08\x05\xac\x01", "a group closed by another field's END_GROUP"), (b"\x0e\x00", "wire type 6")]
    for bad, why in bad_cases:
        try:  # the runtime is the judge of what is malformed
            Message.FromString(bad)
            raise SystemExit(f"the protobuf runtime ACCEPTS {why!r}: not an error vector")
        except SystemExit:
            raise
        except Exception:
            pass
        vectors.append({"hex": bad.hex(), "type": why, "error": True})
    # A field of the schema with the wrong wire type. HERE the judge is not the Python runtime -- it keeps such a field as an
    # unknown one and accepts the message -- but the two Rust codecs of the reference, which both refuse it (see
    # mistyped_known_field): `strict` marks the vectors where the runtimes differ and the decoder follows the reference's.
    strict_cases = [(b"\x08\x04\x1a\x01\x05\x21" + b"\x00" * 8 + b"\x30\x07", "from as bytes, term as fixed64 (MsgAppendResponse otherwise)"),
                    (b"\x08\x04\x22\x01\x07", "term length-delimited"), (b"\x0d\x04\x00\x00\x00", "msg_type as fixed32"),
                    (b"\x38\x05", "entries as a varint"), (b"\x48\x05", "snapshot as a varint"), (b"\x60\x01", "context as a varint"),
                    (b"\x55\x01\x00\x00\x00", "reject as fixed32"), (b"\x13\x14", "to as a group"),
                    (b"\x3a\x03\x12\x01\x00", "an entry whose term is length-delimited"), (b"\x3a\x02\x20\x01", "an entry whose data is a varint"),
                    (b"\x4a\x02\x08\x01", "a snapshot whose data is a varint"), (b"\x4a\x04\x12\x02\x08\x01", "snapshot metadata whose conf_state is a varint"),
                    (b"\x4a\x09\x12\x07\x0a\x05\x0d\x01\x00\x00\x00", "conf_state voters as fixed32")]
    for bad, why in strict_cases:
        Message.FromString(bad)  # (raises if the Python runtime refuses it: then it belongs to bad_cases)
        assert mistyped_known_field(bad, messages, enums), why
        vectors.append({"hex": bad.hex(), "type": why, "error": True, "strict": True})
    # ... and the other side of that rule: a repeated uint64 is accepted packed AND one varint field per element
    f0 = dict(vectors[0]["fields"])
    f0["has_snapshot"] = 1
    for okb, why in ((b"\x4a\x07\x12\x05\x0a\x03\x0a\x01\x09" + base, "conf_state voters packed"),
                     (b"\x4a\x08\x12\x06\x0a\x04\x08\x09\x08\x0a" + base, "conf_state voters unpacked (one varint field per element)")):
        m = Message.FromString(okb)
        assert list(m.snapshot.metadata.conf_state.voters) in ([9], [9, 10]), why
        assert not mistyped_known_field(okb, messages, enums), why
        vectors.append({"hex": okb.hex(), "type": why, "fields": f0})
    for v in vectors:
        if not v.get("error"):
            Message.FromString(bytes.fromhex(v["hex"]))  # (raises if the runtime disagrees)
    doc = {"source": "proto/proto/eraftpb.proto (enum MessageType :49-69, message Message :71-92), serialised by the protobuf "
                     "runtime from descriptors parsed out of that file (tests/golden/make_eraftpb_vectors.py)",
           "message_types": types, "message_fields": {n: num for _, _, n, num in messages["Message"]}, "vectors": vectors}
    # one vector per line: a 250-line fixture instead of a 5 000-line one
    head = {k: v for k, v in doc.items() if k != "vectors"}
    text = (json.dumps(head, sort_keys=True)[:-1] + ', "vectors": [\n' +
            ",\n".join(json.dumps(v, sort_keys=True) for v in vectors) + "\n]}\n")
    if "--check" in sys.argv:
        sys.exit(0 if open(OUT).read() == text else 1)
    open(OUT, "w").write(text)
    print(f"wrote {OUT}: {len(vectors)} vectors")


if __name__ == "__main__":
    main()
