"""The wire format of the path's input: rg_decode_message / rg_step_bytes (RawNode::step on the protobuf bytes a transport
delivers) against vectors the protobuf runtime serialised from the reference's own eraftpb.proto
(tests/golden/make_eraftpb_vectors.py -> tests/golden/eraftpb_messages.json)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "eraftpb_messages.json")
DOC = json.load(open(GOLD))
MT = DOC["message_types"]


def varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def encode(fields):
    """A scalar-only eraftpb::Message as proto3 serialises it: fields in number order, defaults omitted (test
    infrastructure for the GPU test below, where neither the reference nor its .proto exists; pinned against the golden
    bytes by test_the_test_encoder_reproduces_the_runtime)."""
    num = DOC["message_fields"]
    out = bytearray()
    for name, n in sorted(num.items(), key=lambda kv: kv[1]):
        v = int(fields.get(name, 0))
        if v:
            out += varint(n << 3) + varint(v)
    return bytes(out)


def test_decoder_matches_every_golden_vector(rg):
    from raft_rs_amd.engine import decode_message, EngineError
    n_ok = n_err = 0
    for v in DOC["vectors"]:
        data = bytes.fromhex(v["hex"])
        if v.get("error"):
            with pytest.raises(EngineError) as e:
                decode_message(data)
            assert e.value.code == -1, (v["type"], e.value)
            n_err += 1
            continue
        got = decode_message(data)
        assert got == v["fields"], (v["type"], got, v["fields"])
        n_ok += 1
    assert n_ok >= 244 and n_err >= 24
    # (13 of the error vectors are `strict`: a field of the schema with the wrong wire type -- accepted by the Python runtime
    # as an unknown field, refused by both Rust codecs of the reference, hence by this decoder)
    assert sum(1 for v in DOC["vectors"] if v.get("strict")) >= 13
    assert decode_message(b"") == {k: 0 for k in DOC["vectors"][0]["fields"]}  # the empty message: every field default


def test_the_test_encoder_reproduces_the_runtime():
    n = 0
    for v in DOC["vectors"]:
        f = v.get("fields")
        if not f or f["n_entries"] or f["has_snapshot"] or f["context_len"] or v["type"] == "unknown-fields":
            continue
        assert encode(f).hex() == v["hex"], v
        n += 1
    assert n > 100


@pytest.mark.skipif(not os.path.exists("/root/reference/proto/proto/eraftpb.proto"), reason="reference tree not present")
def test_committed_vectors_are_what_the_generator_produces():
    r = subprocess.run([sys.executable, os.path.join(HERE, "golden", "make_eraftpb_vectors.py"), "--check"])
    assert r.returncode == 0, "tests/golden/eraftpb_messages.json is out of date: rerun make_eraftpb_vectors.py"


def _mutations(rng, n):
    for _ in range(n):
        b = bytearray(bytes.fromhex(DOC["vectors"][int(rng.integers(0, len(DOC["vectors"])))]["hex"]))
        k = rng.random()
        if k < 0.4 and b:
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        elif k < 0.6 and b:
            b = b[:int(rng.integers(0, len(b)))]
        elif k < 0.8:
            at = int(rng.integers(0, len(b) + 1))
            b[at:at] = bytes(int(x) for x in rng.integers(0, 256, size=int(rng.integers(1, 5))))
        else:
            b = bytearray(int(x) for x in rng.integers(0, 256, size=int(rng.integers(0, 24))))
        yield bytes(b)


def test_decoder_survives_mutated_and_random_bytes(rg):
    """20 000 mutated golden vectors / random byte strings: every one is either decoded or refused with INVALID_ARG, the
    same way twice (no crash, no read past the end: the library is built with the sanitiser-clean walker)."""
    from raft_rs_amd.engine import decode_message, EngineError
    rng = np.random.default_rng(99)
    n_ok = n_bad = 0
    for data in _mutations(rng, 20000):
        res = []
        for _ in range(2):
            try:
                res.append(decode_message(data))
            except EngineError as e:
                assert e.code == -1
                res.append(None)
        assert res[0] == res[1]
        n_ok += res[0] is not None
        n_bad += res[0] is None
    assert n_ok > 3000 and n_bad > 3000, (n_ok, n_bad)


@pytest.mark.skipif(not os.path.exists("/root/reference/proto/proto/eraftpb.proto"), reason="reference tree not present")
def test_decoder_agrees_with_the_protobuf_runtime_on_mutated_bytes(rg):
    """Differential: the protobuf runtime, with descriptors parsed out of the reference's eraftpb.proto, accepts exactly
    the byte strings rg_decode_message accepts, and reads the same fields (nested entries / snapshots are validated, unknown
    fields and well-formed groups skipped, tags beyond 32 bits refused) -- with the one rule where the Python runtime and the
    reference's Rust codecs part: a field of the schema that arrives with another wire type than the declared one is an
    unknown field to the former and a parse error to the latter (make_eraftpb_vectors.mistyped_known_field), and the decoder
    sides with the reference."""
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_eraftpb_vectors as M
    from raft_rs_amd.engine import decode_message, EngineError
    package, enums, messages = M.parse_proto(open(M.PROTO, encoding="utf-8").read())
    Message = M.build_classes(package, enums, messages)["Message"]
    rng = np.random.default_rng(4242)
    n_ok = n_strict = 0
    for data in _mutations(rng, 20000):
        try:
            m = Message.FromString(data)
        except Exception:  # noqa: BLE001
            m = None
        try:
            d = decode_message(data)
        except EngineError:
            d = None
        strict = m is not None and M.mistyped_known_field(data, messages, enums)
        n_strict += strict
        assert (m is None or strict) == (d is None), (data.hex(), m is None, strict, d is None)
        if d is None:
            continue
        want = {"msg_type": int(m.msg_type) & 0xffffffff, "to": m.to, "from": getattr(m, "from"), "term": m.term,
                "log_term": m.log_term, "index": m.index, "commit": m.commit, "commit_term": m.commit_term,
                "reject": int(m.reject), "reject_hint": m.reject_hint, "request_snapshot": m.request_snapshot,
                "priority": m.priority, "n_entries": len(m.entries), "has_snapshot": int(m.HasField("snapshot")),
                "context_len": len(m.context)}
        assert d == want, (data.hex(), d, want)
        n_ok += 1
    assert n_ok > 3000 and n_strict > 100, (n_ok, n_strict)


def test_wire_code_is_clean_under_the_sanitisers(tmp_path):
    """raft_rs_amd/csrc/rg_wire.h -- the very code rg_decode_message and rg_encode_message run -- built with
    -fsanitize=address,undefined: the decoder over every golden vector (incoming and outgoing) plus 300 seeded mutations of
    each, every input in a heap buffer of exactly its length; the encoder over 20 000 seeded messages into output buffers of
    exactly the computed size, read back by the decoder."""
    import shutil
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    exe = str(tmp_path / "wire_asan")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                        os.path.join(HERE, "host_check", "wire_asan.cpp"), "-o", exe], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0 and "asan" in r.stdout.lower():
        pytest.skip("no sanitiser runtime on this host: " + r.stdout[-200:])
    assert r.returncode == 0, r.stdout
    feed = "\n".join(v["hex"] for v in DOC["vectors"] + OUT_DOC["vectors"]) + "\n"
    r = subprocess.run([exe, "300", "20000"], input=feed, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "WIRE_ASAN_OK" in r.stdout and "encoded 20000" in r.stdout, r.stdout[-2000:]


# ---- the other direction: rg_encode_message / rg_entry_size / rg_limit_size (what the path SENDS) ----
OUT_DOC = json.load(open(os.path.join(HERE, "golden", "eraftpb_outgoing.json")))


def _entries_of(v):
    return [dict(e, data=bytes.fromhex(e.get("data", "")), context=bytes.fromhex(e.get("context", ""))) for e in v["entries"]]


def _encode_vector(v, **kw):
    from raft_rs_amd.engine import encode_message
    return encode_message(v["message"], _entries_of(v), snapshot=None if v["snapshot"] is None else bytes.fromhex(v["snapshot"]),
                          context=bytes.fromhex(v["context"]), **kw)


def test_encoder_reproduces_every_outgoing_vector(rg):
    """rg_encode_message writes, byte for byte, what the protobuf runtime serialised for the same message (entries with their
    payloads, a present-but-empty snapshot, every message type); rg_entry_size is the runtime's Entry.ByteSize() =
    Entry::compute_size(); and the decoder of the input side reads the scalar fields back."""
    from raft_rs_amd.engine import decode_message, entry_size
    n_ent = n_snap = 0
    for v in OUT_DOC["vectors"]:
        assert _encode_vector(v).hex() == v["hex"], v["type"]
        assert [entry_size(e) for e in _entries_of(v)] == v["entry_sizes"]
        back = decode_message(bytes.fromhex(v["hex"]))
        for k, want in v["message"].items():
            assert back[k] == want, (k, v["type"])
        assert back["n_entries"] == len(v["entries"]) and back["has_snapshot"] == int(v["snapshot"] is not None)
        n_ent += len(v["entries"])
        n_snap += v["snapshot"] is not None
    assert len(OUT_DOC["vectors"]) >= 200 and n_ent > 200 and n_snap > 20
    assert entry_size({}) == 0  # Entry::default()


def test_limit_size_is_the_references(rg):
    """rg_limit_size against util::limit_size (src/util.rs:52-76) restated literally by the vector generator: around every
    boundary of every vector's entry sizes (incl. Entry::default() runs, where the running total is still 0), NO_LIMIT,
    and the function's own doc example (five 100-byte entries: Some(220) keeps 2, Some(0) keeps 1)."""
    from raft_rs_amd.engine import limit_size
    for c in OUT_DOC["limit_size"]:
        ents = _entries_of(OUT_DOC["vectors"][c["vector"]])
        assert limit_size(ents, c["max"]) == c["keep"], c
    assert len(OUT_DOC["limit_size"]) > 400
    ex = OUT_DOC["limit_size_doc_example"]
    ents = [{"data": bytes.fromhex(ex["entry"]["data"])}] * ex["n"]
    assert (ex["cases"][0]["max"], ex["cases"][0]["keep"], ex["cases"][1]["keep"]) == (220, 2, 1)
    for c in ex["cases"]:
        assert limit_size(ents, c["max"]) == c["keep"]
    assert limit_size([], 10) == 0 and limit_size(ents[:1], 0) == 1 and limit_size(ents, None) == ex["n"]


def test_encoder_refuses_what_it_cannot_write(rg):
    from raft_rs_amd.engine import encode_message, EngineError, MessageC, EntryC, load_library
    import ctypes as C
    v = next(x for x in OUT_DOC["vectors"] if len(x["entries"]) >= 2)
    size = len(bytes.fromhex(v["hex"]))
    for cap in (0, 1, size - 1):  # too small a buffer: INVALID_ARG, nothing written past it (the ASan harness checks that part)
        with pytest.raises(EngineError) as e:
            _encode_vector(v, cap=cap)
        assert e.value.code == -1 and str(size) in str(e.value)
    assert _encode_vector(v, cap=size + 7).hex() == v["hex"]
    L = load_library()
    m, n = MessageC(), C.c_uint64(0)
    m.n_entries = 2  # a length without its pointer
    assert L.rg_message_size(C.byref(m), C.byref(n)) == -1
    m.n_entries, m.context_len = 0, 5
    assert L.rg_message_size(C.byref(m), C.byref(n)) == -1
    m.context_len = 0
    e = (EntryC * 1)()
    e[0].data_len = 1 << 31  # beyond what a protobuf message may hold
    e[0].data = b"x"
    m.entries, m.n_entries = e, 1
    assert L.rg_message_size(C.byref(m), C.byref(n)) == -1
    assert L.rg_message_size(None, C.byref(n)) == -1 and L.rg_entry_size(None) == 0
    assert encode_message({}) == b""  # Message::default(): zero bytes


@pytest.mark.skipif(not os.path.exists("/root/reference/proto/proto/eraftpb.proto"), reason="reference tree not present")
def test_committed_outgoing_vectors_are_what_the_generator_produces():
    r = subprocess.run([sys.executable, os.path.join(HERE, "golden", "make_eraftpb_outgoing.py"), "--check"])
    assert r.returncode == 0, "tests/golden/eraftpb_outgoing.json is out of date: rerun make_eraftpb_outgoing.py"


@pytest.mark.skipif(not os.path.exists("/root/reference/proto/proto/eraftpb.proto"), reason="reference tree not present")
def test_encoder_agrees_with_the_protobuf_runtime_on_random_messages(rg):
    """Differential, 5 000 seeded messages the committed vectors do not hold: the runtime (descriptors parsed out of the
    reference's eraftpb.proto) parses what rg_encode_message wrote into a message EQUAL to the one it built from the same
    content, serialises that to the same bytes, and its ByteSize() is rg_message_size."""
    import random
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_eraftpb_vectors as M
    import make_eraftpb_outgoing as MO
    from raft_rs_amd.engine import encode_message
    package, enums, messages = M.parse_proto(open(M.PROTO, encoding="utf-8").read())
    cls = M.build_classes(package, enums, messages)
    Message, Entry, Snapshot = cls["Message"], cls["Entry"], cls["Snapshot"]
    rng = random.Random(777)
    n_ent = 0
    for i in range(5000):
        m = Message()
        m.msg_type = rng.randrange(0, 19)
        fields = {"msg_type": int(m.msg_type)}
        for f in MO.SCALARS:
            if rng.random() < 0.6:
                setattr(m, f, M.rand_u64(rng))
            fields[f] = int(getattr(m, f))
        m.reject = rng.random() < 0.3
        fields["reject"] = int(m.reject)
        entries = []
        for k in range(rng.randrange(0, 5) if rng.random() < 0.6 else 0):
            e, d = MO.rand_entry(rng, Entry, rng.randrange(1 << 40) + k, M.rand_u64(rng))
            m.entries.add().CopyFrom(e)
            entries.append(dict(d, data=bytes.fromhex(d.get("data", "")), context=bytes.fromhex(d.get("context", ""))))
        snap = None
        if rng.random() < 0.15:
            s = Snapshot()
            if rng.random() < 0.7:
                s.data = MO.rand_bytes(rng, 50)
                s.metadata.index = M.rand_u64(rng)
            m.snapshot.CopyFrom(s)
            snap = s.SerializeToString(deterministic=True)
        ctx = MO.rand_bytes(rng, 16) if rng.random() < 0.3 else b""
        m.context = ctx
        got = encode_message(fields, entries, snapshot=snap, context=ctx)
        assert got == m.SerializeToString(deterministic=True), i
        assert Message.FromString(got) == m and len(got) == m.ByteSize()
        n_ent += len(entries)
    assert n_ent > 3000


def test_encode_then_decode_is_the_identity_on_the_scalar_fields(rg):
    """Property (hypothesis): for ANY field values -- the whole u64 range, every message type number, entries and payloads
    of any size class -- rg_decode_message reads back what rg_encode_message wrote, rg_message_size is the length, and
    rg_entry_size adds up to the bytes the entries occupy on the wire (minus their tags and length prefixes)."""
    from hypothesis import given, settings, strategies as S
    from raft_rs_amd.engine import decode_message, encode_message, entry_size
    u64 = S.one_of(S.just(0), S.integers(0, 300), S.integers(0, (1 << 64) - 1), S.just((1 << 64) - 1))
    blob = S.binary(max_size=70)
    entry = S.fixed_dictionaries({"entry_type": S.integers(0, 2), "term": u64, "index": u64, "data": blob, "context": S.one_of(S.just(b""), blob),
                                  "sync_log": S.booleans()})
    fields = S.fixed_dictionaries({"msg_type": S.integers(0, 18), "to": u64, "from": u64, "term": u64, "log_term": u64, "index": u64,
                                   "commit": u64, "commit_term": u64, "reject": S.integers(0, 1), "reject_hint": u64,
                                   "request_snapshot": u64, "priority": u64})

    def vlen(v):
        n = 1
        while v >= 0x80:
            v >>= 7
            n += 1
        return n

    @settings(max_examples=400, deadline=None)
    @given(fields, S.lists(entry, max_size=5), S.one_of(S.none(), S.just(b"")), S.one_of(S.just(b""), blob))
    def check(f, ents, snap, ctx):
        data = encode_message(f, ents, snapshot=snap, context=ctx)
        back = decode_message(data)
        for k, v in f.items():
            assert back[k] == v, k
        assert back["n_entries"] == len(ents) and back["has_snapshot"] == int(snap is not None) and back["context_len"] == len(ctx)
        sizes = [entry_size(e) for e in ents]
        scalars = sum(1 + vlen(v) for k, v in f.items() if v)
        want = scalars + sum(1 + vlen(z) + z for z in sizes) + (2 if snap is not None else 0) + ((1 + vlen(len(ctx)) + len(ctx)) if ctx else 0)
        assert len(data) == want

    check()


@pytest.mark.gpu
def test_step_bytes_equals_step(rg):
    """Two engines, the same stream: one stepped through rg_step / rg_step_heartbeat_response, the other through
    rg_step_bytes on the protobuf encoding of the same messages; identical state and results -- with the Inflights on the
    host (max_inflight = 0), so that the caller's Inflights::full() bit travels beside the bytes (it is not on the wire) and
    decides is_paused / free_first_one / the `old_paused` re-send exactly as through rg_step. A third engine gets the same
    bytes WITHOUT the bit and must end somewhere else: the argument is live. Plus RawNode::step's error behaviour on bytes:
    local types, unknown peers, other types, garbage."""
    import fuzz
    import oracle_lib as O
    from raft_rs_amd.engine import EngineError, ERR
    rng = np.random.default_rng(5150)
    G, P, TERM = 600, 5, 7
    st = O.add_term_table(O.alloc_state(G, P))
    st["cfg"][:] = fuzz.random_cfg(rng, G, P)
    fuzz.random_state(rng, st, small_values=True)
    fuzz.random_term_table(rng, st, TERM)
    self_slot = ((st["cfg"] >> 16) & 7).astype(np.int64)
    a, b, c = rg.Engine(G, P), rg.Engine(G, P), rg.Engine(G, P)
    for eng in (a, b, c):
        eng.load_state(st)
        for g in range(G):
            eng.set_peers(g, [11 * (s + 1) for s in range(P)], TERM)
    n_app = n_hb = blind = 0
    for rnd in range(6):
        for g in range(G):
            for s in range(P):
                if s == self_slot[g] or rng.random() < 0.4:
                    continue
                frm = 11 * (s + 1)
                full = bool(rng.random() < 0.3)  # the host's Inflights::full() for this peer
                if rng.random() < 0.2:
                    commit = int(rng.integers(0, 50))
                    a.step_heartbeat_response(g, frm, TERM, commit, ins_full=full)
                    data = encode({"msg_type": MT["MsgHeartbeatResponse"], "from": frm, "to": 1, "term": TERM, "commit": commit})
                    b.step_bytes(g, data, ins_full=full)
                    c.step_bytes(g, data)
                    n_hb += 1
                else:
                    reject = rng.random() < 0.2
                    idx = int(rng.integers(0, 60))
                    f = {"msg_type": MT["MsgAppendResponse"], "from": frm, "to": 1, "term": TERM, "index": idx,
                         "commit": int(rng.integers(0, 40)), "reject": int(reject), "reject_hint": int(rng.integers(0, 60)) if reject else 0,
                         "log_term": int(rng.integers(1, TERM)) if reject and rng.random() < 0.5 else 0,
                         "request_snapshot": int(rng.integers(1, 30)) if reject and rng.random() < 0.1 else 0}
                    a.step(g, frm, TERM, idx, commit=f["commit"], reject=reject, reject_hint=f["reject_hint"],
                           request_snapshot=f["request_snapshot"], log_term=f["log_term"], ins_full=full)
                    b.step_bytes(g, encode(f), ins_full=full)
                    c.step_bytes(g, encode(f))
                    n_app += 1
        a.flush()
        b.flush()
        c.flush()
        ra, rb = a.ingested_results(), b.ingested_results()
        oa, ob = np.argsort(ra[0]), np.argsort(rb[0])
        for x, y in zip(ra, rb):
            assert np.array_equal(x[oa], y[ob]), rnd
        sa, sb = a.read_state(), b.read_state()
        assert not fuzz.diff_states(sa, sb, G, P), rnd
        rc_ = c.ingested_results()
        oc_ = np.argsort(rc_[0])
        blind += int(sum((x[ob] != y[oc_]).sum() for x, y in zip(rb, rc_)))
    assert blind > 50, "without the Inflights::full() bit the same bytes give other results: the bit is not on the wire"
    assert n_app > 3000 and n_hb > 500
    for typ in ("MsgHup", "MsgBeat", "MsgUnreachable", "MsgSnapStatus", "MsgCheckQuorum"):  # is_local_msg (raw_node.rs:57-66)
        with pytest.raises(EngineError) as e:
            b.step_bytes(0, encode({"msg_type": MT[typ], "from": 22, "term": TERM}))
        assert e.value.code == ERR["STEP_LOCAL_MSG"], typ
    with pytest.raises(EngineError) as e:
        b.step_bytes(0, encode({"msg_type": MT["MsgAppendResponse"], "from": 999, "term": TERM, "index": 3}))
    assert e.value.code == ERR["STEP_PEER_NOT_FOUND"]
    with pytest.raises(EngineError) as e:
        b.step_bytes(0, encode({"msg_type": MT["MsgAppendResponse"], "from": 22, "term": TERM + 1, "index": 3}))
    assert e.value.code == ERR["HIGHER_TERM"]
    for typ in ("MsgAppend", "MsgRequestVote", "MsgRequestVoteResponse", "MsgSnapshot", "MsgHeartbeat", "MsgTimeoutNow"):
        with pytest.raises(EngineError) as e:
            b.step_bytes(0, encode({"msg_type": MT[typ], "from": 22, "term": TERM}))
        assert e.value.code == ERR["NOT_ON_PATH"], typ
    with pytest.raises(EngineError) as e:
        b.step_bytes(0, b"\x08")
    assert e.value.code == ERR["INVALID_ARG"]
    a.close()
    b.close()
    c.close()
