"""Helpers shared by the CPU (host_check) and GPU tests of the send stage (device Inflights + the
maybe_send_append decision). Test infrastructure."""
import numpy as np

import fuzz
import oracle_lib as O

PF_INS_FULL = 0x10


def coalesce_oracle(msgs):
    """Oracle messages (one per maybe_send_append that sent) -> {(group, slot): (kind, prev_index, last_index,
    n_msgs)}; checks that one peer's messages are contiguous, which is what lets the engine report one item."""
    out = {}
    for m in msgs:
        key = (int(m["group"]), int(m["to"]) - 1)
        kind, prev, n = int(m["kind"]), int(m["index"]), int(m["n_entries"])
        if key not in out:
            out[key] = [kind, prev, (prev + n) & ((1 << 64) - 1), 1]
        else:
            cur = out[key]
            assert cur[0] == O.SEND_APPEND and kind == O.SEND_APPEND, (key, cur, m)
            assert prev == cur[2], ("messages of one peer must be contiguous", key, cur, m)
            cur[2] = prev + n
            cur[3] += 1
    return {k: tuple(v) for k, v in out.items()}


def items_dict(items):
    out = {}
    for it in items:
        key = (int(it["group"]), int(it["slot"]))
        assert key not in out, ("one item per peer", key)
        out[key] = (int(it["kind"]), int(it["prev_index"]), int(it["last_index"]), int(it["n_msgs"]))
    return out


def compare_items(engine_items, oracle_msgs):
    a, b = items_dict(engine_items), coalesce_oracle(oracle_msgs)
    for k in set(a) | set(b):
        x, y = a.get(k), b.get(k)
        if x is not None and y is not None and x[0] == O.SEND_SNAPSHOT and y[0] == O.SEND_SNAPSHOT:
            assert x[1] == y[1], (k, x, y)  # a snapshot item carries the requested index in last_index
            continue
        assert x == y, (k, "engine", x, "oracle", y)
    return a


def prepare_msgs(msgs):
    """Device-Inflights mode: no host SENT events, no host ins_full bits."""
    msgs["m_flags"][...] &= np.uint8(~(fuzz.MF_INS_FULL | fuzz.MF_SENT) & 0xff)


def ring_contents(meta, ring, g, p, cap):
    m = int(meta[p, g])
    start, count = m & 0xffff, m >> 16
    return [int(ring[g, p, (start + i) % cap]) for i in range(count)]


def compare_rings(cl, meta, ring, st, cap):
    """Every Progress in Replicate: identical Inflights contents (rings of the others are empty in the reference;
    the engine clears them lazily, the first time the send stage looks at the peer)."""
    G, P = st["n_groups"], st["n_slots"]
    present = (st["cfg"] >> 24) & 0xff
    for g in range(G):
        for p in range(P):
            if not (int(present[g]) >> p) & 1:
                continue
            want = cl.ins_contents(g, p + 1)
            if (int(st["pflags"][g, p]) & 3) != O.REPLICATE:
                assert want == [], (g, p, want)
                continue
            got = ring_contents(meta, ring, g, p, cap)
            assert got == want, (g, p, got, want)


def mark_pending_conf(rng, st, frac=0.15):
    """Raft::has_pending_conf() for a random subset of the groups: flag bit 0x20 on the leader's own slot."""
    G = st["n_groups"]
    self_slot = ((st["cfg"] >> 16) & 7).astype(np.int64)
    pick = np.nonzero(rng.random(G) < frac)[0]
    st["pflags"][pick, self_slot[pick]] |= 0x20
