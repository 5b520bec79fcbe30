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


INS_COMPACT, INS_DBITS, INS_DMASK = 0xffff, 21, (1 << 21) - 1  # rg_send.h: RG_INS_COMPACT, RG_INS_DBITS, RG_INS_DMASK


def ring_contents(meta, ring, g, p, cap):
    m = int(meta[p, g])
    start, count = m & 0xffff, m >> 16
    if start == INS_COMPACT:  # the engine-internal form of a window of <= 4 entries (patch_ring has written them out at 0)
        start = 0
    return [int(ring[g, p, (start + i) % cap]) for i in range(count)]


def compact_entries(hd, tail, count):
    """The entries of a compact window, OLDEST first: newest first they are tail, tail - d1, tail - d1 - d2, ... with the
    distances d1 | d2 << 21 | d3 << 42 in the window's `head` cell (rg_send.h)."""
    e, out = int(tail), []
    for i in range(count):
        out.append(e & ((1 << 64) - 1))
        e -= (int(hd) >> (INS_DBITS * i)) & INS_DMASK
    return out[::-1]


def patch_ring(meta, head, tail, ring, cap, G, P):
    """What rg_read_inflights does with the engine-internal columns (host_check twins work on them directly): the oldest
    and the newest entry of a ring window live in the head / tail columns, a compact window lives there entirely -- write
    them into the ring so that ring_contents sees every window whole."""
    for p in range(P):
        m = meta[p, :G]
        live = np.nonzero(m >> 16)[0]
        start, count = (m[live] & 0xffff).astype(np.int64), (m[live] >> 16).astype(np.int64)
        rw = start != INS_COMPACT
        ring[live[rw], p, start[rw]] = head[p, live[rw]]
        ring[live[rw], p, (start[rw] + count[rw] - 1) % cap] = tail[p, live[rw]]
        for g, c in zip(live[~rw].tolist(), count[~rw].tolist()):
            assert 1 <= c <= 4 and c <= cap, (g, p, c)
            ring[g, p, :c] = compact_entries(head[p, g], tail[p, g], c)


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


# ---- byte-accurate Config::max_size_per_msg (RG_SEND_BYTES) ------------------------------------------------
def entry_sizes(rng, G, n_index, zero_frac=0.1, hi=700):
    """Entry::compute_size() of entry `i` of group g -> sizes[g, i] (index 0 = no entry, size 0); a tenth of the entries are
    empty (Entry::default(), the case util::limit_size's `size == 0` test special-cases). Returns (sizes u32, cum u64)."""
    sizes = rng.integers(1, hi, size=(G, n_index), dtype=np.uint32)
    sizes[rng.random((G, n_index)) < zero_frac] = 0
    sizes[:, 0] = 0
    return sizes, np.cumsum(sizes.astype(np.uint64), axis=1)


def fill_size_window(esz, cum, last_index):
    """What the host keeps on the device: the cumulative sizes of (last_index - W, last_index] of every group."""
    G, W = esz.shape
    g = np.arange(G)
    for k in range(W):
        idx = last_index.astype(np.int64) - k
        ok = idx >= 0
        esz[g[ok], idx[ok] & (W - 1)] = (cum[g[ok], idx[ok]] & 0xffffffff).astype(np.uint32)


def split_host_items(engine_items, oracle_msgs):
    """RG_SEND_HOST items (the peer needs entries outside the device's size window): the engine left the Progress alone and
    the host serves the peer. Returns (engine items without them, oracle messages without those peers, {(g, slot):
    [last index of every message the reference sent to that peer]})."""
    host = {(int(it["group"]), int(it["slot"])): it for it in engine_items if int(it["kind"]) == O.SEND_HOST}
    served = {k: [] for k in host}
    keep = np.ones(len(oracle_msgs), dtype=bool)
    for i, m in enumerate(oracle_msgs):
        key = (int(m["group"]), int(m["to"]) - 1)
        if key in host:
            assert int(m["kind"]) == O.SEND_APPEND, (key, m)
            if not served[key]:
                assert int(m["index"]) == int(host[key]["prev_index"]), (key, m, host[key])
            served[key].append(int(m["index"]) + int(m["n_entries"]))
            keep[i] = False
    for k, v in served.items():
        assert v, ("the reference sent nothing to a peer the engine handed to the host", k)
    rest = engine_items[[int(it["kind"]) != O.SEND_HOST for it in engine_items]] if len(engine_items) else engine_items
    return rest, oracle_msgs[keep], served


def host_update_state(st, meta, head, tail, ring, cap, g, p, lasts):
    """Progress::update_state(last) for messages the host sent -- what rg_update_state does on the device, over the
    numpy copies the host_check tests use (Replicate: next = last + 1, ins.add(last); Probe: paused)."""
    state = int(st["pflags"][g, p]) & 3
    if state == O.PROBE:
        st["pflags"][g, p] |= 0x4
        return
    assert state == O.REPLICATE
    m = int(meta[p, g])
    start, count = m & 0xffff, m >> 16
    for last in lasts:
        if count == cap:
            break
        st["next"][p, g] = last + 1
        last = int(last)
        if count == 0:  # an empty window starts compact
            start, head[p, g] = INS_COMPACT, 0
        elif start == INS_COMPACT and count < 4 and last - int(tail[p, g]) <= INS_DMASK:
            head[p, g] = ((int(head[p, g]) << INS_DBITS) & ((1 << 63) - 1)) | (last - int(tail[p, g]))
        else:
            if start == INS_COMPACT:  # a fifth message / a distance beyond 21 bits: the window moves to the ring (start 0)
                e = compact_entries(head[p, g], tail[p, g], count)
                for i in range(1, count - 1):
                    ring[g, p, i] = e[i]
                start, head[p, g] = 0, e[0]
            if count >= 2:
                ring[g, p, (start + count - 1) % cap] = tail[p, g]
        tail[p, g] = last
        count += 1
    meta[p, g] = start | (count << 16)
    st["pflags"][g, p] = (int(st["pflags"][g, p]) & ~PF_INS_FULL) | (PF_INS_FULL if count == cap else 0)
    # the logical view rg_read_inflights gives
    if count:
        if start == INS_COMPACT:
            ring[g, p, :count] = compact_entries(head[p, g], tail[p, g], count)
        else:
            ring[g, p, start] = head[p, g]
            ring[g, p, (start + count - 1) % cap] = tail[p, g]
