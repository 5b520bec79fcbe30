"""GPU parity at BASELINE.json's FULL sizes: every group of every 1 M-group configuration against the oracle
(all host cores, ro_tick_soa_mt), every column and the result word, several ticks of the device-generated stream
the bench replays -- plus the reference's 80 commit / group-commit golden vectors through the engine's kernels.

  config 2  1 000 000 x 5, majority                         (bench.py default)
  config 3  1 000 000 x 5 slots, joint {0,1,2}&&{1,2,3} + learners
  config 4  one rank's shard of 8 M x 7: 1 000 000 x 7, majority
  config 5  1 000 000 mixed 3/5/7 + leader-term rollover, both layouts: one 7-slot engine, and one engine per
            replica-set size (what bench.py --workload 5 runs)
"""
import json
import os

import numpy as np
import pytest

import fuzz
import oracle_lib as O

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
TERM = 5  # RG_WL_TERM0 of the generator
MSG_KEYS = ("m_index", "m_commit", "m_hint", "m_rs")


def _run_full_size(rg, workload, n_groups, n_slots, ticks, first_group=0, fixed_peers=0, variant=0, placed=False, group_commit=False,
                   place_after_load=False):
    import torch
    threads = os.cpu_count() or 8
    eng = rg.Engine(n_groups, n_slots, variant=variant)
    eng.workload_init(workload, first_group=first_group, fixed_peers=fixed_peers, sorted_classes=placed and not place_after_load,
                      group_commit=group_commit)
    if place_after_load:  # loaded in id order (sizes interleaved), then re-placed on the device: rg_plan_placement + rg_permute_groups
        assert placed and eng.size_classes() == []
        eng.place_by_size_class()
    if placed:  # three ranges, one launch per tick (k_tick_classes)
        assert [q for _, _, q in eng.size_classes()] == [3, 5, 7]
    st = eng.read_state()
    cl = O.Cluster(n_groups)
    cl.load_soa(st, term=TERM)
    dev = [torch.zeros((n_slots, eng.stride), dtype=torch.int64, device="cuda") for _ in range(4)]
    dflags = torch.zeros((n_groups, 8), dtype=torch.uint8, device="cuda")
    msgs = {"n_groups": n_groups, "n_slots": n_slots, "stride": eng.stride}
    gout = np.zeros(n_groups, dtype=np.uint32)
    seen = {"changed": 0, "rejects": 0, "elections": 0, "valid": 0}
    for t in range(ticks):
        eng.workload_gen(workload, t, *[d.data_ptr() for d in dev], dflags.data_ptr(), first_group=first_group,
                         fixed_peers=fixed_peers, sorted_classes=placed, group_commit=group_commit)
        eng.sync()
        for k, d in zip(MSG_KEYS, dev):
            msgs[k] = np.ascontiguousarray(d.cpu().numpy().view(np.uint64))
        msgs["m_flags"] = np.ascontiguousarray(dflags.cpu().numpy())
        eng.tick_device(*[d.data_ptr() for d in dev], dflags.data_ptr())
        stepped = cl.tick_soa_mt(msgs, gout, threads)
        got = eng.read_state()
        cl.store_soa(st)
        # ALL groups, all columns: a stride / tail / tile bug anywhere in the 1 M groups fails here
        for k in fuzz.STATE_KEYS:
            if st[k].ndim == 2 and k != "pflags":
                same = np.array_equal(st[k][:, :n_groups], got[k][:, :n_groups])
            else:
                same = np.array_equal(st[k], got[k])
            if not same:
                diffs = fuzz.diff_states(st, got, n_groups, n_slots, keys=(k,))
                # (cells of slots without a Progress are not state: diff_states masks them)
                assert not diffs, f"workload {workload} {n_groups}x{n_slots} tick {t}: " + "; ".join(diffs[:8])
        bad = np.nonzero(got["out"] != gout)[0]
        assert bad.size == 0, (workload, t, bad[:5], [hex(x) for x in got["out"][bad[:5]]], [hex(x) for x in gout[bad[:5]]])
        assert not (gout & 2).any(), "well-formed stream: no fault"
        seen["changed"] += int((gout & 1).sum())
        seen["valid"] += int(stepped)
        seen["rejects"] += int(((msgs["m_flags"][:, 1:] & 3) == 3).sum())
        seen["elections"] += int(((gout & 0x10) != 0).sum())
    commit, out = eng.results()
    assert np.array_equal(commit, st["commit"]) and np.array_equal(out, gout)
    eng.close()
    return seen


@pytest.mark.parametrize("workload,n_slots,name", [(2, 5, "config 2"), (3, 5, "config 3"), (2, 7, "config 4 shard")])
def test_one_million_groups_match_the_oracle(rg, workload, n_slots, name):
    seen = _run_full_size(rg, workload, 1_000_000, n_slots, ticks=4)
    assert seen["changed"] > 2_500_000 and seen["valid"] > 4 * 1_000_000 * (n_slots - 1) * 0.9, (name, seen)


@pytest.mark.parametrize("workload,n_slots,name", [(2, 5, "config 2"), (3, 5, "config 3"), (2, 7, "config 4 shard")])
def test_one_million_groups_with_group_commit_match_the_oracle(rg, workload, n_slots, name):
    """The same streams with ProgressTracker.group_commit on in every group and three commit groups over the peers
    (RG_WL_GROUP_COMMIT; Raft::enable_group_commit + assign_commit_groups, src/raft.rs:513-544): every commit evaluation of
    every group is the group-commit form (src/quorum/majority.rs:99-123; joint: min of the two, AND of the flags,
    joint.rs:47-51) -- 1 M groups, every column and result word against the oracle. The kernels are the GC = true
    instantiations (no scratch since round 6: tests/test_kernel_resources.py)."""
    import raft_rs_amd as R
    eng = R.Engine(1000, n_slots)
    eng.workload_init(workload, group_commit=True)
    st = eng.read_state()
    eng.close()
    used = st["cfg"][:1000]
    assert ((used >> 19) & 1).all() and set(np.unique(st["gid"][:n_slots if workload == 2 else 4, :1000])) <= {1, 2, 3}
    seen = _run_full_size(rg, workload, 1_000_000, n_slots, ticks=4, group_commit=True)
    # (a commit now waits for the slowest commit group: fewer groups move per tick than under the plain quorum, most still do)
    assert seen["changed"] > 1_000_000 and seen["valid"] > 4 * 1_000_000 * (n_slots - 1) * 0.9, (name, seen)


def test_config5_one_engine_matches_the_oracle(rg):
    """1 M groups of 3 / 5 / 7 peers interleaved in one 7-slot engine, elections on every tick."""
    seen = _run_full_size(rg, 5, 1_000_000, 7, ticks=5)
    assert seen["elections"] > 5 * 1_000_000 / 32 * 0.9 and seen["rejects"] > 300_000, seen


def test_config5_placed_by_size_class_one_launch_matches_the_oracle(rg):
    """bench.py --workload 5 --slots 7 --sorted: the same 1 M groups placed by replica-set size class in ONE 7-slot engine,
    one launch per tick whose blocks run the tick instantiated for 3, 5 or 7 slots (k_tick_classes): every group, every
    column, five ticks of the rollover stream."""
    seen = _run_full_size(rg, 5, 1_000_000, 7, ticks=5, placed=True)
    assert seen["elections"] > 5 * 1_000_000 / 32 * 0.9 and seen["rejects"] > 300_000, seen


def test_config5_loaded_interleaved_then_placed_matches_the_oracle(rg):
    """What a real host does: the 1 M groups arrive in id order (sizes interleaved), rg_plan_placement + rg_permute_groups
    re-place them by size class on the device, and from then on the shard runs the one-launch class kernel -- every group, every
    column, five ticks of the rollover stream against the oracle (whose state is the engine's after the permutation: the gather
    itself is checked column by column in tests/test_placement_gpu.py)."""
    seen = _run_full_size(rg, 5, 1_000_000, 7, ticks=5, placed=True, place_after_load=True)
    assert seen["elections"] > 5 * 1_000_000 / 32 * 0.9 and seen["rejects"] > 300_000, seen


def test_config5_size_class_engines_match_the_oracle(rg):
    """bench.py --workload 5: one engine per replica-set size, global group ids as the bench assigns them."""
    G = 1_000_000
    first, total = 0, {"elections": 0, "rejects": 0}
    for slots, n in ((3, G // 3), (5, G // 3), (7, G - 2 * (G // 3))):
        seen = _run_full_size(rg, 5, n, slots, ticks=5, first_group=first, fixed_peers=slots)
        first += n
        for k in total:
            total[k] += seen[k]
    assert total["elections"] > 5 * G / 32 * 0.9 and total["rejects"] > 300_000, total


@pytest.mark.parametrize("layout", ["one-engine", "size-classes"])
def test_config5_compact_variant_full_size(rg, layout):
    """RG_VARIANT_COMPACT (rare groups gathered into one wave per workgroup through LDS) over config 5 at full size:
    the groups change lanes inside the kernel, the results must not."""
    G = 1_000_000
    if layout == "one-engine":
        seen = _run_full_size(rg, 5, G, 7, ticks=4, variant=5)
    else:
        first, seen = 0, {"elections": 0, "rejects": 0}
        for slots, n in ((3, G // 3), (5, G // 3), (7, G - 2 * (G // 3))):
            part = _run_full_size(rg, 5, n, slots, ticks=4, first_group=first, fixed_peers=slots, variant=5)
            first += n
            for k in seen:
                seen[k] += part[k]
    assert seen["elections"] > 4 * G / 32 * 0.9 and seen["rejects"] > 200_000, seen


def test_config2_compact_variant_full_size(rg):
    """... and over the steady stream, where no group changes lanes."""
    _run_full_size(rg, 2, 1_000_000, 5, ticks=2, variant=5)


def test_config2_lds_variant_full_size(rg):
    """The LDS-staged kernel over the same 1 M x 5 stream (two ticks)."""
    _run_full_size(rg, 2, 1_000_000, 5, ticks=2, variant=2)


# ---------------------------------------------------------------------------------------------------------------
# beyond the Infinity Cache: the sizes bench.py reports its HBM-regime figures at, each in the memory regime the engine
# picks for it BY ITSELF (rg_config.cache_policy = auto; rg_get_device_info says which kernel ran) -- against the oracle on
# contiguous sub-ranges (groups are independent: the oracle needs only the groups it is compared on, which bounds host
# memory): the head of the shard, the range that straddles the regime's internal boundary (the end of the cache-resident
# range of k_tick_split; a size-class boundary of k_tick_classes), the tail with the last, partial workgroup.
# ---------------------------------------------------------------------------------------------------------------
PER_SLOT = ("match", "next", "pr_commit", "pend_snap", "pend_rs", "gid")
PER_GROUP = ("commit", "term_lo", "term_hi", "cfg")


def _slice_state(st, a, n, n_slots):
    sub = O.alloc_state(n, n_slots)
    for k in PER_SLOT:
        sub[k][:, :n] = st[k][:, a:a + n]
    sub["pflags"][:] = st["pflags"][a:a + n]
    for k in PER_GROUP:
        sub[k][:] = st[k][a:a + n]
    if "out" in st:
        sub["out"] = st["out"][a:a + n].copy()
    if "run_first" in st:  # the term-run table and the log's dummy entry (the send decision reads first_index = dummy + 1)
        O.add_term_table(sub)
        for k in ("run_first", "run_term"):
            sub[k][:, :n] = st[k][:, a:a + n]
        for k in ("dummy_index", "dummy_term", "cur_term"):
            sub[k][:] = st[k][a:a + n]
    return sub


def _slice_msgs(msgs, a, n, n_slots):
    sub = O.alloc_msgs(n, n_slots)
    for k in MSG_KEYS:
        sub[k][:, :n] = msgs[k][:, a:a + n]
    sub["m_flags"][:] = msgs["m_flags"][a:a + n]
    del sub["m_logterm"]
    return sub


def _run_sub_ranges(rg, eng, workload, ranges, ticks, expect, placed=False, first_group=0):
    """`ticks` ticks of the device-generated stream over the WHOLE engine; every column and the result word of the groups in
    `ranges` ([(first, n), ...]) against the oracle after every tick. `expect`: what rg_get_device_info must report."""
    import torch
    threads = os.cpu_count() or 8
    G, P = eng.n_groups, eng.n_slots
    st = eng.read_state()
    subs = []
    for a, n in ranges:
        assert 0 <= a and a + n <= G, (a, n, G)
        sub = _slice_state(st, a, n, P)
        cl = O.Cluster(n)
        cl.load_soa(sub, term=TERM)
        subs.append((a, n, sub, cl, np.zeros(n, dtype=np.uint32)))
    del st
    dev = [torch.zeros((P, eng.stride), dtype=torch.int64, device="cuda") for _ in range(4)]
    dflags = torch.zeros((G, 8), dtype=torch.uint8, device="cuda")
    seen = {"changed": 0, "rejects": 0, "elections": 0, "valid": 0, "groups": sum(n for _, n in ranges)}
    for t in range(ticks):
        eng.workload_gen(workload, t, *[d.data_ptr() for d in dev], dflags.data_ptr(), first_group=first_group, sorted_classes=placed)
        eng.sync()
        msgs = {"n_groups": G, "n_slots": P, "stride": eng.stride}
        for k, d in zip(MSG_KEYS, dev):
            msgs[k] = d.cpu().numpy().view(np.uint64)
        msgs["m_flags"] = dflags.cpu().numpy()
        eng.tick_device(*[d.data_ptr() for d in dev], dflags.data_ptr())
        got = eng.read_state()
        info = eng.device_info()
        for k, v in expect.items():
            assert info[k] == v, (k, info)
        for a, n, sub, cl, gout in subs:
            m = _slice_msgs(msgs, a, n, P)
            seen["valid"] += int(cl.tick_soa_mt(m, gout, threads))
            cl.store_soa(sub)
            diffs = fuzz.diff_states(sub, _slice_state(got, a, n, P), n, P)
            assert not diffs, f"workload {workload} {G}x{P} groups [{a}, {a + n}) tick {t}: " + "; ".join(diffs[:8])
            bad = np.nonzero(got["out"][a:a + n] != gout)[0]
            assert bad.size == 0, (workload, t, a, bad[:5], [hex(x) for x in got["out"][a:a + n][bad[:5]]], [hex(x) for x in gout[bad[:5]]])
            assert not (gout & 2).any(), "well-formed stream: no fault"
            seen["changed"] += int((gout & 1).sum())
            seen["rejects"] += int(((m["m_flags"][:, 1:] & 3) == 3).sum())
            seen["elections"] += int(((gout & 0x10) != 0).sum())
        assert not (got["out"] & 2).any(), "well-formed stream: no fault anywhere in the shard"
        del got, msgs
    return seen


SUB = 256 * 1024  # groups per compared sub-range


def test_eight_million_groups_all_streamed_match_the_oracle(rg):
    """8 M x 5 (bench.py's out_of_cache figure): 1.28 GB of state, 5 x the Infinity Cache -- the engine streams everything
    (k_tick_lane<5, false, u32, 2>), by its own rule."""
    G, P = 8_000_000, 5
    eng = rg.Engine(G, P)
    assert eng.device_info()["cache_policy"] == "stream_all"
    eng.workload_init(2)
    mid = (G // 2 // 256) * 256 - SUB // 2 + 77  # (not block-aligned on purpose)
    seen = _run_sub_ranges(rg, eng, 2, [(0, SUB), (mid, SUB), (G - SUB, SUB)], 3,
                           {"cache_policy": "stream_all", "last_tick_kernel": "k_tick_lane", "last_tick_streaming": 2})
    eng.close()
    assert seen["changed"] > 0.6 * 3 * seen["groups"] and seen["valid"] > 3 * seen["groups"] * 4 * 0.9, seen


def test_between_the_regimes_partly_resident_matches_the_oracle(rg):
    """2.4 M x 5 (bench.py's between_regimes figure): 384 MB of state, 1.4 x the cache -- an engine that is alone on its device
    keeps a leading range resident and streams the rest in ONE launch (k_tick_split); the compared ranges are the head, the
    256 k groups around the end of the resident range, and the tail."""
    import gc
    gc.collect()
    G, P = 2_400_000, 5
    eng = rg.Engine(G, P)
    info = eng.device_info()
    if info["engines_on_device"] == 1:
        assert info["cache_policy"] == "resident", info  # the rule
    elif info["cache_policy"] != "resident":  # (an engine some other test leaked: ask for the regime instead of being granted it)
        eng.close()
        eng = rg.Engine(G, P, cache_policy=rg.CACHE.RESIDENT)
        info = eng.device_info()
    r = info["resident_groups"]
    assert r % 64 == 0 and 256 * 1024 < r < G - 256 * 1024, info  # (whole workgroups of 64 groups)
    assert abs(r * (24 * P + 40) - 176 * 2**20) < 64 * (24 * P + 40), info  # 176 MB of state
    eng.workload_init(2)
    seen = _run_sub_ranges(rg, eng, 2, [(0, SUB), (r - SUB // 2, SUB), (G - SUB, SUB)], 3,
                           {"cache_policy": "resident", "last_tick_kernel": "k_tick_split", "last_tick_streaming": 2, "resident_groups": r})
    eng.close()
    assert seen["changed"] > 0.6 * 3 * seen["groups"], seen


def test_config5_eight_million_groups_class_placed_match_the_oracle(rg):
    """BASELINE config 5 at 8 M groups, placed by replica-set size class in one 7-slot engine: one launch per tick
    (k_tick_classes<7, u32, 2>, everything streamed), elections and rejects on every tick; compared: the head (3 peers), the
    ranges around both class boundaries (3 -> 5 and 5 -> 7 peers), the tail (7 peers)."""
    G, P = 8_000_000, 7
    eng = rg.Engine(G, P)
    assert eng.device_info()["cache_policy"] == "stream_all"
    eng.workload_init(5, sorted_classes=True)
    cls = eng.size_classes()
    assert [q for _, _, q in cls] == [3, 5, 7], cls
    b35, b57 = cls[1][0], cls[2][0]
    half = SUB // 2
    seen = _run_sub_ranges(rg, eng, 5, [(0, half), (b35 - half, SUB), (b57 - half, SUB), (G - half, half)], 3,
                           {"cache_policy": "stream_all", "last_tick_kernel": "k_tick_classes", "last_tick_streaming": 2}, placed=True)
    eng.close()
    assert seen["elections"] > 3 * seen["groups"] / 32 * 0.8 and seen["rejects"] > 0.05 * 3 * seen["groups"], seen


# ---------------------------------------------------------------------------------------------------------------
# maximum sizes: the two engines either side of the 4 GiB column boundary -- the LAST shard whose kernels address their cells
# with 32-bit offsets (67 108 608 groups: slot / run 7 of the last group sits 2 KiB below 4 GiB) and the FIRST one that takes
# the 64-bit-offset instantiations by its size alone (67 108 864 groups; every other test reaches those kernels through
# RG_CFGF_IX64 on small engines). 35 GB of engine + 11 GB of messages on the device, so nothing of that size crosses to the
# host: the compared sub-ranges are REPLICAS kept on the host -- the generator's host twin lays out the groups [a, a + n) of
# the shard and produces their messages tick by tick from the replica's own state (the generator is a function of the global
# group id and the group's state: tests/test_workload_host.py, test_device_generator_equals_host_generator), the oracle steps
# them, and the device's groups are read back through rg_read_groups.
# ---------------------------------------------------------------------------------------------------------------
def _status_as_state(rows, n, n_slots):
    got = O.alloc_state(n, n_slots)
    for k in ("match", "next", "pr_commit", "pend_snap", "pend_rs"):
        got[k][:, :n] = rows[k][:, :n_slots].T
    got["pflags"][:] = rows["pflags"]
    got["commit"][:] = rows["commit"]
    got["term_lo"][:] = rows["term_lo"]
    got["term_hi"][:] = rows["last_index"]
    got["cfg"][:] = rows["cfg"]
    return got


@pytest.mark.parametrize("bits", [32, 64])
def test_shards_either_side_of_the_four_gib_column_boundary_match_the_oracle(rg, bits):
    import torch
    from raft_rs_amd import engine as E
    P, TICKS, N = 5, 3, 128 * 1024
    rows = max(P, E.TERM_RUNS)
    last32 = (0xFFFFFFFF // (8 * rows)) // 256 * 256  # the largest stride whose farthest cell is below 4 GiB
    G = last32 if bits == 32 else last32 + 256
    assert (rows * G * 8 <= 0xFFFFFFFF) == (bits == 32)
    free, _ = torch.cuda.mem_get_info()
    if free < 80 * 2**30:
        pytest.skip(f"needs ~50 GB of device memory, {free >> 30} GiB free")
    threads = os.cpu_count() or 8
    eng = rg.Engine(G, P)
    assert eng.stride == G and eng.device_info()["engine_bytes"] > 30 * 2**30
    eng.workload_init(2)
    # the head, a range in the middle (slot 4's cells there lie either side of byte offset 2 GiB + ...: a sign-extended 32-bit
    # offset would land elsewhere), two near the end -- not block-aligned -- and the tail: the largest offsets the kernels form
    mid = (G // 2 // 256) * 256 + 77
    ranges = [(0, N), (mid, N), (G - 3 * N - 131, N), (G - N, N)]
    reps = []
    for a, n in ranges:
        sub = O.alloc_state(n, P)
        E.workload_init_host(sub, 2, first_group=a)
        cl = O.Cluster(n)
        cl.load_soa(sub, term=TERM)
        reps.append((a, n, sub, cl, rg.MsgBuffers(n, P, sub["stride"]), np.zeros(n, dtype=np.uint32)))
        got = _status_as_state(eng.read_groups(np.arange(a, a + n, dtype=np.uint64)), n, P)
        assert not fuzz.diff_states(sub, got, n, P, keys=[k for k in fuzz.STATE_KEYS if k != "gid"]), ("initial state", a)
    dev = [torch.zeros((P, eng.stride), dtype=torch.int64, device="cuda") for _ in range(4)]
    dflags = torch.zeros((G, 8), dtype=torch.uint8, device="cuda")
    changed = 0
    for t in range(TICKS):
        eng.workload_gen(2, t, *[d.data_ptr() for d in dev], dflags.data_ptr())
        eng.tick_device(*[d.data_ptr() for d in dev], dflags.data_ptr())
        eng.sync()
        info = eng.device_info()
        assert info["last_tick_kernel"] == "k_tick_lane" and info["last_tick_offset_bits"] == bits, info
        for a, n, sub, cl, host, gout in reps:
            E.workload_gen_host(sub, host, 2, t, first_group=a)
            # (the device's message cells of this range are what the host twin wrote: the first and the last slot's rows)
            for p in (0, P - 1):
                assert np.array_equal(dev[0][p, a:a + n].cpu().numpy().view(np.uint64), host.m_index[p, :n]), (t, a, p)
            m = host.as_dict()
            del m["m_logterm"]
            cl.tick_soa_mt(m, gout, threads)
            cl.store_soa(sub)
            rows_ = eng.read_groups(np.arange(a, a + n, dtype=np.uint64))
            diffs = fuzz.diff_states(sub, _status_as_state(rows_, n, P), n, P, keys=[k for k in fuzz.STATE_KEYS if k != "gid"])
            assert not diffs, f"{G} x {P} ({bits}-bit offsets) groups [{a}, {a + n}) tick {t}: " + "; ".join(diffs[:8])
            bad = np.nonzero(rows_["out"] != gout)[0]
            assert bad.size == 0, (t, a, bad[:5], [hex(x) for x in rows_["out"][bad[:5]]], [hex(x) for x in gout[bad[:5]]])
            assert not (gout & 2).any()
            changed += int((gout & 1).sum())
    # ... and nothing faulted anywhere in the 67 M groups (counted on the device: rg_result_counts)
    n_changed, n_fault = eng.result_counts()
    assert n_fault == 0 and n_changed > 0.6 * G, (n_changed, n_fault)
    eng.close()
    del dev, dflags
    torch.cuda.empty_cache()
    assert changed > 0.6 * TICKS * len(ranges) * N, changed


# ---------------------------------------------------------------------------------------------------------------
# the send stage at the size bench.py measures it: 1 M x 5, Inflights of capacity 256 on the device, both forms
# ---------------------------------------------------------------------------------------------------------------
def _items_by_key(items, a, n):
    """Engine work items of groups [a, a + n), sorted by (group, slot): (key, kind, prev, last, n_msgs)."""
    sel = items[(items["group"] >= a) & (items["group"] < a + n)]
    key = (sel["group"].astype(np.int64) - a) * 8 + sel["slot"].astype(np.int64)
    o = np.argsort(key, kind="stable")
    return key[o], sel["kind"][o].astype(np.int64), sel["prev_index"][o], sel["last_index"][o], sel["n_msgs"][o].astype(np.int64)


@pytest.mark.parametrize("form", ["rg_send_appends", "rg_tick_device_send"])
def test_send_stage_one_million_groups_matches_the_oracle(rg, form):
    """BASELINE config 2 with the Inflights (cap 256) on the device, as bench.py's send-stage lines run it: three ticks of the
    device-generated stream (no host SENT / ins_full events), the stage after every tick as a launch of its own or in the
    tick's launch. EVERY group: result word, every state column, every work item (kind, prev_index, last_index, n_msgs) after
    every tick, and after the last one every window (count and contents) -- against the oracle with its own Inflights
    (src/tracker/inflights.rs:65-110, src/raft.rs:773-819). The oracle replays the recorded ticks chunk by chunk (125 k groups:
    its per-Progress rings of 256 entries are what bounds host memory)."""
    import torch
    G, P, CAP, TICKS, CHUNK = 1_000_000, 5, 256, 3, 125_000
    threads = os.cpu_count() or 8
    eng = rg.Engine(G, P, max_inflight=CAP)
    eng.workload_init(2)
    st0 = eng.read_state()
    # the log as the engine holds it: nothing compacted (dummy index 0), one run [term_lo, last_index] of the leader's term --
    # without the table the SoA adapter would take term_lo - 1 for the dummy entry and every lagging peer for a snapshot case
    for col in (rg.COL.RUN_FIRST, rg.COL.RUN_TERM, rg.COL.DUMMY_INDEX, rg.COL.DUMMY_TERM, rg.COL.CUR_TERM):
        st0[rg.COL.NAMES[col]] = eng.read_column(col)
    assert (st0["cur_term"] == TERM).all() and not st0["dummy_index"].any()
    dev = [torch.zeros((P, eng.stride), dtype=torch.int64, device="cuda") for _ in range(4)]
    dflags = torch.zeros((G, 8), dtype=torch.uint8, device="cuda")
    rec = []  # per tick: messages, the engine's state, result words and work items
    for t in range(TICKS):
        eng.workload_gen(2, t, *[d.data_ptr() for d in dev], dflags.data_ptr())
        dflags &= 0xE7  # no RG_MF_SENT / RG_MF_INS_FULL: the device owns the send path
        eng.sync()
        msgs = {"n_groups": G, "n_slots": P, "stride": eng.stride}
        for k, d in zip(MSG_KEYS, dev):
            msgs[k] = d.cpu().numpy().view(np.uint64)
        msgs["m_flags"] = dflags.cpu().numpy()
        if form == "rg_tick_device_send":
            eng.tick_device_send(*[d.data_ptr() for d in dev], dflags.data_ptr(), max_entries_per_msg=0)
            assert eng.device_info()["last_tick_kernel"] == "k_tick_send"
        else:
            eng.tick_device(*[d.data_ptr() for d in dev], dflags.data_ptr())
            eng.send_appends(0)
        items = eng.send_items().copy()
        got = eng.read_state()
        rec.append((msgs, got, items))
        assert not (got["out"] & 2).any()
        assert len(items) > 3_000_000, len(items)  # (~3.9 M work items per tick)
        assert (items["kind"] == O.SEND_APPEND).all()
    meta, ring = eng.read_inflights()
    eng.close()
    cnt_e = (meta >> 16).astype(np.int64)
    start_e = (meta & 0xffff).astype(np.int64)
    K = 16
    n_items = 0
    for a in range(0, G, CHUNK):
        n = min(CHUNK, G - a)
        sub = _slice_state(st0, a, n, P)
        cl = O.Cluster(n)
        cl.load_soa(sub, term=TERM, max_inflight=CAP)
        cl.set_own_inflights(True)
        gout = np.zeros(n, dtype=np.uint32)
        for t, (msgs, got, items) in enumerate(rec):
            m = _slice_msgs(msgs, a, n, P)
            cl.tick_soa_mt(m, gout, threads)
            assert np.array_equal(got["out"][a:a + n], gout), (form, t, a)
            om = cl.send_stage_soa(gout, 0, capacity=n * P)
            cl.store_soa(sub)
            diffs = fuzz.diff_states(sub, _slice_state(got, a, n, P), n, P)
            assert not diffs, f"{form}: groups [{a}, {a + n}) tick {t}: " + "; ".join(diffs[:8])
            # the work items: one per (group, peer) on both sides (no entry limit: one message takes everything)
            okey = om["group"].astype(np.int64) * 8 + om["to"].astype(np.int64) - 1
            o = np.argsort(okey, kind="stable")
            okey, om = okey[o], om[o]
            assert (np.diff(okey) > 0).all(), "one message per peer"
            ekey, ekind, eprev, elast, en = _items_by_key(items, a, n)
            assert np.array_equal(ekey, okey), (form, t, a, len(ekey), len(okey))
            assert np.array_equal(ekind, om["kind"].astype(np.int64)) and (en == 1).all()
            assert np.array_equal(eprev, om["index"]) and np.array_equal(elast, om["index"] + om["n_entries"]), (form, t, a)
            n_items += len(okey)
        # the windows after the last tick: counts of every Progress, contents of every Replicate one
        counts, first_k, mx = cl.ins_export(P, sub["stride"], K)
        assert mx <= K, mx
        present = np.stack([((sub["cfg"] >> 24) >> p) & 1 for p in range(P)], axis=0).astype(bool)
        repl = ((sub["pflags"][:, :P].T & 3) == O.REPLICATE) & present
        ce, se = cnt_e[:, a:a + n], start_e[:, a:a + n]
        assert np.array_equal(np.where(repl, ce, 0), np.where(repl, counts[:, :n].astype(np.int64), 0)), (form, a)
        gi = np.arange(n)[None, :].repeat(P, 0)
        pi = np.arange(P)[:, None].repeat(n, 1)
        for i in range(K):
            live = repl & (ce > i)
            if not live.any():
                break
            e = ring[a:a + n][gi[live], pi[live], (se[live] + i) % CAP]
            assert np.array_equal(e, first_k[gi[live], pi[live], i]), (form, a, i)
    assert n_items > 3 * 3_000_000, n_items


# ---------------------------------------------------------------------------------------------------------------
# the reference's commit golden vectors (src/quorum/testdata/{majority_commit,joint_commit,joint_group_commit}.txt)
# ---------------------------------------------------------------------------------------------------------------
VEC = json.load(open(os.path.join(HERE, "golden", "quorum_vectors.json"), encoding="utf-8"))


def commit_cases():
    from test_oracle_golden import build_case, parse_result
    cases = []
    for fname, gc in (("majority_commit.txt", False), ("joint_commit.txt", False), ("joint_group_commit.txt", True)):
        for case in VEC[fname]:
            ids, idsj, joint, look = build_case(case["args"])
            cases.append((f"{fname}:{case['line']}", ids, idsj, look, gc, parse_result(case["result"])))
    return cases


@pytest.mark.parametrize("variant", [0, 1, 3])
def test_commit_golden_vectors_on_the_gpu(rg, variant):
    """All 16 + 50 + 14 `committed` / `group_committed` cases, one raft group per case, through
    rg_maximal_committed_index (variant 0: k_recompute2, two groups per lane; 1: k_recompute, one group per lane;
    3: the wave-cooperative rank select, for the cases without group commit) and through rg_recompute (Raft::maybe_commit: the same index, gated by the log)."""
    cases = commit_cases()
    assert len(cases) == 80
    if variant == 3:
        cases = [c for c in cases if not c[4]]  # the cooperative kernel has no group-commit path
    G, P = len(cases), 8
    st = O.alloc_state(G, P)
    for g, (_, ids, idsj, look, gc, _) in enumerate(cases):
        slot = {pid: k for k, pid in enumerate(sorted(set(ids) | set(idsj)))}
        m = lambda s: sum(1 << slot[i] for i in s)
        # a voter without an acked index has no Progress: acked_index() -> None -> {0, 0} (majority.rs:80-82)
        st["cfg"][g] = rg.cfg_make(m(ids), m(idsj), 0, group_commit=gc, present=m(look.keys()))
        for pid, (idx, gid) in look.items():
            st["match"][slot[pid], g] = idx
            st["next"][slot[pid], g] = idx + 1
            st["gid"][slot[pid], g] = gid
        st["term_lo"][g], st["term_hi"][g] = 1, 1 << 40  # every index is of the leader's term
    eng = rg.Engine(G, P, variant=variant)
    eng.load_state(st)
    mci, used = eng.maximal_committed_index(with_flag=True)
    for g, (name, *_rest, want) in enumerate(cases):
        assert int(mci[g]) == want, f"{name}: engine {int(mci[g])}, reference {want}"
    # joint symmetry (datadriven_test.rs:176-181): swap the two majorities
    cfg2 = (((st["cfg"] & 0xff) << 8) | ((st["cfg"] >> 8) & 0xff) | (st["cfg"] & 0xffff0000)).astype(np.uint32)
    eng.load_column(rg.COL.CFG, cfg2)
    assert np.array_equal(eng.maximal_committed_index(), mci)
    eng.load_column(rg.COL.CFG, st["cfg"])
    # Raft::maybe_commit on the same groups: commit = mci where the log holds it (mci <= last_index), else unchanged
    eng.recompute()
    commit, out = eng.results()
    for g, (name, *_rest, want) in enumerate(cases):
        exp = want if 0 < want <= (1 << 40) else 0
        assert int(commit[g]) == exp and bool(out[g] & 1) == (exp > 0), name
    eng.close()
