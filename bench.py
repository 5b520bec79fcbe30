#!/usr/bin/env python3
"""bench.py -- raft-group progress+commit evaluations per second on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--groups G] [--slots P] [--workload 2|3|5]
                    [--fuse T] [--split S] [--variant 0|2|4|5] [--publish-every E] [--one-engine] [--no-cpu-baseline]
                    [--inflights CAP] [--no-extras] [--c5-variant V]

The default (N = 1) line also carries, as sub-objects with their own `roofline` (each labelled with its memory regime):
`recompute_only` (+ `_out_of_cache`), `out_of_cache`, `between_regimes` (2.4 M groups), `other_configs` (every other BASELINE configuration that fits one
GPU, and the headline with the send stage), `small_batch_latency`, `cpu_baseline`.

One "step" = one tick of the hot path over every raft group of the shard: apply each group's
AppendResponse slots (Raft::handle_append_response semantics) and re-evaluate + gate the commit
index (maybe_commit), i.e. one *evaluation* per group per step.

Workload at N=1: BASELINE.json configs[1] = 1,000,000 groups x 5 peers, majority quorum, synthetic
AppendResponse stream (seed 0x5EED5EED). Weak scaling: every rank holds its own 1M-group shard
(disjoint global group ids); the only exchange is the publication of commit indices, natively behind the C ABI
(rg_comm_init / rg_publish_commit: ncclAllGather over xGMI of the ~1 B/group slices the ticks produce), after every
tick wherever an exchange fits behind a tick (`--publish-every auto`, the default: measured before the timed region,
reported as config.publish_every_auto; `--publish-every 1` forces every tick). `--gpus N --slots 7` is BASELINE
configs[3] (8 M x 7 over 8 GPUs at N = 8); `--total-groups G` is strong scaling; `python bench.py --gpus N` without a
launcher starts its own ranks.

Procedure: the W+K ticks of messages are generated on the device from the evolving state in an
untimed pass (generate -> tick -> generate ...), the engine state is restored from a checkpoint,
and the timed region replays exactly the K recorded ticks back to back -- inputs resident in HBM,
nothing but tick kernels (and, for N>1, the commit all-gather) inside the region.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

C5_VARIANT = 0  # RG_VARIANT_* the config-5 sub-measurements run (set from the measured comparison, profiles/r03_*)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (spec), /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes(n_groups, slots_present, valid_msgs, rejects):
    """B = 9*P + 58*A + 8*R + 37 bytes per evaluation, summed over the tick."""
    return 9 * slots_present + 58 * valid_msgs + 8 * rejects + 37 * n_groups


def cpu_baseline(n_groups, n_slots, workload, sample_ticks, seed, threads):
    """Time the CPU oracle (reference-equivalent, message-at-a-time, hash-map-per-group data model)
    on a bounded sample of the same stream. TEST INFRASTRUCTURE used only as a reported baseline."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from raft_rs_amd import engine as E
    st = O.alloc_state(n_groups, n_slots)
    E.workload_init_host(st, workload, seed=seed)
    cl = O.Cluster(n_groups)
    cl.load_soa(st, term=5)
    msgs = E.MsgBuffers(n_groups, n_slots, st["stride"])
    gout = np.zeros(n_groups, dtype=np.uint32)
    md = msgs.as_dict()

    def run_tick(nthreads):
        if nthreads == 1:
            cl.tick_soa(md, gout, 0, n_groups)
        else:
            cl.tick_soa_mt(md, gout, nthreads)  # pthreads inside the oracle, contiguous group ranges

    elapsed = {1: 0.0, threads: 0.0}
    evals = {1: 0, threads: 0}
    for t in range(sample_ticks):
        nthreads = 1 if t % 2 == 0 else threads  # alternate so both lines see the same traffic mix
        E.workload_gen_host(st, msgs, workload, t, seed=seed)
        t0 = time.perf_counter()
        run_tick(nthreads)
        elapsed[nthreads] += time.perf_counter() - t0
        evals[nthreads] += n_groups
        cl.store_soa(st)
    soa = soa_cpu_line(n_groups, n_slots, workload, seed, threads)
    out = {"value": evals[threads] / elapsed[threads] if elapsed[threads] else None, "unit": "group-evals/s",
           "cores": threads, "kind": "port",
           "sample": f"{n_groups} groups x {n_slots} peers, {sample_ticks} ticks of the same stream, alternating 1 / {threads} "
                     f"threads; C port oracle/raft_oracle.c",
           "value_1core": evals[1] / elapsed[1] if elapsed[1] else None,
           "host_cores": os.cpu_count()}
    out.update(soa)
    out["config1"] = cpu_config1(seed, min(threads, 16))
    return out


def soa_cpu_line(n_groups, n_slots, workload, seed, threads):
    """Second CPU line: the engine's own struct-of-arrays arithmetic compiled for the host
    (tests/host_check/, test infrastructure) -- a best-effort CPU implementation of the same algorithm,
    next to the reference-shaped (hash-map, message-at-a-time) port above. Threads are std::threads INSIDE the
    library (rg_host_check_tick_mt: contiguous group ranges), as the C port's are pthreads inside its own."""
    import ctypes as C
    try:
        import test_host_check as H
        import oracle_lib as O
        from raft_rs_amd import engine as E
        if H.build_lib() is None:
            return {}
        fn = C.CDLL(H.LIB).rg_host_check_tick_mt
        fn.restype = C.c_int
        fn.argtypes = [C.c_uint, C.c_ulong, C.c_ulong, C.c_void_p, C.c_void_p, C.c_int, C.c_uint]
        st = O.alloc_state(n_groups, n_slots)
        E.workload_init_host(st, workload, seed=seed)
        msgs = E.MsgBuffers(n_groups, n_slots, st["stride"])
        out = np.zeros(n_groups, dtype=np.uint32)
        sp, mp = H.state_ptrs(st, out), H.msg_ptrs(msgs.as_dict())
        res = {}
        for nthreads in (1, threads):
            total, el = 0, 0.0
            for t in range(6):
                E.workload_gen_host(st, msgs, workload, t if nthreads == 1 else t + 6, seed=seed)
                t0 = time.perf_counter()
                if fn(n_slots, n_groups, st["stride"], sp, mp, 0, nthreads) != 0:
                    raise RuntimeError("rg_host_check_tick_mt failed")
                el += time.perf_counter() - t0
                total += n_groups
            res[nthreads] = total / el
        return {"soa_value_1core": res[1], "soa_value": res[threads], "soa_cores": threads,
                "soa_note": "engine arithmetic (rg_group.h) compiled for the host over the same SoA columns, std::threads "
                            "inside the library over contiguous group ranges"}
    except Exception as e:  # noqa: BLE001
        return {"soa_note": f"unavailable: {type(e).__name__}: {e}"}


def cpu_config1(seed, threads):
    """BASELINE.json configs[0]: 1 000 raft groups x 3 peers, the synthetic AppendResponse stream, on the CPU -- the exact
    workload bench_reference_rust/benches/append_response.rs drives through real RawNode<MemStorage> leaders (shape of
    the reference's own Criterion bench, /root/reference/benches/suites/raw_node.rs:35-79). The Rust side cannot run here
    (no cargo); this is the C port (oracle/raft_oracle.c) on the same stream, so whoever runs cargo has the matched number."""
    import oracle_lib as O
    from raft_rs_amd import engine as E
    G, P, ticks = 1000, 3, 2000
    st = O.alloc_state(G, P)
    E.workload_init_host(st, 2, seed=seed)
    cl = O.Cluster(G)
    cl.load_soa(st, term=5)
    msgs = E.MsgBuffers(G, P, st["stride"])
    gout = np.zeros(G, dtype=np.uint32)
    md = msgs.as_dict()
    el = {1: 0.0, threads: 0.0}
    n = {1: 0, threads: 0}
    for t in range(ticks):
        nt = 1 if t % 2 == 0 else threads
        E.workload_gen_host(st, msgs, 2, t, seed=seed)
        t0 = time.perf_counter()
        if nt == 1:
            cl.tick_soa(md, gout, 0, G)
        else:
            cl.tick_soa_mt(md, gout, nt)
        el[nt] += time.perf_counter() - t0
        n[nt] += G
        cl.store_soa(st)
    return {"workload": "1 000 groups x 3 peers, synthetic AppendResponse stream (BASELINE configs[0])", "ticks": ticks,
            "value_1core": n[1] / el[1], "value": n[threads] / el[threads], "cores": threads, "unit": "group-evals/s",
            "kind": "port",
            "note": "C oracle (message-at-a-time, per-group hash map); at 1 000 groups a tick is ~0.5 ms of work, so the "
                    "all-cores figure is dominated by thread start-up -- the 1-core figure is the one to put next to "
                    "`cargo bench` of bench_reference_rust/ (unrun: no Rust toolchain in this image)"}


def small_batch_latency(rg, torch, n_groups, n_slots, seed):
    """What one RawNode::step -> ready() costs a host: k messages through the host mirror (rg_step), then the wall-clock
    time of rg_flush (ingest + tick of the touched groups + results back), with kernel launches and with the resident
    mailbox workgroup (rg_mailbox_start). Medians over 60 flushes each, 10 warm-up flushes dropped; Python caller."""
    import time
    eng = rg.Engine(n_groups, n_slots, device=torch.cuda.current_device())
    eng.workload_init(2, seed=seed)
    n_mirror = 4096
    for g in range(n_mirror):
        eng.set_peers(g, list(range(1, n_slots + 1)), 4)
    hi = eng.read_column(rg.COL.TERM_HI)
    match = eng.read_column(rg.COL.MATCH)
    rng = np.random.default_rng(7)
    out = {"resident_groups": n_groups, "unit": "us per rg_flush (median)", "caller": "python / ctypes"}
    for mode in ("launch", "mailbox"):
        if mode == "mailbox":
            eng.mailbox_start()
        for k in (1, 10):
            lat = []
            for rep in range(70):
                groups = rng.choice(n_mirror, size=k, replace=False)
                for g in groups.tolist():
                    eng.step(g, 2, 4, int(min(hi[g], match[1, g] + (rep if mode == "launch" else 70 + rep) + 1)))
                t0 = time.perf_counter()
                eng.flush()
                lat.append(time.perf_counter() - t0)
                assert len(eng.ingested_results()[0]) == k
            out[f"{mode}_{k}_groups"] = round(float(np.median(lat[10:])) * 1e6, 2)
    served, launches = eng.mailbox_stats()
    out["mailbox_flushes_served"], out["mailbox_launches"] = served, launches
    eng.close()
    # the same with the Inflights on the device: rg_flush_send = tick + the send stage of the touched groups + results +
    # work items in one round trip (one launch, k_flush_small_send) -- and through the resident workgroup
    eng = rg.Engine(min(n_groups, 200_000), n_slots, device=torch.cuda.current_device(), max_inflight=8)
    eng.workload_init(2, seed=seed)
    for g in range(n_mirror):
        eng.set_peers(g, list(range(1, n_slots + 1)), 4)
    hi = eng.read_column(rg.COL.TERM_HI)
    match = eng.read_column(rg.COL.MATCH)
    for mode in ("launch", "mailbox"):
        if mode == "mailbox":
            eng.mailbox_start()
        lat = []
        for rep in range(70):
            g = int(rng.integers(0, n_mirror))
            eng.step(g, 2, 4, int(min(hi[g], match[1, g] + (rep if mode == "launch" else 70 + rep) + 1)))
            t0 = time.perf_counter()
            eng.flush_send()
            lat.append(time.perf_counter() - t0)
            eng.send_items()
            assert len(eng.ingested_results()[0]) == 1
        out[f"{mode}_flush_send_1_groups"] = round(float(np.median(lat[10:])) * 1e6, 2)
    out["mailbox_flush_send_served"] = eng.mailbox_stats()[0]
    eng.close()
    return out


MALL_BYTES = 256 << 20  # Infinity Cache (MALL) of MI355X, /opt/skills/guides/MI355X_MICROARCH.md; replaced by what the engine asked the device (set_mall_bytes)


def set_mall_bytes(n):
    """The Infinity Cache size the engine asked the device for (rg_device_info.infinity_cache_bytes) replaces the MI355X constant."""
    global MALL_BYTES
    if n:
        MALL_BYTES = int(n)


def regime_of(hot_bytes):
    """Which memory regime a measurement runs in: the columns a kernel touches on EVERY launch (state, not the
    read-once message columns) either fit the 256 MB Infinity Cache -- then part of the traffic never reaches HBM and the
    fraction is not a pure HBM efficiency -- or they do not."""
    return "infinity-cache" if hot_bytes <= MALL_BYTES else "hbm"


def hot_state_bytes(n_groups, n_slots, inflights=False):
    """Bytes of engine state a tick touches every launch: match / next / pr_commit (24 P), flag row, commit, term_lo,
    term_hi, cfg, out (40); with device Inflights also window meta / head / tail (20 P) and the work-item columns (20 P)."""
    return n_groups * (24 * n_slots + 40 + (40 * n_slots if inflights else 0))


def csrc_sha16():
    """What the kernels of this run were built from: sha256 over raft_rs_amd/csrc/* and include/raftgroups.h (file names and
    contents, sorted), first 16 hex digits. The GPU box has no .git, so this -- not a commit -- is what ties a PMC pass to a build."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "raft_rs_amd", "csrc")
    for f in sorted(os.listdir(d)) + [os.path.join("..", "..", "include", "raftgroups.h")]:
        path = os.path.join(d, f)
        if os.path.isfile(path):
            h.update(os.path.basename(f).encode() + b"\0")
            h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def traffic_lookup(key):
    """HBM bytes per launch from the committed PMC passes (profiles/traffic.json) -> (bytes, source, stale). `stale`: the
    passes were taken on kernels built from other sources than this run's (csrc_sha16 differs or was not recorded)."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            ent = json.load(f).get(key, {})
    except (OSError, ValueError):
        return None, None, None
    if ent.get("bytes") is None:
        return None, None, None
    src = f"profiles/traffic.json[{key}]@{ent.get('commit', '?')} csrc {ent.get('csrc_sha16', 'unrecorded')}"
    return ent["bytes"], src[:120], ent.get("csrc_sha16") != csrc_sha16()


def workload_label(workload, n_groups, n_slots, one_engine=False, sorted_classes=False):
    m = n_groups // 1_000_000 if n_groups % 1_000_000 == 0 else None
    size = f"{m}M" if m else str(n_groups)
    if workload == 2 and n_slots == 5:
        return f"{size} groups x 5 peers, majority quorum" + (" (BASELINE configs[1])" if n_groups == 1_000_000 else "")
    if workload == 2 and n_slots == 7:
        return f"{size} groups x 7 peers, majority quorum" + (": ONE rank's shard of BASELINE configs[3] (8M x 7 over 8 GPUs), "
                                                               "no publication" if n_groups == 1_000_000 else "")
    if workload == 3:
        return f"{size} groups x {n_slots} slots, joint {{0,1,2}}&&{{1,2,3}} + learner" + (" (BASELINE configs[2])" if (n_groups, n_slots) == (1_000_000, 5) else "")
    if workload == 5:
        return (f"{size} groups mixed 3/5/7 peers + 10% leader-term rollover" + (" (BASELINE configs[4])" if n_groups == 1_000_000 else "") +
                (f", placed by size class in ONE {n_slots}-slot engine: one launch per tick (k_tick_classes)" if sorted_classes else
                 f", sizes interleaved in one {n_slots}-slot engine" if one_engine else ", one engine per replica-set size"))
    return f"{n_groups} groups x {n_slots} slots, workload {workload}"


class _DevU32:
    """A device range of u32 as something torch.as_tensor accepts (CUDA array interface)."""
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i4", "data": (int(ptr), False), "version": 2}


def send_stage_bytes(rg, eng, n_items, with_work=False):
    """The send stage's own algorithmic bytes for the LAST tick: per group out 4 + cfg 4 +
    last_index 8 + first_index 8 + flag row 8 r + 8 w = 40; per peer in the work set (a send request, an Inflights effect,
    or a broadcast) window meta 4 r + 4 w, oldest / newest inflight 16 r + 16 w, next 8 r + 8 w, pending snapshot request
    8 r, matched 8 r = 72; work items as peer-major columns: 4 B per (slot, group) cell + 8 B per item (prev_index) + 8 B per
    item whose last_index is neither the window's new newest inflight nor its own prev_index (round 5: those do not store it a
    second time -- bits 31 / 30 of the n / kind word, counted here from the column itself)."""
    import torch
    in_tail = 0
    try:
        _, _, pn = eng.send_columns()
        nk = torch.as_tensor(_DevU32(pn, eng.n_slots * eng.stride), device="cuda")
        in_tail = int(((nk < 0) | ((nk >> 30) & 1).bool()).sum().item())  # (bits 31 / 30 of a u32 read as i32: last_index is the window's tail / the item's prev_index)
    except Exception:  # noqa: BLE001 -- a sparse stage has no columns: every item carries both indices
        pass
    _, out = eng.results()
    cfg = eng.read_column(rg.COL.CFG)
    present, self_slot = (cfg >> 24) & 0xff, (cfg >> 16) & 7
    bcast = (out & 0x9) != 0  # CHANGED (skip_bcast_commit off) or APPENDED
    work = ((out >> 8) | (out >> 16) | (out >> 24)) & 0xff
    work = np.where(bcast, work | present, work) & present & ~(1 << self_slot)
    n_work = int(sum(((work >> p) & 1).sum() for p in range(8)))
    b = 40 * eng.n_groups + 72 * n_work + 4 * eng.n_slots * eng.n_groups + 16 * n_items - 8 * in_tail
    return (b, n_work) if with_work else b


SIDE_REPEATS = 3  # timed regions per side measurement (median reported, min / max beside it)


def run_config(rg, torch, n_groups, n_slots, workload, warmup, steps, seed, what="tick", variant=0, one_engine=False,
               inflights=0, fused_send=False, sorted_classes=False, repeats=SIDE_REPEATS, cfg_flags=0, group_commit=False,
               place_after_load=False):
    """A complete, self-contained measurement of one configuration on one GPU, for the bench line's sub-objects (the
    headline has its own region in main(), with the multi-GPU plumbing): engines are created, the W+K ticks of the
    synthetic stream are generated on the device from the evolving state and recorded, the state is restored from a
    checkpoint, and the K recorded ticks are replayed back to back between two HIP events on the engines' streams.
      what == "tick":      the hot path; config 5 runs one engine per replica-set size on its own stream unless one_engine;
                           inflights > 0 adds the send stage (rg_send_appends) after every tick, timed per launch too;
                           fused_send: the timed replay runs the tick and its stage as ONE launch (rg_tick_device_send);
                           sorted_classes (config 5, one engine): the same groups placed by replica-set size class, which the
                           engine runs as ONE launch whose blocks skip the slots their class does not have (k_tick_classes)
      what == "recompute": K launches of rg_recompute -- Raft::maybe_commit for every group with no messages, literally
                           BASELINE's "commit-index recomputes"
    Returns a dict with its own `roofline` object (bound, regime, achieved, peak, frac, traffic...)."""
    main_stream = torch.cuda.current_stream()
    one_engine = one_engine or sorted_classes
    if workload == 5 and not one_engine:
        sizes = [(3, n_groups // 3), (5, n_groups // 3), (7, n_groups - 2 * (n_groups // 3))]
    else:
        sizes = [(n_slots, n_groups)]
    T = warmup + steps

    class Part:
        pass

    parts, first = [], 0
    for slots, n in sizes:
        pt = Part()
        pt.n, pt.slots, pt.first = n, slots, first
        pt.fixed = slots if (workload == 5 and not one_engine) else 0
        pt.eng = rg.Engine(n, slots, device=torch.cuda.current_device(), variant=variant, max_inflight=inflights, flags=cfg_flags)
        pt.eng.set_stream(main_stream.cuda_stream)
        set_mall_bytes(pt.eng.device_info().get("infinity_cache_bytes"))
        pt.eng.workload_init(workload, seed=seed, first_group=first, fixed_peers=pt.fixed,
                             sorted_classes=sorted_classes and not place_after_load, group_commit=group_commit)
        if place_after_load:
            # the shard arrives with its sizes interleaved (id order) and is re-placed on the device: rg_plan_placement from its
            # own cfg column, rg_permute_groups -- after which it IS the class-placed shard (same groups at the same positions)
            t_p = time.perf_counter()
            pt.eng.place_by_size_class()
            pt.place_ms = (time.perf_counter() - t_p) * 1e3
        first += n
        parts.append(pt)
    classes = parts[0].eng.size_classes() if sorted_classes else None
    if sorted_classes and len(classes) < 2:
        raise SystemExit(f"side measurement: the class-placed shard was not recognised as one ({classes})")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    extra = {}
    if what == "recompute":
        pt = parts[0]
        cols = [torch.empty((n_slots, pt.eng.stride), dtype=torch.int64, device="cuda") for _ in range(4)]
        flags = torch.empty((n_groups, 8), dtype=torch.uint8, device="cuda")
        for t in range(3):  # matches as a run leaves them: a few ticks of the stream first (untimed)
            pt.eng.workload_gen(workload, t, *[c.data_ptr() for c in cols], flags.data_ptr(), seed=seed)
            pt.eng.tick_device(*[c.data_ptr() for c in cols], flags.data_ptr())
        for _ in range(warmup):
            pt.eng.recompute()
        us_all = []
        for _ in range(repeats):
            torch.cuda.synchronize()
            e0.record(main_stream)
            for _ in range(steps):
                pt.eng.recompute()
            e1.record(main_stream)
            torch.cuda.synchronize()
            us_all.append(e0.elapsed_time(e1) * 1e3 / steps)
        us = float(np.median(us_all))
        nbytes = (8 * n_slots + 37) * n_groups  # B0(P) = 8 P + 37
        hot = nbytes  # every byte it reads is re-read by the next launch
        kernel, unit = "k_recompute", "commit-index recomputes/s"
        key = f"recompute:{n_groups}:{n_slots}"
    else:
        for pt in parts:
            pt.cols = [torch.empty((T, pt.slots, pt.eng.stride), dtype=torch.int64, device="cuda") for _ in range(4)]
            pt.flags = torch.empty((T, pt.n, 8), dtype=torch.uint8, device="cuda")
            pt.eng.checkpoint()

        def ptrs(pt, t):
            return [c[t].data_ptr() for c in pt.cols] + [pt.flags[t].data_ptr()]

        alg = [0] * T
        census = dict(valid=0, rejects=0, elections=0)
        for t in range(T):
            for pt in parts:
                pt.eng.workload_gen(workload, t, *ptrs(pt, t), seed=seed, first_group=pt.first, fixed_peers=pt.fixed,
                                    sorted_classes=sorted_classes, group_commit=group_commit)
                if inflights:
                    pt.flags[t] &= 0xE7  # no RG_MF_SENT / RG_MF_INS_FULL: the device owns the send path
                s = pt.eng.msg_stats(pt.flags[t].data_ptr())
                alg[t] += algorithmic_bytes(pt.n, s["slots"], s["valid"], s["rejects"])
                if group_commit:  # ... plus Progress.commit_group_id of every peer, read once per group (8 P): SURVEY 8(d)'s
                    alg[t] += 8 * s["slots"]  # formula is the plain quorum's and counts no gid column
                if t >= warmup:
                    for k in census:
                        census[k] += s[k]
                pt.eng.tick_device(*ptrs(pt, t))
                if inflights:
                    pt.eng.send_appends(0)
        for pt in parts:
            pt.eng.sync()
            pt.ref = pt.eng.results()
            if pt.eng.result_counts()[1]:
                raise SystemExit("side measurement: the stream raised faults")
            pt.eng.restore()
        multi = len(parts) > 1
        for pt in parts:  # size-class engines are independent: one HIP stream each
            pt.stream = torch.cuda.Stream() if multi else main_stream
            pt.eng.set_stream(pt.stream.cuda_stream)
        per_launch = []  # (tick start, tick end / stage start, stage end) events of every timed step (inflights only)

        def replay(t0, n, record):
            if multi:
                fork = torch.cuda.Event()
                fork.record(main_stream)
                for pt in parts:
                    pt.stream.wait_event(fork)
            for i in range(n):
                for pt in parts:
                    if record and inflights:
                        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                        ev[0].record(pt.stream)
                    if inflights and fused_send:
                        pt.eng.tick_device_send(*ptrs(pt, t0 + i), max_entries_per_msg=0)
                        if record:
                            ev[1].record(pt.stream)
                            ev[2] = ev[1]
                            per_launch.append(ev)
                        continue
                    pt.eng.tick_device(*ptrs(pt, t0 + i))
                    if inflights:
                        if record:
                            ev[1].record(pt.stream)
                        pt.eng.send_appends(0)
                        if record:
                            ev[2].record(pt.stream)
                            per_launch.append(ev)
            if multi:
                for pt in parts:
                    ev_ = torch.cuda.Event()
                    ev_.record(pt.stream)
                    main_stream.wait_event(ev_)

        us_all = []
        for rep in range(repeats):  # every region: restored state, W warm-up ticks, then exactly K timed ticks
            if rep:
                for pt in parts:
                    pt.eng.restore()
                per_launch.clear()
            replay(0, warmup, False)
            torch.cuda.synchronize()
            e0.record(main_stream)
            replay(warmup, steps, True)
            e1.record(main_stream)
            torch.cuda.synchronize()
            us_all.append(e0.elapsed_time(e1) * 1e3 / steps)
            for pt in parts:  # replay determinism
                c, o = pt.eng.results()
                if not (np.array_equal(c, pt.ref[0]) and np.array_equal(o, pt.ref[1])):
                    raise SystemExit("side measurement: the timed replay diverged from the recorded pass")
        us = float(np.median(us_all))
        nbytes = float(np.mean(alg[warmup:]))
        hot = sum(hot_state_bytes(pt.n, pt.slots, bool(inflights)) for pt in parts)
        if sorted_classes:  # the columns of the slots a class does not have are never touched
            hot = sum(hot_state_bytes(n, q) for _, n, q in classes)
        kernel = {2: "k_tick_lds", 4: "k_tick_lds", 5: "k_tick_compact"}.get(variant, "k_tick_classes" if sorted_classes else "k_tick_lane")
        unit = "group-evals/s"
        key = (f"{workload}:{n_groups}:{n_slots}" + (":sorted" if sorted_classes else ":one-engine" if (workload == 5 and one_engine) else "") +
               (f":v{variant}" if variant else "") + (":inflights" if inflights else "") + (":fused-send" if (inflights and fused_send) else "") +
               (":gc" if group_commit else ""))
        denom = float(n_groups * steps)
        extra = {"acks_per_group": round(census["valid"] / denom, 3), "rejects_per_group": round(census["rejects"] / denom, 5)}
        if workload == 5:
            extra["elections_per_group"] = round(census["elections"] / denom, 5)
        if inflights and fused_send:
            pt = parts[0]
            step_all = sorted(a.elapsed_time(b) * 1e3 for a, b, _ in per_launch)
            step_us = step_all[len(step_all) // 2]
            items = len(pt.eng.send_items())
            sb, n_work = send_stage_bytes(rg, pt.eng, items, with_work=True)
            # what one launch for both removes from the two models' sum: the stage no longer reads the result word, cfg,
            # last_index and the flag row (24 B per group), the flag row is written once (8), and per peer in the work set
            # `next` is neither re-read nor written twice and `matched` is not re-read (24)
            fb = alg[-1] + sb - 32 * n_groups - 24 * n_work
            fg = fb / (step_us * 1e-6) / 1e9
            extra["send_stage"] = {
                "max_inflight": inflights, "max_entries_per_msg": 0, "work_items_last_tick": int(items), "one_launch": True,
                "us_per_step_median": step_us,
                "roofline": {"bound": "hbm", "regime": regime_of(hot), "achieved": fg, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": fg / HBM_PEAK_GBS, "kernel": "k_tick_send", "algorithmic_bytes_per_launch": fb,
                             "bytes_per_group": fb / n_groups, "avg_launch_us": step_us, "traffic": None,
                             "avg_launch_us_min": step_all[0], "avg_launch_us_max": step_all[-1],  # (single launches: the windows fill as the replay goes on)
                             "note": "tick + send stage in ONE launch: the tick's SURVEY 8(d) bytes plus the stage's own byte model "
                                     "(DESIGN.md section 3) minus what the shared registers save (32 B per group, 24 B per peer in "
                                     "the work set), counted on the last tick"}}
        elif inflights:
            pt = parts[0]
            tick_us = sorted(a.elapsed_time(b) for a, b, _ in per_launch)[len(per_launch) // 2] * 1e3
            stage_in_order = [round(b.elapsed_time(c) * 1e3, 1) for _, b, c in per_launch]  # (the last region's launches, in replay order)
            stage_all = sorted(b.elapsed_time(c) * 1e3 for _, b, c in per_launch)
            stage_us = stage_all[len(stage_all) // 2]
            items = len(pt.eng.send_items())
            sb = send_stage_bytes(rg, pt.eng, items)
            sg = sb / (stage_us * 1e-6) / 1e9
            extra["send_stage"] = {
                "max_inflight": inflights, "max_entries_per_msg": 0, "work_items_last_tick": int(items),
                "us_per_tick_median": tick_us, "us_per_stage_median": stage_us,
                # every stage launch of the last region, in order: the spread of k_send_dense (round 5: "bimodal, 57-108 us") is
                # the REPLAY -- each region starts from the restored windows and the time per launch falls monotonically as they
                # settle into the stream's steady state (r06: 86 -> 69 us over 30 launches), not two modes of the hardware
                "us_per_stage_in_replay_order": stage_in_order,
                "roofline": {"bound": "hbm", "regime": regime_of(hot), "achieved": sg, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": sg / HBM_PEAK_GBS, "kernel": "k_send_dense", "algorithmic_bytes_per_launch": sb,
                             "bytes_per_group": sb / n_groups, "avg_launch_us": stage_us, "traffic": None,
                             "avg_launch_us_min": stage_all[0], "avg_launch_us_max": stage_all[-1],  # (single launches: the windows fill as the replay goes on)
                             "note": "the stage's OWN byte model (DESIGN.md section 3), counted on the last tick; SURVEY 8(d) "
                                     "counts no bytes for the send path"}}
    for pt in parts:
        pt.info = pt.eng.device_info()
        pt.eng.close()
    gbs = nbytes / (us * 1e-6) / 1e9
    traffic, traffic_source, traffic_stale = traffic_lookup(key)
    info = parts[0].info if hasattr(parts[0], "info") else {}
    if what == "tick" and info.get("last_tick_kernel") not in (None, "none"):
        kernel = info["last_tick_kernel"]  # what the engine says it launched (k_tick_split / k_tick_classes / k_tick_send ...)
    label = (workload_label(workload, n_groups, n_slots, one_engine, sorted_classes) if what == "tick" else
             f"rg_recompute (Raft::maybe_commit, no messages) over {n_groups} groups x {n_slots} peers")
    if group_commit:
        label += " + group commit in every group, three commit groups over the peers (majority.rs:99-123)"
    if place_after_load:
        label = label.replace("placed by size class", "LOADED INTERLEAVED, then placed by size class on the device (rg_plan_placement + rg_permute_groups)")
        extra["placement_ms"] = round(parts[0].place_ms, 2)
    if inflights:
        label += (f" + Inflights (cap {inflights}) on the device and the send stage after every tick, " +
                  ("tick and stage as ONE launch (rg_tick_device_send)" if fused_send else "as a launch of its own (rg_send_appends)"))
    return {"workload": label,
            "groups": n_groups, "peer_slots": n_slots, "engines": [{"slots": pt.slots, "groups": pt.n} for pt in parts],
            **({"size_classes": [{"first_group": f, "groups": n, "slots": q} for f, n, q in classes]} if classes else {}),
            "steps": steps, "warmup": warmup, "repeats": repeats, "us_per_step": us, "us_per_step_min": min(us_all),
            "us_per_step_max": max(us_all), "value": n_groups / (us * 1e-6), "unit": unit, **extra,
            "engine_reports": {k: info.get(k) for k in ("cache_policy", "resident_groups", "last_tick_kernel", "last_tick_streaming")},
            "roofline": {"bound": "hbm", "regime": regime_of(hot), "hot_state_bytes": hot, "achieved": gbs, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "kernel": ("k_tick_send" if fused_send else kernel + " + k_send_dense") if inflights else kernel,
                         "algorithmic_bytes_per_launch": nbytes, "bytes_per_eval": nbytes / n_groups, "avg_launch_us": us,
                         "avg_launch_us_min": min(us_all), "avg_launch_us_max": max(us_all),
                         "traffic": traffic, "traffic_source": traffic_source, "traffic_stale": traffic_stale,
                         **({"note": "avg_launch_us is the tick AND its send stage; the algorithmic bytes are the tick's, so frac "
                                     "understates this mode -- the stage has its own roofline under send_stage"} if inflights else {})}}


def by_config_summary(result):
    """Every configuration this line measured, compact, INSIDE `roofline` (the driver's record keeps that object whole and
    only the names of the other sub-objects): frac = alg_bytes / us / 8 TB/s can be recomputed from each entry alone.
    `traffic` = PMC HBM bytes per launch from the committed passes (null where not profiled); `regime` as in `roofline`."""
    def entry(obj, r=None):
        r = r or (obj or {}).get("roofline")
        if not r or "frac" not in r:
            return {"error": (obj or {}).get("error", "not measured")[:120]}
        return {"frac": round(r["frac"], 4), "us": round(r["avg_launch_us"], 2),
                "us_min": None if r.get("avg_launch_us_min") is None else round(r["avg_launch_us_min"], 2),
                "us_max": None if r.get("avg_launch_us_max") is None else round(r["avg_launch_us_max"], 2),
                "alg_bytes": int(r["algorithmic_bytes_per_launch"]),
                "traffic": None if r.get("traffic") is None else int(r["traffic"]), "traffic_stale": r.get("traffic_stale"),
                "regime": r.get("regime"), "kernel": r.get("kernel")}
    oc = result.get("other_configs") or {}
    out = {"c2_headline": entry(result)}
    for name, obj in (("c2_hbm_8M", result.get("out_of_cache")), ("c2_resident_2_4M", result.get("between_regimes")),
                      ("c3_joint", oc.get("configs[2] joint")),
                      ("c4_shard", oc.get("configs[3] one rank's shard")), ("c5_one_launch", oc.get("configs[4] one launch, class-sorted")),
                      ("c5_size_class", oc.get("configs[4] size-class engines")), ("c5_interleaved", oc.get("configs[4] one 7-slot engine")),
                      ("c5_one_launch_hbm_8M", oc.get("configs[4] one launch, class-sorted, 8M groups")),
                      ("c5_placed", oc.get("configs[4] interleaved, after rg_permute_groups")),
                      ("c2_group_commit", oc.get("configs[1] group commit")),
                      ("recompute", result.get("recompute_only")), ("recompute_hbm_8M", result.get("recompute_only_out_of_cache"))):
        if obj is not None:
            out[name] = entry(obj)
    for name, key in (("send_two_launch", "configs[1] + send stage"), ("send_one_launch", "configs[1] + send stage, one launch")):
        obj = oc.get(key)
        if obj is None:
            continue
        e = entry(obj, (obj.get("send_stage") or {}).get("roofline"))  # the stage's OWN byte model
        if "frac" in e:
            e["step_us"] = round(obj["us_per_step"], 2)
            e["frac_by_tick_bytes"] = round(obj["roofline"]["frac"], 4)
            # PMC bytes of one STEP (tick + stage; in the one-launch form that is the k_tick_send launch itself)
            st = obj["roofline"].get("traffic")
            e["step_traffic"] = None if st is None else int(st)
            if name == "send_one_launch":
                e["traffic"], e["traffic_stale"] = e["step_traffic"], obj["roofline"].get("traffic_stale")
        out[name] = e
    return out


def flat_config_keys(by_config):
    flat = {}
    for name, e in by_config.items():
        if name == "c2_headline":
            continue  # (roofline's own frac / avg_launch_us / algorithmic_bytes_per_launch / traffic)
        if "frac" not in e:
            flat[f"frac_{name}"] = None
            continue
        flat[f"frac_{name}"] = e["frac"]
        flat[f"us_{name}"] = e["us"]
        flat[f"us_min_{name}"], flat[f"us_max_{name}"] = e.get("us_min"), e.get("us_max")
        flat[f"mb_{name}"] = round(e["alg_bytes"] / 1e6, 2)
        flat[f"traffic_mb_{name}"] = None if e.get("traffic") is None else round(e["traffic"] / 1e6, 2)
        if e.get("traffic_stale"):
            flat[f"traffic_stale_{name}"] = True
        for k in ("step_us", "frac_by_tick_bytes"):
            if k in e:
                flat[f"{k}_{name}"] = e[k]
        if e.get("step_traffic") is not None:
            flat[f"step_traffic_mb_{name}"] = round(e["step_traffic"] / 1e6, 2)
    return flat


LINE_LIMIT = 6000  # bytes of the ONE line on stdout: the driver's record keeps an ~8 KB tail of stdout and parses the line from it
FULL_RESULT = os.path.join("gpurun_out", "bench_full.json")  # every nested sub-object of the run (relative to the repo root)

# what each flat `<prefix>_<cfg>` key of `roofline` means; frac = mb / us / 8 TB/s can be recomputed from the line alone
_FLAT_PREFIXES = ("frac_", "us_", "mb_", "traffic_mb_", "step_us_", "frac_by_tick_bytes_", "step_traffic_mb_", "traffic_stale_")


def _short(v, digits=6, text=160):
    """A scalar as the line carries it: floats to `digits` significant digits, strings cut at `text` characters."""
    if isinstance(v, bool) or v is None or isinstance(v, int):
        return v
    if isinstance(v, float):
        return float(f"{v:.{digits}g}")
    if isinstance(v, str):
        return v if len(v) <= text else v[:text - 3] + "..."
    return None


def _scalars(obj, text=160, skip=(), digits=6):
    return {k: _short(v, digits=digits, text=text) for k, v in (obj or {}).items()
            if k not in skip and (v is None or isinstance(v, (bool, int, float, str)))}


def compact_line(result):
    """The ONE line stdout carries, from the complete (nested) result of a run: the contract's top-level keys, `config` and
    `roofline` and `cpu_baseline` as objects of SCALARS (every side configuration as frac_/us_/mb_/traffic_mb_<cfg> keys of
    `roofline`), nothing nested below them, at most LINE_LIMIT bytes. Everything else the run measured (other_configs,
    out_of_cache, between_regimes, recompute_only*, small_batch_latency, by_config, fused_model, the publication statistics)
    goes to FULL_RESULT and to stderr -- the name of that file is the line's `full` key."""
    line = _scalars(result, skip=("full",), digits=12)  # (value / ms_per_step: the contract's own numbers keep their digits)
    line["full"] = FULL_RESULT
    cfg_in = result.get("config") or {}
    cfg = _scalars(cfg_in, text=220)
    cfg["engines"] = "+".join(f"{e['groups']}x{e['slots']}" for e in cfg_in.get("engines", [])) or None
    dev = cfg_in.get("device") or {}
    for k in ("arch", "compute_units", "cache_policy", "last_tick_kernel", "last_tick_streaming", "resident_groups", "infinity_cache_bytes"):
        if k in dev:
            cfg[k] = _short(dev[k])
    pub = cfg_in.get("publication") or {}
    for k, v in _scalars(pub).items():
        cfg[f"pub_{k}"] = v
    auto = cfg_in.get("publish_every_auto") or {}
    for k in ("tick_us", "exchange_us"):
        if k in auto:
            cfg[f"publish_auto_{k}"] = auto[k]
    pc = cfg_in.get("publication_compare") or {}
    if pc:
        cfg["pub_compare_mode"], cfg["pub_compare_ms_per_step"], cfg["pub_compare_value"] = \
            _short(pc.get("mode"), text=60), _short(pc.get("ms_per_step")), _short(pc.get("value"))
    line["config"] = cfg
    roof_in = result.get("roofline") or {}
    roof = _scalars(roof_in, text=120)
    fm = roof_in.get("fused_model") or {}
    if fm:
        roof["fused_ticks_per_launch"], roof["fused_per_tick_model_bytes"] = fm.get("ticks_per_launch"), _short(fm.get("per_tick_model_bytes"))
    line["roofline"] = roof
    cb = result.get("cpu_baseline")
    if cb is None or "error" in cb:
        line["cpu_baseline"] = cb if cb is None else _scalars(cb, text=200)
    else:
        out = _scalars(cb, text=150, skip=("soa_note",))
        for k, v in _scalars(cb.get("config1") or {}, text=80, skip=("note", "unit", "kind")).items():
            out[f"config1_{k}"] = v
        line["cpu_baseline"] = out
    if result.get("send_stage"):
        line["send_stage"] = _scalars(result["send_stage"])
    lat = result.get("small_batch_latency")
    if lat and "error" not in lat:
        line["latency_us"] = {k: v for k, v in _scalars(lat).items() if isinstance(v, (int, float))}

    def size():
        return len(json.dumps(line, separators=(",", ":"))) + 1

    # Should the line ever outgrow the limit, what goes first is what FULL_RESULT holds anyway -- never a contract key, never
    # frac_/us_/mb_ of a configuration.
    droppers = [lambda: line.pop("latency_us", None),
                lambda: [roof.pop(k) for k in list(roof) if k.startswith(("us_min_", "us_max_"))],
                lambda: [roof.pop(k, None) for k in ("regime_note", "traffic_source", "note")],
                lambda: [line["cpu_baseline"].pop(k, None) for k in ("sample", "config1_workload")] if line.get("cpu_baseline") else None,
                lambda: [cfg.pop(k) for k in list(cfg) if k.startswith(("pub_", "publish_auto_")) and k not in ("pub_compare_value",)],
                lambda: [roof.pop(k) for k in list(roof) if k.startswith(("traffic_mb_", "step_traffic_mb_", "traffic_stale_"))]]
    for drop in droppers:
        if size() <= LINE_LIMIT:
            break
        drop()
    if size() > LINE_LIMIT:
        raise SystemExit(f"bench.py: the JSON line is {size()} bytes (limit {LINE_LIMIT})")
    return line


def line_text(result):
    return json.dumps(compact_line(result), separators=(",", ":")) + "\n"


def write_full(result):
    """The complete nested result: gpurun_out/bench_full.json under the repo root, and stderr."""
    text = json.dumps(result)
    try:
        path = os.path.join(ROOT, FULL_RESULT)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            f.write(text + "\n")
    except OSError as e:
        sys.stderr.write(f"bench.py: could not write {FULL_RESULT}: {e}\n")
    sys.stderr.write("bench.py full result: " + text + "\n")
    sys.stderr.flush()


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: re-run this very command line as N ranks (one per GPU) under
    torch.distributed.run, --nnodes=1, rendezvous on 127.0.0.1 at a port the OS just handed out. stdout / stderr are the
    children's (rank 0 prints the JSON line last); returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs between processes on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def guarded(fn, *a, **kw):
    """A side measurement must never cost the headline its JSON line: whatever goes wrong in it (a device out of memory on
    a shared box, a fault the stream check raises) is reported in its place."""
    try:
        return fn(*a, **kw)
    except (Exception, SystemExit) as e:  # noqa: BLE001
        try:
            import torch
            torch.cuda.empty_cache()
        except Exception:  # noqa: BLE001
            pass
        return {"error": f"{type(e).__name__}: {e}"[:500]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--groups", type=int, default=1_000_000, help="raft groups per GPU")
    ap.add_argument("--slots", type=int, default=5)
    ap.add_argument("--workload", type=int, default=2, choices=[2, 3, 5])
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x5EED5EED)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--publish-every", default="auto",
                    help="N>1: publish the commit advances every E ticks (and after the last tick). Default `auto`: E is picked "
                         "from the exchange time measured on 8 publications before the timed region -- E = ceil(exchange / "
                         "tick), 1 (= every tick) wherever an exchange fits behind a tick; rank 0 decides, the control plane "
                         "broadcasts it, no extra data-path collective. `1` forces every tick.")
    ap.add_argument("--total-groups", type=int, default=0,
                    help="N>1: STRONG scaling -- this many groups in total, split into N disjoint contiguous ranges "
                         "(sharding.strong_shard); `scaling` is then \"strong\". Default 0 = weak scaling, --groups per GPU")
    ap.add_argument("--publish-raw", action="store_true",
                    help="N>1: publish the full 8 B/group commit column every time (RG_PUBLISH_FULL) instead of the ~1 B/group "
                         "delta slices -- the naive exchange, for comparison")
    ap.add_argument("--no-publish-compare", action="store_true",
                    help="N>1: skip the second run of the K ticks with the other publication form (publication_compare)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="timed regions of K steps each (every one behind its own restore + W warm-up steps); the line reports the "
                         "median region and the fastest / slowest beside it")
    ap.add_argument("--fuse", type=int, default=1,
                    help="temporal fusion: process this many consecutive ticks per launch (1..8, default 1)")
    ap.add_argument("--split", type=int, default=1, help="sub-shards per rank, each on its own HIP stream")
    ap.add_argument("--cfg-flags", type=lambda s: int(s, 0), default=0,
                    help="rg_config.flags of the measured engines (RG_CFGF_*: 0x1 = no size classes, 0x2 = class blocks in block order, 0x4 = 64-bit cell offsets)")
    ap.add_argument("--one-engine", action="store_true", help="config 5: keep all sizes interleaved in one engine")
    ap.add_argument("--sorted", action="store_true",
                    help="config 5: ONE 7-slot engine with the groups placed by replica-set size class (one launch per tick, "
                         "k_tick_classes) -- what --workload 5 runs unless --one-engine / --size-class-engines says otherwise")
    ap.add_argument("--c5-sizes", default=None,
                    help="--workload 5 --size-class-engines: the engines as slots:groups,... instead of 3 / 5 / 7 x G/3 (experiment)")
    ap.add_argument("--size-class-engines", action="store_true",
                    help="config 5: one engine per replica-set size (three launches per tick on three streams)")
    ap.add_argument("--inflights", type=int, default=0,
                    help="N > 0: keep the Inflights (cap N) on the device and run the send stage (rg_send_appends: "
                         "maybe_send_append decisions, SURVEY 8f row 3) after every tick, inside the timed region; the "
                         "stream then carries no host SENT events. Not the headline configuration.")
    ap.add_argument("--fused-send", action="store_true",
                    help="with --inflights N: the tick and its send stage as ONE launch (rg_tick_device_send, k_tick_send)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the recompute_only and out_of_cache sub-measurements (N=1 only; they run after the "
                         "timed region and do not touch `value`)")
    ap.add_argument("--place-after-load", action="store_true",
                    help="--side tick --workload 5 --sorted: load the shard interleaved, then rg_plan_placement + rg_permute_groups")
    ap.add_argument("--group-commit", action="store_true",
                    help="--side tick: ProgressTracker.group_commit on in every group, three commit groups over the peers (RG_WL_GROUP_COMMIT)")
    ap.add_argument("--side", default=None, choices=["recompute", "tick"],
                    help="profiling hook: run ONLY one sub-measurement (run_config) of --groups x --slots, workload "
                         "--workload, and print its object -- what the PMC passes behind profiles/traffic.json wrap")
    ap.add_argument("--out-of-cache-groups", type=int, default=8_000_000)
    ap.add_argument("--c5-variant", type=int, default=C5_VARIANT,
                    help="kernel variant of the config-5 lines under other_configs (5 = compact, 0 = lane)")
    ap.add_argument("--cpu-sample-groups", type=int, default=1_000_000)
    ap.add_argument("--cpu-sample-ticks", type=int, default=16)
    args = ap.parse_args()
    if args.workload == 5 and not (args.one_engine or args.size_class_engines):
        args.sorted = True  # config 5's layout of record: placed by size class, ONE launch per tick
    if args.sorted:
        args.one_engine = True
    if args.workload == 5:
        args.slots = max(args.slots, 7)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # The plain command line (`python bench.py --gpus 8 ...`, what works at N = 1) starts its own ranks: one process
        # per GPU under torch.distributed.run on this node, rendezvous on 127.0.0.1 and a free port. Under an external
        # launcher (WORLD_SIZE set) nothing of this runs.
        raise SystemExit(self_launch(args.gpus))

    # fd 1 belongs to the JSON line alone: gloo ("[Gloo] Rank 3 is connected to 7 peer ranks...") and RCCL (its version banner)
    # write progress lines to stdout from every rank. From here on whatever anybody prints goes to stderr; emit() writes the line.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj, whole=False):
        """The headline result goes out as compact_line (<= LINE_LIMIT bytes; the nested rest to FULL_RESULT and stderr);
        `whole`: a --side object, printed as it is (what the profiling tools parse)."""
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:  # noqa: BLE001
            pass
        if whole:
            os.write(json_fd, (json.dumps(obj) + "\n").encode())
            return
        text = line_text(obj)  # (raises before anything is written if the line could not be made to fit)
        write_full(obj)
        os.write(json_fd, text.encode())

    import torch
    import torch.distributed as dist
    import raft_rs_amd as rg

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py --gpus {args.gpus} under a launcher with WORLD_SIZE={world}: one rank per GPU, the two must agree")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    # BENCH_SHARE_GPU=1 (test hook): every rank uses GPU 0 and the collective backend is gloo, so the N>1
    # code path (sharding offsets, publication, max-over-ranks timing) can run on a single-GPU box
    share_gpu = os.environ.get("BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # BENCH_FORCE_DIST=1 runs the N>1 code path (process group + commit all-gather) at world size 1
    distributed = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # torch.distributed is the CONTROL plane only (unique-id hand-over, barriers, max-over-ranks): gloo. The data
        # path -- the commit publication -- is the engine's own RCCL communicator (rg_comm_init), one per engine.
        dist.init_process_group("gloo", rank=rank, world_size=world)

    G, P, W, K = args.groups, args.slots, args.warmup, args.steps
    strong = args.total_groups > 0
    if strong:
        # strong scaling: the SAME total population at every N, rank r holds [r G/N, (r+1) G/N) (sharding.strong_shard). The
        # publication's all-gather moves equal slices, so the total must divide evenly
        if args.total_groups % world:
            raise SystemExit(f"--total-groups {args.total_groups} does not divide into {world} equal ranges")
        from raft_rs_amd.sharding import strong_shard
        sh = strong_shard(rank, world, args.total_groups)
        G = sh.n_groups
        assert sh.first_group == rank * G
    T = W + K
    if args.side:
        torch.cuda.set_stream(torch.cuda.Stream())
        emit(run_config(rg, torch, G, P, args.workload, W, K, args.seed, what=args.side, variant=args.variant,
                        one_engine=args.one_engine, inflights=args.inflights, fused_send=args.fused_send,
                        sorted_classes=args.sorted, repeats=max(1, args.repeats), cfg_flags=args.cfg_flags,
                        group_commit=args.group_commit, place_after_load=args.place_after_load), whole=True)
        return
    # An explicit (non-default) stream for the engines: on the legacy NULL stream every launch orders itself against the
    # other streams of the process, which serialises the publication's side stream with the ticks
    # (tools/probe_publish_host.py: 66 vs 43 us per tick + publication at world size 1).
    torch.cuda.set_stream(torch.cuda.Stream())
    stream = torch.cuda.current_stream()

    # A rank's shard is one engine per replica-set size class. Configs 2-4 have one size; config 5
    # (mixed 3/5/7) places its groups by size class -- three engines of G/3 groups, three launches per
    # tick -- instead of streaming the absent peers' cells of a single 7-slot engine.
    # --one-engine keeps the mixed population interleaved in one P-slot engine for comparison.
    if args.workload == 5 and not args.one_engine:
        sizes = [(3, G // 3), (5, G // 3), (7, G - 2 * (G // 3))]
        if args.c5_sizes:  # experiment: other replica-set sizes per engine, "slots:groups,..." (profiles/r06_c5_lane_pair.txt)
            sizes = [tuple(int(x) for x in part.split(":")) for part in args.c5_sizes.split(",")]
    else:
        sizes = [(P, G)]
    if args.split > 1:
        # Independent sub-shards on separate HIP streams: ticks of one sub-shard are ordered, the
        # sub-shards are not, so one chain's launch tail / kernel boundary overlaps the other's body.
        sizes = [(sl, n // args.split + (1 if k < n % args.split else 0)) for sl, n in sizes for k in range(args.split)]

    class Part:
        pass

    parts, first = [], rank * G  # disjoint global group ids per rank (sharding.weak_shard)
    for slots, n in sizes:
        pt = Part()
        pt.n, pt.slots, pt.first = n, slots, first
        pt.fixed = slots if (args.workload == 5 and not args.one_engine) else 0
        pt.eng = rg.Engine(n, slots, device=local_rank, variant=args.variant, max_inflight=args.inflights, flags=args.cfg_flags)
        pt.eng.set_stream(stream.cuda_stream)
        set_mall_bytes(pt.eng.device_info().get("infinity_cache_bytes"))
        pt.eng.workload_init(args.workload, seed=args.seed, first_group=first, fixed_peers=pt.fixed, sorted_classes=args.sorted)
        pt.eng.checkpoint()
        pt.cols = [torch.empty((T, slots, pt.eng.stride), dtype=torch.int64, device="cuda") for _ in range(4)]
        pt.flags = torch.empty((T, n, 8), dtype=torch.uint8, device="cuda")
        first += n
        parts.append(pt)

    def tick_ptrs(pt, t):
        return [c[t].data_ptr() for c in pt.cols] + [pt.flags[t].data_ptr()]

    # ---- untimed pass: generate + apply W+K ticks, recording the message columns on the device ----
    alg_bytes = [0] * T
    census = [dict(valid=0, rejects=0, slots=0, elections=0) for _ in range(T)]
    for t in range(T):
        for pt in parts:
            pt.eng.workload_gen(args.workload, t, *tick_ptrs(pt, t), seed=args.seed, first_group=pt.first,
                                fixed_peers=pt.fixed, sorted_classes=args.sorted)
            if args.inflights:
                pt.flags[t] &= 0xE7  # no RG_MF_SENT / RG_MF_INS_FULL: the device owns the send path
            s = pt.eng.msg_stats(pt.flags[t].data_ptr())
            for k in census[t]:
                census[t][k] += s[k]
            alg_bytes[t] += algorithmic_bytes(pt.n, s["slots"], s["valid"], s["rejects"])
            pt.eng.tick_device(*tick_ptrs(pt, t))
            if args.inflights:
                pt.eng.send_appends(0)
    n_changed = 0
    for pt in parts:
        pt.eng.sync()
        pt.ref_commit, pt.ref_out = pt.eng.results()
        ch, n_fault = pt.eng.result_counts()
        n_changed += ch
        if n_fault:
            raise SystemExit(f"stream raised {n_fault} faults: malformed workload")

    # ---- commit publication (N>1): natively behind the C ABI ----
    # Every tick writes, fused into its store path, one byte per group whose commit index advanced (include/raftgroups.h,
    # "multi-GPU"); rg_publish_commit all-gathers that ~1 B/group slice on the engine's side stream (ncclAllGather over
    # xGMI) while the next ticks run, and every rank keeps a replica of all commit indices, updated lazily. 1 MB per rank
    # and tick at 1 M groups instead of the 8 MB column: per-tick publication at 8 ranks moves 7 MB into each GPU per
    # ~60 us tick (~120 GB/s of its ~450 GB/s xGMI ingress) instead of 56 MB (~930 GB/s: not feasible).
    auto_E = str(args.publish_every).lower() == "auto"
    E = 1 if auto_E else max(1, int(args.publish_every))
    auto_note = None

    class Dev:  # a device range as a torch tensor (CUDA array interface), for the shared-GPU test transport only
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}

    def gloo_allgather(dev_send, dev_recv, nbytes, hip_stream):
        torch.cuda.synchronize()
        send = torch.as_tensor(Dev(dev_send, nbytes), device="cuda").cpu()
        out = torch.empty(world * nbytes, dtype=torch.uint8)
        dist.all_gather_into_tensor(out, send)
        torch.as_tensor(Dev(dev_recv, world * nbytes), device="cuda").copy_(out)
        torch.cuda.synchronize()
        return 0

    transport_note = None
    if distributed:
        from raft_rs_amd import engine as E_
        if not share_gpu:
            # RCCL's first use in a process maps ~0.5 GB and sets up its transports (seconds, minutes on a cold box): paid here,
            # before anything timed or bounded by a timeout -- a one-rank communicator created and destroyed (rg_comm_warmup)
            t_w = time.perf_counter()
            try:
                E_.comm_warmup()
                sys.stderr.write(f"bench.py rank {rank}: rg_comm_warmup {time.perf_counter() - t_w:.2f} s\n")
            except rg.EngineError as e:
                sys.stderr.write(f"bench.py rank {rank}: rg_comm_warmup failed: {e}\n")
        for pt in parts:
            if share_gpu:  # RCCL refuses two ranks on one device: gloo moves the slices (test hook)
                pt.eng.comm_init(rank, world, transport=gloo_allgather)
                continue
            err = ""
            try:
                box = [E_.comm_unique_id() if rank == 0 else None]
            except rg.EngineError as e:  # RCCL cannot be loaded on this box
                box, err = [None], str(e)
            dist.broadcast_object_list(box, src=0)
            if box[0] is not None:
                try:
                    pt.eng.comm_init(rank, world, unique_id=box[0])
                except rg.EngineError as e:
                    err = str(e)
            # every rank must take the same road: if RCCL failed anywhere, all ranks fall back to the host transport
            # (gloo moving the slices through host memory: correct, slow, and SAID SO in the JSON line) rather than crash
            flags = [None] * world
            dist.all_gather_object(flags, err if box[0] is not None else (err or "no unique id"))
            if any(flags):
                transport_note = "RCCL unavailable (" + next(f for f in flags if f)[:200] + "): gloo host transport fallback"
                try:
                    pt.eng.comm_destroy()
                except rg.EngineError:
                    pass
                pt.eng.comm_init(rank, world, transport=gloo_allgather)

    # Size-class engines are independent, so each runs on its own HIP stream (forked from / joined to the
    # main stream around the region): the tail of one engine's launch overlaps the next engine's head.
    multi = len(parts) > 1
    for pt in parts:
        pt.stream = torch.cuda.Stream() if multi else stream
        pt.eng.set_stream(pt.stream.cuda_stream)

    issued = [0.0]  # when the host had issued the last tick / publication of a run_ticks call

    def run_ticks(t0, n, publish, raw=False):
        if multi:
            fork = torch.cuda.Event()
            fork.record(stream)
            for pt in parts:
                pt.stream.wait_event(fork)
        F = 1 if args.inflights else max(1, min(8, args.fuse))
        if F > 1 and not publish:
            # temporal fusion: F consecutive recorded ticks per launch (state stays in registers between them)
            for pt in parts:
                if not hasattr(pt, "out_t"):
                    pt.out_t = torch.empty((F, pt.n), dtype=torch.int32, device="cuda")
            for i in range(0, n, F):
                k = min(F, n - i)
                for pt in parts:
                    pt.eng.tick_device_fused([tick_ptrs(pt, t0 + i + q) for q in range(k)], pt.out_t.data_ptr())
            n = 0  # nothing left for the per-tick loop below
        for i in range(n):
            pub_now = publish and ((i + 1) % E == 0 or i == n - 1)
            for pt in parts:
                if args.inflights and args.fused_send:
                    pt.eng.tick_device_send(*tick_ptrs(pt, t0 + i), max_entries_per_msg=0)
                else:
                    pt.eng.tick_device(*tick_ptrs(pt, t0 + i))
                    if args.inflights:
                        pt.eng.send_appends(0)
                if pub_now:
                    pt.eng.publish_commit(full=raw)  # raw: the 8 B/group column itself (RG_PUBLISH_FULL) instead of the ~1 B/group slice
        issued[0] = time.perf_counter()
        if publish:  # the region ends when every exchange has landed and the replicas are up to date
            for pt in parts:
                pt.eng.publish_sync()
        if multi:
            for pt in parts:
                ev = torch.cuda.Event()
                ev.record(pt.stream)
                stream.wait_event(ev)

    if distributed and auto_E:
        # --publish-every auto: how long does ONE exchange take on this fabric, next to one tick? Up to 32 ticks without
        # publication, then eight back-to-back publications (slices of the real size; what they carry does not change what
        # travels), each bracketed by a control-plane barrier. Rank 0 decides -- E = ceil(exchange / tick), so that an exchange
        # is hidden behind the E ticks that follow it -- and broadcasts E over the control plane (gloo): every rank publishes
        # at the same ticks, and the data path gets no collective of its own for this.
        def wall_of(fn):
            torch.cuda.synchronize()
            dist.barrier()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            return time.perf_counter() - t0

        def eight_publications():
            for _ in range(8):
                for pt in parts:
                    pt.eng.publish_commit()
            for pt in parts:
                pt.eng.publish_sync()

        for pt in parts:
            pt.eng.restore()
            pt.eng.publish_commit(full=True)
            pt.eng.publish_sync()
        n_cal = min(32, T)  # (enough ticks that the launch / synchronisation overhead of the bracket does not pass for tick time)
        t_tick = wall_of(lambda: run_ticks(0, n_cal, False)) / n_cal
        eight_publications()  # (first use of the slices / the communicator: not timed)
        t_ex = wall_of(eight_publications) / 8
        box = [max(1, min(32, int(np.ceil(t_ex / t_tick)))), t_tick, t_ex] if rank == 0 else [None, None, None]
        dist.broadcast_object_list(box, src=0)
        E = int(box[0])
        auto_note = {"publish_every": E, "tick_us": round(box[1] * 1e6, 2), "exchange_us": round(box[2] * 1e6, 2),
                     "rule": "E = ceil(exchange / tick), 1..32; measured on rank 0 over up to 32 ticks and 8 back-to-back publications "
                             "before the timed region, broadcast over the control plane"}

    # ---- timed region(s) ----
    # R regions (--repeats, default 5), each exactly as the contract says -- restored state, W untimed warm-up ticks, then
    # EXACTLY K ticks bracketed by barrier + synchronize on both sides, max over ranks -- and the line reports the MEDIAN region
    # (ms_per_step) with the fastest and slowest beside it (ms_per_step_min / _max): boxes differ by several per cent and one
    # 1 ms region says nothing about spread.
    launch_mode = "eager"
    REPS = max(1, args.repeats)
    walls, kernel_mss, issues = [], [], []
    for rep in range(REPS):
        for pt in parts:
            pt.eng.restore()
            if distributed:
                pt.eng.publish_commit(full=True)  # the replicas restart from the restored columns (outside the region)
        run_ticks(0, W, distributed, args.publish_raw)
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        wall0 = time.perf_counter()
        e0.record(stream)
        run_ticks(W, K, distributed, args.publish_raw)
        e1.record(stream)
        issues.append(issued[0] - wall0)  # the host's share: how long it took to ISSUE the K steps
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        w = time.perf_counter() - wall0
        if distributed:
            tmax = torch.tensor([w], dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            w = float(tmax.item())
        walls.append(w)
        kernel_mss.append(e0.elapsed_time(e1))  # HIP events on the stream the tick kernels run on
    wall = float(np.median(walls))
    kernel_ms = float(np.median(kernel_mss))
    host_issue_s = float(np.median(issues))

    send_stage_info = None
    if args.inflights:
        send_stage_info = {"max_inflight": args.inflights, "max_entries_per_msg": 0,
                           "work_items_last_tick": int(sum(len(pt.eng.send_items()) for pt in parts)),
                           "full_windows": int(sum((pt.eng.read_column(rg.COL.PFLAGS) & rg.PF.INS_FULL).astype(bool).sum()
                                                   for pt in parts))}
    # replay determinism: the timed replay must land on the state the recorded pass produced
    for j, pt in enumerate(parts):
        commit, out = pt.eng.results()
        if not (np.array_equal(commit, pt.ref_commit) and np.array_equal(out, pt.ref_out)):
            raise SystemExit("timed replay diverged from the recorded pass")
        if distributed and os.environ.get("BENCH_SKIP_VERIFY") != "1":  # (skipped only by RG_PUB_DEBUG measurement builds)
            # every rank's replica must hold every rank's shard: compare against the columns themselves
            rep = pt.eng.published_commit()  # [world][n] out of THIS rank's replica
            cols_all = [torch.empty(pt.n, dtype=torch.int64) for _ in range(world)]
            dist.all_gather(cols_all, torch.from_numpy(commit.view(np.int64)))
            for r in range(world):
                if not np.array_equal(rep[r], cols_all[r].numpy().view(np.uint64)):
                    raise SystemExit(f"rank {rank}: published commit indices of rank {r} differ from its commit column")
    pub_stats = parts[0].eng.publish_stats() if distributed else None
    comm_census = None
    if distributed:
        # what the exchange REALLY spanned: the communicator's own ncclCommCount / ncclCommUserRank of every rank (not what the
        # launcher said), gathered over the control plane -- a line that says "8 ranks over RCCL" has eight distinct RCCL ranks
        # of an 8-rank communicator behind it
        ci = parts[0].eng.comm_info()
        infos = [None] * world
        dist.all_gather_object(infos, ci)
        comm_census = {"transport": ci["transport"], "rccl_ranks": ci["rccl_ranks"], "rccl_rank": ci["rccl_rank"],
                       "rccl_ranks_min": min(i["rccl_ranks"] for i in infos), "rccl_ranks_max": max(i["rccl_ranks"] for i in infos),
                       "rccl_distinct_ranks": len({i["rccl_rank"] for i in infos if i["transport"] == "rccl"}),
                       "rccl_engines": sum(1 for i in infos if i["transport"] == "rccl")}
    pub_compare = None
    if distributed:
        if not args.no_publish_compare:
            # the same K ticks once more with the OTHER publication form (delta slices <-> the raw 8 B/group column), so
            # that a multi-GPU run reports both side by side; outside the headline region, same barriers, max over ranks
            other = not args.publish_raw
            for pt in parts:
                pt.eng.restore()
                pt.eng.publish_commit(full=True)
            run_ticks(0, W, True, other)
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            w0 = time.perf_counter()
            run_ticks(W, K, True, other)
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            w2 = torch.tensor([time.perf_counter() - w0], dtype=torch.float64)
            dist.all_reduce(w2, op=dist.ReduceOp.MAX)
            for pt in parts:  # same final state, whatever travelled
                commit, _ = pt.eng.results()
                if not np.array_equal(commit, pt.ref_commit):
                    raise SystemExit("publication comparison run diverged")
                rep = pt.eng.published_commit(rank)
                if not np.array_equal(rep, commit):
                    raise SystemExit("publication comparison run: replica differs from the commit column")
            st2 = parts[0].eng.publish_stats()
            pub_compare = {"mode": "raw 8 B/group column every tick (RG_PUBLISH_FULL)" if other else "delta slices (~1 B/group)",
                           "ms_per_step": float(w2.item()) * 1e3 / K, "value": world * G * K / float(w2.item()),
                           "bytes_per_rank_per_publication": st2["bytes_per_rank_full"] if other else st2["bytes_per_rank_delta"]}

    if args.c5_sizes:
        G = sum(n for _, n in sizes)  # (experiment: the engines' own group counts)
    evals = world * G * K  # (strong scaling: world * G = --total-groups)
    value = evals / wall
    timed_bytes = float(np.mean(alg_bytes[W:]))
    per_launch_s = kernel_ms / 1e3 / K  # one step = one launch per size class (1 except config 5)
    F_used = 1 if (args.inflights or distributed) else max(1, min(8, args.fuse))
    fused_model = None
    if F_used > 1:
        # Temporal fusion has its own byte model: ONE launch runs F consecutive ticks with the group's state in registers, so
        # per launch the state part of SURVEY 8(d) moves once and the message / result parts once per tick. Splitting
        # B = 9 P + 58 A + 8 R + 37 by stream (SURVEY.md 8d's table): state = 8 P (match) + 17 A (next, flags, pr_commit read)
        # + 25 A (written) + 28 + 8 (commit, log range, masks; commit written) = 8 P + 42 A + 36; messages and result of one
        # tick = P (flag bytes) + 16 A (m_index, m_commit) + 8 R (hints) + 1. A launch of T ticks: state + sum over its ticks
        # (T = 1 gives B back). `A` of the state part = the acked slots of the launch's first tick (the same followers ack in
        # every tick of the stream).
        launches = []
        for i in range(W, W + K, F_used):
            ts = range(i, min(i + F_used, W + K))
            c0 = census[ts[0]]
            launches.append(8 * c0["slots"] + 42 * c0["valid"] + 36 * G +
                            sum(census[t]["slots"] + 16 * census[t]["valid"] + 8 * census[t]["rejects"] + G for t in ts))
        timed_bytes = float(np.mean(launches))
        per_launch_s = kernel_ms / 1e3 / len(launches)
        fused_model = {"ticks_per_launch": F_used, "launches": len(launches),
                       "per_tick_model_bytes": float(np.mean(alg_bytes[W:])),
                       "note": "state moves once per launch, messages + result once per tick: 8P+42A+36 + sum_t(P+16A_t+8R_t+1)"}
    achieved = timed_bytes / per_launch_s / 1e9
    A = float(np.mean([c["valid"] for c in census[W:]])) / G
    R = float(np.mean([c["rejects"] for c in census[W:]])) / G
    EL = float(np.mean([c["elections"] for c in census[W:]])) / G

    # HBM traffic per launch: PMC-measured in separate rocprofv3 passes of this same command
    # (tools/summarize_prof.py -> profiles/traffic.json); null for configurations not profiled.
    traffic, traffic_source, traffic_stale = (None, None, None)
    if not args.inflights and args.split == 1 and args.fuse == 1:
        traffic, traffic_source, traffic_stale = traffic_lookup(
            f"{args.workload}:{G}:{P}" + (":sorted" if args.sorted else ":one-engine" if (args.workload == 5 and args.one_engine) else "") +
            (f":v{args.variant}" if args.variant else ""))
    hot = sum(hot_state_bytes(pt.n, pt.slots, bool(args.inflights)) for pt in parts)

    result = {
        "metric": "raft-group progress+commit evaluations/sec (commit-index recomputes/sec at 1M groups x 5 peers)",
        "value": value, "unit": "group-evals/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": wall * 1e3 / K, "ms_per_step_min": min(walls) * 1e3 / K, "ms_per_step_max": max(walls) * 1e3 / K, "repeats": REPS,
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": {"workload": (f"{world * G} groups x 7 peers sharded over {world} GPUs ({G} per GPU), commit indices published "
                                f"every {'tick' if E == 1 else str(E) + ' ticks'} (BASELINE configs[3]: 8 M x 7 over 8 GPUs"
                                f"{'' if (world, G) == (8, 1_000_000) else ' -- here at ' + str(world) + ' x ' + str(G)})"
                                if (args.workload == 2 and P == 7 and distributed) else
                                workload_label(args.workload, G, P, args.one_engine, args.sorted) +
                                (f", x {world} ranks ({'strong scaling: ' + str(world * G) + ' groups in total' if strong else 'weak scaling'}), commit indices published every "
                                 f"{'tick' if E == 1 else str(E) + ' ticks'}" if distributed else "")),
                   "groups_per_gpu": G, "peer_slots": P, "workload_id": args.workload, "seed": hex(args.seed),
                   "acks_per_group": round(A, 3), "rejects_per_group": round(R, 5),
                   **({"elections_per_group": round(EL, 5),
                       "rollover": "every tick 1/32 of the groups elects a new leader (RG_MF_BECOME_LEADER = Raft::reset + "
                                   "become_leader): ~10% of the groups are between election and the first commit of the new "
                                   "term at any time"} if args.workload == 5 else {}),
                   "kernel_variant": {0: "lane", 1: "lane", 2: "lds", 4: "lds-dma", 5: "compact"}[args.variant],
                   "engines": [{"slots": pt.slots, "groups": pt.n} for pt in parts],
                   "device": {k: v for k, v in parts[0].eng.device_info().items() if k != "engine_bytes"},
                   "engine_hbm_bytes": sum(pt.eng.device_info()["engine_bytes"] for pt in parts),
                   "sharding": f"{world} disjoint group ranges" + (
                       f", commit indices published every {E} tick(s) through rg_publish_commit: "
                       f"{'gloo transport callback (shared-GPU test hook)' if share_gpu else (transport_note or 'ncclAllGather (RCCL)')} of "
                       f"{pub_stats['bytes_per_rank_delta']} B/rank delta slices (full column: {pub_stats['bytes_per_rank_full']} B)"
                       if distributed else ""),
                   **(comm_census or {}),
                   **({"publication": pub_stats,
                       "publication_mode": "raw 8 B/group column every tick (RG_PUBLISH_FULL)" if args.publish_raw else "delta slices (~1 B/group)",
                       "publication_compare": pub_compare, "publish_every": E,
                       **({"publish_every_auto": auto_note} if auto_note else {})} if distributed else {}),
                   "launch": launch_mode, "host_issue_us_per_step": round(host_issue_s * 1e6 / K, 2), "ticks_per_launch": max(1, min(8, args.fuse)) if not (distributed or args.inflights) else 1},
        "roofline": {"bound": "hbm", "regime": regime_of(hot), "hot_state_bytes": hot,
                     "regime_note": "infinity-cache: hot state fits the 256 MB MALL, frac can exceed a pure HBM stream; HBM figure: frac_c2_hbm_8M",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                     "traffic_stale": traffic_stale, "csrc_sha16": csrc_sha16(),
                     "kernel": "k_tick_send" if (args.inflights and args.fused_send) else
                               "k_tick_fused" if (args.fuse > 1 and not distributed and not args.inflights) else
                               ({2: "k_tick_lds", 4: "k_tick_lds", 5: "k_tick_compact"}.get(args.variant, "k_tick_classes" if args.sorted else "k_tick_lane")) +
                               (" + k_send_dense" if args.inflights else ""),
                     "algorithmic_bytes_per_launch": timed_bytes, "bytes_per_eval": timed_bytes / G / F_used,
                     "avg_launch_us": per_launch_s * 1e6,
                     "avg_launch_us_min": min(kernel_mss) * 1e3 / K * (K / max(1, len(range(W, W + K, F_used)))),
                     "avg_launch_us_max": max(kernel_mss) * 1e3 / K * (K / max(1, len(range(W, W + K, F_used)))),
                     **({"fused_model": fused_model} if fused_model else {}),
                     **({"note": "avg_launch_us is the tick AND its send stage; the algorithmic bytes are the tick's "
                                 "(SURVEY 8d counts no bytes for the send path), so frac understates this mode"}
                        if args.inflights else {})},
        **({"send_stage": send_stage_info} if args.inflights else {}),
        "commit_changed_last_tick": int(n_changed),
    }
    for pt in parts:  # free the headline engines' HBM before the side measurements
        pt.eng.close()
        pt.cols = pt.flags = None
    torch.cuda.empty_cache()
    if world == 1 and not distributed and not args.no_extras:
        wl = args.workload if args.workload != 5 else 2
        # the literal BASELINE metric ("commit-index recomputes/sec"): Raft::maybe_commit for every group, no messages --
        # at BASELINE's size (77 MB working set: an Infinity-Cache number) and at 8 M groups (616 MB: an HBM number)
        result["recompute_only"] = guarded(run_config, rg, torch, G, P, wl, 5, 50, args.seed, what="recompute")
        result["recompute_only_out_of_cache"] = guarded(run_config, rg, torch, args.out_of_cache_groups, P, wl, 3, 20, args.seed,
                                                           what="recompute")
        # the headline configuration beyond the 256 MB Infinity Cache (state + one tick of messages >> cache)
        result["out_of_cache"] = guarded(run_config, rg, torch, args.out_of_cache_groups, P, wl, 3, 12, args.seed)
        # ... and just beyond it (1.3 - 2.5 x the cache): a leading range of the groups stays resident, the rest is streamed,
        # one launch (k_tick_split; the engine is alone in the process here: the headline's engines are closed)
        result["between_regimes"] = guarded(run_config, rg, torch, 2_400_000, P, wl, 3, 20, args.seed)
        # every other BASELINE configuration that fits one GPU, each with its own roofline object (driver-timed like the
        # headline): configs[2] joint, one rank's shard of configs[3], configs[4] in both layouts, and configs[1] with the
        # Inflights on the device and the send stage after every tick
        oc = {}
        c5v = args.c5_variant
        for name, kw in (("configs[2] joint", dict(n_groups=1_000_000, n_slots=5, workload=3)),
                         ("configs[1] group commit", dict(n_groups=1_000_000, n_slots=5, workload=2, group_commit=True)),
                         ("configs[3] one rank's shard", dict(n_groups=1_000_000, n_slots=7, workload=2)),
                         ("configs[4] one launch, class-sorted", dict(n_groups=1_000_000, n_slots=7, workload=5, sorted_classes=True)),
                         ("configs[4] interleaved, after rg_permute_groups", dict(n_groups=1_000_000, n_slots=7, workload=5, sorted_classes=True,
                                                                                  place_after_load=True)),
                         ("configs[4] one launch, class-sorted, 8M groups", dict(n_groups=8_000_000, n_slots=7, workload=5, sorted_classes=True,
                                                                                 warmup=3, steps=10)),
                         ("configs[4] size-class engines", dict(n_groups=1_000_000, n_slots=7, workload=5, variant=c5v)),
                         ("configs[4] one 7-slot engine", dict(n_groups=1_000_000, n_slots=7, workload=5, variant=c5v, one_engine=True)),
                         ("configs[1] + send stage", dict(n_groups=1_000_000, n_slots=5, workload=2, inflights=256)),
                         ("configs[1] + send stage, one launch", dict(n_groups=1_000_000, n_slots=5, workload=2, inflights=256,
                                                                      fused_send=True))):
            if (kw["workload"], kw["n_groups"], kw["n_slots"]) == (args.workload, G, P) and not kw.get("inflights") and not kw.get("group_commit") \
                    and kw.get("variant", 0) == args.variant and kw.get("sorted_classes", False) == args.sorted \
                    and (kw.get("one_engine", False) or kw.get("sorted_classes", False)) == args.one_engine:
                continue  # that is the headline itself
            oc[name] = guarded(run_config, rg, torch, seed=args.seed, **{"warmup": 5, "steps": 30, **kw})
            torch.cuda.empty_cache()
        result["other_configs"] = oc
        result["roofline"]["by_config"] = by_config_summary(result)
        # ... and once more FLAT, as scalar keys of `roofline` (frac_<cfg>, us_<cfg>, us_min_/us_max_<cfg>, mb_<cfg> = algorithmic
        # MB per launch, traffic_mb_<cfg> = PMC HBM MB per launch or null): scalars are what the driver's record keeps of a
        # sub-object, so frac = mb / us / 8 TB/s of EVERY configuration can be recomputed from BENCH_rNN.json.parsed alone
        result["roofline"].update(flat_config_keys(result["roofline"]["by_config"]))
        # the other end of the scale: the round trip of a flush that touches 1 / 10 groups (not a throughput number)
        result["small_batch_latency"] = guarded(small_batch_latency, rg, torch, G, P, args.seed)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = guarded(cpu_baseline, min(args.cpu_sample_groups, G), P, args.workload,
                                              args.cpu_sample_ticks, args.seed, os.cpu_count() or 1)
    elif rank == 0:
        result["cpu_baseline"] = None
    if distributed:
        dist.destroy_process_group()
    if rank == 0:
        emit(result)  # (the ONLY thing this process writes to its stdout)


if __name__ == "__main__":
    main()
