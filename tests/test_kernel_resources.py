"""CPU-only guard: the hot kernels keep their register budget (hipcc -Rpass-analysis=kernel-resource-usage).

The tick kernel's occupancy (4 waves/SIMD at P=5) is part of the measured performance; a rare-path
change once pushed it from 110 to 150 VGPRs unnoticed. This test compiles the P=5 instance and pins
VGPRs / scratch of the kernels the benchmark runs."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


_CACHE = {}


def resource_usage(p):
    """Kernel resource remarks of the tick kernels at P = p. The three slot counts the tests look at are compiled side by side
    on first use (a hipcc run per slot count takes ~50 s; one after the other they were half of the CPU suite's wall time)."""
    if not _CACHE:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=3) as ex:
            for q, rows in zip((5, 3, 7), ex.map(_resource_usage, (5, 3, 7))):
                _CACHE[q] = rows
    if _CACHE.get(p) is None:
        pytest.skip("hipcc not available")
    return _CACHE[p]


def _resource_usage(p):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        return None
    import tempfile
    asm = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    # device code only, as assembly: the remarks come from the same compile, and the text says which stores kept their
    # non-temporal bit (test_streamed_stores_are_streamed)
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S",
           os.path.join(ROOT, "raft_rs_amd", "csrc", "tick_inst.hip"),
           "-o", asm, "-Wno-pass-failed", "-Rpass-analysis=kernel-resource-usage", f"-DRG_P={p}"]
    err = subprocess.run(cmd, stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True).stderr
    rows, cur = {}, None
    for line in err.splitlines():
        m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
        if not m:
            continue
        txt = m.group(1).strip()
        if txt.startswith("Function Name:"):
            cur = rows.setdefault(txt.split(":", 1)[1].strip(), {})
        elif cur is not None and ":" in txt:
            k, v = txt.split(":", 1)
            cur[k.strip()] = v.strip()
    name = None
    try:
        with open(asm) as f:
            for line in f:
                m = re.match(r"(_Z\w+):", line)
                if m:
                    name = m.group(1)
                elif name in rows and re.search(r"\bglobal_store_\w+ .* nt\b", line):
                    rows[name]["nt stores"] = rows[name].get("nt stores", 0) + 1
                elif name in rows and re.search(r"\bglobal_load_\w+ .* nt\b", line):
                    rows[name]["nt loads"] = rows[name].get("nt loads", 0) + 1
    finally:
        os.unlink(asm)
    return rows


def test_tick_kernels_keep_their_register_budget():
    rows = resource_usage(5)
    lane = next(v for k, v in rows.items() if "k_tick_laneILi5ELb0EjLi0EE" in k)    # 32-bit cell offsets: what the bench runs
    lane64 = next(v for k, v in rows.items() if "k_tick_laneILi5ELb0EmLi0EE" in k)  # engines beyond 4 GiB per column
    lst = next(v for k, v in rows.items() if "k_tick_listILi5ELb0E" in k)
    fused = next(v for k, v in rows.items() if "k_tick_fusedILi5ELb0E7rg_u32oE" in k)
    # the dense sweep is the bandwidth-bound kernel: 4 waves/SIMD. (Round 2 added the election event, the publication
    # byte and the rare-path prefetch -- reject hints, the election's cold cells, issued with the bulk loads -- to it;
    # the `SGPR base + 32-bit offset` addressing of rg_at paid for their registers: profiles/r02_*.)
    assert int(lane["VGPRs"]) <= 128 and int(lane["Occupancy [waves/SIMD]"]) >= 4, lane
    assert int(lane64["Occupancy [waves/SIMD]"]) >= 3, lane64
    lane_nt = next(v for k, v in rows.items() if "k_tick_laneILi5ELb0EjLi1EE" in k)  # message columns streamed (large engines)
    assert int(lane_nt["VGPRs"]) <= 128 and int(lane_nt["ScratchSize [bytes/lane]"]) == 0, lane_nt
    lane_all = next(v for k, v in rows.items() if "k_tick_laneILi5ELb0EjLi2EE" in k)  # everything streamed (engines far beyond the cache)
    assert int(lane_all["VGPRs"]) <= 128 and int(lane_all["ScratchSize [bytes/lane]"]) == 0, lane_all
    # the sparse-path kernel carries the list / result pointers on top; round 3 (the term-run table read from memory
    # behind the stores instead of prefetched into registers) brought it to 4 waves as well
    assert int(lst["VGPRs"]) <= 128 and int(lst["Occupancy [waves/SIMD]"]) >= 4, lst
    cpt = next(v for k, v in rows.items() if "k_tick_compactILi5ELb0EjE" in k)
    assert int(cpt["Occupancy [waves/SIMD]"]) >= 4 and int(cpt["ScratchSize [bytes/lane]"]) == 0, cpt
    for name, r in (("k_tick_lane<5,false,u32>", lane), ("k_tick_lane<5,false,u64>", lane64), ("k_tick_list<5,false>", lst)):
        assert int(r["ScratchSize [bytes/lane]"]) == 0, (name, r)
    # the fused kernel (8 message sets, the election event included since round 3): 4 waves/SIMD since its cell offsets are
    # opaque next to every access (rg_u32o: SGPR-base + VGPR-offset addressing instead of a 64-bit VGPR address pair per cell;
    # 156 -> 115 VGPRs), no scratch
    assert int(fused["VGPRs"]) <= 128 and int(fused["Occupancy [waves/SIMD]"]) >= 4 and int(fused["ScratchSize [bytes/lane]"]) == 0, fused
    # round 4: the tick and its send stage in one launch at FOUR waves per SIMD (its phases read their column pointers from the
    # kernarg segment themselves: 145 -> 123 VGPRs), and the one-launch kernel for shards placed by size class at the plain
    # kernel's occupancy (its bodies do the same: the first form carried 1 566 spill-lane instructions)
    send = next(v for k, v in rows.items() if "k_tick_sendILi5ELb0EjLb0EE" in k)
    assert int(send["VGPRs"]) <= 128 and int(send["Occupancy [waves/SIMD]"]) >= 4 and int(send["ScratchSize [bytes/lane]"]) == 0, send
    send_nt = next(v for k, v in rows.items() if "k_tick_sendILi5ELb0EjLb1EE" in k)  # round 5: the tick's state columns streamed as well
    assert int(send_nt["VGPRs"]) <= 128 and int(send_nt["Occupancy [waves/SIMD]"]) >= 4 and int(send_nt["ScratchSize [bytes/lane]"]) == 0, send_nt
    cls = next(v for k, v in rows.items() if "k_tick_classesILi5EjLi0EE" in k)
    assert int(cls["VGPRs"]) <= 128 and int(cls["Occupancy [waves/SIMD]"]) >= 4 and int(cls["ScratchSize [bytes/lane]"]) == 0, cls


def test_occupancy_of_the_other_slot_counts():
    # P = 3 runs at 5 waves/SIMD, P = 7 (config 4's shard) at 3
    for p, waves in ((3, 5), (7, 3)):
        rows = resource_usage(p)
        lane = next(v for k, v in rows.items() if f"k_tick_laneILi{p}ELb0EjLi0EE" in k)
        assert int(lane["Occupancy [waves/SIMD]"]) >= waves and int(lane["ScratchSize [bytes/lane]"]) == 0, (p, lane)
        for name in (f"k_tick_sendILi{p}ELb0EjLb0EE", f"k_tick_listILi{p}ELb0EjE"):  # no scratch at any slot count
            r = next(v for k, v in rows.items() if name in k)
            assert int(r["ScratchSize [bytes/lane]"]) == 0, (p, name, r)
    # config 5's one launch: the 7-slot body sets the allocation of every class. Round 5: it loads committed_index / Message.commit
    # behind the commit phase (RgLatePc, opaque offsets) in the cached regimes -- FOUR waves per SIMD, no scratch; the all-streamed
    # instantiation keeps the plain body (three waves)
    for ntm, waves, vgprs in ((0, 4, 128), (1, 4, 128), (2, 3, 168)):
        cls7 = next(v for k, v in resource_usage(7).items() if f"k_tick_classesILi7EjLi{ntm}EE" in k)
        assert int(cls7["VGPRs"]) <= vgprs and int(cls7["Occupancy [waves/SIMD]"]) >= waves and int(cls7["ScratchSize [bytes/lane]"]) == 0, (ntm, cls7)


def test_group_commit_kernels_carry_no_scratch():
    """Round 6: every GC = true instantiation (the kernels an engine launches once some group has ProgressTracker.group_commit
    on) keeps the group in registers -- through round 5 the literal group-commit evaluation was a __noinline__ function taking
    the match and gid arrays by reference, which put 400-512 B per lane into scratch in all of them. The dense 5-slot kernel
    (what the bench's group-commit figure runs) holds three waves per SIMD."""
    for p in (5, 7):
        rows = resource_usage(p)
        gc = {k: v for k, v in rows.items() if re.search(r"k_tick_(lane|list|send|compact)ILi%dELb1E" % p, k) or
              re.search(r"k_tick_fusedILi%dELb1E" % p, k)}
        assert len(gc) >= 6, sorted(gc)
        for k, v in gc.items():
            assert int(v["ScratchSize [bytes/lane]"]) == 0, (k, v)
    lane = next(v for k, v in resource_usage(5).items() if "k_tick_laneILi5ELb1EjLi0EE" in k)
    assert int(lane["VGPRs"]) <= 160 and int(lane["Occupancy [waves/SIMD]"]) >= 3, lane


def test_streamed_stores_are_streamed():
    """The non-temporal bit of the stores the design streams is IN the ISA. Round 4 found it was not: rg_st took the choice as
    a run-time bool that was constant at every call site, and after inlining LLVM merged the two stores of the if/else into
    one plain store -- k_tick_send wrote its item columns through the cache since round 3 (the run-to-run bimodality of the
    one-launch send form), and the first all-streamed build had streamed loads only (profiles/r04_nt_state.txt)."""
    rows = resource_usage(5)
    def one(tag):
        return next(v for k, v in rows.items() if tag in k)
    plain = one("k_tick_laneILi5ELb0EjLi0EE")
    assert plain.get("nt stores", 0) == 0 and plain.get("nt loads", 0) == 0, plain
    msgs = one("k_tick_laneILi5ELb0EjLi1EE")      # messages streamed: 11 loads (5 slots x 2 + the election word), no store
    assert msgs.get("nt loads", 0) >= 11 and msgs.get("nt stores", 0) == 0, msgs
    alls = one("k_tick_laneILi5ELb0EjLi2EE")      # everything streamed: every state store of the dense path (3 P + 1 at least)
    assert alls.get("nt stores", 0) >= 16 and alls.get("nt loads", 0) >= 11 + 15, alls
    cls = one("k_tick_classesILi5EjLi2EE")
    assert cls.get("nt stores", 0) >= 16, cls
    split = one("k_tick_splitILi5EjE")            # both bodies in one kernel: the streamed one's stores, at the lane kernel's budget
    assert split.get("nt stores", 0) >= 16 and int(split["VGPRs"]) <= 128 and int(split["ScratchSize [bytes/lane]"]) == 0, split
    send = one("k_tick_sendILi5ELb0EjLb0EE")      # the item columns (3 x 5) and the window columns (3 x 5, stored at two places);
    assert send.get("nt stores", 0) >= 45, send   # loads: the messages (11) and the window columns (15)
    assert send.get("nt loads", 0) >= 26, send
    send_all = one("k_tick_sendILi5ELb0EjLb1EE")  # ... and the tick's own state columns, loads and stores (RG_CACHE_STREAM_ALL)
    assert send_all.get("nt stores", 0) >= send.get("nt stores", 0) + 16 and send_all.get("nt loads", 0) >= send.get("nt loads", 0) + 15, (send, send_all)
