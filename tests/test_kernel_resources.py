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


def resource_usage(p):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-c", os.path.join(ROOT, "raft_rs_amd", "csrc", "tick_inst.hip"),
           "-o", os.devnull, "-Wno-pass-failed", "-Rpass-analysis=kernel-resource-usage", f"-DRG_P={p}"]
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
    return rows


def test_tick_kernels_keep_their_register_budget():
    rows = resource_usage(5)
    lane = next(v for k, v in rows.items() if "k_tick_laneILi5ELb0E" in k)
    lst = next(v for k, v in rows.items() if "k_tick_listILi5ELb0E" in k)
    fused = next(v for k, v in rows.items() if "k_tick_fusedILi5ELb0E" in k)
    # the dense sweep is the bandwidth-bound kernel: 4 waves/SIMD. (Round 2 added two rare paths to it -- the election
    # event, in registers because a reload would cost every wave of config 5 a second memory round trip, and the
    # publication byte: 110 -> 125 VGPRs, same occupancy, same measured time: profiles/r02_*.)
    assert int(lane["VGPRs"]) <= 128 and int(lane["Occupancy [waves/SIMD]"]) >= 4, lane
    # the sparse-path kernel carries the list / result-gather pointers on top: latency-bound, 3 waves/SIMD is fine
    assert int(lst["VGPRs"]) <= 136 and int(lst["Occupancy [waves/SIMD]"]) >= 3, lst
    for name, r in (("k_tick_lane<5,false>", lane), ("k_tick_list<5,false>", lst)):
        assert int(r["ScratchSize [bytes/lane]"]) == 0, (name, r)
    assert int(fused["VGPRs"]) <= 168 and int(fused["ScratchSize [bytes/lane]"]) == 0, fused  # 3 waves/SIMD
