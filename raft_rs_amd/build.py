"""Build the HIP engine (libraftgroups.so) in-tree with hipcc for gfx950.

    python -m raft_rs_amd.build [--force]

hipcc cross-compiles without a GPU. The .so is git-ignored but travels with gpurun snapshots.
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libraftgroups.so")
SOURCES = ["engine.hip"]
HEADERS = ["rg_common.h", "rg_group.h", "rg_workload.h", os.path.join("..", "..", "include", "raftgroups.h")]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", "--offload-arch=" + ARCH, "-Wall", "-Wno-unused-function"]


def hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP engine cannot be built (there is no CPU fallback)")


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force=False, extra_flags=(), verbose=False):
    if not force and not is_stale():
        return LIB
    if not all(os.path.exists(os.path.join(CSRC, f)) for f in SOURCES):
        raise RuntimeError("engine sources missing under " + CSRC)
    cmd = [hipcc()] + FLAGS + list(extra_flags) + [os.path.join(CSRC, f) for f in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
