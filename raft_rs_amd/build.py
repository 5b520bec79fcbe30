"""Build the HIP engine (libraftgroups.so) in-tree with hipcc for gfx950.

    python -m raft_rs_amd.build [--force] [--opt N ...]

hipcc cross-compiles without a GPU. The tick kernels are instantiated once per slot count
(csrc/tick_inst.hip, -DRG_P=1..8) and compiled in parallel with the ABI units (csrc/abi_*.hip: state / tick /
send / mirror / wire / publish / placement, along the sections of include/raftgroups.h), then linked. Everything is built with
-fvisibility=hidden: the exported symbols are exactly the entry points the public header declares.
The .so is git-ignored but travels with gpurun snapshots.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(PKG, "build")
LIB = os.path.join(PKG, "libraftgroups.so")
UNITS = ["abi_state.hip", "abi_tick.hip", "abi_send.hip", "abi_mirror.hip", "abi_wire.hip", "abi_publish.hip", "abi_placement.hip"]
DEPS = UNITS + ["tick_inst.hip", "rg_engine.h", "rg_abi_guard.h", "rg_common.h", "rg_group.h", "rg_send.h", "rg_wire.h", "rg_workload.h", "rg_tick_kernels.h",
                "rg_publish.h", "rg_kernels_quorum.h", "rg_kernels_sparse.h", "rg_kernels_send.h", "rg_kernels_state.h",
                "rg_kernels_workload.h", "rg_kernels_publish.h", "rg_kernels_placement.h", os.path.join("..", "..", "include", "raftgroups.h")]
ARCH = "gfx950"
CFLAGS = ["-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "--offload-arch=" + ARCH, "-Wall", "-Wno-unused-function",
          "-Wno-pass-failed"]


FORCE = [False]  # --force: recompile every object


def hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP engine cannot be built (there is no CPU fallback)")


def is_stale(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, f) for f in DEPS]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def _compile(args):
    src, obj, defs, verbose = args
    # an object is up to date when it is newer than its own source and every header (the units share them all)
    deps = [src, os.path.abspath(__file__)] + [os.path.join(CSRC, f) for f in DEPS if f.endswith(".h")]
    if not FORCE[0] and os.path.exists(obj) and all(os.path.getmtime(obj) > os.path.getmtime(d) for d in deps if os.path.exists(d)):
        return obj
    cmd = [hipcc()] + CFLAGS + defs + ["-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return obj


def _build(lib, tag, defs, verbose):
    if not all(os.path.exists(os.path.join(CSRC, f)) for f in UNITS + ["tick_inst.hip"]):
        raise RuntimeError("engine sources missing under " + CSRC)
    os.makedirs(OBJ, exist_ok=True)
    jobs = []
    for p in (7, 8, 5, 6):  # (the slowest units first)
        jobs.append((os.path.join(CSRC, "tick_inst.hip"), os.path.join(OBJ, f"tick_p{p}{tag}.o"),
                     defs + [f"-DRG_P={p}"], verbose))
    jobs += [(os.path.join(CSRC, u), os.path.join(OBJ, u.replace(".hip", f"{tag}.o")), defs, verbose) for u in UNITS]
    for p in range(1, 5):
        jobs.append((os.path.join(CSRC, "tick_inst.hip"), os.path.join(OBJ, f"tick_p{p}{tag}.o"),
                     defs + [f"-DRG_P={p}"], verbose))
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 2)) as ex:
        objs = list(ex.map(_compile, jobs))
    cmd = [hipcc(), "-shared", "-fPIC", "--offload-arch=" + ARCH] + objs + ["-o", lib]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return lib


def build_opt(opt, verbose=False):
    """Experiment build: libraftgroups_opt<N>.so with -DRG_OPT=N (select it with RG_LIB_PATH)."""
    return _build(os.path.join(PKG, f"libraftgroups_opt{opt}.so"), f"_opt{opt}", [f"-DRG_OPT={opt}"], verbose)


def build_exp(name, defs, verbose=False):
    """Experiment build with arbitrary -D flags: libraftgroups_<name>.so."""
    return _build(os.path.join(PKG, f"libraftgroups_{name}.so"), "_" + name, list(defs), verbose)


def build(force=False, verbose=False):
    if not force and not is_stale():
        return LIB
    FORCE[0] = bool(force)
    return _build(LIB, "", [], verbose)


if __name__ == "__main__":
    if "--exp" in sys.argv:  # --exp name -DFOO=1 -DBAR=2
        i = sys.argv.index("--exp")
        print(build_exp(sys.argv[i + 1], sys.argv[i + 2:]))
    elif "--opt" in sys.argv:
        for o in sys.argv[sys.argv.index("--opt") + 1:]:
            print(build_opt(int(o), verbose=False))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
