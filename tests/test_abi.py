"""CPU-only: the C-ABI library loads and exports every symbol include/raftgroups.h declares; the bit
layouts the oracle's SoA adapter uses are the header's; without a GPU the engine fails loudly."""
import ctypes
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "raftgroups.h")


def declared_functions():
    """Every function prototype of the header (parsed by the binding generator: any return type)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_rust_bindings
    funcs = gen_rust_bindings.parse(open(HEADER, encoding="utf-8").read())[3]
    return sorted({name for name, _, _ in funcs})


def header_defines():
    src = open(HEADER).read()
    return {k: int(v.rstrip("u"), 0) for k, v in re.findall(r"#define\s+(RG_\w+)\s+(0x[0-9a-fA-F]+u?|\d+u?)\s", src)}


def test_header_declares_the_expected_surface():
    names = declared_functions()
    assert len(names) >= 30
    for must in ("rg_create", "rg_tick", "rg_tick_device", "rg_recompute", "rg_results", "rg_step", "rg_flush",
                 "rg_vote_result", "rg_workload_gen"):
        assert must in names


def test_library_exports_every_declared_symbol(rg):
    lib = ctypes.CDLL(rg.LIB_PATH)
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, missing
    from raft_rs_amd import engine as E
    unbound = [n for n in declared_functions() if n not in E.SYMBOLS]
    assert not unbound, f"python binding lacks {unbound}"


def test_abi_version_is_the_headers_everywhere(rg):
    """The ABI is source-compatible only (rg_config / rg_device_info grow in place): the header's RG_ABI_VERSION, what the built
    library reports, the Python binding's layouts and the generated Rust constant name the same version."""
    from raft_rs_amd import engine as E
    v = header_defines()["RG_ABI_VERSION"]
    lib = ctypes.CDLL(rg.LIB_PATH)
    lib.rg_abi_version.restype = ctypes.c_uint32
    assert lib.rg_abi_version() == v == E.ABI_VERSION
    rs = open(os.path.join(os.path.dirname(HEADER), "..", "bindings", "raftgroups.rs")).read()
    assert re.search(r"pub const RG_ABI_VERSION: u32 = %d;" % v, rs)


def test_oracle_adapter_uses_the_header_bit_layouts():
    d = header_defines()
    osrc = open(os.path.join(ROOT, "oracle", "raft_oracle.c")).read()
    odef = {k: int(v.rstrip("u"), 0) for k, v in re.findall(r"#define\s+(RO_\w+)\s+(0x[0-9a-fA-F]+u?)\s", osrc)}
    for name in ("PF_STATE_MASK", "PF_PAUSED", "PF_RECENT_ACTIVE", "MF_VALID", "MF_REJECT", "MF_HAS_RS",
                 "MF_INS_FULL", "MF_SENT", "MF_APPEND", "MF_HEARTBEAT", "OUT_CHANGED", "OUT_FAULT", "OUT_TIMEOUT_NOW"):
        assert d["RG_" + name] == odef["RO_" + name], name
    from raft_rs_amd import engine as E
    assert (E.MF.VALID, E.MF.REJECT, E.MF.HAS_RS, E.MF.INS_FULL, E.MF.SENT, E.MF.APPEND) == tuple(
        d["RG_MF_" + k] for k in ("VALID", "REJECT", "HAS_RS", "INS_FULL", "SENT", "APPEND"))
    assert E.cfg_make(0x07, 0x0e, 2, True, 3, 0x1f) == (0x07 | (0x0e << 8) | (2 << 16) | 0x80000 | (3 << 20) | (0x1f << 24))


def test_term_run_table_depth_is_the_same_everywhere():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fuzz
    import oracle_lib as O
    from raft_rs_amd import engine as E
    n = header_defines()["RG_TERM_RUNS"]
    ro = int(re.search(r"#define RO_TERM_RUNS (\d+)", open(os.path.join(ROOT, "oracle", "raft_oracle.h")).read()).group(1))
    assert n == ro == O.TERM_RUNS == fuzz.TERM_RUNS == E.TERM_RUNS == 8


def test_engine_fails_loudly_without_a_gpu(rg):
    lib = rg.load_library()
    if lib.rg_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(rg.EngineError) as ei:
        rg.Engine(16, 3)
    assert ei.value.code == -2 and "no CPU fallback" in str(ei.value)


def test_library_reads_no_environment(rg):
    """Which kernel an engine runs is decided by rg_config alone: the shipped library neither imports getenv nor carries the
    name of one of the old measurement hooks (RG_NT_*, RG_NO_CLASSES, ...; the experiment builds of raft_rs_amd.build --exp
    may, under their own macros)."""
    import subprocess
    undefined = subprocess.run(["nm", "-D", "-u", rg.LIB_PATH], stdout=subprocess.PIPE, text=True, check=True).stdout
    assert "getenv" not in undefined
    blob = open(rg.LIB_PATH, "rb").read()
    for name in (b"RG_NT_", b"RG_NO_CLASSES", b"RG_CLASS_ORDER", b"RG_FORCE_IX64", b"RG_PUB_DEBUG"):
        assert name not in blob, name
    from raft_rs_amd import engine as E
    assert C.sizeof(E._Config) == 40 and C.sizeof(E.DeviceInfo) == 112


def test_product_package_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "raft_rs_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in txt.lower(), os.path.join(dirpath, f)


def test_oracle_flag_bits_are_the_headers():
    """The oracle's SoA adapter speaks the engine's flag byte: every RO_PF_* / RO_MF_* / RO_OUT_* constant of
    oracle/raft_oracle.c that has an RG_* namesake in include/raftgroups.h carries the same value."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "raftgroups.h")).read()
    src = open(os.path.join(root, "oracle", "raft_oracle.c")).read() + open(os.path.join(root, "oracle", "raft_oracle.h")).read()
    rg = {m.group(1): int(m.group(2), 16) for m in re.finditer(r"#define RG_((?:PF|MF|OUT)_\w+)\s+(0x[0-9a-fA-F]+)u", hdr)}
    ro = {m.group(1): int(m.group(2), 16) for m in re.finditer(r"#define RO_((?:PF|MF|OUT)_\w+)\s+(0x[0-9a-fA-F]+)u", src)}
    common = sorted(set(rg) & set(ro))
    assert len(common) >= 12, common
    assert {"PF_PEND_SNAP", "PF_PEND_RS", "PF_INS_FULL", "PF_PAUSED"} <= set(common)
    for k in common:
        assert rg[k] == ro[k], (k, hex(rg[k]), hex(ro[k]))


def test_rust_binding_is_generated_from_the_header_and_complete(rg):
    """INTEGRATION.md's `extern "C"` block and bindings/raftgroups.rs are OUTPUT of tools/gen_rust_bindings.py: up to date,
    one declaration per function the header declares = per symbol libraftgroups.so exports, with the header's arity."""
    import subprocess
    import sys
    gen = os.path.join(ROOT, "tools", "gen_rust_bindings.py")
    r = subprocess.run([sys.executable, gen, "--check"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    rs = open(os.path.join(ROOT, "bindings", "raftgroups.rs")).read()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert rs in doc, "INTEGRATION.md does not carry the generated block verbatim"
    rust = {m.group(1): m.group(2) for m in re.finditer(r"pub fn (rg_\w+)\((.*?)\)(?: -> [^;]+)?;", rs)}
    declared = declared_functions()
    assert sorted(rust) == declared, (sorted(set(declared) - set(rust)), sorted(set(rust) - set(declared)))
    # arity: count the parameters of every prototype in the header and in the Rust block
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    for name in declared:
        m = re.search(r"\b" + name + r"\s*\(([^()]*)\)\s*;", src, flags=re.S)
        assert m, name
        c_args = [a for a in m.group(1).split(",") if a.strip() and a.strip() != "void"]
        r_args = [a for a in rust[name].split(",") if a.strip()]
        assert len(c_args) == len(r_args), (name, c_args, r_args)
    lib = ctypes.CDLL(rg.LIB_PATH)
    assert all(hasattr(lib, n) for n in rust)
    # ... and the library exports nothing rg_* beyond them
    out = subprocess.run(["nm", "-D", "--defined-only", rg.LIB_PATH], stdout=subprocess.PIPE, text=True).stdout
    exported = sorted({l.split()[-1] for l in out.splitlines() if l.split() and l.split()[-1].startswith("rg_") and " T " in l})
    assert exported == declared, (sorted(set(exported) - set(declared)), sorted(set(declared) - set(exported)))


def test_no_cpp_exception_can_leave_the_c_abi(tmp_path):
    """A Rust / C / ctypes caller cannot unwind through `extern "C"`: every int-returning entry point of csrc/abi_*.hip is a
    function-try-block that ends in RG_ABI_GUARD (csrc/rg_abi_guard.h), which turns std::bad_alloc into RG_ERR_OUT_OF_MEMORY and
    anything else into RG_ERR_STATE. Two halves: (1) the sources -- no entry point without the guard; (2) the guard itself,
    compiled with g++ around functions that throw."""
    import glob
    import subprocess
    root = ROOT
    n = 0
    for path in sorted(glob.glob(os.path.join(root, "raft_rs_amd", "csrc", "abi_*.hip"))):
        lines = open(path).read().split("\n")
        i = 0
        while i < len(lines):
            if lines[i].startswith('extern "C" int '):
                j = i
                while not lines[j].rstrip().endswith(("{", "}", ";", "RG_ABI_GUARD")):
                    j += 1
                if lines[j].rstrip().endswith(";") and "{" not in lines[j]:  # a forward declaration
                    i = j + 1
                    continue
                assert "try {" in lines[j], (os.path.basename(path), i + 1, lines[i][:80])
                if not lines[j].rstrip().endswith("RG_ABI_GUARD"):  # (a one-line body carries both on its line)
                    k = j + 1
                    while not lines[k].startswith("}"):
                        k += 1
                    assert lines[k] == "} RG_ABI_GUARD", (os.path.basename(path), k + 1, lines[k])
                n += 1
                i = j + 1
            else:
                i += 1
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_rust_bindings
    funcs = gen_rust_bindings.parse(open(HEADER, encoding="utf-8").read())[3]
    n_int = len({name for name, ret, _ in funcs if str(ret).strip() in ("int", "i32", "c_int")})
    assert n >= 85 and (n_int == 0 or n == n_int), (n, n_int, sorted({str(r) for _, r, _ in funcs}))  # every int entry point the header declares is defined behind the guard
    src = tmp_path / "guard_probe.cpp"
    src.write_text("""
#include <cstdarg>
#include <cstdio>
#include <stdexcept>
#include <vector>
#include "rg_abi_guard.h"
static char text[512];
int rg_fail(int code, const char *fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(text, sizeof(text), fmt, ap); va_end(ap); return code; }
extern "C" int probe(int kind) try {
    if (kind == 1) throw std::bad_alloc();
    if (kind == 2) { std::vector<char> v; v.reserve(v.max_size() + 1); }   // std::length_error
    if (kind == 3) throw 42;
    if (kind == 4) { std::vector<long> v; v.resize((size_t)1 << 50); }      // 8 PB: a real allocation failure, std::bad_alloc
    return RG_OK;
} RG_ABI_GUARD
int main() {
    if (probe(0) != RG_OK) return 1;
    if (probe(1) != RG_ERR_OUT_OF_MEMORY) return 2;
    if (probe(2) != RG_ERR_STATE || !text[0]) return 3;
    if (probe(3) != RG_ERR_STATE) return 4;
    if (probe(4) != RG_ERR_OUT_OF_MEMORY) return 5;
    return 0;
}
""")
    exe = tmp_path / "guard_probe"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(root, "raft_rs_amd", "csrc"), str(src), "-o", str(exe)])
    assert subprocess.run([str(exe)]).returncode == 0
