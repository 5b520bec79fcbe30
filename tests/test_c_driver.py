"""The C ABI from plain C: include/raftgroups.h is valid C99 and examples/c_driver.c links against the
in-tree library (CPU); on a GPU the driver runs and checks its own results."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_driver(tmp_path, rg):
    exe = str(tmp_path / "c_driver")
    libdir = os.path.dirname(rg.LIB_PATH)
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "c_driver.c"), "-o", exe, "-L", libdir, "-lraftgroups",
           "-Wl,-rpath," + libdir]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    return exe


def test_header_is_c99_and_driver_links(tmp_path, rg):
    exe = build_driver(tmp_path, rg)
    if rg.load_library().rg_device_count() == 0:
        r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 2 and "no CPU fallback" in r.stdout  # fails loudly without a GPU


@pytest.mark.gpu
def test_c_driver_runs_on_gpu(tmp_path, rg):
    exe = build_driver(tmp_path, rg)
    # (line-buffered, so that a run that does not come back shows how far it got)
    try:
        r = subprocess.run(["stdbuf", "-oL", "-eL", exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    except subprocess.TimeoutExpired as e:
        out = e.output.decode(errors="replace") if isinstance(e.output, bytes) else (e.output or "")
        pytest.fail("examples/c_driver did not finish in 600 s; its output so far:\n" + out[-3000:])
    assert r.returncode == 0 and "C_DRIVER_OK" in r.stdout, r.stdout
