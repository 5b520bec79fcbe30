#!/usr/bin/env python3
"""How long does a process WITHOUT torch take to get RCCL on a fresh box? (`rg_comm_init` dlopens librccl.so.1 -- 573 MB under
/opt/rocm/lib -- the first time a communicator is asked for; a python process that imported torch has torch's own copy mapped
already.) Prints the dlopen time of a cold and of a warm load, then the time of the C driver's whole run."""
import ctypes
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
t0 = time.time()
r = subprocess.run([sys.executable, "-c", "import ctypes,time; t=time.time(); ctypes.CDLL('librccl.so.1', mode=ctypes.RTLD_GLOBAL); print('%.1f' % (time.time()-t))"],
                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
print("cold dlopen(librccl.so.1) in a fresh process:", r.stdout.strip(), "s")
r = subprocess.run([sys.executable, "-c", "import ctypes,time; t=time.time(); ctypes.CDLL('librccl.so.1', mode=ctypes.RTLD_GLOBAL); print('%.1f' % (time.time()-t))"],
                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
print("again (page cache warm):", r.stdout.strip(), "s")
libdir = os.path.join(ROOT, "raft_rs_amd")
with tempfile.TemporaryDirectory() as d:
    exe = os.path.join(d, "c_driver")
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_driver.c"), "-o", exe,
                           "-L", libdir, "-lraftgroups", "-Wl,-rpath," + libdir])
    t = time.time()
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    print("c_driver: rc %d in %.1f s" % (r.returncode, time.time() - t))
    print(r.stdout[-600:])
