#!/usr/bin/env python3
"""Print VGPR/SGPR/scratch/LDS/occupancy per kernel of the HIP engine (hipcc -Rpass-analysis)."""
import re
import subprocess
import sys

# usage: tools/kernel_resources.py [-DRG_P=5] [-DRG_OPT=6] ...   (default: the P=5 tick kernels)
SRC = "raft_rs_amd/csrc/tick_inst.hip"
defs = [a for a in sys.argv[1:] if a.startswith("-")]
if not any(d.startswith("-DRG_P=") for d in defs):
    defs.append("-DRG_P=5")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-c", SRC, "-o", "/dev/null",
       "-Wno-pass-failed", "-Rpass-analysis=kernel-resource-usage"] + defs
out = subprocess.run(cmd, stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    txt = m.group(1).strip()
    if txt.startswith("Function Name:"):
        cur = {"name": txt.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in txt:
        k, v = txt.split(":", 1)
        cur[k.strip()] = v.strip()
flt = [a for a in sys.argv[1:] if not a.startswith("-")]
print(f"{'kernel':58s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'LDS':>7s} {'occ':>4s}")
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], stdout=subprocess.PIPE, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name).replace("void ", "")
    print(f"{name:58s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} {r.get('TotalSGPRs', r.get('SGPRs','?')):>5s} "
          f"{r.get('ScratchSize [bytes/lane]','?'):>8s} {r.get('LDS Size [bytes/block]','?'):>7s} {r.get('Occupancy [waves/SIMD]','?'):>4s}")
