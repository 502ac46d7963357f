#!/usr/bin/env python3
"""developer helper: registers / occupancy / LDS of every kernel as hipcc reports them (-Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_resources.py [csrc root (default: repo)] [-DX=1 ...]   (cross-compiles, no GPU needed)"""
import os
import re
import subprocess
import sys

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
args = sys.argv[1:]
if args and not args[0].startswith("-"):
    root = args.pop(0)
src = os.path.join(root, "smalify_amd", "csrc", "smalfit_kernels.hip")
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-fPIC", "-shared",
                      "-Rpass-analysis=kernel-resource-usage", src, "-o", "/tmp/_kres.so"] + args, capture_output=True, text=True).stderr
cur = {}
rows = []
for ln in out.splitlines():
    m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", ln)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.split("(")[0].replace("smalfit::", "").strip()}
        rows.append(cur)
    else:
        cur[k.split(" ")[0]] = v
print("%-34s %5s %5s %4s %7s %7s" % ("kernel", "VGPR", "SGPR", "occ", "LDS", "scratch"))
for r in rows:
    print("%-34s %5s %5s %4s %7s %7s" % (r["name"][:34], r.get("VGPRs"), r.get("TotalSGPRs"), r.get("Occupancy"), r.get("LDS"), r.get("ScratchSize")))
