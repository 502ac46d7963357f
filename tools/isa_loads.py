#!/usr/bin/env python3
"""developer helper (no GPU needed): the order of global loads, returning atomics, vector-memory waits and branches in each kernel's
gfx950 assembly, as one compact string per kernel -- shows at a glance whether independent loads are issued back to back
('LLLLLLLL w7 w6 ...') or each one is waited for before the next is issued ('L w0 L w0 L w0': one dependent trip to memory per
load; this is how the round-3 packed-box experiment lost 13 %: sixteen conditional loads compiled into sixteen waits).

usage: python tools/isa_loads.py [kernel-name-substring ...] [-- -DX=1 ...]
  L  global / buffer load          A  returning global atomic         wN  s_waitcnt vmcnt(N)
  |  branch (runs of branches collapsed)                              S  store / non-returning atomic (vmcnt on gfx9 counts them too)
"""
import os
import re
import subprocess
import sys

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
args = sys.argv[1:]
defs = []
if "--" in args:
    i = args.index("--")
    args, defs = args[:i], args[i + 1:]
src = os.path.join(root, "smalify_amd", "csrc", "smalfit_kernels.hip")
asm = "/tmp/_isa_loads.s"
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-S", "--cuda-device-only", "-o", asm, src] + defs,
                   capture_output=True, text=True)
if r.returncode != 0:
    raise SystemExit(r.stderr[-2000:])
text = open(asm).read()
for m in re.finditer(r"^(_Z\w*kernel\w*):[^\n]*\n(.*?)^\.Lfunc_end", text, re.S | re.M):      # (a kernel may hold several s_endpgm)
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.split("(")[0].replace("smalfit::", "").replace("void ", "").strip()
    if args and not any(a in name for a in args):
        continue
    ev = []
    n_inst = 0
    for ln in m.group(2).splitlines():
        ln = ln.strip()
        if re.match(r"(v_|s_|ds_|global_|buffer_|flat_)", ln):
            n_inst += 1
        if ln.startswith(("global_load", "buffer_load", "flat_load")):
            ev.append("L")
        elif ln.startswith(("global_atomic", "buffer_atomic", "flat_atomic")):
            ev.append("A" if " sc0" in ln or "glc" in ln else "S")
        elif ln.startswith(("global_store", "buffer_store", "flat_store")):
            ev.append("S")
        elif ln.startswith("s_waitcnt") and "vmcnt" in ln:
            ev.append(" w" + re.search(r"vmcnt\((\d+)\)", ln).group(1) + " ")
        elif ln.startswith("s_cbranch") or ln.startswith("s_branch"):
            ev.append("|")
    s = re.sub(r"\|+", "|", "".join(ev))
    s = re.sub(r"\s+", " ", s)
    print("%s  (%d instructions)" % (name, n_inst))
    print("    " + s)
