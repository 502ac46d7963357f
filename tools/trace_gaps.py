#!/usr/bin/env python3
"""developer helper (GPU box): kernel-by-kernel timeline (start, duration, gap before) of iterations 0, 1, 2 and 8 of the COLD timed
run of `bench.py --steps STEPS` from a rocprofv3 --kernel-trace CSV -- shows where a window loses time BETWEEN kernels (host-bound
stage 0, event records).  usage: python tools/trace_gaps.py KERNEL_TRACE.csv STEPS"""
import csv, sys
path, steps = sys.argv[1], int(sys.argv[2])
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows=[r for r in rows if "smalfit" in r["Kernel_Name"]]
its, cur = [], []
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("smalfit::", "").replace("void ", "")
    cur.append((name, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    if name.startswith("adam_segments"):
        its.append(cur); cur = []
sel = its[-3 * steps:-2 * steps]          # the cold run (bench.py ends with cold, primed, profiled)
t0 = sel[0][0][1]
prev_end = None
for i in (0, 1, 2, 8):
    print("--- iteration", i)
    for name, s, e in sel[i]:
        gap = (s - prev_end) / 1e3 if prev_end else 0.0
        print("  %-22s start %9.1f dur %7.1f gap_before %6.1f" % (name[:22], (s - t0) / 1e3, (e - s) / 1e3, gap))
        prev_end = e
