#!/usr/bin/env python3
"""developer helper (GPU box): per-iteration kernel durations of a bench window from a rocprofv3 --kernel-trace CSV.
usage: python tools/trace_window.py KERNEL_TRACE.csv STEPS [which]    which = cold (default) | primed | profiled
An iteration ends with adam_segments_kernel; the last 3*STEPS iterations of the trace are the cold, the primed and the profiled runs."""
import csv
import sys

path, steps = sys.argv[1], int(sys.argv[2])
which = sys.argv[3] if len(sys.argv) > 3 else "cold"
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
its, cur = [], []
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("smalfit::", "").replace("void ", "")
    if "smalfit" not in r["Kernel_Name"]:
        continue
    cur.append((name, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    if name.startswith("adam_segments"):
        its.append(cur)
        cur = []
# bench.py times three runs of STEPS iterations at the end of the process: cold, primed, profiled (section events)
sel = {"cold": its[-3 * steps:-2 * steps], "primed": its[-2 * steps:-steps], "profiled": its[-steps:]}[which]
cols = ["lbs_head", "skin_mfma", "face_bbox", "raster_sweep", "raster_resolve", "raster_band", "raster_select", "raster_bwd", "vertex_bwd", "lbs_bwd_mid", "chain_bwd", "assemble", "adam_segments"]
print("%3s %8s | " % ("it", "wall") + " ".join("%7s" % c.replace("raster_", "")[:7] for c in cols))
for i, it in enumerate(sel):
    d = {}
    for name, us, s, e in it:
        key = next((c for c in cols if name.startswith(c)), name)
        d[key] = d.get(key, 0.0) + us
    wall = (it[-1][3] - it[0][2]) / 1e3
    print("%3d %8.1f | " % (i, wall) + " ".join("%7.1f" % d.get(c, 0.0) for c in cols))
