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
sel = its[-2 * steps:-steps]
t0 = sel[0][0][1]
prev_end = None
for i in (0, 1, 2, 8):
    print("--- iteration", i)
    for name, s, e in sel[i]:
        gap = (s - prev_end) / 1e3 if prev_end else 0.0
        print("  %-22s start %9.1f dur %7.1f gap_before %6.1f" % (name[:22], (s - t0) / 1e3, (e - s) / 1e3, gap))
        prev_end = e
