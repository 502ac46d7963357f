#!/bin/bash
# HBM traffic of the mesh-objective kernels (fitter_3d path) via rocprofv3 PMC counters, FETCH_SIZE and WRITE_SIZE in
# separate passes like tools/pmc_traffic.sh (KiB, raw values; see that script for the gfx950 caveat).
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_fit3d
mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o pmc -- python $GRAFT_REPO_ROOT/tools/fit3d_bench.py --meshes 8 --iters 20 > $OUT/$c.json 2> $OUT/$c.err || true
done
python - <<PY
import csv, glob, collections, json
out = {}
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob("$OUT/%s/*counter_collection.csv" % name):
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "").split("(")[0].replace("void ", "")
            if "smalfit" not in k: continue
            agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
    out[name] = {k: {"sum_KiB": v[0], "dispatch_rows": v[1], "avg_KiB_per_row": v[0] / max(v[1], 1)} for k, v in agg.items()}
json.dump(out, open("$OUT/pmc_fit3d_summary.json", "w"), indent=1)
for name in out:
    for k, v in sorted(out[name].items(), key=lambda kv: -kv[1]["sum_KiB"])[:6]:
        print(name, k, round(v["avg_KiB_per_row"], 1), v["dispatch_rows"])
PY
