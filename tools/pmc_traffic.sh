#!/bin/bash
# HBM traffic of the rasteriser kernels via rocprofv3 PMC counters (separate passes, no tracing domains combined):
# FETCH_SIZE / WRITE_SIZE are in KiB summed over the L2 memory-side requests; on gfx950 FETCH_SIZE reports half of a
# wide coalesced read stream (MI355X_MICROARCH.md §HBM) -- the summary keeps raw values and states the correction.
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_r1
mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 39 --warmup 4 --no-cpu-baseline > $OUT/fetch.json 2> $OUT/fetch.err || true
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 39 --warmup 4 --no-cpu-baseline > $OUT/write.json 2> $OUT/write.err || true
python - <<PY
import csv, glob, collections, json
out = {}
for name in ("fetch", "write"):
    files = glob.glob("$OUT/%s/*counter_collection.csv" % name)
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in files:
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "").split("(")[0]
            if "smalfit" not in k: continue
            agg[k][0] += float(row["Counter_Value"]); agg[k][1] += 1
    out[name] = {k: {"sum": v[0], "dispatch_rows": v[1], "avg_per_row": v[0] / max(v[1], 1)} for k, v in agg.items()}
json.dump(out, open("$OUT/pmc_summary.json", "w"), indent=1)
for name in out:
    for k, v in sorted(out[name].items(), key=lambda kv: -kv[1]["sum"])[:8]:
        print(name, k, v)
PY
