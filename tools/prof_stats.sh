#!/bin/bash
# developer helper (GPU box): rocprofv3 kernel stats of a short bench run -> gpurun_out/prof_<tag>/ ; prints the top kernels
tag=${1:-dev}; steps=${2:-390}; scene=${3:-survey}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-crop --steps $steps --scene $scene > $out/bench.json 2> $out/bench.err
f=$(find $out -name '*kernel_stats.csv' | head -1)
cp $f $out/bench_kernel_stats.csv 2>/dev/null
python - <<PY
import csv
rows=list(csv.DictReader(open("$out/bench_kernel_stats.csv")))
for r in rows[:24]:
    print("%-40s calls %5s avg_us %8.1f  %5s%%" % (r["Name"].split("(")[0].replace("smalfit::","").replace("void ","")[:40], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"][:5]))
PY
tail -c 600 $out/bench.json
