#!/bin/bash
# developer helper (GPU box): rocprofv3 kernel stats of an arbitrary command -> prints the top kernels.  usage: tools/prof_cmd.sh TAG cmd...
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
( cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats --output-format csv -d $out -o run -- "$@" > $out/stdout.txt 2> $out/stderr.txt )
f=$(find $out -name '*kernel_stats.csv' | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows[:22]:
    print("%-40s calls %5s avg_us %8.1f  %5s%%" % (r["Name"].split("(")[0].replace("smalfit::","").replace("void ","")[:40], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"][:5]))
PY
tail -2 $out/stdout.txt
