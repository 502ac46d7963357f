#!/bin/bash
# developer helper (GPU box): everything a round's profiles/ are made from, in one gpurun call.
#   usage: bash tools/round_profiles.sh rN      -> gpurun_out/rN_*  (copy what is wanted into profiles/)
# Every step has its own timeout; logs are written as they are produced.
tag=${1:-rX}
cd $GRAFT_REPO_ROOT
out=gpurun_out
if [ "$2" != "notests" ]; then timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tee $out/${tag}_gpu_tests.log | tail -40; fi
timeout 240 python bench.py > $out/${tag}_bench_default.json 2> $out/${tag}_bench_default.err
timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/${tag}_bench_20_driver_args.json 2>> $out/${tag}_bench_default.err
timeout 90 python bench.py --steps 1950 --no-cpu-baseline --no-crop > $out/${tag}_bench_1950_survey.json 2>> $out/${tag}_bench_default.err
timeout 120 python bench.py --steps 1950 --no-cpu-baseline --no-crop --scene crop > $out/${tag}_bench_1950_crop.json 2>> $out/${tag}_bench_default.err
SMALFIT_BENCH_FORCE_DIST=1 timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 390 --no-cpu-baseline --no-crop > $out/${tag}_bench_rccl_world1.json 2>> $out/${tag}_bench_default.err
for n in 1 2 4 8; do timeout 100 python tools/rank_sim.py $n 390 2>&1 | tail -1; done > $out/${tag}_rank_sim.txt
bash tools/prof_stats.sh ${tag}_390 390 survey > $out/${tag}_prof390.log 2>&1
timeout 250 python tools/pmc_sq.py ${tag}_pmc > $out/${tag}_pmc.log 2>&1
PMC_SCRIPT="tools/crop_fit.py crop 0.3" timeout 250 python tools/pmc_sq.py ${tag}_pmc_crop > $out/${tag}_pmc_crop.log 2>&1
bash tools/prof_cmd.sh ${tag}_crop python tools/crop_fit.py crop 1.0 > $out/${tag}_prof_crop.log 2>&1
bash tools/prof_cmd.sh ${tag}_8frames python tools/rank_sim.py 8 390 > $out/${tag}_prof_8frames.log 2>&1
( cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/trace20; timeout 100 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace20 -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-crop > /dev/null 2>&1; f=$(find /tmp/trace20 -name "*kernel_trace.csv" | head -1); python $GRAFT_REPO_ROOT/tools/trace_window.py $f 20 cold > $GRAFT_REPO_ROOT/$out/${tag}_trace20_cold.txt; python $GRAFT_REPO_ROOT/tools/trace_gaps.py $f 20 > $GRAFT_REPO_ROOT/$out/${tag}_trace20_gaps.txt )
timeout 300 python -m smalify_amd.tools.work_stats --synthetic --frames 4 --out $out/${tag}_work_stats_standin.txt > /dev/null 2>&1
if [ -f smalify_amd/_variants/phases.so ]; then SMALFIT_LIB=$PWD/smalify_amd/_variants/phases.so timeout 300 python tools/lbs_phases.py 195 > $out/${tag}_lbs_phase_breakdown.txt 2>&1; fi   # tools/build_variant.sh phases -DSMALFIT_DEV_PROBES -DSMALFIT_PHASES
# gpurun copies back at most 64 MiB: keep the summaries, drop the raw per-dispatch tables
rm -rf $out/${tag}_pmc/g[0-9] $out/${tag}_pmc_crop/g[0-9] $out/${tag}_pmc/counters_list.txt $out/${tag}_pmc_crop/counters_list.txt
for d in $out/prof_${tag}_390 $out/prof_${tag}_crop $out/prof_${tag}_8frames; do
  f=$(find $d -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $d.kernel_stats.csv; rm -rf $d
done
du -sh $out | tail -1
for f in bench_default bench_20_driver_args bench_1950_survey bench_1950_crop bench_rccl_world1; do python - <<PY
import json
try:
    d = json.load(open("$out/${tag}_$f.json"))
    print("$f", round(d["value"], 1), round(d.get("value_primed", 0), 1), "crop", d.get("value_crop"), d["per_stage_iterations_per_s"], d.get("status_bits"))
except Exception as e:
    print("$f FAILED", e)
PY
done
cat $out/${tag}_rank_sim.txt
