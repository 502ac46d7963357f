cd $GRAFT_REPO_ROOT
O=gpurun_out/c14; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err
python bench.py --no-cpu-baseline > $O/bench390.json 2> $O/bench390.err
python bench.py --no-cpu-baseline --steps 1950 > $O/bench1950.json 2> $O/bench1950.err
python bench.py --no-cpu-baseline --steps 1950 --scene crop > $O/bench1950c.json 2> $O/bench1950c.err
bash tools/prof_stats.sh c14prof 390 > $O/prof.txt 2>&1
cp gpurun_out/prof_c14prof/bench_kernel_stats.csv $O/kernel_stats.csv
env -u PMC_GROUPS timeout 400 python tools/pmc_sq.py c14/pmc > $O/pmc_stdout.txt 2>&1
python -m pytest tests -x -q -m gpu -k "not full_schedule" > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt; cat $O/bench20.json | cut -c1-600; for f in bench390 bench1950 bench1950c; do python -c "
import sys, json
d = json.load(open('$O/$f.json')); print(d['steps'], round(d['value'],1), 'host', round(d['host_issue_ms_per_step'],3), d['per_stage_iterations_per_s'], {k: round(v,4) for k,v in d['section_ms'].items() if v})
"; done; head -20 $O/prof.txt
