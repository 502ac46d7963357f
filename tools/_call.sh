cd $GRAFT_REPO_ROOT
SMALFIT_LIB=$PWD/smalify_amd/_variants/probes.so PROBE_FLAGS=0 python tools/raster_probe.py 2>&1 | grep flags | cut -c1-250
python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['steps'], round(d['value'],1), d['per_stage_iterations_per_s'], {k: round(v,4) for k,v in d['section_ms'].items() if v})
"
python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['steps'], round(d['value'],1), d['per_stage_iterations_per_s'], {k: round(v,4) for k,v in d['section_ms'].items() if v})
"
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_anchors.py -x -q -m gpu -k "not full_schedule" 2>&1 | tail -3
