cd $GRAFT_REPO_ROOT
for v in probes probes_t4; do
SMALFIT_LIB=$PWD/smalify_amd/_variants/$v.so PROBE_COLD=1 PROBE_FLAGS=0 python tools/raster_probe.py 2>&1 | grep "survey flags" | cut -c1-250
done
SMALFIT_LIB=$PWD/smalify_amd/_variants/probes_t4.so PROBE_STATS=1 python tools/band_probe.py 390 2>&1 | grep "stage [123]" | cut -c1-170
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cache or render or fit_full or trajectory" 2>&1 | tail -2
