cd $GRAFT_REPO_ROOT
SMALFIT_LIB=$PWD/smalify_amd/_variants/probes.so PROBE_FLAGS=0 python tools/raster_probe.py 2>&1 | grep flags | cut -c1-250
