set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/c2; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/pytest.txt 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench20.json 2> $O/bench20.err
python bench.py --no-cpu-baseline > $O/bench390.json 2> $O/bench390.err
SMALFIT_LIB=$PWD/smalify_amd/_variants/work.so python tools/work_stats.py 195 > $O/work_stats.txt 2>&1
SMALFIT_LIB=$PWD/smalify_amd/_variants/probes.so python tools/band_probe.py 390 > $O/band_probe.txt 2>&1
tail -3 $O/pytest.txt; cat $O/bench20.json $O/bench390.json | cut -c1-1500; cat $O/work_stats.txt $O/band_probe.txt
