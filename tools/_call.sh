cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_dropin.py tests/test_gpu_fit3d.py -x -q -m gpu 2>&1 | tail -8
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lbs" 2>&1 | tail -3
python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "graph" 2>&1 | tail -3
