cd $GRAFT_REPO_ROOT
O=gpurun_out/c15; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1
tail -12 $O/pytest.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c1-900
