cd $GRAFT_REPO_ROOT
O=gpurun_out/c11; mkdir -p $O
python -m pytest tests -x -q -m gpu -k "not full_schedule" > $O/pytest.txt 2>&1
tail -25 $O/pytest.txt
