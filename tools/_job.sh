cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_sync_free.py -m gpu -x -q 2>&1 | tail -3
