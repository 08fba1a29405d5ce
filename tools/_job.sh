cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -3
