cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "one_gpu or exchange" 2>&1 | tail -8
