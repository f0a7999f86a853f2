cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 600 python tests/run_config_human.py 20 3000 4 -p0 -a1 2>/dev/null | tee gpurun_out/human20.json
timeout 900 python tests/run_config_human.py 47 20000 1 2>/dev/null | tee gpurun_out/human47.json
