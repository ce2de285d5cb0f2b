cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_pseudo_transient.py -x -q 2>&1 | tail -25 | cut -c1-500
