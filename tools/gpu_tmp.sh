cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_polyalg.py tests/test_gpu_direct.py tests/test_gpu_pseudo_transient.py tests/test_gpu_lm.py tests/test_gpu_solvers.py tests/test_gpu_round2.py -x -q 2>&1 | tail -30 | cut -c1-500
