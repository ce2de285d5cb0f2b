set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_lm.py tests/test_gpu_multirank.py tests/test_gpu_round2.py -x -q > $O/pytest_lm.log 2>&1; tail -30 $O/pytest_lm.log
