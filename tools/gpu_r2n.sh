set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2n; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_lm.py tests/test_gpu_round2.py tests/test_gpu_direct.py -x -q > $O/pytest.log 2>&1; tail -25 $O/pytest.log
