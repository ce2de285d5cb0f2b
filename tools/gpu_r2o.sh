set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2o; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_linesearch.py -x -q > $O/pytest.log 2>&1; tail -25 $O/pytest.log
