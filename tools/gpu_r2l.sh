set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2l; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_direct.py -x -q -s > $O/pytest_direct.log 2>&1; tail -40 $O/pytest_direct.log
