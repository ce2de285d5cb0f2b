set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2x; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_direct.py -x -q -k random > $O/pytest.log 2>&1; tail -12 $O/pytest.log | cut -c1-600
