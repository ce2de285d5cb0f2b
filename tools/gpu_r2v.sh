set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2v; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_brus_mg.py -x -q > $O/pytest.log 2>&1; tail -25 $O/pytest.log | cut -c1-1500
