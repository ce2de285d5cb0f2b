set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2r; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_solvers.py tests/test_gpu_direct.py -x -q -k "zero_right or started_at or cyclic" > $O/pytest.log 2>&1; tail -25 $O/pytest.log
