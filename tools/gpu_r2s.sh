set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2s; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_direct.py tests/test_gpu_lm.py -x -q -s > $O/pytest.log 2>&1; tail -25 $O/pytest.log
NK_BCR_PIVOT=never timeout 300 python -m pytest tests/test_gpu_direct.py -q -k "pivots_inside or bratu256 or c2_direct" -s > $O/pytest_nopivot.log 2>&1; tail -8 $O/pytest_nopivot.log
