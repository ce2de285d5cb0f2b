set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_multirank.py -x -q -k "four" > $O/pytest.log 2>&1; tail -30 $O/pytest.log | cut -c1-2500
