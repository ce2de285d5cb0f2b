set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2u; mkdir -p $O
for i in 1 2 3 4 5 6; do NK_BCR_PIVOT=always timeout 300 python -m pytest tests/test_gpu_direct.py -x -q 2>&1 | tail -1; done
for i in 1 2 3; do timeout 300 python -m pytest tests/test_gpu_direct.py tests/test_gpu_lm.py tests/test_gpu_solvers.py -x -q 2>&1 | tail -1; done
