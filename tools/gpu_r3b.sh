set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3b; mkdir -p $O
NK_DIRECT=band timeout 600 python -m pytest tests/test_gpu_solvers.py tests/test_gpu_round2.py tests/test_gpu_lm.py -q -k "direct or c2 or band or lu or default_linsolve" 2>&1 | tail -2
NK_BCR_PIVOT=always timeout 600 python -m pytest tests/test_gpu_solvers.py tests/test_gpu_round2.py tests/test_gpu_lm.py tests/test_gpu_direct.py -q -k "direct or c2 or band or lu or default_linsolve or cyclic" 2>&1 | tail -2
NK_AXPY_DESC=0 NK_DCGS2_SPLIT_TAIL=1 timeout 600 python -m pytest tests/test_gpu_solvers.py tests/test_gpu_fullsize.py -q 2>&1 | tail -2
NK_GMRES_GRAPH=1 timeout 600 python -m pytest tests/test_gpu_solvers.py tests/test_gpu_round2.py -q 2>&1 | tail -2
