#!/bin/bash
# Round 6, GPU call H: the norms' stage-2 reduction folded into the fill kernel (A/B), non-temporal hints on sweep B (A/B, three
# development builds), the pruned sweep widths (tests), the timeline and the scalar launches' phase stamps of the new dispatch
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_sstep.py tests/test_gpu_solvers.py tests/test_gpu_round2.py tests/test_gpu_determinism.py tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -5 > $OUT/r06_h_pytest_subset.txt
cat $OUT/r06_h_pytest_subset.txt
L=nonlinearsolve.jl_amd/lib
tools/gpu_ab.sh r06_h_fold 3 "NK_FOLD_NORMS=0" ""
tools/gpu_ab.sh r06_h_nt 2 "" "NK_LIB_PATH=$L/libmi355x_nk_nt1.so" "NK_LIB_PATH=$L/libmi355x_nk_nt2.so" "NK_LIB_PATH=$L/libmi355x_nk_nt3.so"
bash tools/step_timeline.sh r06_h > /dev/null 2>&1
cat $OUT/r06_h_step_timeline.md | tail -24
NK_LIB_PATH=$L/libmi355x_nk_stamps.so timeout 200 python tools/ss_stamps.py > $OUT/r06_h_ss_stamps.txt 2>&1
tail -9 $OUT/r06_h_ss_stamps.txt
