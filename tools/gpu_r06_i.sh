#!/bin/bash
# Round 6, GPU call I: the cycle begin run ahead inside the fill kernel — tests, A/B, timeline
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_sstep.py tests/test_gpu_solvers.py tests/test_gpu_round2.py tests/test_gpu_determinism.py tests/test_gpu_kernels.py tests/test_gpu_powers.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -5 > $OUT/r06_i_pytest_subset.txt
cat $OUT/r06_i_pytest_subset.txt
tools/gpu_ab.sh r06_i_begin 3 "NK_BEGIN_AHEAD=0" ""
bash tools/step_timeline.sh r06_i > /dev/null 2>&1
head -22 $OUT/r06_i_step_timeline.md
