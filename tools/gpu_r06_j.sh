#!/bin/bash
# Round 6, GPU call J: sweep B of the cycle's last block stores nothing — tests, A/B, timeline
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_sstep.py tests/test_gpu_solvers.py tests/test_gpu_round2.py tests/test_gpu_determinism.py tests/test_gpu_fullsize.py tests/test_gpu_multirank.py -x -q -m gpu 2>&1 | tail -5 > $OUT/r06_j_pytest_subset.txt
cat $OUT/r06_j_pytest_subset.txt
tools/gpu_ab.sh r06_j_nostore 3 "NK_SS_NOSTORE=0" ""
bash tools/step_timeline.sh r06_j > /dev/null 2>&1
head -20 $OUT/r06_j_step_timeline.md
