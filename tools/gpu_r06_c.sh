#!/bin/bash
# Round 6, GPU call C: (1) the tests of the touched paths, (2) the matrix-powers hand-off forms: stamps + timing + A/B,
# (3) the head of the next linear solve: A/B, (4) host/kernel timeline of a step.
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_powers.py tests/test_gpu_sstep.py tests/test_gpu_determinism.py tests/test_gpu_solvers.py tests/test_gpu_round2.py -x -q -m gpu 2>&1 | tail -8 > $OUT/r06_c_pytest_subset.txt
cat $OUT/r06_c_pytest_subset.txt
STL=nonlinearsolve.jl_amd/lib/libmi355x_nk_stamps.so
for g in 0 1; do
  NK_PW_GRAN=$g NK_LIB_PATH=$STL timeout 200 python tools/pw_stamps.py 1024 15 20 > $OUT/r06_c_pw_stamps_gran$g.txt 2>&1
  tail -20 $OUT/r06_c_pw_stamps_gran$g.txt
done
: > $OUT/r06_c_powers_bench.jsonl
for r in 1 2; do for g in 0 1; do
  NK_PW_GRAN=$g timeout 200 python tools/powers_bench.py 1024 15 60 2>/dev/null | tail -1 | sed "s/^{/{\"gran\": $g, /" >> $OUT/r06_c_powers_bench.jsonl
done; done
cat $OUT/r06_c_powers_bench.jsonl
tools/gpu_ab.sh r06_c 2 "NK_PW_GRAN=0 NK_SOLVE_HEAD=0" "NK_PW_GRAN=0" "NK_SOLVE_HEAD=0" ""
bash tools/step_host_timeline.sh r06_c
