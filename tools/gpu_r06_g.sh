#!/bin/bash
# Round 6, GPU call G: the whole GPU suite on the cleaned tree, the ILU(τ) cost table, the quick bench line
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
mkdir -p $OUT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > $OUT/r06_g_pytest_gpu_tail.txt
cat $OUT/r06_g_pytest_gpu_tail.txt
timeout 600 python tools/ilut_time.py 32 128 256 512 > $OUT/r06_g_ilut_time.md 2>$OUT/r06_g_ilut_time.err
cat $OUT/r06_g_ilut_time.md
python bench.py --steps 300 --warmup 20 --cpu-seconds 0 --no-ttt --no-spmv-hbm --pmc off 2>/dev/null | tail -1 > $OUT/r06_g_bench_quick.json
cut -c1-600 $OUT/r06_g_bench_quick.json
