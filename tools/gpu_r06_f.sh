#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
Q="--cpu-seconds 0 --no-profile-pass --no-ttt --no-spmv-hbm --pmc off"
for g in 0 1 0 1; do
  BENCH_GC=$g python bench.py --steps 300 --warmup 20 $Q 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('gc_off_in_timed_region=$g', 'ms/step', d['ms_per_step'], d['step_time_stats'])"
done
