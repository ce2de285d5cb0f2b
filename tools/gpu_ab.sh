#!/bin/bash
# A/B of bench.py's headline line on ONE box, alternating: tools/gpu_ab.sh TAG ROUNDS "ENV_A" "ENV_B" ["ENV_C" …] [-- bench flags]
#   e.g. tools/gpu_ab.sh r06_c_head 3 "NK_SOLVE_HEAD=0" ""      (an empty string = the defaults)
# One line per run into gpurun_out/TAG_ab.txt: the variant, steps/s, ms per step, ‖F‖∞ after the timed steps.
set -u
cd "$(dirname "$0")/.."
TAG=$1; ROUNDS=$2; shift 2
VARS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do VARS+=("$1"); shift; done
[ $# -gt 0 ] && shift
OUT=gpurun_out/${TAG}_ab.txt
mkdir -p gpurun_out
: > $OUT
for r in $(seq 1 $ROUNDS); do
  for v in "${VARS[@]}"; do
    line=$(env $v timeout 300 python bench.py --steps 300 --warmup 20 --cpu-seconds 0 --no-profile-pass --no-ttt --no-spmv-hbm --pmc off "$@" 2>/dev/null | tail -1)
    python - "$v" "$line" >> $OUT <<'PY'
import json, sys
v, line = sys.argv[1], sys.argv[2]
try:
    d = json.loads(line)
    print(f"{v or 'default':40s} {d['value']:9.1f} steps/s  {d['ms_per_step']:.4f} ms  fnorm {d['check']['fnorm_inf_after_timed_steps']:.12e}  stats {d.get('step_time_stats')}")
except Exception as ex:
    print(f"{v or 'default':40s} FAILED {ex}: {line[:200]}")
PY
  done
done
cat $OUT
