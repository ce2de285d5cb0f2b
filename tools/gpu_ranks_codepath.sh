#!/bin/bash
# The one-process-per-GPU code path through bench.py with 1 / 2 / 4 / 8 processes on ONE GPU (peer arenas, NK_DEVICE_SHARED=0 so
# that the ranks take what ranks on GPUs of their own take): counts per step and residual agreement. gpurun_out/<tag>_ranks_codepath.txt
set -u
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=${1:-r06_z}
OUT=gpurun_out/${TAG}_ranks_codepath.txt
: > $OUT
for N in 1 2 4 8; do
  line=$(BENCH_BACKEND=gloo NK_COMM=peer NK_DEVICE_SHARED=0 timeout 300 python bench.py --gpus $N --grid 256 --steps 20 --warmup 2 --cpu-seconds 0 --no-ttt --no-spmv-hbm --pmc off --no-profile-pass --no-weak 2>/dev/null | tail -1)
  python - "$N" "$line" >> $OUT <<'PY'
import json, sys
n, line = sys.argv[1], sys.argv[2]
try:
    d = json.loads(line)
    c = d["check"]
    print(f"{n:>3} ranks  {d['value']:8.1f} steps/s  {d['config']['comm']:12s} fnorm {c['fnorm_inf_after_timed_steps']:.15e}  all-reduces {c['allreduces']:4d}  launches with halo exchanges {c['halo_exchanges']:4d}  selfcheck {json.dumps(d['config'].get('comm_selfcheck'))[:80]}")
except Exception as ex:
    print(f"{n:>3} ranks FAILED {ex}: {line[:300]}")
PY
done
cat $OUT
