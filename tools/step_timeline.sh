#!/bin/bash
# Timeline of ONE steady-state fixed-work Newton step from a rocprofv3 kernel trace: every launch with its start offset,
# duration and the idle gap in front of it (gpurun_out/<tag>_step_timeline.md) — where the step's time goes BETWEEN kernels.
#   bash tools/step_timeline.sh r03_n [bench.py flags]
set -u
TAG=${1:-r03_x}
shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o tl -- python $REPO/bench.py --steps 12 --warmup 3 --cpu-seconds 0 --no-profile-pass --no-ttt --no-spmv-hbm --pmc off "$@" > /dev/null 2>&1
F=$(find /tmp/tl -name "tl_kernel_trace.csv" | head -1)
python $REPO/tools/step_timeline.py "$F" > $OUT/${TAG}_step_timeline.md
tail -25 $OUT/${TAG}_step_timeline.md
