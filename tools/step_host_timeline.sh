#!/bin/bash
# One steady-state Newton step: the host's launch calls against the kernels' execution (rocprofv3 kernel + HIP runtime trace).
#   bash tools/step_host_timeline.sh r06_a [bench.py flags]
set -u
TAG=${1:-r06_x}
shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tlh
rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d /tmp/tlh -o tl -- python $REPO/bench.py --steps 12 --warmup 3 --cpu-seconds 0 --no-profile-pass --no-ttt --no-spmv-hbm --pmc off "$@" > /dev/null 2>&1
K=$(find /tmp/tlh -name "tl_kernel_trace.csv" | head -1)
A=$(find /tmp/tlh -name "tl_hip_api_trace.csv" | head -1)
python $REPO/tools/step_host_timeline.py "$K" "$A" > $OUT/${TAG}_step_host_timeline.md
cat $OUT/${TAG}_step_host_timeline.md
