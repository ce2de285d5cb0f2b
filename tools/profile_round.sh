#!/bin/bash
# Evidence refresh on a GPU box: rocprofv3 kernel trace + separate FETCH_SIZE / WRITE_SIZE PMC passes of the SAME
# bench.py command (counters in their own passes, kernel trace only), summarised into
# gpurun_out/<tag>_kernel_stats_pmc.{md,json}.
#   bash tools/profile_round.sh r02_a            # config C3 (Bratu 1024²)
#   bash tools/profile_round.sh r02_a_c4size_1gpu --workload c4 --steps 4 --warmup 1
set -u
TAG=${1:-r02_x}
shift || true
EXTRA="$*"
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 12 --warmup 3 --cpu-seconds 0 --no-profile-pass --no-ttt --no-spmv-hbm --pmc off $EXTRA"
rm -rf /tmp/kt /tmp/pf /tmp/pw /tmp/merge
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o fetch -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw -o write -- $CMD > /dev/null 2>&1
mkdir -p /tmp/merge
find /tmp/kt -name "kt_kernel_stats.csv" -exec cp {} /tmp/merge/ \;
find /tmp/pf -name "fetch_counter_collection.csv" -exec cp {} /tmp/merge/ \;
find /tmp/pw -name "write_counter_collection.csv" -exec cp {} /tmp/merge/ \;
ls /tmp/merge
python $REPO/tools/pmc_summary.py /tmp/merge $OUT/${TAG}_kernel_stats_pmc | head -14
