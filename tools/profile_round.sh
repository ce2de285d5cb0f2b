#!/bin/bash
# Evidence refresh on a GPU box: rocprofv3 kernel trace + separate FETCH_SIZE / WRITE_SIZE PMC passes of the SAME
# bench.py command, summarised into gpurun_out/<tag>_kernel_stats_pmc.{md,json}; then the full bench lines.
#   bash tools/profile_round.sh r01_f
set -u
TAG=${1:-r01_x}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 12 --warmup 3 --cpu-steps 0 --no-profile-pass"
rm -rf /tmp/kt /tmp/pf /tmp/pw
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o fetch -- $CMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pw -o write -- $CMD > /dev/null 2>&1
mkdir -p /tmp/merge
find /tmp/kt -name "kt_kernel_stats.csv" -exec cp {} /tmp/merge/ \;
find /tmp/pf -name "fetch_counter_collection.csv" -exec cp {} /tmp/merge/ \;
find /tmp/pw -name "write_counter_collection.csv" -exec cp {} /tmp/merge/ \;
ls /tmp/merge
python $REPO/tools/pmc_summary.py /tmp/merge $OUT/${TAG}_kernel_stats_pmc | head -16
cd $REPO
python bench.py > $OUT/${TAG}_bench_csr.json 2> $OUT/${TAG}_bench_csr.err; tail -c 600 $OUT/${TAG}_bench_csr.json
python bench.py --matfree > $OUT/${TAG}_bench_matfree.json 2> /dev/null; tail -c 300 $OUT/${TAG}_bench_matfree.json
