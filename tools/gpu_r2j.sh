set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2j; mkdir -p $O
python tools/c5_mg_time.py 512 > $O/c5_mg_time.txt 2>&1
python tools/c5_mg_time.py 512 matfree >> $O/c5_mg_time.txt 2>&1
python tools/c5_mg_time.py 2048 matfree >> $O/c5_mg_time.txt 2>&1
cat $O/c5_mg_time.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/tools/c5_mg_time.py 512 matfree > /dev/null 2>&1
find /tmp/kt -name "kt_kernel_stats.csv" -exec head -25 {} \; > $GRAFT_REPO_ROOT/$O/c5_mg_kernel_stats.csv
cat $GRAFT_REPO_ROOT/$O/c5_mg_kernel_stats.csv | cut -c1-160
