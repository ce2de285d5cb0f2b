mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 10 --warmup 2 --cpu-steps 0 --no-profile-pass"
rm -rf $R/gpurun_out/prof3
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof3 -o kt -- $B > $R/gpurun_out/prof3_kt.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof3 -o fetch -- $B > $R/gpurun_out/prof3_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof3 -o write -- $B > $R/gpurun_out/prof3_write.log 2>&1
cd $R
timeout 250 python bench.py > gpurun_out/bench_full2.log 2>&1; tail -1 gpurun_out/bench_full2.log | cut -c1-200
timeout 250 python bench.py --matfree > gpurun_out/bench_full2_matfree.log 2>&1; tail -1 gpurun_out/bench_full2_matfree.log | cut -c1-200
