set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2y; mkdir -p $O
timeout 600 bash tools/profile_round.sh r02_d_c4size_1gpu --workload c4 --steps 4 --warmup 1
timeout 300 python tools/spmv_bench.py > $O/spmv_bench.jsonl 2>/dev/null; tail -6 $O/spmv_bench.jsonl | cut -c1-400
