set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2b; mkdir -p $O
timeout 1000 python -m pytest tests -m gpu -q 2>&1 | tail -60 > $O/pytest.log
tail -8 $O/pytest.log
timeout 400 python tools/comm_bench.py 2 > $O/comm_bench.jsonl 2> $O/comm_bench.err; cat $O/comm_bench.jsonl; tail -5 $O/comm_bench.err
BENCH_BACKEND=gloo timeout 400 python bench.py --gpus 2 --steps 8 --warmup 2 > $O/bench_x2_gloo.json 2> $O/bench_x2_gloo.err; tail -c 900 $O/bench_x2_gloo.json; tail -5 $O/bench_x2_gloo.err
BENCH_BACKEND=gloo NK_COMM=peer timeout 400 python bench.py --gpus 2 --steps 8 --warmup 2 --no-weak > $O/bench_x2_peer.json 2> $O/bench_x2_peer.err; tail -c 900 $O/bench_x2_peer.json; tail -5 $O/bench_x2_peer.err
timeout 300 python bench.py > $O/bench_csr.json 2> $O/bench_csr.err; tail -c 1200 $O/bench_csr.json
timeout 400 bash tools/profile_round.sh r02_a
timeout 500 bash tools/profile_round.sh r02_a_c4size_1gpu --workload c4 --steps 4 --warmup 1
timeout 300 python bench.py --workload c4 --steps 6 --warmup 2 --no-ttt > $O/bench_c4size_1gpu.json 2> $O/bench_c4.err; tail -c 600 $O/bench_c4size_1gpu.json
