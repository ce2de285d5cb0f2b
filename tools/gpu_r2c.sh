set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2c; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -80 > $O/pytest.log
tail -6 $O/pytest.log
timeout 400 python tools/comm_bench.py 2 > $O/comm_bench.jsonl 2> $O/comm_bench.err; cat $O/comm_bench.jsonl
timeout 300 python bench.py > $O/bench_csr.json 2> $O/bench_csr.err; tail -c 1500 $O/bench_csr.json
timeout 300 python bench.py --workload c4 --steps 6 --warmup 2 --no-ttt --cpu-seconds 6 > $O/bench_c4size_1gpu.json 2> $O/bench_c4.err; tail -c 700 $O/bench_c4size_1gpu.json
timeout 300 python bench.py --workload c5 --steps 10 --warmup 2 --no-ttt > $O/bench_c5.json 2> $O/bench_c5.err; tail -c 700 $O/bench_c5.json; tail -3 $O/bench_c5.err
BENCH_BACKEND=gloo NK_COMM=peer timeout 400 python bench.py --gpus 2 --steps 8 --warmup 2 > $O/bench_x2_peer.json 2> $O/bench_x2_peer.err; tail -c 900 $O/bench_x2_peer.json
BENCH_BACKEND=gloo NK_COMM=peer timeout 400 python bench.py --gpus 2 --workload c5 --steps 6 --warmup 2 > $O/bench_c5_x2_peer.json 2> $O/bench_c5_x2_peer.err; tail -c 600 $O/bench_c5_x2_peer.json; tail -3 $O/bench_c5_x2_peer.err
