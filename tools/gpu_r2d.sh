set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2d; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_round2.py -m gpu -q 2>&1 | tail -40 > $O/pytest.log
tail -5 $O/pytest.log
timeout 400 python tools/comm_bench.py 2 > $O/comm_bench.jsonl 2> $O/comm_bench.err; cat $O/comm_bench.jsonl
NK_PEER_UNFUSED=1 timeout 400 python tools/comm_bench.py 2 > $O/comm_bench_unfused.jsonl 2> $O/comm_bench_unfused.err; cat $O/comm_bench_unfused.jsonl
timeout 300 python bench.py > $O/bench_csr.json 2> $O/bench_csr.err; python -c "
import json; d=json.loads([x for x in open('$O/bench_csr.json') if x.startswith('{')][-1]); print(d['value'], d['cpu_baseline'])"
