set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2p; mkdir -p $O
BENCH_BACKEND=gloo NK_COMM=peer timeout 300 python bench.py --gpus 2 --cpu-seconds 0 --no-ttt > $O/bench_x2_peer.json 2> $O/bench_x2_peer.err; python -c "
import json; d=json.loads([x for x in open('$O/bench_x2_peer.json') if x.startswith('{')][-1]); print('x2 peer', d['value'], d['config']['comm'], d['check'], d['weak_scaling'])"
tail -3 $O/bench_x2_peer.err
BENCH_BACKEND=gloo NK_COMM=peer timeout 300 python bench.py --gpus 2 --workload c5 --cpu-seconds 0 --no-ttt > $O/bench_c5_x2_peer.json 2> /dev/null; python -c "
import json; d=json.loads([x for x in open('$O/bench_c5_x2_peer.json') if x.startswith('{')][-1]); print('c5 x2 peer', d['value'], d['config']['comm'], d['check'])"
python tools/comm_bench.py > $O/comm_bench.jsonl 2>/dev/null; tail -4 $O/comm_bench.jsonl | cut -c1-300
