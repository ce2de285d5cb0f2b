cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_g; mkdir -p $O
timeout 60 python bench.py --workload c4 --steps 4 --warmup 1 --cpu-seconds 0 --no-ttt < /dev/null > $O/bench_c4size_1gpu.json 2> /dev/null
BENCH_BACKEND=gloo NK_COMM=peer timeout 60 python bench.py --gpus 2 --cpu-seconds 0 --no-ttt --no-weak < /dev/null > $O/bench_x2_peer_shared_gpu.json 2> /dev/null
for f in $O/bench_c4size_1gpu.json $O/bench_x2_peer_shared_gpu.json; do python -c "
import json,sys; d=json.loads([x for x in open('$f') if x.startswith('{')][-1]); print('$f', d['value'], d['n_gpus'])" < /dev/null; done
