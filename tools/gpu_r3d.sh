set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3d; mkdir -p $O
for n in 4 8; do
  BENCH_BACKEND=gloo NK_COMM=peer timeout 400 python bench.py --gpus $n --cpu-seconds 0 --no-ttt --no-weak > $O/bench_x${n}.json 2> $O/bench_x${n}.err; python -c "
import json; d=json.loads([x for x in open('$O/bench_x${n}.json') if x.startswith('{')][-1]); print('x$n', d['value'], d['n_gpus'], d['config']['comm'], d['config']['unknowns_per_gpu'], d['check'])"
done
BENCH_BACKEND=gloo NK_COMM=peer timeout 400 python bench.py --gpus 8 --workload c4 --steps 3 --warmup 1 --cpu-seconds 0 --no-ttt --no-weak > $O/bench_c4_x8.json 2> /dev/null; python -c "
import json; d=json.loads([x for x in open('$O/bench_c4_x8.json') if x.startswith('{')][-1]); print('c4 x8', d['value'], d['config']['comm'], d['check'])"
