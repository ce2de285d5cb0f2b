set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2h; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
for d in 0 1 0 1; do
  NK_AXPY_DESC=$d timeout 200 python bench.py --cpu-seconds 0 --no-ttt > $O/bench_desc${d}.json 2> /dev/null
  python -c "
import json; d=json.loads([x for x in open('$O/bench_desc${d}.json') if x.startswith('{')][-1]); k=d['kernels']; print('desc$d', d['value'], {n:k[n]['avg_us'] for n in ('spmv','multidot','multiaxpy','reduce_small')})"
done
for d in 0 1; do
  NK_AXPY_DESC=$d timeout 200 python bench.py --cpu-seconds 0 --no-ttt --workload c4 --steps 4 --warmup 1 > $O/bench_c4_desc${d}.json 2> /dev/null
  python -c "
import json; d=json.loads([x for x in open('$O/bench_c4_desc${d}.json') if x.startswith('{')][-1]); k=d['kernels']; print('c4 desc$d', d['value'], {n:k[n]['avg_us'] for n in ('spmv','multidot','multiaxpy','reduce_small')})"
done
