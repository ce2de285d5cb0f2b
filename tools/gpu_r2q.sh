set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2q; mkdir -p $O
for nt in 0 1 0 1; do
  NK_SPMV_NT=$nt timeout 200 python bench.py --cpu-seconds 0 --no-ttt > $O/bench_nt${nt}.json 2> /dev/null
  python -c "
import json; d=json.loads([x for x in open('$O/bench_nt${nt}.json') if x.startswith('{')][-1]); k=d['kernels']; print('nt$nt', d['value'], {n:k[n]['avg_us'] for n in ('spmv','multidot','multiaxpy','reduce_small')})"
done
