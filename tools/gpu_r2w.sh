set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2w; mkdir -p $O
( time timeout 400 python bench.py > $O/bench_csr.json 2> $O/bench_csr.err ) 2>&1 | tail -3; python -c "
import json; d=json.loads([x for x in open('$O/bench_csr.json') if x.startswith('{')][-1]); print(d['value'], d['roofline']['frac'], d['cpu_baseline']['value'], json.dumps(d['time_to_tolerance'])[:900])"
tail -3 $O/bench_csr.err
