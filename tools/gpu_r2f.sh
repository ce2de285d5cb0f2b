set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2f; mkdir -p $O
OMP_PLACES=cores OMP_PROC_BIND=spread python tools/cpu_leg_diag.py 1024 > $O/cpu_diag.jsonl 2>&1; cat $O/cpu_diag.jsonl
timeout 600 python -m pytest tests/test_gpu_multirank.py -m gpu -q 2>&1 | tail -30 > $O/pytest.log; tail -4 $O/pytest.log
timeout 300 python bench.py > $O/bench_csr.json 2> $O/bench_csr.err; python -c "
import json; d=json.loads([x for x in open('$O/bench_csr.json') if x.startswith('{')][-1]); print(d['value'], d['cpu_baseline'])"
