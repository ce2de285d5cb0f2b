set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2e; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -60 > $O/pytest.log
tail -6 $O/pytest.log
timeout 300 python bench.py > $O/bench_csr.json 2> $O/bench_csr.err; python -c "
import json; d=json.loads([x for x in open('$O/bench_csr.json') if x.startswith('{')][-1]); print(d['value'], d['cpu_baseline'])"
