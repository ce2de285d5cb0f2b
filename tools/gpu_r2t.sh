set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2t; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 400 python bench.py > $O/bench_csr.json 2> $O/bench_csr.err; python -c "
import json; d=json.loads([x for x in open('$O/bench_csr.json') if x.startswith('{')][-1]); print(d['value'], d['roofline']['frac'], d['cpu_baseline']['value'], d['time_to_tolerance'])"
