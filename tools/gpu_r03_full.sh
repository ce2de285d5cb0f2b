set -x
TAG=${1:-r03_g}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
timeout 2000 python -m pytest tests -m gpu -q -x --timeout 600 > $O/pytest_gpu.log 2>&1; tail -12 $O/pytest_gpu.log
timeout 200 python bench.py --cpu-seconds 0 --no-ttt > $O/bench_default.json 2> /dev/null
python -c "
import json
d=json.loads([x for x in open('$O/bench_default.json') if x.startswith('{')][-1]); print(d['value'], d['ms_per_step'], {k:(v['avg_us'],v['launches']) for k,v in d['kernels'].items()})"
