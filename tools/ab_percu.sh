cd $GRAFT_REPO_ROOT
for i in 1 2; do
for v in 0 3 2; do
NK_SS_PER_CU=$v python bench.py --cpu-seconds 0 --no-ttt --pmc off 2>/dev/null | python -c "
import json,sys
d=json.loads([x for x in sys.stdin if x.startswith(chr(123))][-1]); print('per_cu=$v', d['value'], d['step_time_stats']['median_ms'], d['check']['fnorm_inf_after_timed_steps'], {k:v['avg_us'] for k,v in d['kernels'].items() if k in ('multiaxpy','reduce_small','multidot')})"
done; done
python -m pytest tests/test_gpu_sstep.py -q -x 2>&1 | tail -2
