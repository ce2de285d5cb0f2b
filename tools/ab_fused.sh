cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
for v in 1 0; do
NK_SS_FUSED=$v python bench.py --cpu-seconds 0 --no-ttt --pmc off 2>/dev/null | python -c "
import json,sys
d=json.loads([x for x in sys.stdin if x.startswith(chr(123))][-1]); print('fused=$v', d['value'], d['step_time_stats']['median_ms'], d['check']['fnorm_inf_after_timed_steps'], {k:v['avg_us'] for k,v in d['kernels'].items() if k in ('multiaxpy','reduce_small','multidot')})"
done; done
python tools/ss_stamps.py 2>/dev/null | tail -2
