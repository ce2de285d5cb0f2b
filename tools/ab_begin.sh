cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_sstep.py tests/test_gpu_solvers.py tests/test_gpu_precond.py tests/test_gpu_round2.py -q -x 2>&1 | tail -3
for i in 1 2; do
for v in "" "NK_SS_FUSED_BEGIN=0 NK_NORMS_SEPARATE_PUBLISH=1"; do
env $v python bench.py --cpu-seconds 0 --no-ttt --pmc off 2>/dev/null | python -c "
import json,sys
d=json.loads([x for x in sys.stdin if x.startswith(chr(123))][-1]); print('[$v]', d['value'], d['step_time_stats']['median_ms'], d['check']['fnorm_inf_after_timed_steps'], {k:v['avg_us'] for k,v in d['kernels'].items() if k in ('multiaxpy','reduce_small','multidot')})"
done; done
bash tools/step_timeline.sh r03_o > /dev/null; head -12 gpurun_out/r03_o_step_timeline.md; tail -32 gpurun_out/r03_o_step_timeline.md | head -14
