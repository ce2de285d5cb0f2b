set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3a; mkdir -p $O
NK_FUSED_REDUCE=1 timeout 600 python -m pytest tests/test_gpu_solvers.py tests/test_gpu_fullsize.py -x -q > $O/pytest_fused.log 2>&1; tail -3 $O/pytest_fused.log
for f in 0 1 0 1; do
  NK_FUSED_REDUCE=$f timeout 200 python bench.py --cpu-seconds 0 --no-ttt --no-profile-pass > $O/bench_fused${f}.json 2> /dev/null
  python -c "
import json; d=json.loads([x for x in open('$O/bench_fused${f}.json') if x.startswith('{')][-1]); print('fused$f', d['value'], d['check']['fnorm_inf_after_timed_steps'])"
done
