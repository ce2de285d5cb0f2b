# round 5, pass A: the deferred second factorisation — parity suite, A/B bench lines, one step's timeline
set -x
TAG=${1:-r05_a}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sstep.py tests/test_gpu_powers.py tests/test_gpu_solvers.py tests/test_gpu_fullsize.py tests/test_gpu_multirank.py -q -x < /dev/null > $O/pytest_core.log 2>&1; tail -15 $O/pytest_core.log
B="--cpu-seconds 0 --no-ttt --no-spmv-hbm --pmc off --steps 200 --warmup 20 --no-profile-pass"
for rep in 1 2; do
timeout 200 python bench.py $B < /dev/null > $O/bench_default_$rep.json 2> $O/bench_default.err
NK_SS_DEFER=0 timeout 200 python bench.py $B < /dev/null > $O/bench_nodefer_$rep.json 2> /dev/null
NK_SS_TAIL_BACK=0 timeout 200 python bench.py $B < /dev/null > $O/bench_notailback_$rep.json 2> /dev/null
NK_SS_DEFER_HESS=0 timeout 200 python bench.py $B < /dev/null > $O/bench_hessinjob_$rep.json 2> /dev/null
done
timeout 200 python bench.py $B --matfree < /dev/null > $O/bench_matfree.json 2> /dev/null
timeout 200 python bench.py $B --workload c5 < /dev/null > $O/bench_c5.json 2> /dev/null
timeout 300 bash tools/step_timeline.sh ${TAG} < /dev/null
NK_SS_DEFER_HESS=0 timeout 300 bash tools/step_timeline.sh ${TAG}_hessinjob < /dev/null
for f in $O/bench_*.json; do python -c "
import json,sys
try:
    d=json.loads([x for x in open('$f') if x.startswith('{')][-1]); print('$f', d['value'], d['n_gpus'], d['ms_per_step'])
except Exception as e: print('$f FAILED', e)"; done
