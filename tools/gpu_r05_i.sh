# round 5, pass I: the Newton update inside the last multiaxpy, the norms' stage 1 inside the residual kernel
set -x
TAG=${1:-r05_i}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_solvers.py tests/test_gpu_sstep.py tests/test_gpu_fullsize.py tests/test_gpu_linesearch.py tests/test_gpu_multirank.py -q -x < /dev/null > $O/pytest_core.log 2>&1; tail -8 $O/pytest_core.log
B="--cpu-seconds 0 --no-ttt --no-spmv-hbm --pmc off --steps 200 --warmup 20 --no-profile-pass"
for rep in 1 2 3; do
NK_PRELOADED_RHS=0 timeout 200 python bench.py $B < /dev/null > $O/bench_nopre_$rep.json 2> /dev/null
timeout 200 python bench.py $B < /dev/null > $O/bench_default_$rep.json 2> $O/bench_default.err


NK_FUSED_UPDATE=0 NK_FUSED_RESIDUAL_NORMS=0 timeout 200 python bench.py $B < /dev/null > $O/bench_neither_$rep.json 2> /dev/null
done
timeout 200 python bench.py $B --matfree < /dev/null > $O/bench_matfree.json 2> /dev/null
timeout 300 bash tools/step_timeline.sh ${TAG} < /dev/null > /dev/null 2>&1
head -22 gpurun_out/${TAG}_step_timeline.md
for f in $O/bench_*.json; do python -c "
import json,sys
try:
    d=json.loads([x for x in open('$f') if x.startswith('{')][-1]); print('$f', d['value'], d['n_gpus'], d['ms_per_step'])
except Exception as e: print('$f FAILED', e)"; done
