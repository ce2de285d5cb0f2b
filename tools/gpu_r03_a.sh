# round 3, first GPU pass: the Newton-basis s-step path — its tests, then A/B bench lines and one kernel trace
set -x
TAG=${1:-r03_a}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sstep.py -x -q > $O/pytest_sstep.log 2>&1; tail -15 $O/pytest_sstep.log
B="--cpu-seconds 0 --no-ttt"
timeout 200 python bench.py $B > $O/bench_default.json 2> $O/bench_default.err
timeout 200 python bench.py $B --sstep 12 > $O/bench_s12.json 2> /dev/null
timeout 200 python bench.py $B --sstep 10 > $O/bench_s10.json 2> /dev/null
timeout 200 python bench.py $B --sstep 8 > $O/bench_s8.json 2> /dev/null
timeout 200 python bench.py $B --sstep 6 --sstep-basis monomial > $O/bench_mono6.json 2> /dev/null
timeout 200 python bench.py $B --ortho dcgs2 > $O/bench_dcgs2.json 2> /dev/null
timeout 200 python bench.py $B --matfree > $O/bench_matfree.json 2> /dev/null
timeout 200 python bench.py $B --workload c5 > $O/bench_c5.json 2> /dev/null
timeout 300 python bench.py $B --workload c4 --steps 4 --warmup 1 > $O/bench_c4.json 2> /dev/null
timeout 500 bash tools/profile_round.sh ${TAG}
for f in $O/bench_*.json; do python -c "
import json,sys
try:
    d=json.loads([x for x in open('$f') if x.startswith('{')][-1]); print('$f', d['value'], d['ms_per_step'], d['check']['fnorm_inf_after_timed_steps'], {k:(v['avg_us'],v['launches']) for k,v in d['kernels'].items()})
except Exception as e: print('$f', 'FAILED', e)"; done
tail -5 $O/bench_default.err
