# round 6, final evidence pass — on the tree as committed:
#   smoke(), the default bench line (cpu_baseline, live PMC, profile pass), rocprofv3 kernel statistics + separate
#   FETCH_SIZE / WRITE_SIZE passes of the same workload (tools/profile_round.sh), the timeline of one Newton step
#   (tools/step_timeline.sh), A/B lines of the round's forms, the scalar launches' phase stamps, the two-rank code path, and
#   (argument 2 = tests) the whole GPU suite. Copy what should be judged from gpurun_out/ into profiles/.
set -x
TAG=${1:-r06_z}
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/$TAG; mkdir -p $O
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 300 python bench.py < /dev/null > $O/bench_default.json 2> $O/bench_default.err; tail -c 400 $O/bench_default.json
timeout 200 bash tools/profile_round.sh ${TAG} < /dev/null
timeout 120 bash tools/step_timeline.sh ${TAG} < /dev/null | head -24
B="--cpu-seconds 0 --no-ttt --no-spmv-hbm --pmc off --steps 300 --warmup 20 --no-profile-pass"
timeout 100 python bench.py $B --matfree < /dev/null > $O/bench_matfree.json 2> /dev/null
timeout 100 python bench.py $B --workload c5 < /dev/null > $O/bench_c5.json 2> /dev/null
NK_SS_NOSTORE=0 timeout 100 python bench.py $B < /dev/null > $O/bench_last_block_stored.json 2> /dev/null
NK_SS_RO=0 NK_SS_RO_GRID=0 timeout 100 python bench.py $B < /dev/null > $O/bench_last_block_staging_kernel.json 2> /dev/null
NK_SS_NOSTORE=0 NK_BEGIN_AHEAD=0 NK_FOLD_NORMS=0 timeout 100 python bench.py $B < /dev/null > $O/bench_round5_dispatch.json 2> /dev/null
timeout 100 python bench.py $B < /dev/null > $O/bench_quick_default.json 2> /dev/null
timeout 100 python bench.py --workload c4 --steps 10 --warmup 2 --cpu-seconds 0 --no-ttt --no-spmv-hbm --pmc off --no-profile-pass < /dev/null > $O/bench_c4size_1gpu.json 2> /dev/null
BENCH_BACKEND=gloo timeout 200 python bench.py --gpus 2 --steps 50 --warmup 5 --cpu-seconds 0 --no-ttt --no-spmv-hbm --pmc off --no-profile-pass --no-weak < /dev/null > $O/bench_x2_shared_gpu.json 2> $O/bench_x2_shared_gpu.err
NK_LIB_PATH=nonlinearsolve.jl_amd/lib/libmi355x_nk_stamps.so timeout 100 python tools/ss_stamps.py > $O/ss_stamps.txt 2>&1; tail -4 $O/ss_stamps.txt
for f in $O/bench_*.json; do python -c "
import json
try:
    d=json.loads([x for x in open('$f') if x.startswith('{')][-1]); print('$f'.split('/')[-1], d['value'], d['ms_per_step'], d['check']['fnorm_inf_after_timed_steps'])
except Exception as e: print('$f FAILED', e)"; done | tee $O/bench_lines.txt
if [ "${2:-}" = "tests" ]; then
  timeout ${3:-900} python -m pytest tests -m gpu -q -p no:cacheprovider < /dev/null > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
fi
