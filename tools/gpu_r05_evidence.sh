# round 5, final evidence pass — the LAST GPU run of the round, on the tree as committed:
#   smoke(), the default bench line (cpu_baseline, live PMC, profile pass), rocprofv3 kernel statistics +
#   separate FETCH_SIZE / WRITE_SIZE passes of the same workload (tools/profile_round.sh), the timeline of one Newton step
#   (tools/step_timeline.sh), A/B lines, the AMG set-up times. Copy what should be judged from gpurun_out/ into profiles/.
set -x
TAG=${1:-r05_z}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 300 python bench.py < /dev/null > $O/bench_default.json 2> $O/bench_default.err; tail -c 400 $O/bench_default.json
timeout 200 bash tools/profile_round.sh ${TAG} < /dev/null
timeout 120 bash tools/step_timeline.sh ${TAG} < /dev/null | head -24
B="--cpu-seconds 0 --no-ttt --no-spmv-hbm --pmc off --steps 200 --warmup 20 --no-profile-pass"
timeout 100 python bench.py $B --matfree < /dev/null > $O/bench_matfree.json 2> /dev/null
timeout 100 python bench.py $B --workload c5 < /dev/null > $O/bench_c5.json 2> /dev/null
NK_SS_HOST_A=0 NK_FUSED_UPDATE=0 NK_FUSED_RESIDUAL_NORMS=0 NK_PRELOADED_RHS=0 timeout 100 python bench.py $B < /dev/null > $O/bench_without_late_round5_fusions.json 2> /dev/null
NK_SS_DEFER=0 NK_SS_HOST_A=0 NK_FUSED_UPDATE=0 NK_FUSED_RESIDUAL_NORMS=0 NK_PRELOADED_RHS=0 timeout 100 python bench.py $B < /dev/null > $O/bench_round4_dispatch.json 2> /dev/null
timeout 100 python bench.py --workload c4 --steps 10 --warmup 2 --cpu-seconds 0 --no-ttt --no-spmv-hbm --pmc off --no-profile-pass < /dev/null > $O/bench_c4size_1gpu.json 2> /dev/null
for g in 512 1024 2048; do timeout 60 python tools/amg_setup_time.py $g 2>&1 | tail -2; done > $O/amg_setup.txt; cat $O/amg_setup.txt
timeout 100 python tools/amg_time.py > $O/amg_time.txt 2>&1; tail -9 $O/amg_time.txt
for f in $O/bench_*.json; do python -c "
import json
try:
    d=json.loads([x for x in open('$f') if x.startswith('{')][-1]); print('$f'.split('/')[-1], d['value'], d['ms_per_step'])
except Exception as e: print('$f FAILED', e)"; done | tee $O/bench_lines.txt
# the whole GPU suite last (the files that start several processes first): [ "$2" = "tests" ] runs it
if [ "${2:-}" = "tests" ]; then
  timeout ${3:-700} python -m pytest tests/test_gpu_multirank.py tests/test_gpu_fullsize.py tests -m gpu -q -p no:cacheprovider < /dev/null > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
fi
