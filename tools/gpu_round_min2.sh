# Evidence refresh for the s-step default (≈ 2.5 GPU-minutes): the two-rank test, the default bench line, the A/B lines
# (fused / un-fused scalar work, s = 5 and 8, column-by-column), matrix-free and C5 lines, rocprofv3 kernel trace + PMC passes.
set -x
TAG=${1:-r02_x}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
timeout 60 python -m pytest tests/test_gpu_multirank.py -q -x -k two_ranks < /dev/null > $O/pytest_two_ranks.log 2>&1; tail -2 $O/pytest_two_ranks.log
timeout 90 python bench.py < /dev/null > $O/bench_csr.json 2> $O/bench_csr.err
NK_SS_FUSED=0 timeout 30 python bench.py --cpu-seconds 0 --no-ttt < /dev/null > $O/bench_csr_unfused_tails.json 2> /dev/null
timeout 30 python bench.py --sstep 5 --cpu-seconds 0 --no-ttt < /dev/null > $O/bench_csr_sstep5.json 2> /dev/null
timeout 30 python bench.py --sstep 8 --cpu-seconds 0 --no-ttt < /dev/null > $O/bench_csr_sstep8.json 2> /dev/null
timeout 30 python bench.py --ortho dcgs2 --cpu-seconds 0 --no-ttt < /dev/null > $O/bench_csr_dcgs2.json 2> /dev/null
timeout 30 python bench.py --matfree --cpu-seconds 0 --no-ttt < /dev/null > $O/bench_matfree.json 2> /dev/null
timeout 30 python bench.py --workload c5 --cpu-seconds 0 --no-ttt < /dev/null > $O/bench_c5_1gpu.json 2> /dev/null
timeout 100 bash tools/profile_round.sh ${TAG} < /dev/null
for f in $O/bench_*.json; do python -c "
import json,sys; d=json.loads([x for x in open('$f') if x.startswith('{')][-1]); print('$f', d['value'], d['n_gpus'])" < /dev/null; done
