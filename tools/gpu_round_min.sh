# The short evidence pass (≈ 4–5 GPU-minutes): the whole -m gpu suite, smoke(), the default bench line (s-step Arnoldi), the
# column-by-column A/B line, and the rocprofv3 kernel trace + FETCH_SIZE / WRITE_SIZE passes of the default command.
#   gpurun --timeout 540 -- 'bash tools/gpu_round_min.sh r02_f'
set -x
TAG=${1:-r02_x}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
timeout 260 python -m pytest tests -m gpu -q -x < /dev/null > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" < /dev/null 2>&1 | tail -1
timeout 90 python bench.py < /dev/null > $O/bench_csr.json 2> $O/bench_csr.err
timeout 40 python bench.py --ortho dcgs2 --cpu-seconds 0 --no-ttt < /dev/null > $O/bench_csr_dcgs2.json 2> /dev/null
timeout 100 bash tools/profile_round.sh ${TAG} < /dev/null
for f in $O/bench_*.json; do python -c "
import json,sys; d=json.loads([x for x in open('$f') if x.startswith('{')][-1]); print('$f', d['value'], d['n_gpus'])" < /dev/null; done
