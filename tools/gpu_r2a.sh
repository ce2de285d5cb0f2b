set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > gpurun_out/r2a/pytest.log
tail -5 gpurun_out/r2a/pytest.log
timeout 300 python bench.py --steps 20 --warmup 3 > gpurun_out/r2a/bench_csr.json 2> gpurun_out/r2a/bench_csr.err
tail -c 1500 gpurun_out/r2a/bench_csr.json
NK_DCGS2_SPLIT_TAIL=1 timeout 200 python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --no-ttt > gpurun_out/r2a/bench_split.json 2>&1
NK_FETCH_MEMCPY=1 timeout 200 python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --no-ttt > gpurun_out/r2a/bench_memcpy.json 2>&1
timeout 200 python bench.py --steps 20 --warmup 3 --cpu-seconds 0 --no-ttt --matfree > gpurun_out/r2a/bench_matfree.json 2>&1
for f in split memcpy matfree; do python -c "
import json,sys
try:
  l=[x for x in open('gpurun_out/r2a/bench_$f.json') if x.startswith('{')][-1]; d=json.loads(l); print('$f', d['value'], d['ms_per_step'])
except Exception as e: print('$f failed', e)
"; done
