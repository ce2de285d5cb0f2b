# round 4, pass A: the resident matrix-powers kernel — parity tests, micro-benchmark (variants), short bench lines A/B
set -x
TAG=${1:-r04_a}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_powers.py -q -x < /dev/null > $O/pytest_powers.log 2>&1; tail -15 $O/pytest_powers.log
for v in 0 1; do NK_PW_VARIANT=$v timeout 120 python tools/powers_bench.py 1024 15 100 < /dev/null 2>&1 | tail -1 | tee -a $O/powers_bench.jsonl; done
NK_SPMV_POWERS=0 timeout 120 python tools/powers_bench.py 1024 15 100 < /dev/null 2>&1 | tail -1 | tee -a $O/powers_bench.jsonl
timeout 120 python tools/powers_bench.py 512 15 100 < /dev/null 2>&1 | tail -1 | tee -a $O/powers_bench.jsonl
timeout 120 python tools/powers_bench.py 256 15 100 < /dev/null 2>&1 | tail -1 | tee -a $O/powers_bench.jsonl
B="--cpu-seconds 0 --no-ttt --pmc off --steps 100 --warmup 10"
for v in 0 1; do NK_PW_VARIANT=$v timeout 200 python bench.py $B < /dev/null > $O/bench_csr_pw$v.json 2> $O/bench_csr_pw$v.err; done
NK_SPMV_POWERS=0 timeout 200 python bench.py $B < /dev/null > $O/bench_csr_stream.json 2> $O/bench_csr_stream.err
for f in $O/bench_*.json; do python -c "
import json,sys
try:
    d=json.loads([x for x in open('$f') if x.startswith('{')][-1]); print('$f', d['value'], d['ms_per_step'], d.get('check'))
except Exception as e: print('$f FAILED', e)"; done
