# One GPU-box pass that regenerates round 4's evidence (gpurun --timeout 2400 -- 'bash tools/gpu_r04_evidence.sh r04_x [quick]'):
# the whole -m gpu suite, smoke(), the default bench line (CPU leg, time-to-tolerance, live PMC traffic, HBM-resident SpMV), the
# A/B lines (resident matrix powers off; column-by-column; matrix-free; C5; 4096² on one GPU), the rocprofv3 kernel trace with
# separate FETCH_SIZE / WRITE_SIZE passes, the launch-by-launch timeline of one Newton step, the matrix-powers micro-benchmark.
# Copy what should be judged from gpurun_out/ into profiles/.
set -x
TAG=${1:-r04_x}
MODE=${2:-full}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q < /dev/null > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" < /dev/null 2>&1 | tail -1 | tee $O/smoke.txt
B0="--cpu-seconds 0 --no-ttt --no-spmv-hbm --pmc off --steps 100 --warmup 10"
timeout 700 python bench.py < /dev/null > $O/bench_csr.json 2> $O/bench_csr.err
NK_SS_IMPLICIT=0 timeout 200 python bench.py $B0 < /dev/null > $O/bench_csr_explicit_second_pass.json 2> /dev/null
B="--cpu-seconds 0 --no-ttt --no-spmv-hbm --pmc off --steps 100 --warmup 10"
NK_SPMV_POWERS=0 timeout 200 python bench.py $B < /dev/null > $O/bench_csr_streaming_spmv.json 2> /dev/null
timeout 200 python bench.py $B --matfree < /dev/null > $O/bench_matfree.json 2> /dev/null
timeout 200 python bench.py $B --workload c5 < /dev/null > $O/bench_c5_1gpu.json 2> /dev/null
NK_SPMV_POWERS=0 timeout 200 python bench.py $B --workload c5 < /dev/null > $O/bench_c5_1gpu_streaming_spmv.json 2> /dev/null
NK_SS_MM=0 NK_SS_KCONST=0 timeout 200 python bench.py $B0 < /dev/null > $O/bench_csr_round3_sweeps.json 2> /dev/null
( echo "== round-3 sweep kernels (NK_SS_MM=0 NK_SS_KCONST=0)"; NK_SS_MM=0 NK_SS_KCONST=0 timeout 200 python tools/sweep_shapes_bench.py 1048576 16777216 2>&1 | grep sweep;
  echo "== round-4 sweep kernels (default)"; timeout 200 python tools/sweep_shapes_bench.py 1048576 16777216 2>&1 | grep sweep ) > $O/sweep_shapes.txt
if [ "$MODE" = "full" ]; then
  timeout 200 python bench.py $B --ortho dcgs2 < /dev/null > $O/bench_csr_dcgs2.json 2> /dev/null
  timeout 300 python bench.py $B --workload c4 --steps 10 --warmup 2 < /dev/null > $O/bench_c4size_1gpu.json 2> /dev/null
  timeout 700 bash tools/profile_round.sh ${TAG}_c4size_1gpu --workload c4 --steps 4 --warmup 1 < /dev/null
  timeout 300 python tools/spmv_bench.py 100 < /dev/null > $O/spmv_bench.jsonl 2>/dev/null
  timeout 300 python tools/amg_time.py < /dev/null > $O/amg_time.txt 2>&1
  timeout 200 python tools/ss_stamps.py < /dev/null > $O/ss_stamps.txt 2>/dev/null
  for n in 2 8; do
    BENCH_BACKEND=gloo NK_COMM=peer timeout 500 python bench.py --gpus $n --steps 20 --warmup 3 $B --no-weak < /dev/null > $O/bench_x${n}_peer_shared_gpu.json 2> $O/bench_x${n}.err
  done
fi
timeout 500 bash tools/profile_round.sh ${TAG} < /dev/null
timeout 300 bash tools/step_timeline.sh ${TAG} < /dev/null
for s in 1 2 4 8 15; do timeout 120 python tools/powers_bench.py 1024 $s 60 < /dev/null 2>&1 | tail -1 >> $O/powers_bench.jsonl; done
NK_SPMV_POWERS=0 timeout 120 python tools/powers_bench.py 1024 15 60 < /dev/null 2>&1 | tail -1 >> $O/powers_bench.jsonl
cat $O/powers_bench.jsonl
for f in $O/bench_*.json; do python -c "
import json,sys
try:
    d=json.loads([x for x in open('$f') if x.startswith('{')][-1]); print('$f', d['value'], d['n_gpus'], d['ms_per_step'])
except Exception as e: print('$f FAILED', e)"; done
