# One GPU-box pass that regenerates the round's evidence (run through `gpurun --timeout 2400 -- 'bash tools/gpu_round.sh r02_e'`):
# the whole -m gpu suite, smoke(), the default bench line plus the matrix-free / C4-size / C5 lines, the rocprofv3 kernel trace
# with separate FETCH_SIZE / WRITE_SIZE passes at 1024² and 4096², the SpMV warm/cold hygiene run and the 2/4/8-rank
# shared-GPU code-path runs. Copy what should be judged from gpurun_out/ into profiles/.
set -x
TAG=${1:-r02_x}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 400 python bench.py > $O/bench_csr.json 2> $O/bench_csr.err
timeout 200 python bench.py --matfree --cpu-seconds 0 --no-ttt > $O/bench_matfree.json 2> /dev/null
timeout 200 python bench.py --ortho dcgs2 --cpu-seconds 0 --no-ttt > $O/bench_csr_dcgs2.json 2> /dev/null
NK_SS_FUSED=0 timeout 200 python bench.py --cpu-seconds 0 --no-ttt > $O/bench_csr_unfused_tails.json 2> /dev/null
timeout 200 python bench.py --workload c5 --cpu-seconds 0 --no-ttt > $O/bench_c5_1gpu.json 2> /dev/null
timeout 200 python bench.py --workload c4 --steps 4 --warmup 1 --cpu-seconds 0 --no-ttt > $O/bench_c4size_1gpu.json 2> /dev/null
timeout 500 bash tools/profile_round.sh ${TAG}
timeout 600 bash tools/profile_round.sh ${TAG}_c4size_1gpu --workload c4 --steps 4 --warmup 1
timeout 300 python tools/spmv_bench.py > $O/spmv_bench.jsonl 2>/dev/null
for n in 2 4 8; do
  BENCH_BACKEND=gloo NK_COMM=peer timeout 400 python bench.py --gpus $n --cpu-seconds 0 --no-ttt --no-weak > $O/bench_x${n}_peer_shared_gpu.json 2> /dev/null
done
python tools/c5_mg_time.py 512 > $O/c5_mg_time.txt 2>&1
python tools/c2_direct.py 256 > $O/c2_direct.txt 2>&1
for f in $O/bench_*.json; do python -c "
import json,sys; d=json.loads([x for x in open('$f') if x.startswith('{')][-1]); print('$f', d['value'], d['n_gpus'])"; done
