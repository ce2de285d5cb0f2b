# One GPU-box pass that regenerates round 3's evidence (gpurun --timeout 2400 -- 'bash tools/gpu_r03_evidence.sh r03_k'):
# the whole -m gpu suite, smoke(), the default bench line (CPU leg and time-to-tolerance included) plus the A/B lines, the
# rocprofv3 kernel trace with separate FETCH_SIZE / WRITE_SIZE passes at 1024² AND 4096² (the s-step kernels HBM-resident), the
# 2/4/8-rank shared-GPU code-path runs with the self-check. Copy what should be judged from gpurun_out/ into profiles/.
set -x
TAG=${1:-r03_k}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
timeout 700 python bench.py > $O/bench_csr.json 2> $O/bench_csr.err
B="--cpu-seconds 0 --no-ttt --pmc off"
timeout 200 python bench.py $B --matfree > $O/bench_matfree.json 2> /dev/null
timeout 200 python bench.py $B --ortho dcgs2 > $O/bench_csr_dcgs2.json 2> /dev/null
timeout 200 python bench.py $B --sstep 6 --sstep-basis monomial > $O/bench_csr_sstep6_monomial.json 2> /dev/null
timeout 200 python bench.py $B --sstep 10 > $O/bench_csr_sstep10.json 2> /dev/null
NK_SS_FUSED=0 timeout 200 python bench.py $B > $O/bench_csr_unfused_scalar_work.json 2> /dev/null
timeout 200 python bench.py $B --workload c5 > $O/bench_c5_1gpu.json 2> /dev/null
timeout 300 python bench.py $B --workload c4 --steps 10 --warmup 2 > $O/bench_c4size_1gpu.json 2> /dev/null
timeout 500 bash tools/profile_round.sh ${TAG}
timeout 700 bash tools/profile_round.sh ${TAG}_c4size_1gpu --workload c4 --steps 4 --warmup 1
timeout 300 python tools/spmv_bench.py > $O/spmv_bench.jsonl 2>/dev/null
for n in 2 4 8; do
  BENCH_BACKEND=gloo NK_COMM=peer timeout 500 python bench.py --gpus $n --steps 20 --warmup 3 $B --no-weak > $O/bench_x${n}_peer_shared_gpu.json 2> $O/bench_x${n}.err
done
BENCH_BACKEND=gloo NK_COMM=peer timeout 500 python bench.py --gpus 8 --workload c4 --steps 3 --warmup 1 $B --no-weak > $O/bench_c4_x8_peer_shared_gpu.json 2> /dev/null
python tools/c5_mg_time.py 512 > $O/c5_mg_time.txt 2>&1
python tools/c2_direct.py 256 > $O/c2_direct.txt 2>&1
python tools/mg_time.py > $O/mg_time.txt 2>&1
python tools/ss_stamps.py > $O/ss_stamps.txt 2>/dev/null
for f in $O/bench_*.json; do python -c "
import json,sys
try:
    d=json.loads([x for x in open('$f') if x.startswith('{')][-1]); print('$f', d['value'], d['n_gpus'], d['ms_per_step'])
except Exception as e: print('$f FAILED', e)"; done
