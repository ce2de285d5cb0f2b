cd $GRAFT_REPO_ROOT
export BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
run() { local t0=$(date +%s.%N); env $1 timeout 75 python bench.py --gpus $2 $3 2>gpurun_out/_two_err.log | python -c "import sys,json
ls=[l for l in sys.stdin if l.startswith('{')]
print('$1', '$2', (lambda d:(d['value'], d['check']))(json.loads(ls[-1])) if ls else 'NO OUTPUT')"; echo "   rc=$? elapsed $(python -c "import time; print(round(time.time()-$t0,1))")"; tail -3 gpurun_out/_two_err.log | cut -c1-300; }
C4="--workload c4 --steps 3 --warmup 1 --cpu-seconds 0 --no-ttt --no-weak --no-profile-pass"
D="--steps 40 --warmup 5 --cpu-seconds 0 --no-ttt --no-weak --no-profile-pass --no-spmv-hbm --pmc off"
run X=1 1 "$D"
run X=1 2 "$D"
run X=1 2 "$D"
run X=1 2 "$C4"
run X=1 2 "$C4"
run NK_FUSED_UPDATE=0 2 "$C4"
