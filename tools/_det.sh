cd $GRAFT_REPO_ROOT
export BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
run() { env $1 timeout 30 python bench.py --gpus $4 $2 2>/dev/null | python -c "import sys,json
ls=[l for l in sys.stdin if l.startswith('{')]
print('$3', (lambda d:(d['check']['fnorm_inf_after_timed_steps'], d['check']['allreduces'], d['check']['halo_exchanges'], d['value']))(json.loads(ls[-1])) if ls else 'NO OUTPUT')"; }
D="--grid 512 --steps 20 --warmup 2 --cpu-seconds 0 --no-ttt --no-weak --no-profile-pass --no-spmv-hbm --pmc off"
for i in 1 2 3 4 5 6 7 8 9 10; do run "X=1" "$D" two_default 2; done
for i in 1 2 3 4; do run "NK_COMM=peer" "$D" two_peer 2; done
for i in 1 2 3; do run "NK_COMM=peer" "--grid 1024 --steps 30 --warmup 2 --cpu-seconds 0 --no-ttt --no-weak --no-profile-pass --no-spmv-hbm --pmc off" two_peer_1024 2; done
