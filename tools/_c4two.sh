cd $GRAFT_REPO_ROOT
export BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
C="--workload c4 --steps 3 --warmup 1 --cpu-seconds 0 --no-ttt --no-weak --no-profile-pass"
run() { env $1 python bench.py --gpus 2 $C 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', d['value'], d['check'])"; }
for i in 1 2 3; do run X=1; done
for i in 1 2; do run NK_PW_RANKS=0; done
for i in 1 2; do run NK_FUSED_UPDATE=0; done
for i in 1 2; do run NK_FUSED_RESIDUAL_NORMS=0; done
