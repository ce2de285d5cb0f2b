set -x
TAG=${1:-r05_h2}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
B="--cpu-seconds 0 --no-ttt --no-spmv-hbm --pmc off --steps 200 --warmup 20 --no-profile-pass"
for rep in 1 2; do
for nj in 8 16 30; do
NK_SS_HOST_A_WGS=$nj timeout 200 python bench.py $B < /dev/null > $O/bench_nj${nj}_$rep.json 2> $O/bench_default.err
done
NK_SS_HOST_A=0 timeout 200 python bench.py $B < /dev/null > $O/bench_nohosta_$rep.json 2> /dev/null
done
timeout 300 bash tools/step_timeline.sh ${TAG} < /dev/null > /dev/null 2>&1
head -24 gpurun_out/${TAG}_step_timeline.md
for f in $O/bench_*.json; do python -c "
import json,sys
try:
    d=json.loads([x for x in open('$f') if x.startswith('{')][-1]); print('$f', d['value'], d['n_gpus'], d['ms_per_step'])
except Exception as e: print('$f FAILED', e)"; done
