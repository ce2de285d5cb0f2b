set -x
TAG=${1:-r03_b}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sstep.py -q > $O/pytest_sstep.log 2>&1; tail -8 $O/pytest_sstep.log
B="--cpu-seconds 0 --no-ttt"
NK_SS_FUSED=0 timeout 200 python bench.py $B > $O/bench_unfused.json 2> /dev/null
timeout 200 python bench.py $B > $O/bench_default.json 2> /dev/null
cd /tmp && export TMPDIR=/tmp
NK_SS_FUSED=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --cpu-seconds 0 --no-profile-pass --no-ttt > /dev/null 2>&1
find /tmp/kt2 -name "kt_kernel_stats.csv" -exec cp {} $GRAFT_REPO_ROOT/$O/unfused_kernel_stats.csv \;
head -20 $GRAFT_REPO_ROOT/$O/unfused_kernel_stats.csv | cut -c1-150
cd $GRAFT_REPO_ROOT
for f in $O/bench_*.json; do python -c "
import json,sys
try:
    d=json.loads([x for x in open('$f') if x.startswith('{')][-1]); print('$f', d['value'], d['ms_per_step'], d['check']['fnorm_inf_after_timed_steps'], {k:(v['avg_us'],v['launches']) for k,v in d['kernels'].items()})
except Exception as e: print('$f', 'FAILED', e)"; done
python tools/ss_stamps.py 2>/dev/null | tail -2
