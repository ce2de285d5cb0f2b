timeout 300 python -m pytest tests/test_gpu_solvers.py -m gpu -q -x --timeout 300 -k "direct or default_linsolve or banded or tiny or operator_jac" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof6 -o kt -- python $GRAFT_REPO_ROOT/tools/c2_direct.py > $GRAFT_REPO_ROOT/gpurun_out/prof6.log 2>&1
tail -2 $GRAFT_REPO_ROOT/gpurun_out/prof6.log
python - <<PY
import csv, os
for r in list(csv.DictReader(open(os.environ["GRAFT_REPO_ROOT"]+'/gpurun_out/prof6/kt_kernel_stats.csv')))[:4]:
    print(r['Name'][:40], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage'])
PY
