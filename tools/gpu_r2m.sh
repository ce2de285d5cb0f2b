set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2m; mkdir -p $O
NK_GMRES_GRAPH=1 timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -k "fixed or c3 or 1024" > $O/pytest_graph.log 2>&1; tail -3 $O/pytest_graph.log
for gph in 0 1 0 1; do
  NK_GMRES_GRAPH=$gph timeout 200 python bench.py --cpu-seconds 0 --no-ttt --no-profile-pass > $O/bench_graph${gph}.json 2> /dev/null
  python -c "
import json; d=json.loads([x for x in open('$O/bench_graph${gph}.json') if x.startswith('{')][-1]); print('graph$gph', d['value'], d['check'])"
done
