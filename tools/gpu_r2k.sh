set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2k; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 400 python bench.py > $O/bench_csr.json 2> $O/bench_csr.err; python -c "
import json; d=json.loads([x for x in open('$O/bench_csr.json') if x.startswith('{')][-1]); print(d['value'], d['roofline'], d['cpu_baseline']['value'])"
timeout 200 python bench.py --matfree --cpu-seconds 0 --no-ttt > $O/bench_matfree.json 2> /dev/null; python -c "
import json; d=json.loads([x for x in open('$O/bench_matfree.json') if x.startswith('{')][-1]); print('matfree', d['value'])"
timeout 200 python bench.py --workload c5 --cpu-seconds 0 --no-ttt > $O/bench_c5.json 2> /dev/null; python -c "
import json; d=json.loads([x for x in open('$O/bench_c5.json') if x.startswith('{')][-1]); print('c5', d['value'])"
timeout 200 python bench.py --workload c4 --steps 4 --warmup 1 --cpu-seconds 0 --no-ttt > $O/bench_c4.json 2> /dev/null; python -c "
import json; d=json.loads([x for x in open('$O/bench_c4.json') if x.startswith('{')][-1]); print('c4', d['value'])"
timeout 500 bash tools/profile_round.sh r02_d
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ht; rocprofv3 --hip-trace --stats --output-format csv -d /tmp/ht -o ht -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --cpu-seconds 0 --no-profile-pass --no-ttt > /dev/null 2>&1
find /tmp/ht -name "ht_hip_api_stats.csv" -exec head -12 {} \; > $GRAFT_REPO_ROOT/gpurun_out/r02_d_hip_api_stats.csv; cat $GRAFT_REPO_ROOT/gpurun_out/r02_d_hip_api_stats.csv | cut -c1-150
