set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2g; mkdir -p $O
timeout 300 python bench.py > $O/bench_csr.json 2> $O/bench_csr.err; python -c "
import json; d=json.loads([x for x in open('$O/bench_csr.json') if x.startswith('{')][-1]); print(d['value'], d['cpu_baseline'])"
timeout 200 python bench.py --matfree --cpu-seconds 0 --no-ttt > $O/bench_matfree.json 2> /dev/null; python -c "
import json; d=json.loads([x for x in open('$O/bench_matfree.json') if x.startswith('{')][-1]); print('matfree', d['value'])"
timeout 400 bash tools/profile_round.sh r02_c
timeout 300 python bench.py --cpu-seconds 0 --no-ttt > $O/bench_csr2.json 2> /dev/null; python -c "
import json; d=json.loads([x for x in open('$O/bench_csr2.json') if x.startswith('{')][-1]); print('again', d['value'])"
