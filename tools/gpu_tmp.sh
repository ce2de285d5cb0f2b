cd $GRAFT_REPO_ROOT
for o in "--ortho dcgs2" "--ortho sstep --sstep 6" "--ortho sstep --sstep 5" "--ortho sstep --sstep 3" "--ortho sstep --sstep 8"; do
  timeout 300 python bench.py --cpu-seconds 0 --no-ttt $o 2>&1 | tail -1 | python -c "
import sys, json
l = sys.stdin.read().strip()
try:
    d = json.loads(l); print('$o', d['value'], d['ms_per_step'], {k: (v['avg_us'], v['launches']) for k, v in d.get('kernels', {}).items()})
except Exception as e:
    print('$o', 'ERR', l[-400:])
"
done
