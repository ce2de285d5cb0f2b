for o in cgs2 cgs; do timeout 120 python bench.py --steps 20 --warmup 3 --cpu-steps 0 --ortho $o 2>&1 | grep -v amdgpu.ids | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$o', d['value'], d['ms_per_step'], d['check'], {k:(v['avg_us'],v['launches']) for k,v in d['kernels'].items()})"; done
