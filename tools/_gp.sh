timeout 100 python tools/microbench.py --mode spmv 2>&1 | grep -v amdgpu.ids
timeout 120 python bench.py --steps 20 --warmup 3 --cpu-steps 0 --matfree 2>&1 | grep -v amdgpu.ids | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('matfree', d['value'], d['ms_per_step'], d['roofline']['avg_us'], d['roofline']['achieved'])"
timeout 300 python -m pytest tests -m gpu -q --timeout 300 2>&1 | tail -2
