python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 200 python bench.py --cpu-steps 6 2>&1 | grep -v amdgpu.ids > gpurun_out/bench5.log; python -c "
import json
d=json.loads(open('gpurun_out/bench5.log').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']); print({k:(v['avg_us'],v['launches'],v['GB/s']) for k,v in d['kernels'].items()})"
timeout 300 python -m pytest tests -m gpu -q --timeout 250 2>&1 | tail -2
