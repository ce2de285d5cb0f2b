timeout 300 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x --timeout 300 2>&1 | tail -2
timeout 250 python bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/bench7.log; python -c "
import json
d=json.loads(open('gpurun_out/bench7.log').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['cpu_baseline']['value']); print(d['time_to_tolerance'])"
