NK_SPMV_TILE=1024 timeout 100 python tools/microbench.py --mode spmv 2>&1 | grep -v amdgpu.ids
timeout 200 python bench.py --cpu-steps 0 2>&1 | grep -v amdgpu.ids > gpurun_out/bench6.log; python -c "
import json
d=json.loads(open('gpurun_out/bench6.log').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_us']); print({k:(v['avg_us'],v['launches'],v['GB/s']) for k,v in d['kernels'].items()})"
timeout 200 python bench.py --cpu-steps 0 --matfree 2>&1 | grep -v amdgpu.ids > gpurun_out/bench6m.log; python -c "
import json
d=json.loads(open('gpurun_out/bench6m.log').read().strip().splitlines()[-1]); print('matfree', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_us'])"
timeout 200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -q -x --timeout 200 2>&1 | tail -2
