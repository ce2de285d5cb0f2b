import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch, ctypes as C
import ensemble_sources as E
import nonlinearsolve_jl_amd as nls
from nonlinearsolve_jl_amd import _lib as L
from nonlinearsolve_jl_amd.core import _BatchKernel
# a dense, coupled residual: f_i = u_i^2 - p_i + 0.1*sum_j u_j/n   (dense Jacobian)
DENSE = """
template <typename T> __device__ void nk_f(const T *u, const double *p, T *f) {
  T s = u[0];
  for (int i = 1; i < NK_N; ++i) s = s + u[i];
  for (int i = 0; i < NK_N; ++i) f[i] = u[i] * u[i] - p[i] + (0.1 / NK_N) * s;
}
"""
ctx = nls.default_context()
for n in (4, 8, 16, 32, 64):
    nb = (1 << 20) // max(1, n // 4)
    P = torch.tensor(np.random.default_rng(0).uniform(1, 4, (nb, n)), device="cuda")
    u0 = torch.ones(n, dtype=torch.float64, device="cuda")
    h = _BatchKernel.get(ctx, DENSE, n, n, 0)
    du = torch.empty((nb, n), dtype=torch.float64, device="cuda"); dr = torch.empty_like(du)
    drc = torch.empty(nb, dtype=torch.int32, device="cuda"); dit = torch.empty(nb, dtype=torch.int32, device="cuda")
    ptr = lambda x: C.c_void_p(x.data_ptr())
    ts = []
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        st = L.lib().nk_batch_solve(h, nb, ptr(u0), 0, ptr(P), L.DEVICE, 0.0, 100, ptr(du), ptr(dr), ptr(drc), ptr(dit))
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e-3)
    dt = min(ts); it = dit.cpu().numpy().astype(np.int64)
    flops = it.sum() * ((2 / 3) * n ** 3 + 2 * n * n * (n + 1))   # LU + dual-number Jacobian sweeps (n partials × n-term residual)
    print(f"n={n}: {nb} systems, {dt*1e3:.2f} ms, {nb/dt/1e6:.2f} M systems/s, mean iters {it.mean():.1f}, ok {(drc.cpu().numpy()==1).mean()*100:.0f}%, ~{flops/dt/1e12:.2f} TFLOP/s", flush=True)
