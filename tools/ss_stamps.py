"""Development: where the time of the s-step block's scalar work goes (phase stamps of k_ss_reduce_factor's last workgroup,
100 MHz wall clock). python tools/ss_stamps.py"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import nonlinearsolve_jl_amd as nls
from nonlinearsolve_jl_amd import _lib as L
f = L.lib().nk_ss_debug_stamps
f.argtypes = [C.c_int, C.POINTER(C.c_ulonglong)]
ns = 1024
prob = nls.NonlinearProblem(nls.Bratu2D(ns, 6.0))
prob.u0 = torch.zeros(ns * ns, dtype=torch.float64, device="cuda")
cache = nls.init(prob, nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(fixed_iters=30, maxiters=30), concrete_jac=True), abstol=1e-300, maxiters=10**9)
for _ in range(3):
    cache.step()
out = (C.c_ulonglong * 16)()
f(1, out)
for _ in range(3):
    cache.step()
    f(1, out)
    v = list(out)
    names = ["kernel start(wg0)", "last ticket", "red in LDS", "factor done", "end", "f:Ct/U done", "f:frame ready", "(unused)"]
    base = v[0]
    print({n: round((x - base) / 100.0, 2) for n, x in zip(names, v)})
    hn = ["hess: start", "C1 loaded", "F = [C; R] ready", "recurrence done", "H stored + old rotations", "new rotations done"]
    print({n: round((x - v[8]) / 100.0, 2) for n, x in zip(hn, v[8:14])})
