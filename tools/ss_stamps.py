"""(needs the stamps build: make -C nonlinearsolve.jl_amd/csrc stamps; NK_LIB_PATH=nonlinearsolve.jl_amd/lib/libmi355x_nk_stamps.so)
Development: where the time of the s-step block's scalar work goes — phase stamps of k_ss_job's last workgroup (100 MHz wall
clock), one bank per kind of launch: the first block's (first factorisation only), the second block's (the pending block's second
factorisation + this block's first), the cycle's last (second factorisation, Hessenberg columns, back-substitution).
python tools/ss_stamps.py"""
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
out = (C.c_ulonglong * 48)()
f(1, out)
names = {0: "start(wg0)", 1: "last ticket", 2: "operands in LDS", 5: "f:Ct/U done (last factorisation)", 6: "f:frame ready", 3: "second factorisation (+Wi, D) done",
         4: "first factorisation done", 8: "hess: start", 9: "hess: after barrier", 10: "hess: A v_k", 11: "hess: recurrence done",
         12: "hess: H stored + old rotations", 7: "hess: rotations done", 13: "hess: rotations + verdict", 14: "outcome published", 15: "y done"}
for _ in range(3):
    cache.step()
    f(1, out)
    v = list(out)
    for bank, what in enumerate(("first block", "second block / the job hosted by sweep A", "cycle's last launch")):
        b = v[16 * bank:16 * bank + 16]
        if b[0] == 0:
            continue
        order = sorted((x, i) for i, x in enumerate(b) if x >= b[0] and i in names and (x - b[0]) < 100000)
        print(what + ": " + ", ".join(f"{names[i]} {round((x - b[0]) / 100.0, 2)}" for x, i in order))
