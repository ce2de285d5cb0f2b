"""Times the s-step sweeps of the default cycle's shapes in isolation (nk_ss_sweep_test, development harness): with NK_LIB_PATH
pointing at a -DNK_SS_EXP=1 / 2 build of nk_sstep.hip the sweep-B loop runs without its matrix instructions / without its loads —
what each resource costs on its own (tools/gpu_sweep_decomposition.sh). Results of those builds are not checked."""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, '.')
import nonlinearsolve_jl_amd as nls
from nonlinearsolve_jl_amd import _lib as L
lib = L.lib()
ctx = nls.default_context()
f = lib.nk_ss_sweep_test
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_double)]
rng = np.random.default_rng(0)
n = 1 << 20
tag = os.path.basename(os.environ.get("NK_LIB_PATH", "default"))
for (mode, k, s, flags) in [(0, 1, 15, 0), (1, 1, 15, 0), (0, 16, 15, 0), (1, 16, 15, 0), (1, 16, 15, 1)]:
    V = np.asfortranarray(rng.standard_normal((n, k + s)))
    U = rng.standard_normal((k, s)) * 0.01
    R = np.triu(rng.standard_normal((s, s)) * 0.01) + np.eye(s)
    coef = np.concatenate([U.ravel(), R.ravel()])
    gram = np.zeros((k + s, s))
    best = 1e30
    for rep in range(3):
        us = C.c_double(0)
        rc = f(ctx._h, 3 if flags else mode, n, k, s, V.ctypes.data, coef.ctypes.data, gram.ctypes.data, 40, C.byref(us))
        assert rc == 0
        best = min(best, us.value)
    rd = (k + s) * n * 8
    wr = s * n * 8 if (mode == 1 and not flags) else 0
    print(f"{tag:28s} mode={'A' if mode == 0 else 'B'} k={k:2d} s={s} {'stores nothing' if flags else '':14s} {best:7.1f} us  "
          f"{(rd + wr) / best / 1e6:5.2f} TB/s", flush=True)
