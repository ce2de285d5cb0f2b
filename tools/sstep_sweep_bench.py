import sys, ctypes as C, numpy as np
sys.path.insert(0, '.')
import nonlinearsolve_jl_amd as nls
from nonlinearsolve_jl_amd import _lib as L
lib = L.lib()
ctx = nls.default_context()
f = lib.nk_ss_sweep_test
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_double)]
rng = np.random.default_rng(0)
def run(mode, n, k, s, iters=0):
    V = np.asfortranarray(rng.standard_normal((n, k + s)))
    U = rng.standard_normal((k, s)) * 0.1
    Rinv = np.triu(rng.standard_normal((s, s))) + 2 * np.eye(s)
    coef = np.concatenate([U.ravel(), Rinv.ravel()])
    V0 = V.copy()
    gram = np.zeros(((k + s), s))
    us = C.c_double(0)
    rc = f(ctx._h, mode, n, k, s, V.ctypes.data, coef.ctypes.data, gram.ctypes.data, iters, C.byref(us))
    assert rc == 0, lib.nk_last_error()
    if mode == 0:
        Wn = V0[:, k:]
    else:
        Wn = (V0[:, k:] - V0[:, :k] @ U) @ Rinv
    eV = np.max(np.abs(V[:, k:] - Wn)) / np.max(np.abs(Wn))
    assert np.array_equal(V[:, :k], V0[:, :k])
    eg = 0.0
    if mode != 2:
        X = np.concatenate([V0[:, :k], Wn], axis=1)
        gref = X.T @ Wn
        eg = np.max(np.abs(gram - gref)) / np.max(np.abs(gref))
    return eV, eg, us.value
for (n, k, s) in [(1000, 3, 2), (5000, 1, 6), (70001, 17, 5), (4096, 30, 6), (300, 40, 8), (257, 7, 1), (10000, 60, 3)]:
    for mode in (0, 1, 2):
        eV, eg, _ = run(mode, n, k, s)
        print(n, k, s, mode, f"{eV:.2e} {eg:.2e}")
        assert __import__("os").environ.get("NK_SS_DBG","0") != "0" or (eV < 1e-13 and eg < 1e-12)
n = 1 << 20
for k in (6, 18, 30):
    for mode in (0, 1, 2):
        _, _, us = run(mode, n, k, 6, iters=20)
        rd = (k + 6) * n * 8; wr = 6 * n * 8 if mode else 0
        print(f"n=2^20 k={k} s=6 mode={mode}: {us:.1f} us  {(rd+wr)/us/1e6:.2f} TB/s")
