"""Times the s-step sweeps of the default cycle's shapes (blocks of 15 behind 1 and 16 columns, n = 2^20 and 2^24) through the
development harness nk_ss_sweep_test, checks them against NumPy, prints one line per (shape, sweep): µs and TB/s of the
algorithmic bytes. Environment switches of csrc/nk_sstep.hip (NK_SS_KCONST, NK_SS_BARRIERS, NK_SS_MM) select variants."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, '.')
import nonlinearsolve_jl_amd as nls
from nonlinearsolve_jl_amd import _lib as L

lib = L.lib()
ctx = nls.default_context()
f = lib.nk_ss_sweep_test
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_double)]
rng = np.random.default_rng(0)


def run(mode, n, k, s, iters):
    V = np.asfortranarray(rng.standard_normal((n, k + s)))
    U = rng.standard_normal((k, s)) * 0.1
    Rup = np.triu(rng.standard_normal((s, s))) * 0.3 + 2 * np.eye(s)
    coef = np.concatenate([U.ravel(), Rup.ravel()])
    V0 = V.copy()
    gram = np.zeros(((k + s), s))
    us = C.c_double(0)
    rc = f(ctx._h, mode, n, k, s, V.ctypes.data, coef.ctypes.data, gram.ctypes.data, iters, C.byref(us))
    assert rc == 0, lib.nk_last_error()
    Wn = V0[:, k:] if mode == 0 else (V0[:, k:] - V0[:, :k] @ U) @ np.linalg.inv(Rup)
    eV = np.max(np.abs(V[:, k:] - Wn)) / np.max(np.abs(Wn))
    eg = 0.0
    if mode != 2:
        X = np.concatenate([V0[:, :k], Wn], axis=1)
        gref = X.T @ Wn
        eg = np.max(np.abs(gram - gref)) / np.max(np.abs(gref))
    return eV, eg, us.value


sizes = [int(a) for a in sys.argv[1:]] or [1 << 20]
for n in sizes:
    for (k, s) in ((1, 15), (16, 15)):
        for mode in (0, 1):
            eV, eg, us = run(mode, n, k, s, 30 if n <= (1 << 21) else 8)
            by = 8.0 * n * ((k + s) + (s if mode else 0))
            print(f"n={n} k={k} s={s} sweep {'AB'[mode]}: {us:8.2f} us  {by / us / 1e6:5.2f} TB/s   err {eV:.1e} {eg:.1e}", flush=True)
            assert eV < 1e-13 and eg < 1e-12
