"""ctypes wrapper around oracle/libnk_oracle.so (C99 + OpenMP restatement). TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libnk_oracle.so")
_lib = None

_d = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


def build(force=False):
    src = os.path.join(_HERE, "nk_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libnk_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.orc_num_threads.restype = C.c_int
        L.orc_set_num_threads.argtypes = [C.c_int]
        L.orc_dot.restype = C.c_double
        L.orc_dot.argtypes = [C.c_int64, _d, _d]
        L.orc_norm_inf.restype = C.c_double
        L.orc_norm_inf.argtypes = [C.c_int64, _d]
        L.orc_spmv.argtypes = [C.c_int64, _i, _i, _d, _d, _d]
        L.orc_spmv_t.argtypes = [C.c_int64, C.c_int64, _i, _i, _d, _d, _d]
        L.orc_bratu_residual.argtypes = [C.c_int64, C.c_double, C.c_double, _d, _d]
        L.orc_bratu_jvp.argtypes = [C.c_int64, C.c_double, C.c_double, _d, _d, _d]
        L.orc_bratu_nnz.restype = C.c_int64
        L.orc_bratu_nnz.argtypes = [C.c_int64]
        L.orc_bratu_pattern.argtypes = [C.c_int64, _i, _i]
        L.orc_bratu_jac_values.argtypes = [C.c_int64, C.c_double, C.c_double, _d, _i, _d]
        L.orc_brusselator_residual.argtypes = [C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double, _d, _d]
        L.orc_gmres_csr.restype = C.c_int
        L.orc_gmres_csr.argtypes = [C.c_int64, _i, _i, _d, _d, _d, C.c_double, C.c_double, C.c_int, C.c_int,
                                    C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_ensemble_newton.argtypes = [C.c_int, C.c_int, C.c_int64, _d, C.c_int, _d, C.c_int, C.c_double, C.c_int, _d, _d,
                                          _i, _i]
        L.orc_bratu_newton.restype = C.c_int
        L.orc_bratu_newton.argtypes = [C.c_int64, C.c_double, C.c_double, _d, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_int, C.c_double, _d, _i, _d]
        L.orc_bratu_newton_cheb.restype = C.c_int
        L.orc_bratu_newton_cheb.argtypes = [C.c_int64, C.c_double, C.c_double, _d, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_int, C.c_double, C.c_double, _d, _i]
        L.orc_stream_triad.restype = C.c_double
        L.orc_stream_triad.argtypes = [C.c_int64, C.c_int]
        L.orc_spmv_rate.restype = C.c_double
        L.orc_spmv_rate.argtypes = [C.c_int64, C.c_int]
        L.orc_phase_times.argtypes = [_d]
        L.orc_bratu_newton_fast.restype = C.c_double
        L.orc_bratu_newton_fast.argtypes = [C.c_int64, C.c_double, C.c_double, _d, C.c_int, C.c_int, C.c_int, _d]
        L.orc_bratu_newton_fast_sstep.restype = C.c_double
        L.orc_bratu_newton_fast_sstep.argtypes = [C.c_int64, C.c_double, C.c_double, _d, C.c_int, C.c_int, C.c_int, C.c_int, _d]
        L.orc_bratu_newton_fast_sstep2.restype = C.c_double
        L.orc_bratu_newton_fast_sstep2.argtypes = [C.c_int64, C.c_double, C.c_double, _d, C.c_int, C.c_int, C.c_int, C.c_int,
                                                   C.c_int, _d]
        L.orc_leja_nodes.restype = None
        L.orc_leja_nodes.argtypes = [C.c_int, _d]
        _lib = L
    return _lib


def num_threads():
    return lib().orc_num_threads()


def set_num_threads(t):
    lib().orc_set_num_threads(int(t))


def spmv(rowptr, col, val, x):
    y = np.empty(len(rowptr) - 1)
    lib().orc_spmv(len(rowptr) - 1, rowptr, col, val, x, y)
    return y


def spmv_t(rowptr, col, val, x, ncols):
    y = np.empty(ncols)
    lib().orc_spmv_t(len(rowptr) - 1, ncols, rowptr, col, val, x, y)
    return y


def bratu_residual(ns, lam, scale, u):
    f = np.empty_like(u)
    lib().orc_bratu_residual(ns, lam, scale, u, f)
    return f


def bratu_jvp(ns, lam, scale, u, v):
    jv = np.empty_like(u)
    lib().orc_bratu_jvp(ns, lam, scale, u, v, jv)
    return jv


def bratu_pattern(ns):
    nnz = lib().orc_bratu_nnz(ns)
    rowptr = np.empty(ns * ns + 1, dtype=np.int32)
    col = np.empty(nnz, dtype=np.int32)
    lib().orc_bratu_pattern(ns, rowptr, col)
    return rowptr, col


def bratu_jac_values(ns, lam, scale, u, rowptr):
    val = np.empty(int(rowptr[-1]))
    lib().orc_bratu_jac_values(ns, lam, scale, u, rowptr, val)
    return val


def brusselator_residual(N, A, B, alpha, dx, u):
    du = np.empty_like(u)
    lib().orc_brusselator_residual(N, A, B, alpha, dx, u, du)
    return du


def gmres_csr(rowptr, col, val, b, atol=0.0, rtol=1e-8, m=30, itmax=300, fixed_iters=0):
    x = np.empty_like(b)
    conv, r0, r1 = C.c_int(0), C.c_double(0), C.c_double(0)
    it = lib().orc_gmres_csr(len(b), rowptr, col, val, b, x, atol, rtol, m, itmax, fixed_iters,
                             C.byref(conv), C.byref(r0), C.byref(r1))
    return x, dict(iters=it, converged=bool(conv.value), rnorm0=r0.value, rnorm=r1.value)


def bratu_newton(ns, lam, scale, u0, nsteps, use_csr=True, m=30, itmax=300, fixed_iters=0, forcing=True,
                 rtol=1e-4):
    u = np.array(u0, dtype=np.float64, copy=True)
    fn = np.zeros(nsteps)
    gi = np.zeros(nsteps, dtype=np.int32)
    eta = np.zeros(nsteps)
    lib().orc_bratu_newton(ns, lam, scale, u, nsteps, int(use_csr), m, itmax, fixed_iters, int(forcing), rtol,
                           fn, gi, eta)
    return u, fn, gi, eta


def bratu_newton_cheb(ns, lam, scale, u0, maxsteps=50, use_csr=True, m=30, itmax=300, cheb_degree=32, cheb_ratio=300.0,
                      abstol=1e-8):
    """NewtonRaphson + GMRES(m) + Eisenstat–Walker + Chebyshev(degree, ratio) right preconditioner, stop at
    ‖f‖∞ ≤ abstol — the CPU counterpart of the device `precs` path. Returns (u, fnorm trace, gmres iters)."""
    u = np.array(u0, dtype=np.float64, copy=True)
    fn = np.zeros(maxsteps)
    gi = np.zeros(maxsteps, dtype=np.int32)
    k = lib().orc_bratu_newton_cheb(ns, lam, scale, u, maxsteps, int(use_csr), m, itmax, cheb_degree, cheb_ratio, abstol,
                                    fn, gi)
    return u, fn[:k], gi[:k]


def ensemble_newton(kind, u0, P, abstol=None, maxiters=1000):
    """SimpleNewtonRaphson over the rows of P (kind 0: u.*u .- p, kind 1: tutorial p2_f); OpenMP over systems.
    Returns (u, resid, retcode, iters)."""
    P = np.ascontiguousarray(P, dtype=np.float64)
    u0 = np.ascontiguousarray(u0, dtype=np.float64)
    nb, npar = P.shape
    n = u0.shape[-1]
    u, r = np.empty((nb, n)), np.empty((nb, n))
    rc, it = np.empty(nb, dtype=np.int32), np.empty(nb, dtype=np.int32)
    if abstol is None:
        abstol = float(np.finfo(float).eps) ** 0.8
    lib().orc_ensemble_newton(kind, n, nb, u0, int(u0.ndim == 2), P, npar, abstol, maxiters, u, r, rc, it)
    return u, r, rc, it


def stream_triad(n=1 << 26, reps=5):
    """Best-of-reps STREAM triad GB/s of this box (24 n bytes per pass, first-touch placement)."""
    return lib().orc_stream_triad(int(n), int(reps))


def spmv_rate(ns, reps=10):
    """Best-of-reps CSR SpMV GB/s on the Bratu ns×ns pattern (algorithmic bytes 12 nnz + 4 (n+1) + 16 n)."""
    return lib().orc_spmv_rate(int(ns), int(reps))


def bratu_newton_fast(ns, lam, scale, u0, nsteps, use_csr=True, m=30):
    """Tuned CPU leg: nsteps fixed-work Newton steps (m Arnoldi steps of DCGS2-1R GMRES each), one persistent OpenMP
    region with first-touch placement. Returns (u, fnorm trace, seconds of the step loop)."""
    u = np.array(u0, dtype=np.float64, copy=True)
    fn = np.zeros(nsteps)
    sec = lib().orc_bratu_newton_fast(ns, lam, scale, u, nsteps, int(use_csr), m, fn)
    if sec < 0:
        raise ValueError("bad restart length")
    return u, fn, sec


def bratu_newton_fast_sstep(ns, lam, scale, u0, nsteps, use_csr=True, m=30, s=6, basis="monomial"):
    """The same fixed-work Newton steps with the s-step Arnoldi process (the device's NK_ORTHO_SSTEP; C restatement of
    reference_restatement.gmres_sstep). basis = "newton": Leja-ordered Chebyshev shifts on the Gershgorin interval of every
    Jacobian (the device's default for this operator, with s = 15). Returns (u, fnorm trace, seconds of the step loop)."""
    u = np.array(u0, dtype=np.float64, copy=True)
    fn = np.zeros(nsteps)
    sec = lib().orc_bratu_newton_fast_sstep2(ns, lam, scale, u, nsteps, int(use_csr), m, int(s),
                                             {"monomial": 0, "newton": 1}[basis], fn)
    if sec == -2.0:
        raise ArithmeticError("a block of the monomial basis lost rank numerically (Cholesky breakdown)")
    if sec < 0:
        raise ValueError("bad restart length or block size")
    return u, fn, sec


def leja_nodes(s):
    out = np.zeros(16)
    lib().orc_leja_nodes(int(s), out)
    return out[:s]


def phase_times():
    """Seconds thread 0 spent in [operator, dot sweep, reductions + tail (+ barrier waits), axpy sweep, rest] of the last
    bratu_newton_fast run (diagnostics)."""
    out = np.zeros(5)
    lib().orc_phase_times(out)
    return out
