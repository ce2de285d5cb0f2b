"""Host-side mirror of the reference's interface for the first-order step path, above the C ABI.

The names, argument meaning and error behaviour follow NonlinearSolve.jl so that the parity tests read like
the reference's own tests:

    prob = NonlinearProblem(NonlinearFunction(f!; jvp = jvp!), u0, p)          # SciMLBase
    sol  = solve(prob, NewtonRaphson(; linsolve = KrylovJL_GMRES(), forcing = EisenstatWalkerForcing2());
                 abstol = 1e-8)                                                 # lib/NonlinearSolveFirstOrder
    cache = init(prob, TrustRegion(; linsolve = KrylovJL_GMRES())); step!(cache); solve!(cache); reinit!(cache, u0; p)
    J = StatefulJacobianOperator(JacobianOperator(prob, fu, u), u, p);  J * v;  J' * v   # lib/SciMLJacobianOperators

Everything numerical happens in libmi355x_nk.so (HIP kernels); this file only marshals pointers. Vectors may
be NumPy arrays (host; copied per call) or torch CUDA tensors (device; zero-copy). torch is used for device
memory and streams only.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Callable, Optional, Sequence

import numpy as np

from . import _lib as L
from ._lib import NKError, check

try:  # torch is plumbing: device buffers + current stream
    import torch
except Exception:  # pragma: no cover
    torch = None


# ------------------------------------------------------------------------------------------- pointers
def _is_torch(x):
    return torch is not None and isinstance(x, torch.Tensor)


def _ptr(x, n: Optional[int] = None):
    """(address, memspace, keepalive) of a float64 vector."""
    if _is_torch(x):
        if x.dtype != torch.float64:
            raise TypeError("device vectors must be float64")
        if not x.is_contiguous():
            raise ValueError("device vectors must be contiguous")
        if n is not None and x.numel() != n:
            raise ValueError(f"expected {n} elements, got {x.numel()}")
        return C.c_void_p(x.data_ptr()), (L.DEVICE if x.is_cuda else L.HOST), x
    a = np.ascontiguousarray(x, dtype=np.float64)
    if n is not None and a.size != n:
        raise ValueError(f"expected {n} elements, got {a.size}")
    return C.c_void_p(a.ctypes.data), L.HOST, a


def _like(x, n: int):
    if _is_torch(x):
        return torch.empty(n, dtype=torch.float64, device=x.device)
    return np.empty(n, dtype=np.float64)


class _DevView:
    """Zero-copy torch view of a raw device pointer handed to a callback (via __cuda_array_interface__)."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


def _view(ptr, n):
    return torch.as_tensor(_DevView(ptr, n), device="cuda")


# ------------------------------------------------------------------------------------------- context
class Context:
    """nk_ctx: one HIP device + stream (+ communicator). One process per GPU."""

    def __init__(self, device: Optional[int] = None, stream: Optional[int] = None):
        lib = L.lib()
        if device is None:
            device = torch.cuda.current_device() if (torch is not None and torch.cuda.is_available()) else 0
        if stream is None and torch is not None and torch.cuda.is_available():
            stream = torch.cuda.current_stream(device).cuda_stream
        h = C.c_void_p()
        check(lib.nk_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(h)))
        self._h, self.device = h, int(device)
        self._keep = []

    def synchronize(self):
        check(L.lib().nk_ctx_synchronize(self._h))

    def set_deterministic(self, flag: bool = True):
        check(L.lib().nk_ctx_set_deterministic(self._h, int(bool(flag))))

    def set_halo_overlap(self, flag: bool = True):
        """Multi-rank CSR SpMV: halo exchange on a second stream, overlapped with the interior row blocks."""
        check(L.lib().nk_ctx_set_halo_overlap(self._h, int(bool(flag))))

    # -- per-kernel-family HIP-event timing (bench.py roofline evidence)
    def profile_enable(self, on: bool = True):
        check(L.lib().nk_ctx_profile_enable(self._h, int(bool(on))))

    def profile_report(self):
        out = {}
        for k in range(L.lib().nk_ctx_profile_kernel_count()):
            name, cnt, ms, by = C.c_char_p(), C.c_int64(), C.c_double(), C.c_double()
            check(L.lib().nk_ctx_profile_query(self._h, k, C.byref(name), C.byref(cnt), C.byref(ms), C.byref(by)))
            if cnt.value:
                out[name.value.decode()] = dict(launches=cnt.value, total_ms=ms.value, bytes=by.value,
                                                avg_us=1e3 * ms.value / cnt.value,
                                                gbps=by.value / (ms.value * 1e-3) / 1e9 if ms.value > 0 else 0.0)
        return out

    # -- communicator
    def comm_init_rccl(self, nranks: int, rank: int, unique_id: bytes):
        check(L.lib().nk_ctx_comm_init_rccl(self._h, nranks, rank, unique_id))

    def comm_init_callbacks(self, nranks: int, rank: int, allreduce: Callable, alltoallv: Callable):
        cb = L.CommCallbacks(L.ALLREDUCE_FN(allreduce), L.ALLTOALLV_FN(alltoallv), None)
        self._keep.append(cb)
        check(L.lib().nk_ctx_comm_init_callbacks(self._h, nranks, rank, C.byref(cb)))

    def comm_peer_handle(self, arena_bytes: int = 0) -> bytes:
        """Allocate this rank's uncached arena and return its 64-byte hipIpc handle (all-gather it, then comm_enable_peer)."""
        buf = C.create_string_buffer(64)
        check(L.lib().nk_ctx_comm_peer_handle(self._h, int(arena_bytes), buf))
        return buf.raw

    def comm_enable_peer(self, handles: bytes):
        check(L.lib().nk_ctx_comm_enable_peer(self._h, handles))

    def comm_peer_disable(self):
        check(L.lib().nk_ctx_comm_peer_disable(self._h))

    def comm_peer_selftest(self) -> bool:
        ok = C.c_int()
        check(L.lib().nk_ctx_comm_peer_selftest(self._h, C.byref(ok)))
        return bool(ok.value)

    def comm_peer_status(self):
        en, err = C.c_int(), C.c_int64()
        check(L.lib().nk_ctx_comm_peer_status(self._h, C.byref(en), C.byref(err)))
        return bool(en.value), int(err.value)

    def allreduce(self, x, op: str = "sum"):
        """in-place all-reduce of a float64 device tensor over the ranks, through the context's transport; returns x"""
        check(L.lib().nk_ctx_comm_allreduce(self._h, C.c_void_p(x.data_ptr()), int(x.numel()), {"sum": 0, "max": 1}[op]))
        return x

    def comm_info(self):
        k, n, r = C.c_int(), C.c_int(), C.c_int()
        check(L.lib().nk_ctx_comm_info(self._h, C.byref(k), C.byref(n), C.byref(r)))
        return k.value, n.value, r.value

    def comm_device_shared(self) -> bool:
        """several ranks of the communicator run on one device (processes time-slicing a GPU)"""
        s = C.c_int(0)
        check(L.lib().nk_ctx_comm_device_shared(self._h, C.byref(s)))
        return bool(s.value)

    def close(self):
        if self._h:
            L.lib().nk_ctx_destroy(self._h)
            self._h = None

    # -- BLAS-1 on device tensors
    def dot(self, x, y):
        r = C.c_double()
        check(L.lib().nk_dot(self._h, x.numel(), C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), C.byref(r)))
        return r.value

    def nrm2(self, x):
        r = C.c_double()
        check(L.lib().nk_nrm2(self._h, x.numel(), C.c_void_p(x.data_ptr()), C.byref(r)))
        return r.value

    def norm_inf(self, x):
        r = C.c_double()
        check(L.lib().nk_norm_inf(self._h, x.numel(), C.c_void_p(x.data_ptr()), C.byref(r)))
        return r.value

    def axpy(self, a, x, y):
        check(L.lib().nk_axpy(self._h, x.numel(), float(a), C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr())))

    def multidot(self, V, w):
        """V: (nv, ldv) row-major torch tensor = column-major n×nv basis; returns h[j] = V[j]·w."""
        nv, ldv = V.shape
        h = (C.c_double * nv)()
        check(L.lib().nk_multidot(self._h, w.numel(), nv, C.c_void_p(V.data_ptr()), ldv, C.c_void_p(w.data_ptr()), h))
        return np.array(h[:])

    def multiaxpy(self, V, h, w, want_norm2=False):
        nv, ldv = V.shape
        hh = (C.c_double * nv)(*[float(t) for t in h])
        r = C.c_double()
        check(L.lib().nk_multiaxpy(self._h, w.numel(), nv, C.c_void_p(V.data_ptr()), ldv, hh,
                                   C.c_void_p(w.data_ptr()), C.byref(r) if want_norm2 else None))
        return r.value if want_norm2 else None


def _fused_axpy_dot(self, V, h, s, w):
    """w -= V (h∘s); returns (h2, ‖w_new‖²) with h2[j] = s_j V[j]·w_new — the fused CGS2 pass."""
    nv, ldv = V.shape
    hh = (C.c_double * nv)(*[float(t) for t in h])
    ss = (C.c_double * nv)(*[float(t) for t in s])
    out = (C.c_double * (nv + 1))()
    check(L.lib().nk_fused_axpy_dot(self._h, w.numel(), nv, C.c_void_p(V.data_ptr()), ldv, hh, ss,
                                    C.c_void_p(w.data_ptr()), out))
    arr = np.array(out[:])
    return arr[:nv], float(arr[nv])


Context.fused_axpy_dot = _fused_axpy_dot

_default_ctx: Optional[Context] = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context()
    return _default_ctx


def set_default_context(ctx: Optional[Context]):
    global _default_ctx
    _default_ctx = ctx


def partition_range(n_global: int, granule: int, nranks: int, rank: int):
    b, e = C.c_int64(), C.c_int64()
    check(L.lib().nk_partition_range(n_global, granule, nranks, rank, C.byref(b), C.byref(e)))
    return b.value, e.value


def comm_unique_id() -> bytes:
    buf = C.create_string_buffer(128)
    check(L.lib().nk_comm_unique_id(buf))
    return buf.raw


# ------------------------------------------------------------------------------------------- CSR
class CSRMatrix:
    """nk_csr: row-partitioned CSR with f64 values / i32 indices on the device (SparseMatrixCSC stand-in)."""

    def __init__(self, handle, ctx: Context, owned=True):
        self._h, self.ctx, self._owned = handle, ctx, owned

    @classmethod
    def from_arrays(cls, rowptr, colind, vals=None, n_global=None, row_begin=0, index_base=0, ctx=None):
        ctx = ctx or default_context()
        rowptr = np.ascontiguousarray(rowptr)
        colind = np.ascontiguousarray(colind)
        if rowptr.dtype not in (np.int32, np.int64):
            rowptr = rowptr.astype(np.int64)
        colind = colind.astype(rowptr.dtype, copy=False)
        bits = 32 if rowptr.dtype == np.int32 else 64
        nrows = rowptr.size - 1
        nnz = int(colind.size)
        if n_global is None:
            n_global = nrows
        v = None if vals is None else np.ascontiguousarray(vals, dtype=np.float64)
        h = C.c_void_p()
        check(L.lib().nk_csr_create(ctx._h, nrows, n_global, row_begin, nnz, bits, index_base,
                                    C.c_void_p(rowptr.ctypes.data), C.c_void_p(colind.ctypes.data),
                                    None if v is None else C.c_void_p(v.ctypes.data), L.HOST, C.byref(h)))
        return cls(h, ctx)

    @classmethod
    def from_scipy(cls, A, ctx=None):
        A = A.tocsr()
        A.sort_indices()
        return cls.from_arrays(A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data, ctx=ctx)

    def set_values_csc(self, nzval):
        """New values in the CSC order of `from_csc` (Julia: `nonzeros(J)` after `f.jac(J, u, p)`): one gather on the device
        through the permutation remembered at creation — no new conversion, no new pattern upload."""
        if _is_torch(nzval) and nzval.is_cuda:
            check(L.lib().nk_csr_set_values_csc(self._h, C.c_void_p(nzval.data_ptr()), int(nzval.numel()), L.DEVICE))
        else:
            v = np.ascontiguousarray(nzval.cpu().numpy() if _is_torch(nzval) else nzval, dtype=np.float64)
            check(L.lib().nk_csr_set_values_csc(self._h, C.c_void_p(v.ctypes.data), int(v.size), L.HOST))
        return self

    @classmethod
    def from_csc(cls, colptr, rowval, nzval, index_base=1, ctx=None, row_range=None):
        """Julia's SparseMatrixCSC fields (1-based Int64 by default) of the whole matrix; on several ranks this rank keeps
        `row_range = (begin, end)` (default: the library's contiguous partition)."""
        ctx = ctx or default_context()
        colptr = np.ascontiguousarray(colptr, dtype=np.int64)
        rowval = np.ascontiguousarray(rowval, dtype=np.int64)
        nz = np.ascontiguousarray(nzval, dtype=np.float64)
        h = C.c_void_p()
        if row_range is None:
            check(L.lib().nk_csr_create_from_csc(ctx._h, colptr.size - 1, rowval.size, 64, index_base,
                                                 C.c_void_p(colptr.ctypes.data), C.c_void_p(rowval.ctypes.data),
                                                 C.c_void_p(nz.ctypes.data), C.byref(h)))
        else:
            check(L.lib().nk_csr_create_from_csc_rows(ctx._h, colptr.size - 1, rowval.size, 64, index_base,
                                                      C.c_void_p(colptr.ctypes.data), C.c_void_p(rowval.ctypes.data),
                                                      C.c_void_p(nz.ctypes.data), int(row_range[0]),
                                                      int(row_range[1] - row_range[0]), C.byref(h)))
        return cls(h, ctx)

    def info(self):
        a, b, c, d = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        check(L.lib().nk_csr_info(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return dict(nrows_local=a.value, n_global=b.value, nnz=c.value, n_halo=d.value)

    @property
    def shape(self):
        i = self.info()
        return (i["nrows_local"], i["n_global"])

    def set_values(self, vals):
        p, ms, _k = _ptr(vals, self.info()["nnz"])
        check(L.lib().nk_csr_set_values(self._h, p, ms))

    def values(self):
        out = np.empty(self.info()["nnz"])
        check(L.lib().nk_csr_get_values(self._h, C.c_void_p(out.ctypes.data), L.HOST))
        return out

    def values_device(self):
        return _view(L.lib().nk_csr_values_device(self._h), self.info()["nnz"])

    def matvec(self, x, out=None):
        n = self.info()["nrows_local"]
        px, ms, _k = _ptr(x, n)
        y = _like(x, n) if out is None else out
        py, ms2, _k2 = _ptr(y, n)
        if ms != ms2:
            raise ValueError("x and out must live in the same memory space")
        check(L.lib().nk_spmv(self._h, px, py, ms))
        return y

    def powers(self, x, s, theta=None, scale=1.0):
        """Matrix powers Y[:, p] = scale·(A − θ_p I) Y[:, p−1], Y[:, −1] = x (theta None: plain powers) — the s operator
        applications of an s-step Arnoldi block. Returns (Y as an (s, n) array whose row p is power p, resident) where
        `resident` says whether the one-launch kernel that holds the matrix in registers ran (csrc/nk_powers.hip)."""
        n = self.info()["nrows_local"]
        px, ms, _k = _ptr(x, n)
        Y = _like(x, n * s)
        pY, _ms2, _k2 = _ptr(Y, n * s)
        th = None if theta is None else np.ascontiguousarray(theta, dtype=np.float64)
        if th is not None and th.size != s:
            raise ValueError("theta must have s entries")
        res = C.c_int(0)
        check(L.lib().nk_csr_powers(self._h, px, pY, n, int(s), None if th is None else C.c_void_p(th.ctypes.data),
                                    float(scale), ms, C.byref(res)))
        return Y.reshape(s, n), bool(res.value)

    @staticmethod
    def powers_layout(A, num_cus: int = 256):
        """Which form of the resident matrix-powers kernel a SciPy CSR pattern fits on a device with `num_cus` compute units
        (host arithmetic only, nk_csr_powers_layout): dict(kind = 'none' | 'bands' | 'segments', slices, slots, bands, segments, ring)"""
        import numpy as _np
        A = A.tocsr()
        rp = _np.ascontiguousarray(A.indptr, dtype=_np.int32)
        ci = _np.ascontiguousarray(A.indices, dtype=_np.int32)
        out = (C.c_int * 6)()
        check(L.lib().nk_csr_powers_layout(int(A.shape[0]), C.c_void_p(rp.ctypes.data), C.c_void_p(ci.ctypes.data), int(num_cus), out))
        return {"kind": ("none", "bands", "segments")[out[0]], "slices": out[1], "slots": out[2], "bands": out[3],
                "segments": out[4], "ring": bool(out[5])}

    def rmatvec(self, x, out=None):
        n = self.info()["nrows_local"]
        px, ms, _k = _ptr(x, n)
        y = _like(x, n) if out is None else out
        py, _ms2, _k2 = _ptr(y, n)
        check(L.lib().nk_spmv_t(self._h, px, py, ms))
        return y

    def colsumsq(self, like=None):
        """diag(AᵀA): out_j = Σ_i A_ij² (LevenbergMarquardt's DᵀD source, levenberg_marquardt.jl:133-148)."""
        n = self.info()["nrows_local"]
        y = np.empty(n) if like is None else _like(like, n)
        py, ms, _k = _ptr(y, n)
        check(L.lib().nk_csr_colsumsq(self._h, py, ms))
        return y

    __matmul__ = matvec

    def close(self):
        if self._h and self._owned:
            L.lib().nk_csr_destroy(self._h)
        self._h = None


# ------------------------------------------------------------------------------------------- problems
class DeviceProblem:
    """nk_problem: residual / JVP / VJP / Jacobian provider living on the device."""

    def __init__(self, handle, ctx: Context, keep=()):
        self._h, self.ctx, self._keep = handle, ctx, list(keep)
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        check(L.lib().nk_problem_size(self._h, C.byref(a), C.byref(b), C.byref(c)))
        self.n_local, self.n_global, self.row_begin = a.value, b.value, c.value

    def residual(self, u):
        f = _like(u, self.n_local)
        pu, ms, _a = _ptr(u, self.n_local)
        pf, _m, _b = _ptr(f)
        check(L.lib().nk_residual(self._h, pu, pf, ms))
        return f

    def _jv(self, fn, u, v):
        out = _like(u, self.n_local)
        pu, ms, _a = _ptr(u, self.n_local)
        pv, ms2, _b = _ptr(v, self.n_local)
        po, _m, _c = _ptr(out)
        if ms != ms2:
            raise ValueError("u and v must live in the same memory space")
        check(fn(self._h, pu, pv, po, ms))
        return out

    def jvp(self, v, u):  # argument order of the reference: jvp(v, u, p)
        return self._jv(L.lib().nk_jvp, u, v)

    def vjp(self, v, u):
        return self._jv(L.lib().nk_vjp, u, v)

    def jac_csr(self) -> CSRMatrix:
        h = C.c_void_p()
        check(L.lib().nk_problem_jac_csr(self._h, C.byref(h)))
        return CSRMatrix(h, self.ctx)

    def jac_values(self, u, J: CSRMatrix, colored: bool = False):
        pu, ms, _a = _ptr(u, self.n_local)
        if colored:
            nc = C.c_int()
            check(L.lib().nk_jac_values_colored(self._h, pu, ms, J._h, C.byref(nc)))
            return nc.value
        check(L.lib().nk_jac_values(self._h, pu, ms, J._h))
        return None

    def initial_guess(self, device: bool = False):
        u0 = torch.empty(self.n_local, dtype=torch.float64, device=f"cuda:{self.ctx.device}") if device \
            else np.empty(self.n_local)
        p, ms, _a = _ptr(u0)
        check(L.lib().nk_problem_initial_guess(self._h, p, ms))
        return u0

    def set_params(self, params: Sequence[float]):
        arr = (C.c_double * len(params))(*[float(x) for x in params])
        check(L.lib().nk_problem_set_params(self._h, arr, len(params)))

    def close(self):
        if self._h:
            L.lib().nk_problem_destroy(self._h)
            self._h = None


def _builtin(kind, params, ctx):
    ctx = ctx or default_context()
    arr = (C.c_double * len(params))(*[float(x) for x in params])
    h = C.c_void_p()
    check(L.lib().nk_problem_create(ctx._h, kind, arr, len(params), C.byref(h)))
    return DeviceProblem(h, ctx)


def Quadratic(n: int, p: float = 2.0, ctx=None) -> DeviceProblem:
    """quadratic_f(u, p) = u .* u .- p   (common/common_rootfind_testing.jl:15-17)"""
    P = _builtin(L.PROBLEM_QUADRATIC, [n, p], ctx)
    P.params = [n, p]
    return P


def Bratu2D(n: int, lam: float = 6.0, scale: float = 0.0, ctx=None) -> DeviceProblem:
    """2-D Bratu, 5-point stencil; scale = 0 → h²-scaled residual."""
    P = _builtin(L.PROBLEM_BRATU2D, [n, lam, scale], ctx)
    P.params = [n, lam, scale]
    return P


def Brusselator2D(N: int, A=3.4, B=1.0, alpha=10.0, dx=None, ctx=None) -> DeviceProblem:
    """brusselator_2d_loop (lib/NonlinearSolveFirstOrder/test/sparsity_tests__item1.jl:13-36)"""
    dx = 1.0 / (N - 1) if dx is None else dx
    P = _builtin(L.PROBLEM_BRUSSELATOR2D, [N, A, B, alpha, dx], ctx)
    P.params = [N, A, B, alpha, dx]
    return P


@dataclass
class NonlinearFunction:
    """NonlinearFunction{true}(f!; jvp, vjp, jac, jac_prototype) with torch-tensor callbacks on the device:
    f(du, u, p), jvp(Jv, v, u, p), vjp(vJ, v, u, p), jac(nzval, u, p) (values of jac_prototype, a CSRMatrix)."""
    f: Callable
    jvp: Optional[Callable] = None
    vjp: Optional[Callable] = None
    jac: Optional[Callable] = None
    jac_prototype: Optional[CSRMatrix] = None
    mass_matrix: object = None   # picked up by PseudoTransient when the algorithm names none (number or diagonal vector)


class NonlinearProblem:
    """NonlinearProblem(f, u0, p). `f` is a built-in DeviceProblem or a NonlinearFunction of torch callbacks.
    (NonlinearLeastSquaresProblem below is the same container with `least_squares = True`.)"""
    least_squares = False

    def __init__(self, f, u0=None, p=None, ctx: Optional[Context] = None):
        self.p = p
        if isinstance(f, DeviceProblem):
            self.device_problem = f
            self.ctx = f.ctx
            self.u0 = f.initial_guess() if u0 is None else u0
            if p is not None:
                self.set_p(p)
            return
        if not isinstance(f, NonlinearFunction):
            f = NonlinearFunction(f)
        if u0 is None:
            raise ValueError("u0 is required for a NonlinearFunction problem")
        self.ctx = ctx or default_context()
        self.u0 = u0
        n = int(u0.numel() if _is_torch(u0) else np.asarray(u0).size)
        prob = self

        def wrap(fn, nargs):
            if fn is None:
                return None

            def cb(*a):
                try:
                    with torch.cuda.stream(torch.cuda.ExternalStream(a[-1])) if a[-1] else _nullctx():
                        tensors = [_view(q, n) for q in a[1:-1]]
                        if nargs == 2:      # residual(u, f) → f!(du, u, p)
                            fn(tensors[1], tensors[0], prob.p)
                        elif nargs == 3:    # (v, u, out) → jvp!(out, v, u, p)
                            fn(tensors[2], tensors[0], tensors[1], prob.p)
                    return 0
                except Exception as e:  # pragma: no cover - surfaced as NK_E_CALLBACK
                    import traceback
                    traceback.print_exc()
                    return 1
            return cb

        def jac_cb(user, u_ptr, vals_ptr, stream):
            try:
                nnz = f.jac_prototype.info()["nnz"]
                # the fill must be ordered against the library's SpMV / LU kernels: run it on the library's stream
                with torch.cuda.stream(torch.cuda.ExternalStream(stream)) if stream else _nullctx():
                    f.jac(_view(vals_ptr, nnz), _view(u_ptr, n), prob.p)
                return 0
            except Exception:  # pragma: no cover
                import traceback
                traceback.print_exc()
                return 1

        cbs = L.UserCallbacks(
            L.RESIDUAL_FN(wrap(f.f, 2)),
            L.JVP_FN(wrap(f.jvp, 3)) if f.jvp else L.JVP_FN(),
            L.JVP_FN(wrap(f.vjp, 3)) if f.vjp else L.JVP_FN(),
            L.JACVALS_FN(jac_cb) if (f.jac and f.jac_prototype is not None) else L.JACVALS_FN(),
        )
        h = C.c_void_p()
        check(L.lib().nk_problem_create_user(self.ctx._h, n, n, 0, C.byref(cbs), None,
                                             f.jac_prototype._h if f.jac_prototype is not None else None,
                                             C.byref(h)))
        self.device_problem = DeviceProblem(h, self.ctx, keep=[cbs, f])
        self.f = f

    def set_p(self, p):
        self.p = p
        dp = self.device_problem
        if hasattr(dp, "params"):
            params = list(dp.params)
            if isinstance(p, (int, float)):
                params[1] = float(p)           # quadratic p / Bratu λ
            else:
                params[1:1 + len(p)] = [float(x) for x in p]
            dp.params = params
            dp.set_params(params)


class _nullctx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


# ------------------------------------------------------------------------------------------- algorithms
@dataclass
class ChebyshevPrecs:
    """`precs` for KrylovJL_GMRES: right preconditioner M⁻¹ = p_d(A), d Chebyshev steps on [λmax/ratio, λmax]
    (λmax bounded/estimated by the library for every new Jacobian). Operator applications only."""
    degree: int = 16
    ratio: float = 100.0


@dataclass
class MultigridPrecs:
    """`precs` for KrylovJL_GMRES on Bratu2D problems: one geometric multigrid V-cycle as the right preconditioner
    (nu Chebyshev smoothing steps, coarsest grid side ≤ coarse_max, banded LU there) — the device counterpart of an
    algebraic-multigrid `precs` (docs/src/tutorials/large_systems.md:244-316). Mesh-independent iteration counts."""
    nu: int = 2
    coarse_max: int = 31


@dataclass
class LinearSolveParameters:
    """lib/NonlinearSolveBase/src/linear_solve.jl:1-4: what `precs(A, p)` receives as p — the current iterate u (a device
    tensor) and the nonlinear problem's parameters p."""
    u: object
    p: object


class Preconditioner:
    """nk_precond: a preconditioner object built from a CSRMatrix on the device — what a `precs(A, p)` returns for a general
    sparse Jacobian (docs/src/tutorials/large_systems.md:252-316 fills the slot with IncompleteLU.ilu / an algebraic
    multigrid). Usable as Pl or Pr of the device GMRES and standalone (`apply`)."""

    def __init__(self, handle, A):
        self._h, self.A, self.n = handle, A, None

    def update(self):
        """refactorise for the matrix's current values (`precs` is re-evaluated for every new A)"""
        check(L.lib().nk_precond_update(self._h))
        return self

    def apply(self, x):
        """y = M⁻¹ x (device tensor in → device tensor out; NumPy in → NumPy out)"""
        if _is_torch(x):
            y = torch.empty_like(x)
            check(L.lib().nk_precond_apply(self._h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()),
                                           L.DEVICE if x.is_cuda else L.HOST))
            return y
        xa = np.ascontiguousarray(x, dtype=np.float64)
        ya = np.empty_like(xa)
        check(L.lib().nk_precond_apply(self._h, C.c_void_p(xa.ctypes.data), C.c_void_p(ya.ctypes.data), L.HOST))
        return ya

    __call__ = apply

    def info(self):
        k, ll, lu, nc = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        check(L.lib().nk_precond_info(self._h, C.byref(k), C.byref(ll), C.byref(lu), C.byref(nc)))
        return dict(kind={1: "jacobi", 2: "ilu0", 3: "amg", 4: "ilut"}[k.value], levels_lower=ll.value, levels_upper=lu.value, ncolors=nc.value)

    def close(self):
        if self._h:
            L.lib().nk_precond_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


class JacobiPreconditioner(Preconditioner):
    """M = diag(A)"""

    def __init__(self, A: "CSRMatrix"):
        h = C.c_void_p()
        check(L.lib().nk_precond_create_jacobi(A._h, C.byref(h)))
        super().__init__(h, A)


class ILU0Preconditioner(Preconditioner):
    """A ≈ LU on the pattern of A (no fill, no pivoting) of the rank's local block, level-scheduled on the device.
    ordering = "natural": the classical preconditioner in the matrix's ordering (a lexicographic stencil is a dependency chain
    of 2n − 1 levels: milliseconds per application at n = 1024²); "multicolor": rows permuted by a greedy colouring (as many
    levels as colours — the GPU form)."""

    def __init__(self, A: "CSRMatrix", ordering: str = "multicolor"):
        h = C.c_void_p()
        check(L.lib().nk_precond_create_ilu0(A._h, {"natural": 0, "multicolor": 1}[ordering], C.byref(h)))
        super().__init__(h, A)
        self.ordering = ordering

    def factors(self):
        """(L, U, perm) as SciPy CSR matrices in the permuted ordering: perm[permuted row] = original row."""
        import scipy.sparse as sp
        nnz = C.c_int64(0)
        check(L.lib().nk_precond_ilu0_factors(self._h, C.byref(nnz), None, None, None, None))
        n = self.A.shape[0]
        rp, ci = np.zeros(n + 1, dtype=np.int32), np.zeros(nnz.value, dtype=np.int32)
        v, perm = np.zeros(nnz.value), np.zeros(n, dtype=np.int32)
        check(L.lib().nk_precond_ilu0_factors(self._h, None, C.c_void_p(rp.ctypes.data), C.c_void_p(ci.ctypes.data),
                                              C.c_void_p(v.ctypes.data), C.c_void_p(perm.ctypes.data)))
        M = sp.csr_matrix((v, ci, rp), shape=(n, n))
        return sp.tril(M, -1).tocsr() + sp.identity(n, format="csr"), sp.triu(M, 0).tocsr(), perm


class ILUTPreconditioner(ILU0Preconditioner):
    """Crout ILU with the drop tolerance `tau` — the tutorial's `incompletelu(W, p) = (ilu(W, τ = 50.0), I)`
    (docs/src/tutorials/large_systems.md:252-260): A ≈ (I + L) U with fill, an entry kept if its magnitude before the division by
    the pivot is ≥ tau. Factorised on the host for every `update()` (the pattern depends on the numbers; the reference's
    IncompleteLU.jl runs on the CPU as well), applied on the device by two level-scheduled triangular solves."""

    def __init__(self, A: "CSRMatrix", tau: float):
        h = C.c_void_p()
        check(L.lib().nk_precond_create_ilut(A._h, float(tau), C.byref(h)))
        Preconditioner.__init__(self, h, A)
        self.ordering, self.tau = "natural", float(tau)


class AMGPreconditioner(Preconditioner):
    """Aggregation algebraic multigrid built from the matrix alone (csrc/nk_amg.hip) — what the reference's tutorial returns from
    `precs(A, p)` as Pl for a large sparse Jacobian (`aspreconditioner(ruge_stuben(A))`, docs/src/tutorials/large_systems.md:276-316):
    pairwise aggregation (aggregates of ≤ 2^passes rows, formed on the device at creation), Galerkin coarse matrices refreshed on the device by
    `update()`, ν Chebyshev steps on D⁻¹A per side, over-corrected coarse correction, dense inverse on the ≤ 128 coarsest rows."""

    def __init__(self, A: "CSRMatrix", nu: int = 0, passes: int = 0, theta: float = 0.0, overcorrection: float = 0.0,
                 cheb_ratio: float = 0.0, coarse_max: int = 0, matching: str = "auto"):
        # matching: "auto" (the device set-up's handshaking on one rank, the host's sequential pass on a rank's local block),
        # "greedy" (host) or "handshake" (device)
        mt = {"auto": 0, "greedy": 1, "handshake": 2}[matching]
        prm = L.AMGParams(int(nu), int(passes), int(coarse_max), mt, float(theta), float(overcorrection), float(cheb_ratio))
        h = C.c_void_p()
        check(L.lib().nk_precond_create_amg(A._h, C.byref(prm), C.byref(h)))
        super().__init__(h, A)

    def hierarchy(self):
        """[(rows, non-zeros, Gershgorin bound of D⁻¹A)] per level, finest first, coarsest last"""
        nl = C.c_int(0)
        check(L.lib().nk_precond_amg_info(self._h, C.byref(nl), 0, None, None, None))
        n, z, lm = (C.c_int64 * nl.value)(), (C.c_int64 * nl.value)(), (C.c_double * nl.value)()
        check(L.lib().nk_precond_amg_info(self._h, C.byref(nl), nl.value, n, z, lm))
        return [(int(n[i]), int(z[i]), float(lm[i])) for i in range(nl.value)]

    @property
    def matching(self) -> str:
        """how the aggregates were formed: "greedy" (sequential pairwise pass, host set-up) or "handshake" (device set-up)"""
        m = C.c_int(0)
        check(L.lib().nk_precond_amg_matching(self._h, C.byref(m)))
        return {1: "greedy", 2: "handshake"}[m.value]

    def aggregates(self, level: int):
        """row → coarse row of `level` (NumPy int32)"""
        rows = self.hierarchy()[level][0]
        out = np.empty(rows, dtype=np.int32)
        check(L.lib().nk_precond_amg_aggregates(self._h, int(level), C.c_void_p(out.ctypes.data), rows))
        return out


@dataclass
class ObjectPrecs:
    """`precs` through nk_options: a built-in object (kind = "jacobi" | "ilu0" | "ilu0_natural" | "amg") on the concrete Jacobian,
    refactorised inside the solver for every new J — no callback into the host language. side = "left" as the reference's
    tutorial precs (`(Pl, I)`), or "right"."""
    kind: str = "ilu0"
    side: str = "left"


@dataclass
class KrylovJL_GMRES:
    """LinearSolve.KrylovJL_GMRES stand-in executed by the device GMRES (protocol: SURVEY.md §8d)."""
    gmres_restart: int = 30
    maxiters: int = 300
    ortho: str = "sstep"       # "mgs" (Krylov.jl's structure) | "cgs2" | "dcgs2" (= cgs2, delayed 2nd pass) | "cgs" | "sstep" (the library default)
    fixed_iters: int = 0
    sstep: int = 0             # ortho = "sstep": basis columns per block (1..16); 0 = automatic (15 Newton basis, 6 monomial)
    sstep_basis: str = "auto"  # "auto" (Newton where the spectrum can be bounded) | "monomial" | "newton"
    abstol: Optional[float] = None   # None → the nonlinear tolerances are forwarded (FirstOrder/src/solve.jl:203)
    reltol: Optional[float] = None
    # `precs`: a callable precs(A, p::LinearSolveParameters) -> (Pl, Pr) as in the reference (re-evaluated for every new A;
    # Pl / Pr: None (identity), a Preconditioner object, or a callable y = P⁻¹ x on device tensors), or one of the built-in
    # descriptors ChebyshevPrecs / MultigridPrecs (right) / ObjectPrecs (either side)
    precs: object = None


@dataclass
class EisenstatWalkerForcing2:  # eisenstat_walker.jl:18-29
    eta0: float = 0.5
    eta_max: float = 0.9
    gamma: float = 0.9
    alpha: float = 2.0
    safeguard: bool = True
    safeguard_threshold: float = 0.1


@dataclass
class BackTracking:
    """LineSearch.jl / LineSearches.jl BackTracking [EXT]: `NewtonRaphson(linesearch = BackTracking())`."""
    c_1: float = 1e-4
    rho_hi: float = 0.5
    rho_lo: float = 0.1
    order: int = 3
    maxiters: int = 1000


@dataclass
class LineSearchesJL:
    """LineSearch.jl's wrapper around LineSearches.jl [EXT]: `NewtonRaphson(linesearch = LineSearchesJL(; method = …))` with
    method "Static" | "BackTracking" | "StrongWolfe" | "MoreThuente" | "HagerZhang" (LineSearches.jl default parameters) — the methods of lib/NonlinearSolveFirstOrder/test/rootfind_tests__item2.jl:40-46."""
    method: str = "BackTracking"


class RadiusUpdateSchemes:  # trust_region.jl:59-147
    Simple, NLsolve, NocedalWright, Hei, Yuan, Bastin, Fan = range(7)


@dataclass
class NewtonRaphson:  # raphson.jl:30-43
    linsolve: Optional[KrylovJL_GMRES] = None
    forcing: Optional[EisenstatWalkerForcing2] = None
    concrete_jac: Optional[bool] = None
    linesearch: Optional[BackTracking] = None   # `missing` in the reference = no line search
    jac_colored: bool = False
    name: str = "NewtonRaphson"


@dataclass
class TrustRegion:  # trust_region.jl:25-43
    linsolve: Optional[KrylovJL_GMRES] = None
    radius_update_scheme: int = RadiusUpdateSchemes.Simple
    max_trust_radius: float = 0.0
    initial_trust_radius: float = 0.0
    step_threshold: float = 1.0 / 10000
    shrink_threshold: float = 1.0 / 4
    expand_threshold: float = 3.0 / 4
    shrink_factor: float = 1.0 / 4
    expand_factor: float = 2.0
    max_shrink_times: int = 32
    concrete_jac: Optional[bool] = None
    jac_colored: bool = False   # concrete J by colour-compressed assembly (sparse AD analogue) instead of closed-form values
    name: str = "TrustRegion"


@dataclass
class GaussNewton:  # gauss_newton.jl:11-23: NewtonDescent; on a least-squares problem with a Krylov linsolve: normal form
    linsolve: Optional[KrylovJL_GMRES] = None
    concrete_jac: Optional[bool] = None
    linesearch: Optional[BackTracking] = None
    forcing: Optional[EisenstatWalkerForcing2] = None
    name: str = "GaussNewton"


@dataclass
class LevenbergMarquardt:  # levenberg_marquardt.jl:37-64 (keyword names of the reference; α_geodesic spelt alpha_geodesic)
    linsolve: Optional[KrylovJL_GMRES] = None
    damping_initial: float = 1.0
    alpha_geodesic: float = 0.75
    disable_geodesic: bool = False
    damping_increase_factor: float = 2.0
    damping_decrease_factor: float = 3.0
    finite_diff_step_geodesic: float = 0.1
    b_uphill: float = 1.0
    min_damping_D: float = 1e-8
    concrete_jac: Optional[bool] = True   # concrete_jac = Val(true) in the reference's constructor
    jac_colored: bool = False
    name: str = "LevenbergMarquardt"


@dataclass
class PseudoTransient:  # pseudo_transient.jl:37-57
    """mass_matrix: None / 1.0 (identity), a number λ (λ·I), or a vector m of the local length of u (Diagonal(m)); the
    damping term is α⁻¹·M. A general (non-diagonal) M would change the sparsity pattern of the damped J — not on this path."""
    linsolve: Optional[KrylovJL_GMRES] = None
    alpha_initial: float = 1e-3
    mass_matrix: object = None
    concrete_jac: Optional[bool] = None
    linesearch: Optional[BackTracking] = None
    jac_colored: bool = False
    name: str = "PseudoTransient"


@dataclass
class _TerminationMode:
    """SciMLBase termination modes (lib/NonlinearSolveBase/src/termination_conditions.jl); `internalnorm` is
    "inf" (Base.Fix1(maximum, abs), the reference default) or "l2"."""
    internalnorm: str = "inf"
    patience_steps: int = 100
    patience_objective_multiplier: float = 3.0
    min_max_factor: float = 1.3
    max_stalled_steps: Optional[int] = None
    protective_threshold: Optional[float] = None
    code = 0


class AbsNormSafeBestTerminationMode(_TerminationMode):
    code = 0


class NormTerminationMode(_TerminationMode):
    code = 1


class RelTerminationMode(_TerminationMode):
    code = 2


class RelNormTerminationMode(_TerminationMode):
    code = 3


class RelNormSafeTerminationMode(_TerminationMode):
    code = 4


class RelNormSafeBestTerminationMode(_TerminationMode):
    code = 5


class AbsTerminationMode(_TerminationMode):
    code = 6


class AbsNormTerminationMode(_TerminationMode):
    code = 7


class AbsNormSafeTerminationMode(_TerminationMode):
    code = 8


TERMINATION_CONDITIONS = [  # common/common_rootfind_testing.jl:3-13
    NormTerminationMode, RelTerminationMode, RelNormTerminationMode, RelNormSafeTerminationMode,
    RelNormSafeBestTerminationMode, AbsTerminationMode, AbsNormTerminationMode, AbsNormSafeTerminationMode,
    AbsNormSafeBestTerminationMode,
]

_SS_BASIS = {"auto": 0, "monomial": 1, "newton": 2}
_ORTHO = {"mgs": L.ORTHO_MGS, "cgs2": L.ORTHO_CGS2, "cgs": L.ORTHO_CGS, "dcgs2": L.ORTHO_DCGS2, "dcgs2_1r": L.ORTHO_DCGS2_1R,
          "sstep": L.ORTHO_SSTEP}


def _options(alg, abstol, reltol, maxiters, maxtime, store_trace, termination_kwargs,
             termination_condition=None, least_squares=False) -> L.Options:
    o = L.Options()
    check(L.lib().nk_options_default(C.byref(o)))
    ls = alg.linsolve
    o.algorithm = L.ALG_TRUST_REGION if isinstance(alg, TrustRegion) else L.ALG_NEWTON_RAPHSON
    if least_squares:
        o.termination_norm = 1  # default_termination_mode(::NonlinearLeastSquaresProblem): AbsNormSafeBest on the 2-norm
    if isinstance(alg, GaussNewton):   # (linsolve = None: a factorising solver takes J δ = f as it is — no normal form)
        o.algorithm = L.ALG_GAUSS_NEWTON
        o.termination_norm = 1  # (Gauss–Newton is defined on least-squares problems only)
    if isinstance(alg, PseudoTransient):
        o.algorithm = L.ALG_PSEUDO_TRANSIENT
        o.pt_alpha_initial = float(alg.alpha_initial)
    if isinstance(alg, LevenbergMarquardt):   # (linsolve = None: JᵀJ + λDᵀD is assembled and factorised on the device)
        o.algorithm = L.ALG_LEVENBERG_MARQUARDT
        o.lm_disable_geodesic = int(bool(alg.disable_geodesic))
        o.lm_damping_initial = float(alg.damping_initial)
        o.lm_damping_increase_factor = float(alg.damping_increase_factor)
        o.lm_damping_decrease_factor = float(alg.damping_decrease_factor)
        o.lm_min_damping_D = float(alg.min_damping_D)
        o.lm_alpha_geodesic = float(alg.alpha_geodesic)
        o.lm_finite_diff_step_geodesic = float(alg.finite_diff_step_geodesic)
        o.lm_b_uphill = float(alg.b_uphill)
    if ls is None:
        # linsolve = nothing: LinearSolve's default factorisation of the concrete sparse J → banded LU on device
        o.linsolve = L.LINSOLVE_BANDED_LU
        ls = KrylovJL_GMRES()  # unused Krylov fields keep their defaults
    else:
        o.linsolve = L.LINSOLVE_GMRES_CSR if (alg.concrete_jac or isinstance(alg, LevenbergMarquardt)) \
            else L.LINSOLVE_GMRES_MATFREE
    o.maxiters = int(maxiters)
    o.abstol = 0.0 if abstol is None else float(abstol)
    o.reltol = 0.0 if reltol is None else float(reltol)
    o.maxtime = 0.0 if maxtime is None else float(maxtime)
    o.gmres_restart, o.gmres_maxiters = int(ls.gmres_restart), int(ls.maxiters)
    o.gmres_ortho, o.gmres_fixed_iters = _ORTHO[ls.ortho], int(ls.fixed_iters)
    o.gmres_sstep = int(getattr(ls, "sstep", 0))
    o.gmres_sstep_basis = _SS_BASIS[getattr(ls, "sstep_basis", "auto")]
    o.lin_abstol = -1.0 if ls.abstol is None else float(ls.abstol)
    o.lin_reltol = -1.0 if ls.reltol is None else float(ls.reltol)
    lsr = getattr(alg, "linesearch", None)
    if isinstance(lsr, LineSearchesJL):
        o.linesearch = {"BackTracking": 1, "Static": 2, "StrongWolfe": 3, "MoreThuente": 4, "HagerZhang": 5}[lsr.method]
    elif lsr is not None:
        o.linesearch = 1
        o.ls_c1, o.ls_rho_hi, o.ls_rho_lo = float(lsr.c_1), float(lsr.rho_hi), float(lsr.rho_lo)
        o.ls_order, o.ls_maxiters = int(lsr.order), int(lsr.maxiters)
    precs = getattr(ls, "precs", None)
    if isinstance(precs, MultigridPrecs):
        o.mg_nu, o.mg_coarse = int(precs.nu), int(precs.coarse_max)
    elif isinstance(precs, ChebyshevPrecs):
        o.cheb_degree, o.cheb_ratio = int(precs.degree), float(precs.ratio)
    elif isinstance(precs, ObjectPrecs):
        o.precond_kind = {"jacobi": 1, "ilu0_natural": 2, "ilu0": 3, "ilu0_multicolor": 3, "amg": 4}[precs.kind]
        o.precond_side = {"right": 0, "left": 1}[precs.side]
    elif precs is not None and not callable(precs):
        raise TypeError("KrylovJL_GMRES(precs=…): a callable precs(A, p) -> (Pl, Pr) or a built-in descriptor")
    o.jac_colored = int(bool(getattr(alg, "jac_colored", False)))
    fo = getattr(alg, "forcing", None)
    if fo is not None:
        o.forcing = L.FORCING_EW2
        o.ew_eta0, o.ew_eta_max, o.ew_gamma, o.ew_alpha = fo.eta0, fo.eta_max, fo.gamma, fo.alpha
        o.ew_safeguard, o.ew_safeguard_threshold = int(fo.safeguard), fo.safeguard_threshold
    if isinstance(alg, TrustRegion):
        o.radius_update_scheme = int(alg.radius_update_scheme)
        o.max_shrink_times = int(alg.max_shrink_times)
        for k in ("max_trust_radius", "initial_trust_radius", "step_threshold", "shrink_threshold",
                  "expand_threshold", "shrink_factor", "expand_factor"):
            setattr(o, k, float(getattr(alg, k)))
    if termination_condition is not None:
        tc = termination_condition() if isinstance(termination_condition, type) else termination_condition
        o.termination_mode = tc.code
        o.termination_norm = {"inf": 0, "l2": 1}[tc.internalnorm]
        o.patience_steps = int(tc.patience_steps)
        o.patience_objective_multiplier = float(tc.patience_objective_multiplier)
        o.min_max_factor = float(tc.min_max_factor)
        # an explicitly constructed mode has max_stalled_steps = nothing unless given (the default *mode* of the
        # solver is built with max_stalled_steps = 32, termination_conditions.jl:385-389)
        o.max_stalled_steps = -1 if tc.max_stalled_steps is None else int(tc.max_stalled_steps)
        o.protective_threshold = 0.0 if tc.protective_threshold is None else float(tc.protective_threshold)
    for k, v in (termination_kwargs or {}).items():
        setattr(o, k, v)
    o.store_trace = int(bool(store_trace))
    return o


@dataclass
class NLStats:
    nf: int = 0
    njacs: int = 0
    nfactors: int = 0
    nsolve: int = 0
    nsteps: int = 0
    gmres_iters: int = 0
    op_applies: int = 0
    allreduces: int = 0
    halo_exchanges: int = 0


@dataclass
class NonlinearSolution:
    u: object
    resid: object
    retcode: str
    stats: NLStats
    trace: list = field(default_factory=list)
    alg: object = None

    @property
    def successful_retcode(self):  # SciMLBase.successful_retcode
        return self.retcode == "Success"


class FirstOrderCache:
    """GeneralizedFirstOrderAlgorithmCache behind nk_solver: init / step! / solve! / reinit!."""

    def __init__(self, prob: NonlinearProblem, alg, abstol=None, reltol=None, maxiters=1000, maxtime=None,
                 store_trace=False, termination_kwargs=None, termination_condition=None):
        self.prob, self.alg = prob, alg
        self._opts = _options(alg, abstol, reltol, maxiters, maxtime, store_trace, termination_kwargs,
                              termination_condition, least_squares=getattr(prob, "least_squares", False))
        self._u0_is_torch = _is_torch(prob.u0)
        p, ms, _k = _ptr(prob.u0, prob.device_problem.n_local)
        h = C.c_void_p()
        check(L.lib().nk_solver_init(prob.device_problem._h, p, ms, C.byref(self._opts), C.byref(h)))
        self._h = h
        self.n = prob.device_problem.n_local
        self._install_precs_hook()
        M = getattr(alg, "mass_matrix", None)
        if M is None and isinstance(alg, PseudoTransient):   # resolve_ser_mass_matrix(::Nothing, prob): fall back to prob.f's
            M = getattr(getattr(prob, "f", None), "mass_matrix", None)
        if M is not None:
            if np.ndim(M) == 0:
                M = None if float(M) == 1.0 else np.full(self.n, float(M))       # resolve_ser_mass_matrix: I ≡ nothing
            elif np.ndim(M) != 1:
                raise NKError("PseudoTransient(mass_matrix=…): only the identity, λ·I and Diagonal(m) run on the device "
                              "(a general M changes the sparsity pattern of J + α⁻¹ M)")
            elif len(M) != self.n:  # DimensionMismatch in the reference (pseudo_transient.jl:110-118)
                raise NKError(f"mass matrix has {len(M)} diagonal entries but the problem has {self.n} local unknowns")
            if M is not None:
                pm, msm, self._mass_keep = _ptr(M, self.n)
                check(L.lib().nk_solver_set_mass_matrix_diagonal(self._h, pm, msm))

    def _out(self, fn):
        if self._u0_is_torch and self.prob.u0.is_cuda:
            out = torch.empty(self.n, dtype=torch.float64, device=self.prob.u0.device)
        else:
            out = np.empty(self.n)
        p, ms, _k = _ptr(out)
        check(fn(self._h, p, ms))
        return out

    @property
    def u(self):
        return self._out(L.lib().nk_solver_get_u)

    @property
    def fu(self):
        return self._out(L.lib().nk_solver_get_resid)

    @property
    def stats(self) -> NLStats:
        s = L.Stats()
        check(L.lib().nk_solver_get_stats(self._h, C.byref(s)))
        return NLStats(**s.as_dict())

    def _ret(self):
        r, n, f = C.c_int(), C.c_int(), C.c_int()
        check(L.lib().nk_solver_get_retcode(self._h, C.byref(r), C.byref(n), C.byref(f)))
        return r.value, n.value, bool(f.value)

    @property
    def retcode(self):
        return L.RET_NAMES[self._ret()[0]]

    @property
    def nsteps(self):
        return self._ret()[1]

    @property
    def force_stop(self):
        return self._ret()[2]

    @property
    def trust_region(self):
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        check(L.lib().nk_solver_get_scalars(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return b.value

    @property
    def eta(self):
        """the Eisenstat–Walker forcing cache's η (`cache.forcing_cache.η`, eisenstat_walker.jl:32-39)"""
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        check(L.lib().nk_solver_get_scalars(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return c.value

    @property
    def fnorm_inf(self):
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        check(L.lib().nk_solver_get_scalars(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value

    @property
    def trace(self):
        n = C.c_int()
        check(L.lib().nk_solver_get_trace(self._h, None, 0, C.byref(n)))
        rows = (L.TraceEntry * max(n.value, 1))()
        check(L.lib().nk_solver_get_trace(self._h, rows, n.value, C.byref(n)))
        return [dict(iter=r.iter, gmres_iters=r.gmres_iters, accepted=bool(r.accepted), fnorm_inf=r.fnorm_inf,
                     step_norm2=r.step_norm2, eta=r.eta, trust_region=r.trust_region, rho=r.rho)
                for r in rows[: n.value]]

    def step(self, recompute_jacobian: Optional[bool] = None, evaluate_residual: bool = True):
        """step!(cache; recompute_jacobian = nothing, evaluate_residual = true)"""
        rj = -1 if recompute_jacobian is None else int(bool(recompute_jacobian))
        check(L.lib().nk_solver_step_ex(self._h, rj, int(bool(evaluate_residual))))
        return self

    def supports_deferred_residual(self) -> bool:
        y = C.c_int()
        check(L.lib().nk_solver_supports_deferred_residual(self._h, C.byref(y)))
        return bool(y.value)

    def refresh_residual(self):
        """refresh_residual!(cache): settle a residual evaluation deferred by step(evaluate_residual=False)."""
        check(L.lib().nk_solver_refresh_residual(self._h))
        return None

    def _install_precs_hook(self):
        """KrylovJL_GMRES(precs = callable): `precs(A, LinearSolveParameters(u, p)) -> (Pl, Pr)` is evaluated when the
        linear cache is built and again for every new Jacobian (never by reinit!) — the protocol test/Core/
        core_tests__item21.jl pins through its call counts. A = the concrete Jacobian (a CSRMatrix view of the solver's J),
        or the StatefulJacobianOperator of the matrix-free path."""
        ls = getattr(self.alg, "linsolve", None)
        precs = getattr(ls, "precs", None)
        if precs is None or isinstance(precs, (ChebyshevPrecs, MultigridPrecs, ObjectPrecs)) or not callable(precs):
            return
        n, prob = self.n, self.prob
        self._precs_keep = []

        def hook(user, gh, ah, uptr):
            try:
                u = _view(uptr, n)
                if ah:
                    A = CSRMatrix(C.c_void_p(ah), prob.ctx, owned=False)
                else:
                    A = StatefulJacobianOperator(JacobianOperator(prob), u)
                out = precs(A, LinearSolveParameters(u, prob.p))
                Pl, Pr = out if isinstance(out, tuple) else (out, None)
                keep = []
                _install_preconditioner(C.c_void_p(gh), "left", Pl, n, keep)
                _install_preconditioner(C.c_void_p(gh), "right", Pr, n, keep)
                self._precs_keep = keep      # (the previous pair may be released now: the GMRES object holds the new one)
                return 0
            except Exception:  # pragma: no cover
                import traceback
                traceback.print_exc()
                return 1
        self._precs_fn = L.PRECS_FN(hook)
        check(L.lib().nk_solver_set_precs(self._h, self._precs_fn, None))

    def solve(self) -> NonlinearSolution:
        r = C.c_int()
        check(L.lib().nk_solver_solve(self._h, C.byref(r)))
        return NonlinearSolution(self.u, self.fu, L.RET_NAMES[r.value], self.stats,
                                 self.trace if self._opts.store_trace else [], self.alg)

    def reinit(self, u0=None, p=None):
        params, npar = None, 0
        if p is not None:
            self.prob.set_p(p)  # built-ins push the new parameters to the device problem
        if u0 is not None:
            pu, ms, _k = _ptr(u0, self.n)
        else:
            pu, ms = None, L.HOST
        check(L.lib().nk_solver_reinit(self._h, pu, ms, params, npar))
        return self

    def close(self):
        if self._h:
            L.lib().nk_solver_destroy(self._h)
            self._h = None


class NonlinearLeastSquaresProblem(NonlinearProblem):
    """NonlinearLeastSquaresProblem(f, u0, p): residual count = unknown count on this path (row-partitioned square J). What the
    type changes is what the reference derives from it: the default termination mode measures the residual in the 2-norm
    (default_termination_mode(::NonlinearLeastSquaresProblem)) for EVERY algorithm, and a polyalgorithm's best-of fallback ranks
    its sub-solvers by the 2-norm (findmin_resids, NonlinearSolveBase/src/polyalg.jl)."""
    least_squares = True


def init(prob: NonlinearProblem, alg, **kw):
    from . import polyalg
    if isinstance(alg, polyalg.NonlinearSolvePolyAlgorithm):
        return polyalg.PolyAlgorithmCache(prob, alg, least_squares=getattr(prob, "least_squares", False), **kw)
    return FirstOrderCache(prob, alg, **kw)


def solve(prob: NonlinearProblem, alg, **kw) -> NonlinearSolution:
    from . import polyalg
    if isinstance(alg, polyalg.NonlinearSolvePolyAlgorithm):
        return polyalg.polysolve(prob, alg, least_squares=getattr(prob, "least_squares", False), **kw)
    cache = FirstOrderCache(prob, alg, **kw)
    try:
        return cache.solve()
    finally:
        cache.close()


def step_(cache: FirstOrderCache, **kw):   # step!(cache; recompute_jacobian, evaluate_residual)
    return cache.step(**kw)


def supports_deferred_residual(cache: FirstOrderCache) -> bool:
    return cache.supports_deferred_residual()


def refresh_residual(cache: FirstOrderCache):   # refresh_residual!(cache)
    return cache.refresh_residual()


def solve_(cache: FirstOrderCache):  # solve!(cache)
    return cache.solve()


def reinit_(cache, u0=None, p=None, **kw):  # reinit!(cache, u0; p[, retain_best])
    return cache.reinit(u0, p, **kw)


# ------------------------------------------------------------------------------------------- linear solve seam
def _install_preconditioner(gh, side: str, M, n: int, keep: list):
    """Pl or Pr on the nk_gmres `gh`: None / identity removes that side; a Preconditioner object goes in as it is (no host
    round trip per application); any other callable y = P⁻¹ x on device tensors becomes a device callback."""
    lib = L.lib()
    left = side == "left"
    if M is None or M is IDENTITY:
        check(lib.nk_gmres_set_preconditioner(gh, 1 if left else 0, None))
        return
    if isinstance(M, Preconditioner):
        keep.append(M)
        check(lib.nk_gmres_set_preconditioner(gh, 1 if left else 0, M._h))
        return
    if not callable(M):
        raise TypeError(f"a preconditioner must be None, a Preconditioner or a callable, not {type(M)}")

    def cb(user, x, y, stream):
        try:
            _view(y, n).copy_(M(_view(x, n)))
            return 0
        except Exception:  # pragma: no cover
            import traceback
            traceback.print_exc()
            return 1
    fn = L.MATVEC_FN(cb)
    keep.append(fn)
    check((lib.nk_gmres_set_left_preconditioner if left else lib.nk_gmres_set_right_preconditioner)(gh, fn, None))


class _Identity:
    """LinearAlgebra.I as a preconditioner (what the reference's DummyPreconditioners returns)"""

    def __call__(self, x):
        return x

    def __repr__(self):
        return "I"


IDENTITY = _Identity()


class GMRES:
    """nk_gmres: the LinearCache analogue NonlinearSolveBase drives (A, b, u, reltol; solve!)."""

    def __init__(self, n: int, restart: int = 30, ortho: str = "sstep", ctx: Optional[Context] = None, sstep: int = 0,
                 sstep_basis: str = "auto"):
        self.ctx = ctx or default_context()
        h = C.c_void_p()
        check(L.lib().nk_gmres_create(self.ctx._h, n, restart, _ORTHO[ortho], C.byref(h)))
        if ortho == "sstep":
            check(L.lib().nk_gmres_set_block_size(h, int(sstep)))
            check(L.lib().nk_gmres_set_sstep_basis(h, _SS_BASIS[sstep_basis]))
        self._h, self.n, self._keep = h, n, []
        self.abstol, self.reltol, self.maxiters = 0.0, 1e-8, 300

    def set_operator(self, A, u=None):
        """A: CSRMatrix (concrete J), StatefulJacobianOperator (matrix-free), or callable y = A(x) on tensors."""
        self._keep = [A]
        if isinstance(A, CSRMatrix):
            check(L.lib().nk_gmres_set_operator_csr(self._h, A._h))
        elif isinstance(A, StatefulJacobianOperator):
            if A.mode != "jvp":
                raise NotImplementedError("GMRES on a transposed operator")
            p, ms, k = _ptr(A.u, self.n)
            self._keep.append(k)
            check(L.lib().nk_gmres_set_operator_jvp(self._h, A.jac_op.prob.device_problem._h, p, ms))
        elif callable(A):
            n = self.n

            def cb(user, x, y, stream):
                try:
                    out = A(_view(x, n))
                    _view(y, n).copy_(out)
                    return 0
                except Exception:  # pragma: no cover
                    import traceback
                    traceback.print_exc()
                    return 1
            fn = L.MATVEC_FN(cb)
            self._keep.append(fn)
            check(L.lib().nk_gmres_set_operator_fn(self._h, fn, None))
        else:
            raise TypeError(f"unsupported operator {type(A)}")
        return self

    def set_right_preconditioner(self, M: Optional[Callable]):
        """precs hook (test/Core/core_tests__item21.jl): M(x) ≈ A⁻¹ x on device tensors."""
        if M is None:
            check(L.lib().nk_gmres_set_right_preconditioner(self._h, L.MATVEC_FN(), None))
            return self
        n = self.n

        def cb(user, x, y, stream):
            try:
                _view(y, n).copy_(M(_view(x, n)))
                return 0
            except Exception:  # pragma: no cover
                import traceback
                traceback.print_exc()
                return 1
        fn = L.MATVEC_FN(cb)
        self._keep.append(fn)
        check(L.lib().nk_gmres_set_right_preconditioner(self._h, fn, None))
        return self

    def set_left_preconditioner(self, M):
        """Pl of `precs(A, p) -> (Pl, Pr)`: GMRES runs on Pl⁻¹ A Pr⁻¹ and stops on the preconditioned residual. M: None, a
        Preconditioner object, or a callable y = Pl⁻¹ x on device tensors."""
        _install_preconditioner(self._h, "left", M, self.n, self._keep)
        return self

    def set_preconditioner(self, P, side: str = "left"):
        """a Preconditioner object (JacobiPreconditioner / ILU0Preconditioner) on either side"""
        _install_preconditioner(self._h, side, P, self.n, self._keep)
        return self

    def set_chebyshev_preconditioner(self, degree: int, lambda_min: float = 0.0, lambda_max: float = 0.0,
                                     ratio: float = 30.0):
        check(L.lib().nk_gmres_set_chebyshev_preconditioner(self._h, int(degree), float(lambda_min), float(lambda_max),
                                                            float(ratio)))
        return self

    def set_multigrid_preconditioner(self, problem, u, nu: int = 2, coarse_max: int = 31):
        """One V-cycle of the built-in geometric multigrid (Bratu2D problems) as the right preconditioner, linearised at u."""
        dp = problem.device_problem if hasattr(problem, "device_problem") else problem
        pu, ms, keep = _ptr(u, self.n)
        self._mg_u = (u, keep)  # the hierarchy keeps reading the fine-level u: keep it alive
        check(L.lib().nk_gmres_set_multigrid_preconditioner(self._h, dp._h, pu, ms, int(nu), int(coarse_max)))
        return self

    def set_spectrum_interval(self, lo: float, hi: float):
        """Real bounds of the operator's spectrum for operators the library cannot bound itself (callbacks, matrix-free):
        they place the Newton-basis shifts of the s-step Arnoldi process. lo = hi = 0 forgets them."""
        check(L.lib().nk_gmres_set_spectrum_interval(self._h, float(lo), float(hi)))

    def sstep_state(self):
        """(block size in effect, Newton basis in the last solve, blocks that lost rank so far) of the s-step process"""
        bs, nb, bd = C.c_int(0), C.c_int(0), C.c_int(0)
        check(L.lib().nk_gmres_get_sstep_state(self._h, C.byref(bs), C.byref(nb), C.byref(bd)))
        return bs.value, bool(nb.value), bd.value

    def chebyshev_interval(self):
        a, b = C.c_double(), C.c_double()
        check(L.lib().nk_gmres_get_chebyshev_interval(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def update_tolerances(self, abstol=None, reltol=None):  # LinearSolve.update_tolerances!
        if abstol is not None:
            self.abstol = float(abstol)
        if reltol is not None:
            self.reltol = float(reltol)

    def solve(self, b, x0=None, abstol=None, reltol=None, maxiters=None, fixed_iters=0):
        x = _like(b, self.n)
        use_x0 = 0
        if x0 is not None:
            if _is_torch(x):
                x.copy_(x0)
            else:
                x[...] = x0
            use_x0 = 1
        pb, ms, _a = _ptr(b, self.n)
        px, _m, _b = _ptr(x)
        info = L.GmresInfo()
        check(L.lib().nk_gmres_solve(self._h, pb, px, ms, use_x0,
                                     self.abstol if abstol is None else abstol,
                                     self.reltol if reltol is None else reltol,
                                     self.maxiters if maxiters is None else maxiters, fixed_iters, C.byref(info)))
        return x, dict(iters=info.iters, restarts=info.restarts, converged=bool(info.converged),
                       failed=bool(info.failed), rnorm0=info.rnorm0, rnorm=info.rnorm)

    def close(self):
        if self._h:
            L.lib().nk_gmres_destroy(self._h)
            self._h = None


class BandedLU:
    """nk_lu: direct factorisation of a concrete sparse J (the LinearSolve factorisation cache analogue)."""

    def __init__(self, A: CSRMatrix):
        h = C.c_void_p()
        check(L.lib().nk_lu_create(A._h, C.byref(h)))
        self._h, self.A, self.n = h, A, A.info()["nrows_local"]
        self.factor()

    def factor(self, A: Optional[CSRMatrix] = None):
        ok = C.c_int()
        check(L.lib().nk_lu_factor(self._h, (A or self.A)._h, C.byref(ok)))
        if not ok.value:
            raise NKError("banded LU: zero or non-finite pivot")
        return self

    def solve(self, b):
        x = _like(b, self.n)
        pb, ms, _a = _ptr(b, self.n)
        px, _m, _b = _ptr(x)
        check(L.lib().nk_lu_solve(self._h, pb, px, ms))
        return x

    def info(self):
        kl, ku, by = C.c_int(), C.c_int(), C.c_int64()
        check(L.lib().nk_lu_info(self._h, C.byref(kl), C.byref(ku), C.byref(by)))
        e, blk, lev = C.c_int(), C.c_int(), C.c_int()
        check(L.lib().nk_lu_engine(self._h, C.byref(e), C.byref(blk), C.byref(lev)))
        return dict(kl=kl.value, ku=ku.value, band_bytes=by.value, engine=("band_lu", "block_cyclic_reduction")[e.value],
                    block=blk.value, levels=lev.value)

    def close(self):
        if self._h:
            L.lib().nk_lu_destroy(self._h)
            self._h = None


# ------------------------------------------------------------------------------------------- ensembles of small systems
@dataclass
class SimpleNewtonRaphson:
    """lib/SimpleNonlinearSolve/src/raphson.jl:23-25 — the algorithm the reference allows inside GPU kernels.
    `autodiff=None` = AutoForwardDiff (dual numbers in the kernel); `jac=True` uses the `nk_jac` of the source."""
    jac: bool = False
    name: str = "SimpleNewtonRaphson"


@dataclass
class SimpleTrustRegion:
    """lib/SimpleNonlinearSolve/src/trust_region.jl:44-55 (default radius update rule); None = reference default."""
    jac: bool = False
    step_threshold: float = 1e-4
    shrink_threshold: Optional[float] = None   # 0.25
    expand_threshold: Optional[float] = None   # 0.75
    shrink_factor: Optional[float] = None      # 0.25
    expand_factor: float = 2.0
    max_shrink_times: int = 32
    name: str = "SimpleTrustRegion"


class ImmutableNonlinearProblem:
    """SciMLBase.ImmutableNonlinearProblem{false}(f, u0, p) for the kernel-generation path
    (docs/src/tutorials/nonlinear_solve_gpus.md:120-140). `f_source` is HIP C++ defining
    `template <typename T> __device__ void nk_f(const T *u, const double *p, T *f)`; `u0` has n entries (shared by
    every system) or shape (nbatch, n); `p` has shape (nbatch, nparams) — one parameter set per system."""

    def __init__(self, f_source: str, u0, p, ctx: Optional[Context] = None):
        self.f_source, self.ctx = f_source, ctx or default_context()
        self.u0, self.p = u0, p
        pshape = tuple(p.shape)
        if len(pshape) != 2:
            raise ValueError("p must have shape (nbatch, nparams)")
        self.nbatch, self.nparams = int(pshape[0]), int(pshape[1])
        ushape = tuple(u0.shape)
        self.n = int(ushape[-1])
        self.u0_per_system = len(ushape) == 2
        if self.u0_per_system and ushape[0] != self.nbatch:
            raise ValueError("u0 and p disagree on the number of systems")


@dataclass
class EnsembleSolution:
    u: object          # (nbatch, n)
    resid: object      # (nbatch, n): residual at the last evaluated iterate (reference: `fx` returned by check_termination)
    retcode: np.ndarray  # (nbatch,) strings
    iters: np.ndarray    # (nbatch,)
    retcode_raw: np.ndarray = None


class _BatchKernel:
    _cache: dict = {}

    @classmethod
    def get(cls, ctx, source, n, nparams, flags):
        key = (id(ctx), source, n, nparams, flags)
        if key not in cls._cache:
            h = C.c_void_p()
            check(L.lib().nk_batch_create(ctx._h, source.encode(), n, nparams, flags, C.byref(h)))
            cls._cache[key] = h
        return cls._cache[key]


def vectorized_solve(prob: ImmutableNonlinearProblem, alg=None, abstol=None, maxiters=1000):
    """`vectorized_solve(prob, alg; backend = ROCBackend())` of the tutorial (nonlinear_solve_gpus.md:106-114): solve
    every parameter set with SimpleNewtonRaphson, one system per GPU thread, in one kernel launch."""
    alg = alg or SimpleNewtonRaphson()
    h = _BatchKernel.get(prob.ctx, prob.f_source, prob.n, prob.nparams, 1 if alg.jac else 0)
    on_dev = _is_torch(prob.p) and prob.p.is_cuda
    nb, n = prob.nbatch, prob.n
    if on_dev:
        u0 = prob.u0 if (_is_torch(prob.u0) and prob.u0.is_cuda) else torch.as_tensor(np.asarray(prob.u0), device=prob.p.device)
        u0 = u0.to(torch.float64).contiguous()
        pp = prob.p.to(torch.float64).contiguous()
        u = torch.empty((nb, n), dtype=torch.float64, device=pp.device)
        r = torch.empty_like(u)
        rc = torch.empty(nb, dtype=torch.int32, device=pp.device)
        it = torch.empty(nb, dtype=torch.int32, device=pp.device)
        ptr = lambda x: C.c_void_p(x.data_ptr())
        ms = L.DEVICE
    else:
        u0 = np.ascontiguousarray(np.asarray(prob.u0.cpu() if _is_torch(prob.u0) else prob.u0), dtype=np.float64)
        pp = np.ascontiguousarray(np.asarray(prob.p.cpu() if _is_torch(prob.p) else prob.p), dtype=np.float64)
        u, r = np.empty((nb, n)), np.empty((nb, n))
        rc, it = np.empty(nb, dtype=np.int32), np.empty(nb, dtype=np.int32)
        ptr = lambda x: C.c_void_p(x.ctypes.data)
        ms = L.HOST
    if isinstance(alg, SimpleTrustRegion):
        d = lambda v: -1.0 if v is None else float(v)
        check(L.lib().nk_batch_solve_trust_region(h, nb, ptr(u0), 1 if prob.u0_per_system else 0, ptr(pp), ms,
                                                  0.0 if abstol is None else float(abstol), int(maxiters),
                                                  d(alg.step_threshold), d(alg.shrink_threshold), d(alg.expand_threshold),
                                                  d(alg.shrink_factor), d(alg.expand_factor), int(alg.max_shrink_times),
                                                  ptr(u), ptr(r), ptr(rc), ptr(it)))
    else:
        check(L.lib().nk_batch_solve(h, nb, ptr(u0), 1 if prob.u0_per_system else 0, ptr(pp), ms,
                                     0.0 if abstol is None else float(abstol), int(maxiters), ptr(u), ptr(r), ptr(rc), ptr(it)))
    rch = rc.cpu().numpy() if on_dev else rc
    ith = it.cpu().numpy() if on_dev else it
    return EnsembleSolution(u, r, np.asarray(L.RET_NAMES)[rch], ith.copy(), rch.copy())


# ------------------------------------------------------------------------------------------- Jacobian operators
class JacobianOperator:
    """JacobianOperator(prob, fu, u): JVP by default, `.T` flips to VJP
    (lib/SciMLJacobianOperators/src/SciMLJacobianOperators.jl:86-143)."""

    def __init__(self, prob: NonlinearProblem, fu=None, u=None, mode: str = "jvp"):
        self.prob, self.mode = prob, mode
        n = prob.device_problem.n_local
        self.size = (n, n)

    @property
    def T(self):
        return JacobianOperator(self.prob, mode="vjp" if self.mode == "jvp" else "jvp")

    adjoint = transpose = T

    def __call__(self, v, u, p=None):  # (op::JacobianOperator)(v, u, p)
        dp = self.prob.device_problem
        return dp.jvp(v, u) if self.mode == "jvp" else dp.vjp(v, u)


def JacVecOperator(prob, fu=None, u=None):
    return JacobianOperator(prob, fu, u, "jvp")


def VecJacOperator(prob, fu=None, u=None):
    return JacobianOperator(prob, fu, u, "vjp")


class StatefulJacobianOperator:
    """StatefulJacobianOperator(jac_op, u, p) with `*` / mul! (SciMLJacobianOperators.jl:210-243)."""

    def __init__(self, jac_op: JacobianOperator, u, p=None):
        self.jac_op, self.u, self.p, self.mode = jac_op, u, p, jac_op.mode

    @property
    def T(self):
        return StatefulJacobianOperator(self.jac_op.T, self.u, self.p)

    def __matmul__(self, v):
        if isinstance(v, StatefulJacobianOperator):
            return StatefulJacobianNormalFormOperator(self, v)
        return self.jac_op(v, self.u, self.p)

    __mul__ = __matmul__

    def mul_(self, out, v):  # mul!(Jv, J, v)
        r = self @ v
        if _is_torch(out):
            out.copy_(r)
        else:
            out[...] = r
        return out


class StatefulJacobianNormalFormOperator:
    """JᵀJ x via JVP then VJP (SciMLJacobianOperators.jl:252-291)."""

    def __init__(self, vjp_op, jvp_op):
        self.vjp_operator, self.jvp_operator = vjp_op, jvp_op

    def __matmul__(self, x):
        return self.vjp_operator @ (self.jvp_operator @ x)

    __mul__ = __matmul__
