"""GPU parity tests of the HIP kernels against the CPU oracle (through the C ABI).
Tolerances: SpMV/JVP/VJP ‖y−y_ref‖∞ ≤ 1e-13‖y_ref‖∞; reductions ≤ 1e-13 relative."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import reference_restatement as R
from oracle import c_oracle as CO

pytestmark = pytest.mark.gpu

RTOL = 1e-13


def relerr(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def test_library_loaded(nls):
    from nonlinearsolve_jl_amd import _lib
    assert _lib.lib().nk_version().decode().startswith("mi355x_nk")


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 1000, 4097, 1 << 20])
def test_blas1(nls, dev, n):
    import torch
    ctx = nls.default_context()
    rng = np.random.default_rng(n)
    x, y = rng.standard_normal(n), rng.standard_normal(n)
    dx, dy = torch.tensor(x, device=dev), torch.tensor(y, device=dev)
    assert abs(ctx.dot(dx, dy) - np.dot(x, y)) <= 1e-12 * max(1.0, np.abs(x * y).sum())
    assert abs(ctx.nrm2(dx) - np.linalg.norm(x)) <= 1e-13 * np.linalg.norm(x)
    assert ctx.norm_inf(dx) == np.max(np.abs(x))
    ctx.axpy(-0.75, dx, dy)
    assert relerr(dy.cpu().numpy(), y - 0.75 * x) <= RTOL


def test_norm_inf_nan(nls, dev):
    import torch
    x = torch.ones(5000, dtype=torch.float64, device=dev)
    x[1234] = float("nan")
    assert np.isnan(nls.default_context().norm_inf(x))


@pytest.mark.parametrize("n,nv", [(7, 1), (1000, 3), (4096, 8), (100003, 17), (1 << 18, 31)])
def test_multidot_multiaxpy(nls, dev, n, nv):
    import torch
    ctx = nls.default_context()
    rng = np.random.default_rng(nv)
    ldv = (n + 31) // 32 * 32
    V = np.zeros((nv, ldv))
    V[:, :n] = rng.standard_normal((nv, n))
    w = rng.standard_normal(n)
    dV, dw = torch.tensor(V, device=dev), torch.tensor(w, device=dev)
    h = ctx.multidot(dV, dw)
    href = V[:, :n] @ w
    assert np.max(np.abs(h - href)) <= 1e-12 * np.sqrt(n)
    nrm2 = ctx.multiaxpy(dV, h, dw, want_norm2=True)
    wref = w - V[:, :n].T @ h
    assert relerr(dw.cpu().numpy(), wref) <= 1e-12
    assert abs(nrm2 - wref @ wref) <= 1e-12 * (wref @ wref)


def _random_csr(n, density, seed):
    A = sp.random(n, n, density=density, format="csr", random_state=seed, dtype=np.float64)
    A = A + sp.identity(n, format="csr")
    A.sort_indices()
    return sp.csr_matrix(A)


@pytest.mark.parametrize("ns", [3, 32, 100, 257])
def test_spmv_bratu_bit_exact(nls, dev, ns):
    """Row sums are accumulated in CSR order from LDS ⇒ bit-identical to the sequential CPU row sum."""
    import torch
    p = R.Bratu2D(ns)
    rng = np.random.default_rng(ns)
    u = rng.standard_normal(p.n) * 0.1
    J = p.jac(u)
    A = nls.CSRMatrix.from_scipy(J)
    x = rng.standard_normal(p.n)
    yref = CO.spmv(J.indptr.astype(np.int32), J.indices.astype(np.int32), J.data, x)
    y = A.matvec(x)
    assert np.array_equal(y, yref)
    yd = A.matvec(torch.tensor(x, device=dev))
    assert np.array_equal(yd.cpu().numpy(), yref)
    yt = A.rmatvec(x)
    assert relerr(yt, J.T @ x) <= RTOL


@pytest.mark.parametrize("n,density", [(1, 1.0), (50, 0.3), (3000, 0.01), (20000, 0.0005)])
def test_spmv_random(nls, n, density):
    A = _random_csr(n, density, 7)
    x = np.random.default_rng(1).standard_normal(n)
    M = nls.CSRMatrix.from_scipy(A)
    assert relerr(M.matvec(x), A @ x) <= RTOL
    assert relerr(M.rmatvec(x), A.T @ x) <= RTOL


def test_spmv_long_rows_and_empty_rows(nls):
    """ragged input: empty rows, and rows longer than one LDS tile (2048 nnz)."""
    n = 6000
    rng = np.random.default_rng(3)
    rows = [np.array([], dtype=np.int64)] * n
    rows[0] = np.arange(n)                  # dense row (long-row path)
    rows[17] = np.sort(rng.choice(n, 2049, replace=False))
    rows[18] = np.sort(rng.choice(n, 2048, replace=False))
    for r in range(100, 200):
        rows[r] = np.sort(rng.choice(n, 5, replace=False))
    indptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
    indices = np.concatenate(rows).astype(np.int64)
    data = rng.standard_normal(indices.size)
    A = sp.csr_matrix((data, indices, indptr), shape=(n, n))
    x = rng.standard_normal(n)
    M = nls.CSRMatrix.from_arrays(indptr, indices, data)
    assert relerr(M.matvec(x), A @ x) <= 1e-12


def test_csc_ingest_julia_layout(nls):
    """SparseMatrixCSC{Float64,Int64} fields, 1-based, straight from Julia."""
    A = _random_csr(300, 0.05, 11)
    Ac = A.tocsc()
    M = nls.CSRMatrix.from_csc(Ac.indptr + 1, Ac.indices + 1, Ac.data, index_base=1)
    x = np.random.default_rng(5).standard_normal(300)
    assert relerr(M.matvec(x), A @ x) <= RTOL


@pytest.mark.parametrize("ns", [4, 33, 128])
def test_bratu_kernels(nls, dev, ns):
    import torch
    p = R.Bratu2D(ns, 6.0)
    P = nls.Bratu2D(ns, 6.0)
    rng = np.random.default_rng(ns)
    u, v = rng.standard_normal(p.n) * 0.3, rng.standard_normal(p.n)
    assert relerr(P.residual(u), CO.bratu_residual(ns, 6.0, 0.0, u)) <= RTOL
    assert relerr(P.jvp(v, u), CO.bratu_jvp(ns, 6.0, 0.0, u, v)) <= RTOL
    assert relerr(P.vjp(v, u), p.vjp(v, u)) <= RTOL
    du, dv = torch.tensor(u, device=dev), torch.tensor(v, device=dev)
    assert relerr(P.jvp(dv, du).cpu().numpy(), p.jvp(v, u)) <= RTOL
    J = P.jac_csr()
    P.jac_values(u, J)
    rp, ci = CO.bratu_pattern(ns)
    assert relerr(J.values(), CO.bratu_jac_values(ns, 6.0, 0.0, u, rp)) <= RTOL
    assert relerr(J.matvec(v), p.jac(u) @ v) <= RTOL
    # colour-compressed assembly (AutoSparse structure) reproduces the closed-form fill
    J2 = P.jac_csr()
    ncol = P.jac_values(u, J2, colored=True)
    assert ncol <= 7
    assert relerr(J2.values(), J.values()) <= 1e-12
    assert np.all(P.initial_guess() == 0.0)


@pytest.mark.parametrize("N", [3, 8, 32])
def test_brusselator_kernels(nls, N):
    b = R.Brusselator2D(N)
    P = nls.Brusselator2D(N)
    u0 = P.initial_guess()
    assert relerr(u0, b.u0()) <= 1e-14
    rng = np.random.default_rng(N)
    u, v = b.u0() + 0.1 * rng.standard_normal(b.n), rng.standard_normal(b.n)
    assert relerr(P.residual(u), CO.brusselator_residual(N, 3.4, 1.0, 10.0, 1.0 / (N - 1), u)) <= RTOL
    assert relerr(P.residual(u), b.f(u)) <= RTOL
    assert relerr(P.jvp(v, u), b.jvp(v, u)) <= RTOL
    assert relerr(P.vjp(v, u), b.vjp(v, u)) <= RTOL
    J = P.jac_csr()
    P.jac_values(u, J)
    assert J.info()["nnz"] == 6 * b.n
    assert relerr(J.matvec(v), b.jac(u) @ v) <= 1e-12
    assert relerr(J.rmatvec(v), b.jac(u).T @ v) <= 1e-12


def test_quadratic_kernels(nls):
    P = nls.Quadratic(1000, 2.0)
    u = np.linspace(0.5, 2.0, 1000)
    v = np.cos(np.arange(1000.0))
    assert relerr(P.residual(u), u * u - 2.0) <= RTOL
    assert relerr(P.jvp(v, u), 2 * u * v) <= RTOL
    assert np.all(P.initial_guess() == 1.0)


def test_against_committed_golden_fixtures(nls):
    """tests/golden/oracle_golden.npz (made by tests/golden/make_golden.py from the oracle, cross-checked with SciPy)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_golden.npz"))
    P = nls.Bratu2D(16, 6.0)
    assert relerr(P.jvp(g["v256"], g["bratu16_u"]), g["bratu16_Jv"]) <= 1e-13
    J = P.jac_csr()
    P.jac_values(g["bratu16_u"], J)
    assert relerr(J.matvec(g["v256"]), g["bratu16_Jv"]) <= 1e-13
    assert np.max(np.abs(P.residual(g["bratu16_u"]))) <= 1e-10          # the golden u is a root
    sol = nls.solve(nls.NonlinearProblem(P), nls.NewtonRaphson(), abstol=1e-10, maxiters=50)
    assert np.max(np.abs(sol.u - g["bratu16_u"])) <= 1e-11
    B = nls.Brusselator2D(8)
    assert relerr(B.initial_guess(), g["brus8_u0"]) <= 1e-14
    assert relerr(B.residual(g["brus8_u0"]), g["brus8_f0"]) <= 1e-13
    solb = nls.solve(nls.NonlinearProblem(B), nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(maxiters=4000, reltol=1e-12, abstol=0.0)),
                     abstol=1e-10, maxiters=30)
    assert solb.retcode == "Success" and np.max(np.abs(solb.u - g["brus8_u"])) <= 1e-8
