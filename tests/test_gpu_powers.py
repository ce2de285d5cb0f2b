"""Resident matrix-powers kernel (csrc/nk_powers.hip) — the s operator applications of an s-step Arnoldi block in ONE launch,
the matrix held in registers, neighbouring row bands handing their boundary rows over through write-through stores and flags.

Parity bar: BIT-IDENTICAL to s applications of the sequential CPU row sum of the C oracle (`CO.spmv`, the same bar as the
streaming SpMV, tests/test_gpu_kernels.py::test_spmv_bratu_bit_exact) with the Newton-basis epilogue
y = scale·(A x − θ x) formed as the streaming kernel forms it — every word of every column is compared, repeatedly (the
hand-off is a race if it is wrong: MI355X_MICROARCH.md asks for tests under uneven load that check every word)."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import reference_restatement as R
from oracle import c_oracle as CO

pytestmark = pytest.mark.gpu


def _powers_ref(J, x, s, theta, scale):
    """s sequential applications with the oracle's CSR-order row sum; epilogue as nk_csr.hip::spmv_store_row (mode 3 / 0)."""
    ip, ix, dv = J.indptr.astype(np.int32), J.indices.astype(np.int32), J.data
    out = np.empty((s, x.size))
    cur = x
    for p in range(s):
        y = CO.spmv(ip, ix, dv, cur)
        if theta is not None:
            y = y - theta[p] * cur
        y = scale * y
        out[p] = y
        cur = y
    return out


def _banded(n, half_bw, per_row, seed):
    """random banded matrix: 1 … per_row entries per row (ragged) within ±half_bw of the diagonal, diagonal always stored"""
    rng = np.random.default_rng(seed)
    r = np.repeat(np.arange(n, dtype=np.int64), per_row)
    off = rng.integers(-half_bw, half_bw + 1, size=n * per_row)
    off[::per_row] = 0
    c = np.clip(r + off, 0, n - 1)
    keep = rng.random(n * per_row) < 0.7
    keep[::per_row] = True
    A = sp.csr_matrix((rng.standard_normal(int(keep.sum())) * 0.3, (r[keep], c[keep])), shape=(n, n))
    A.sum_duplicates()
    A.sort_indices()
    return A


@pytest.mark.parametrize("ns,s", [(8, 3), (64, 15), (200, 15), (257, 7), (512, 15), (1024, 15)])
def test_resident_powers_bratu_bit_exact(nls, dev, ns, s):
    """Bratu Jacobians from 64 rows (one band) to config C3's 1024² (256 bands × 4096 rows, the headline matrix)."""
    import torch
    p = R.Bratu2D(ns)
    rng = np.random.default_rng(ns)
    J = p.jac(rng.standard_normal(p.n) * 0.1)
    A = nls.CSRMatrix.from_scipy(J)
    x = rng.standard_normal(p.n)
    lam = float(abs(J).sum(axis=1).max())
    theta = (0.5 + 0.4 * np.cos(np.arange(s))) * lam
    scale = 2.0 / lam
    ref = _powers_ref(J, x, s, theta, scale)
    dx = torch.tensor(x, device=dev)
    for rep in range(3):   # repeated launches: the flag epochs must keep the launches apart
        Y, resident = A.powers(dx, s, theta=theta, scale=scale)
        assert resident, "Bratu Jacobians up to 1024² are the matrices this kernel is for"
        assert np.array_equal(Y.cpu().numpy(), ref), f"rep {rep}: first differing power " \
            f"{int(np.argmax((Y.cpu().numpy() != ref).any(axis=1)))}"
    Yh, resident = A.powers(x, 4, theta=None, scale=1.0)   # host vectors, plain powers
    assert resident and np.array_equal(Yh, _powers_ref(J, x, 4, None, 1.0))


@pytest.mark.parametrize("n,hbw,per_row", [(5000, 3, 5), (70000, 1000, 5), (300000, 1024, 8), (150000, 700, 16),
                                            (1048576, 1024, 5), (1300000, 900, 4), (500000, 1000, 5), (523000, 1024, 3),
                                            (1040000, 17, 5)])
def test_resident_powers_random_banded(nls, dev, n, hbw, per_row):
    """Ragged rows (1 … W entries), partial last band, every register layout (1 / 2 / 4 / 6 rows per thread; 5 / 8 / 16 slots)."""
    import torch
    A = _banded(n, hbw, per_row, seed=n % 97)
    M = nls.CSRMatrix.from_scipy(A)
    x = np.random.default_rng(5).standard_normal(n)
    s = 6
    theta = np.linspace(-0.3, 0.4, s)
    ref = _powers_ref(A, x, s, theta, 0.7)
    Y, resident = M.powers(torch.tensor(x, device=dev), s, theta=theta, scale=0.7)
    assert resident
    assert np.array_equal(Y.cpu().numpy(), ref)


def _periodic_laplacian(N, seed):
    """5-point Laplacian on an N × N torus with a random positive diagonal shift: banded only on a RING of bands"""
    rng = np.random.default_rng(seed)
    idx = np.arange(N * N).reshape(N, N)
    rows, cols, vals = [], [], []
    for dj, di, w in ((0, 0, 4.0), (0, 1, -1.0), (0, -1, -1.0), (1, 0, -1.0), (-1, 0, -1.0)):
        rows.append(idx.ravel())
        cols.append(np.roll(np.roll(idx, -dj, axis=0), -di, axis=1).ravel())
        vals.append(np.full(N * N, w) + (rng.random(N * N) if (dj, di) == (0, 0) else 0.0))
    A = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(N * N, N * N))
    A.sort_indices()
    return A


def _two_segments(M, hbw, seed):
    """n = 2 M rows: two banded segments (≤ 3 entries within ±hbw, ragged) coupled row by row (r ↔ r ± M, and a few rows away)"""
    rng = np.random.default_rng(seed)
    blocks = []
    for sgm in range(2):
        r = np.repeat(np.arange(M, dtype=np.int64), 3)
        off = rng.integers(-hbw, hbw + 1, size=3 * M)
        off[::3] = 0
        c = np.clip(r + off, 0, M - 1)
        keep = rng.random(3 * M) < 0.75
        keep[::3] = True
        blocks.append(sp.csr_matrix((rng.standard_normal(int(keep.sum())) * 0.3, (r[keep], c[keep])), shape=(M, M)))
    cpl = [sp.diags(rng.standard_normal(M) * 0.2, 0, shape=(M, M)).tocsr() for _ in range(2)]
    cpl[0] = cpl[0] + sp.diags(rng.standard_normal(M - 3) * 0.1, 3, shape=(M, M))     # u_r ← v_{r+3}
    A = sp.bmat([[blocks[0], cpl[0]], [cpl[1], blocks[1]]], format="csr")
    A.sum_duplicates()
    A.sort_indices()
    return A


@pytest.mark.parametrize("case", ["brusselator64", "brusselator512", "torus256", "torus1024", "two_segments", "two_segments_small"])
def test_resident_powers_segmented_and_ring_layouts(nls, dev, case):
    """Matrices that are banded only segment by segment (the species of a reaction–diffusion system in (i, j, species) order: the
    Brusselator of config C5 / docs/src/tutorials/large_systems.md) and / or on a ring (periodic boundaries): the workgroup that
    owns band b of EVERY segment keeps the coupling on the chip — k_spmv_powers_seg, bit-identical to streaming launches."""
    import torch
    rng = np.random.default_rng(3)
    if case.startswith("brusselator"):
        P = R.Brusselator2D(int(case[len("brusselator"):]))
        A = P.jac(P.u0() + 0.1 * rng.standard_normal(P.n)).tocsr()     # periodic 5-point stencil per species + the u ↔ v coupling
    elif case.startswith("torus"):
        A = _periodic_laplacian(int(case[len("torus"):]), 4)
    else:
        A = _two_segments(300000 if case == "two_segments" else 5000, 1000 if case == "two_segments" else 40, 5)
    A.sort_indices()
    n = A.shape[0]
    M = nls.CSRMatrix.from_scipy(A)
    x = rng.standard_normal(n)
    lam = float(abs(A).sum(axis=1).max())
    s = 15 if n > 100000 else 6
    theta = (0.5 + 0.4 * np.cos(np.arange(s))) * lam
    ref = _powers_ref(A, x, s, theta, 2.0 / lam)
    dx = torch.tensor(x, device=dev)
    for rep in range(3):
        Y, resident = M.powers(dx, s, theta=theta, scale=2.0 / lam)
        assert resident, case
        assert np.array_equal(Y.cpu().numpy(), ref), (case, rep)


def test_streaming_fallback_for_ineligible_matrices(nls, dev):
    """wide band, long rows, too many rows: s streaming launches, same bits"""
    import torch
    for A in (_banded(20000, 5000, 5, 1), _banded(3000, 40, 24, 2)):
        M = nls.CSRMatrix.from_scipy(A)
        x = np.random.default_rng(6).standard_normal(A.shape[0])
        theta = np.array([0.1, -0.2, 0.3])
        Y, resident = M.powers(torch.tensor(x, device=dev), 3, theta=theta, scale=1.5)
        assert not resident
        assert np.array_equal(Y.cpu().numpy(), _powers_ref(A, x, 3, theta, 1.5))


@pytest.mark.parametrize("ns", [512, 1024])
def test_resident_powers_under_uneven_load(nls, dev, ns):
    """a second stream keeps part of the chip busy while the band hand-offs run: every word of 15 powers, 20 launches
    (512²: bands of one slice; 1024²: the headline shape, four)"""
    import torch
    p = R.Bratu2D(ns)
    rng = np.random.default_rng(11)
    J = p.jac(rng.standard_normal(p.n) * 0.1)
    A = nls.CSRMatrix.from_scipy(J)
    x = rng.standard_normal(p.n)
    lam = float(abs(J).sum(axis=1).max())
    theta = (0.5 + 0.4 * np.cos(np.arange(15))) * lam
    ref = _powers_ref(J, x, 15, theta, 2.0 / lam)
    dx = torch.tensor(x, device=dev)
    side = torch.cuda.Stream()
    big = torch.randn(1 << 24, device=dev, dtype=torch.float64)
    for rep in range(20):
        with torch.cuda.stream(side):
            for _ in range(rep % 4):
                big = big * 1.0000001
        Y, resident = A.powers(dx, 15, theta=theta, scale=2.0 / lam)
        assert resident and np.array_equal(Y.cpu().numpy(), ref), rep
    torch.cuda.synchronize()


def test_sstep_gmres_with_resident_powers_equals_streaming(nls, dev, monkeypatch):
    """the s-step solver's iterates do not depend on which kernel built the block: NK_SPMV_POWERS=0 in a child process"""
    import os
    import subprocess
    import sys
    code = (
        "import numpy as np, nonlinearsolve_jl_amd as nls\n"
        "prob = nls.NonlinearProblem(nls.Bratu2D(256, 6.0))\n"
        "alg = nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(gmres_restart=30, maxiters=30, ortho='sstep', fixed_iters=30),"
        " concrete_jac=True)\n"
        "cache = nls.init(prob, alg, abstol=1e-300, maxiters=50)\n"
        "for _ in range(3): cache.step()\n"
        "u = cache.u\n"
        "u = np.asarray(u.cpu() if hasattr(u, 'cpu') else u)\n"
        "import hashlib; print('HASH', hashlib.sha256(u.tobytes()).hexdigest(), float(np.abs(u).max()))\n")
    outs = []
    for flag in ("1", "0"):
        env = dict(os.environ, NK_SPMV_POWERS=flag)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([l for l in r.stdout.splitlines() if l.startswith("HASH")][-1])
    assert outs[0] == outs[1], outs


def test_brusselator_trust_region_with_resident_powers_equals_streaming(nls, dev):
    """config C5's shape (Brusselator, TrustRegion + s-step GMRES on the CSR Jacobian): the segmented resident kernel builds the
    blocks — same iterates, bit for bit, as with NK_SPMV_POWERS=0"""
    import os
    import subprocess
    import sys
    code = (
        "import numpy as np, nonlinearsolve_jl_amd as nls\n"
        "ctx = nls.default_context()\n"
        "prob = nls.NonlinearProblem(nls.Brusselator2D(128))\n"
        "alg = nls.TrustRegion(linsolve=nls.KrylovJL_GMRES(gmres_restart=30, maxiters=30, ortho='sstep', fixed_iters=30),"
        " concrete_jac=True)\n"
        "cache = nls.init(prob, alg, abstol=1e-300, maxiters=50)\n"
        "ctx.profile_enable(True)\n"
        "for _ in range(3): cache.step()\n"
        "rep = ctx.profile_report()\n"
        "u = cache.u\n"
        "u = np.asarray(u.cpu() if hasattr(u, 'cpu') else u)\n"
        "import hashlib; print('HASH', hashlib.sha256(u.tobytes()).hexdigest(), float(np.abs(u).max()), 'spmv_powers' in rep)\n")
    outs = []
    for flag in ("1", "0"):
        env = dict(os.environ, NK_SPMV_POWERS=flag)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([l for l in r.stdout.splitlines() if l.startswith("HASH")][-1].split())
    assert outs[0][1:3] == outs[1][1:3], outs
    assert outs[0][3] == "True" and outs[1][3] == "False", outs


def test_matrix_free_operator_takes_the_resident_kernel_too(nls, dev):
    """config C3 as BASELINE.json words it (matrix-free JacVecOperator): the stencil Jacobian's rows are GENERATED into the kernel's
    registers instead of loaded — the s-step solver's iterates against the per-column JVP launches (NK_SPMV_POWERS=0), to rounding
    (the generated row sums add the five products in CSR order, the stencil kernel factors c_lap out: 1e-13 relative per product)."""
    import os
    import subprocess
    import sys
    code = (
        "import numpy as np, nonlinearsolve_jl_amd as nls\n"
        "ctx = nls.default_context()\n"
        "prob = nls.NonlinearProblem(nls.Bratu2D(256, 6.0))\n"
        "alg = nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(gmres_restart=30, maxiters=30, ortho='sstep', fixed_iters=30),"
        " concrete_jac=False)\n"
        "cache = nls.init(prob, alg, abstol=1e-300, maxiters=50)\n"
        "ctx.profile_enable(True)\n"
        "for _ in range(3): cache.step()\n"
        "rep = ctx.profile_report()\n"
        "u = cache.u\n"
        "u = np.asarray(u.cpu() if hasattr(u, 'cpu') else u)\n"
        "np.save(__import__('sys').argv[1], u)\n"
        "print('FAMILIES', sorted(rep))\n")
    outs = []
    import tempfile
    for flag in ("1", "0"):
        with tempfile.NamedTemporaryFile(suffix=".npy") as tf:
            env = dict(os.environ, NK_SPMV_POWERS=flag)
            r = subprocess.run([sys.executable, "-c", code, tf.name], env=env, capture_output=True, text=True, timeout=300,
                               cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            assert r.returncode == 0, r.stderr[-2000:]
            fam = [l for l in r.stdout.splitlines() if l.startswith("FAMILIES")][-1]
            assert ("spmv_powers" in fam) == (flag == "1") and ("'jvp'" in fam) == (flag == "0"), fam
            outs.append(np.load(tf.name))
    assert np.max(np.abs(outs[0] - outs[1])) <= 1e-10 * np.max(np.abs(outs[1]))


def test_a_timed_out_launch_is_noticed_and_the_solve_rerun_on_the_streaming_kernel(nls, dev):
    """The resident kernel needs every workgroup on the chip at once; if one never shows up (something else holds its compute
    unit) its neighbours give up after NK_PW_TIMEOUT_MS instead of hanging the GPU, the columns of that launch are garbage — and the
    library notices: the plan is parked, the linear solve is run again on the streaming kernel, the caller sees the same iterates as
    with NK_SPMV_POWERS=0. The plan comes back with the next linear solve (what held the compute units is usually gone by then);
    three torn launches switch it off for the life of the matrix. Provoked by the development hook that makes band 0 of chosen
    launches withhold its flag."""
    import os
    import subprocess
    import sys
    code = (
        "import numpy as np, hashlib, nonlinearsolve_jl_amd as nls\n"
        "prob = nls.NonlinearProblem(nls.Bratu2D(128, 6.0))\n"
        "alg = nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(gmres_restart=30, maxiters=30, ortho='sstep', fixed_iters=30), concrete_jac=True)\n"
        "cache = nls.init(prob, alg, abstol=1e-300, maxiters=50)\n"
        "for _ in range(8): cache.step()\n"
        "u = cache.u\n"
        "u = np.asarray(u.cpu() if hasattr(u, 'cpu') else u)\n"
        "ctx = nls.default_context(); ctx.profile_enable(True); cache.step(); fam = sorted(ctx.profile_report())\n"
        "print('FAMILIES', fam)\n"
        "print('HASH', hashlib.sha256(u.tobytes()).hexdigest())\n")
    outs = {}
    # one torn launch (the 5th: the third step's first block): parked, rerun, back for the fourth step;
    # three of them (launches 5, 8, 11 — each in a later solve): the ninth step still runs the streaming kernel
    for name, env in (("stalled", dict(NK_PW_DEBUG_STALL_LAUNCH="5", NK_PW_TIMEOUT_MS="20")),
                      ("stalled3", dict(NK_PW_DEBUG_STALL_LAUNCH="5,8,11", NK_PW_TIMEOUT_MS="20")),
                      ("streaming", dict(NK_SPMV_POWERS="0"))):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0, r.stderr[-2000:]
        outs[name] = [l for l in r.stdout.splitlines() if l.startswith("HASH")][-1]
        fam = [l for l in r.stdout.splitlines() if l.startswith("FAMILIES")][-1]
        assert ("spmv_powers" in fam) == (name == "stalled"), (name, fam)
    assert outs["stalled"] == outs["streaming"] == outs["stalled3"], outs


def test_a_torn_launch_of_the_public_entry_point_falls_back_to_streaming_launches():
    """nk_csr_powers itself: the launch that times out is answered with the s streaming launches — same bits, resident = False —
    instead of an error; the next call is resident again."""
    import os
    import subprocess
    import sys
    code = (
        "import numpy as np, scipy.sparse as sp, torch, nonlinearsolve_jl_amd as nls\n"
        "from oracle import reference_restatement as R\n"
        "from tests.test_gpu_powers import _powers_ref\n"
        "p = R.Bratu2D(64); rng = np.random.default_rng(1); J = p.jac(rng.standard_normal(p.n) * 0.1)\n"
        "A = nls.CSRMatrix.from_scipy(J); x = rng.standard_normal(p.n); ref = _powers_ref(J, x, 7, None, 0.01)\n"
        "flags = []\n"
        "for rep in range(3):\n"
        "    Y, res = A.powers(torch.tensor(x, device='cuda'), 7, theta=None, scale=0.01)\n"
        "    assert np.array_equal(Y.cpu().numpy(), ref), rep\n"
        "    flags.append(bool(res))\n"
        "print('FLAGS', flags)\n")
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, NK_PW_DEBUG_STALL_LAUNCH="2", NK_PW_TIMEOUT_MS="20"),
                       capture_output=True, text=True, timeout=300, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stderr[-2000:]
    assert "FLAGS [True, False, True]" in r.stdout, r.stdout
