"""Which form of the resident matrix-powers kernel a CSR pattern fits (csrc/nk_powers.hip::pw_find_layout through
nk_csr_powers_layout) — host arithmetic only: runs without a GPU. The device tests of the kernels themselves are in
tests/test_gpu_powers.py."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import reference_restatement as R


@pytest.fixture(scope="module")
def layout():
    import nonlinearsolve_jl_amd as nls
    return lambda A, cus=256: nls.CSRMatrix.powers_layout(A, cus)


def _torus(N):
    idx = np.arange(N * N).reshape(N, N)
    rows = [idx.ravel()] * 5
    cols = [idx.ravel(), np.roll(idx, -1, 1).ravel(), np.roll(idx, 1, 1).ravel(), np.roll(idx, -1, 0).ravel(), np.roll(idx, 1, 0).ravel()]
    A = sp.csr_matrix((np.ones(5 * N * N), (np.concatenate(rows), np.concatenate(cols))), shape=(N * N, N * N))
    A.sort_indices()
    return A


def test_bratu_jacobians_are_plain_bands(layout):
    for ns, slices, bands in ((64, 1, 4), (512, 1, 256), (1024, 4, 256)):
        L = layout(R.Bratu2D(ns).jac(np.zeros(ns * ns)).tocsr())
        assert L == {"kind": "bands", "slices": slices, "slots": 5, "bands": bands, "segments": 1, "ring": False}, (ns, L)
    # config C4's 4096² is 8 × what the register file holds: streaming launches
    n = 4096 * 4096
    rp = np.arange(0, 5 * n + 1, 5, dtype=np.int64)
    assert rp[-1] < 2 ** 31
    band = sp.diags([1.0] * 5, [-4096, -1, 0, 1, 4096], shape=(n, n), format="csr")
    assert layout(band)["kind"] == "none"


def test_the_brusselator_is_two_segments_on_rings(layout):
    """config C5 / the reference's large-systems tutorial: (i, j, species) ordering, periodic boundaries"""
    for N in (64, 512):
        P = R.Brusselator2D(N)
        A = P.jac(P.u0() + 0.1).tocsr()
        L = layout(A)
        assert L == {"kind": "segments", "slices": 1, "slots": 8, "bands": N * N // 1024, "segments": 2, "ring": True}, (N, L)
    # N = 16: all 512 rows are one band of the plain layout
    P = R.Brusselator2D(16)
    assert layout(P.jac(P.u0() + 0.1).tocsr()) == {"kind": "bands", "slices": 1, "slots": 8, "bands": 1, "segments": 1, "ring": False}
    # N = 48: a segment of 2304 rows is not a whole number of 1024-row bands — no ring neighbours: streaming launches
    P = R.Brusselator2D(48)
    assert layout(P.jac(P.u0() + 0.1).tocsr())["kind"] == "none"


def test_torus_and_two_plain_segments(layout):
    assert layout(_torus(256)) == {"kind": "segments", "slices": 1, "slots": 5, "bands": 64, "segments": 1, "ring": True}
    M = 5000
    blk = sp.diags([1.0, 1.0, 1.0], [-40, 0, 40], shape=(M, M), format="csr")
    A = sp.bmat([[blk, sp.identity(M)], [sp.identity(M), blk]], format="csr")
    assert layout(A) == {"kind": "segments", "slices": 1, "slots": 5, "bands": 5, "segments": 2, "ring": False}
    # fewer compute units: more rows per band
    assert layout(A, cus=2)["slices"] == 4 or layout(A, cus=2)["kind"] == "none"


def test_what_does_not_fit(layout):
    n = 20000
    wide = sp.diags([1.0, 1.0, 1.0], [-5000, 0, 5000], shape=(n, n), format="csr")          # band wider than a slice
    assert layout(wide)["kind"] == "none"
    rng = np.random.default_rng(0)
    long_rows = sp.random(3000, 3000, density=24 / 3000, random_state=rng, format="csr") + sp.identity(3000, format="csr")
    assert layout(long_rows)["kind"] == "none"                                               # > 16 entries in a row
