"""The launches of ONE fixed-work Newton step of bench.py's default protocol (Bratu, NewtonRaphson, GMRES(m) with exactly m
Arnoldi steps, the s-step Arnoldi process) and the bytes each of them has to move — the table `roofline_step` of the bench line
is computed from, and the one tests/test_step_model.py checks kernel by kernel against the rocprofv3 timeline committed under
profiles/ (so the model cannot charge for work the code no longer launches — round 3's line did).

Two byte counts per launch:
* `hbm`  — what the launch must move between HBM and the chip given what it keeps on the chip (the resident matrix-powers
  kernel reads the matrix once per block, csrc/nk_powers.hip);
* `alg`  — SURVEY.md §8(d)'s algorithmic figure: every operator application charged at the CSR SpMV's 12 nnz + 4 (n + 1) + 16 n
  bytes however it is executed (what `roofline.achieved` of an SpMV-type kernel is computed from).
For every kernel except the resident matrix powers the two are equal.
"""


def sstep_blocks(arnoldi, s):
    """(k, width) of the blocks the s-step cycle builds (csrc/nk_sstep.hip::nk_ss_cycle / nk_ss_block_width)"""
    out, k = [], 1
    while k - 1 < arnoldi:
        w = min(s, arnoldi - (k - 1))
        w = 15 if w >= 15 else 8 if w >= 8 else 6 if w >= 6 else 4 if w >= 4 else 2 if w >= 2 else 1
        if k + w > 48 and w > 8:
            w = 8
        out.append((k, w))
        k += w
    return out


def spmv_bytes(n, nnz):
    return 12.0 * nnz + 4.0 * (n + 1) + 16.0 * n


def step_launches(n, nnz, arnoldi=30, s=15, matfree=False, resident_powers=True, newton_basis=True, implicit=True, deferred=True,
                  fused_tail=True, preloaded_rhs=True, folded_norms=True, begin_ahead=True, last_block_unstored=True):
    """[(kernel name prefix, hbm bytes, algorithmic bytes)] in launch order.
    folded_norms (round 6): the stage-2 reduction of the step's norms and their delivery to the host ride in workgroup 0 of the
    NEXT Jacobian's fill kernel (k_bratu_jac, enqueued directly behind the residual kernel) — no k_reduce_inf2 launch.
    last_block_unstored (round 6, with deferred): sweep B of the cycle's LAST block stores nothing (its columns are read once more
    only, by x = V y, and enter that product as the matrix powers left them — one more entry in the back-substitution's list):
    8 n w bytes less, for the launch and for the algorithm.
    begin_ahead (round 6, with folded_norms): the linear solve's cycle begin rides there as well (its inputs — Σ f² and the
    Gershgorin partials of the new Jacobian — are left by the residual kernel) — no k_ss_cycle_begin launch.
    fused_tail (round 5): the Newton update u_new = u − x rides in the pass that forms x = V y (k_multiaxpy), and the residual
    kernel leaves the stage-1 partials of its own norms and a second copy of f in column 0 of the Krylov basis — no
    k_newton_update, no k_absmax_sumsq, no k_copy_sumsq launch.
    deferred (round 5, the default with the implicit second pass): a block's second factorisation rides in the NEXT block's scalar
    launch (k_ss_job), the previous block's Hessenberg columns in workgroup 0 of this block's sweep B, and the cycle's last scalar
    launch also back-substitutes — three k_ss_job launches per two-block cycle where rounds 3–4 had four k_ss_reduce_factor,
    k_ss_hess and k_backsolve. The byte counts are the same (scalar launches move nothing that counts)."""
    deferred = deferred and implicit
    m = arnoldi
    L = []

    def add(name, hbm, alg=None):
        L.append((name, float(hbm), float(hbm if alg is None else alg)))

    if not matfree:
        add("k_bratu_jac", 8.0 * nnz + 8.0 * n)                 # u in, values out (the Gershgorin partials ride along)
    preloaded_rhs = preloaded_rhs and fused_tail
    if not preloaded_rhs:
        add("k_copy_sumsq", 16.0 * n)                           # b → column 0, ‖b‖² (preloaded: the residual kernel left both)
    folded_norms = folded_norms and fused_tail and not matfree
    begin_ahead = begin_ahead and folded_norms and preloaded_rhs and newton_basis
    if not begin_ahead:
        add("k_ss_cycle_begin", 0)
    b_op = 24.0 * n if matfree else spmv_bytes(n, nnz)
    blocks = sstep_blocks(m, s)
    for bi, (k, w) in enumerate(blocks):
        if resident_powers and w >= 2:
            # the matrix once (matrix-free: the diagonal d = c·exp(u) once), the start column once, w new columns written
            add("k_spmv_powers", (8.0 * n if matfree else 12.0 * nnz + 4.0 * (n + 1)) + 8.0 * n + 8.0 * n * w, w * b_op)
        else:
            for _ in range(w):
                add("k_bratu_jvp" if matfree else "k_spmv_stream", b_op)
        add("k_ss_block<A>", 8.0 * n * (k + w))                 # Gram of [V X]ᵀX: k + w columns read
        add("k_ss_job" if deferred else "k_ss_reduce_factor", 0)
        unstored = last_block_unstored and deferred and newton_basis and bi + 1 == len(blocks) and w == 15 and k in (1, 16)
        add("k_ss_block<B>", 8.0 * n * (k + (1 if unstored else 2) * w))   # update (k + w read, w written — or not) + Gram of the result
        if deferred:
            continue                                            # (its reduction rides in the next scalar launch)
        add("k_ss_reduce_factor", 0)
        if bi + 1 < len(blocks):
            if not implicit:
                add("k_ss_block<C>", 8.0 * n * (k + 2 * w))     # second update (NK_SS_IMPLICIT=0)
            # implicit second pass: no third sweep, and the block's Hessenberg columns are derived inside the next sweep A
        else:
            add("k_ss_hess", 0)                                 # the cycle's last block: its Hessenberg columns, own launch
    if deferred:
        add("k_ss_job", 0)                                      # the cycle's last launch: reduce, factor, Hessenberg columns, y
    else:
        add("k_backsolve", 0)
    if fused_tail:
        add("k_multiaxpy", 8.0 * n * (m + 4))                   # x = V y (m + 1 columns read, x written) + u read, u_new written
        # f(u_new) and the partials of ‖f‖∞, ‖f‖₂² — preloaded_rhs: f stored twice (fu and column 0 of the next solve's basis)
        add("k_bratu_residual_norms", (24.0 if preloaded_rhs else 16.0) * n)
    else:
        add("k_multiaxpy", 8.0 * n * (m + 2))                   # x = V y: m + 1 columns read, x written
        add("k_newton_update", 24.0 * n)
        add("k_bratu_residual", 16.0 * n)
        add("k_absmax_sumsq", 8.0 * n)
    if not folded_norms:
        add("k_reduce_inf2", 0)
    return L


def step_bytes(*a, **kw):
    L = step_launches(*a, **kw)
    return sum(x[1] for x in L), sum(x[2] for x in L)


def canonical(kernel_name):
    """a rocprofv3 kernel name → the model's name (sweeps: A = Gram only, B = update + Gram, C = update only)"""
    import re
    nm = re.sub(r"^void ", "", kernel_name)
    nm = re.sub(r"\(.*$", "", nm)
    if nm.startswith("k_ss_block_mm<") or nm.startswith("k_ss_block_ro<"):   # sweep B of the default cycle's shapes: update and Gram
        return "k_ss_block<B>"                                                 # block on the matrix cores (_ro: the last block's, read-only)
    if nm.startswith("k_ss_block<"):
        f = [x.strip() for x in nm[len("k_ss_block<"):].rstrip(">").split(",")]
        upd, gram = f[1] in ("true", "1"), f[2] in ("true", "1")
        return "k_ss_block<%s>" % ("A" if (gram and not upd) else "B" if (gram and upd) else "C")
    return re.sub(r"<.*$", "", nm)
