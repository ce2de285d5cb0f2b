"""Multi-rank code path of libmi355x_nk on ONE GPU: two processes share cuda:0, collectives go through the
library's callback communicator over torch.distributed/gloo. Everything except the RCCL calls themselves is the
code that runs on 2–8 GPUs: row-range partition, halo plans (grid lines and general CSR), halo gather kernels,
placement of the all-reduces in GMRES and in the Newton driver. Results must equal the serial oracle.
A second test drives the dlopen'ed RCCL entry points on a 1-rank communicator."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, transport="torch"):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import hashlib
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    digest = {}

    def note(key, arr):   # what the parent compares bit for bit between the transports
        digest[key] = hashlib.sha1(np.ascontiguousarray(arr).tobytes()).hexdigest()

    try:
        import nonlinearsolve_jl_amd as nls
        from oracle import reference_restatement as R
        torch.cuda.set_device(0)
        ctx = nls.Context(device=0)
        nls.set_default_context(ctx)
        assert nls.dist.init_comm(ctx, transport) == ("torch" if transport == "torch" else "peer+torch")
        assert ctx.comm_info() == (2, world, rank)
        assert ctx.comm_peer_status() == (transport == "peer", 0)
        dev = torch.device("cuda:0")
        rng = np.random.default_rng(7)

        # ---------------- Bratu: partition by grid lines, halo = one line per neighbour
        ns = 20
        pb = R.Bratu2D(ns)
        P = nls.Bratu2D(ns)
        b, e = P.row_begin, P.row_begin + P.n_local
        assert (b, e) == tuple(x for x in nls.partition_range(ns * ns, ns, world, rank))
        u, v = 0.2 * rng.standard_normal(pb.n), rng.standard_normal(pb.n)
        ul, vl = torch.tensor(u[b:e], device=dev), torch.tensor(v[b:e], device=dev)
        assert np.allclose(P.residual(ul).cpu().numpy(), pb.f(u)[b:e], rtol=1e-13, atol=1e-14)
        assert np.allclose(P.jvp(vl, ul).cpu().numpy(), pb.jvp(v, u)[b:e], rtol=1e-13, atol=1e-12)
        # assembled, row-partitioned CSR with a general halo plan (collective setup)
        J = P.jac_csr()
        P.jac_values(ul, J)
        info = J.info()
        assert info["nrows_local"] == e - b and info["n_halo"] == ns  # one neighbour line
        assert np.allclose(J.matvec(vl).cpu().numpy(), (pb.jac(u) @ v)[b:e], rtol=1e-13, atol=1e-11)
        # distributed GMRES on the CSR operator vs the serial oracle
        rhs = rng.standard_normal(pb.n)
        xref, iref = R.gmres(lambda z: pb.jac(u) @ z, rhs, rtol=1e-9, restart=30, itmax=3000)
        for ortho in ("cgs2", "mgs", "cgs", "dcgs2", "dcgs2_1r"):   # "dcgs2" itself runs as dcgs2_1r on several ranks
            G = nls.GMRES(e - b, restart=30, ortho=ortho).set_operator(J)
            x, gi = G.solve(torch.tensor(rhs[b:e], device=dev), reltol=1e-9, maxiters=3000)
            xg = nls.dist.gather_vector(x, pb.n, b)
            note("gmres_" + ortho, xg)
            assert gi["converged"] and np.linalg.norm(xg - xref) <= 1e-7 * np.linalg.norm(xref), ortho
            assert abs(gi["iters"] - iref.iters) <= max(3, iref.iters // 20)
        # distributed Newton–Krylov (matrix-free and concrete J) vs the serial oracle
        ref = R.solve(pb, R.NewtonRaphson(linsolve=R.KrylovJL_GMRES(), forcing=R.EisenstatWalkerForcing2()),
                      abstol=1e-9, maxiters=50)
        for concrete in (False, True):
            prob = nls.NonlinearProblem(P, u0=torch.zeros(e - b, dtype=torch.float64, device=dev))
            sol = nls.solve(prob, nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(), forcing=nls.EisenstatWalkerForcing2(),
                                                    concrete_jac=concrete), abstol=1e-9, maxiters=50)
            ug = nls.dist.gather_vector(sol.u, pb.n, b)
            note(f"newton_{concrete}", ug)
            assert sol.retcode == "Success"
            assert np.max(np.abs(ug - ref.u)) <= 1e-7
            assert abs(sol.stats.nsteps - ref.stats.nsteps) <= 1
            assert sol.stats.allreduces > 0 and sol.stats.halo_exchanges > 0
            # the default orthogonalisation on several ranks is DCGS2 with ONE all-reduce per Arnoldi step (+ one per
            # cycle for ‖r0‖, one for the flush, a handful per Newton step); counters are per solver cache
            bound = sol.stats.gmres_iters + 24 * (sol.stats.nsteps + 1)   # (two reductions per step would exceed it)
            assert sol.stats.allreduces <= bound, (sol.stats.allreduces, sol.stats.gmres_iters, sol.stats.nsteps, bound)

        # ---------------- synchronisation points of the headline protocol, counted (the Krylov inner products are
        # the path's only collective). One fixed-work Newton step = GMRES(30) in two s-step blocks of 15: ONE all-reduce per Gram
        # sweep (A and B of each block; the cycle's last block has no sweep C) = 4, + the Gershgorin bounds of the new Jacobian
        # (max), + ‖b‖² at the cycle's start, + the update's ‖δu‖₂, + the step's norm batch (‖f‖∞, Σf²) = 8; 31 halo exchanges
        # (30 operator applications + the residual). Column by column it is 30 + a handful. BASELINE.md §6 prices the 8-GPU
        # strong-scaling point with these numbers: a change of them is a change of the scaling model and must be deliberate.
        probc = nls.NonlinearProblem(P, u0=torch.zeros(e - b, dtype=torch.float64, device=dev))
        cch = nls.init(probc, nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(gmres_restart=30, maxiters=30, fixed_iters=30),
                                                concrete_jac=True), abstol=1e-300, maxiters=10 ** 6)
        cch.step()
        a0, h0 = cch.stats.allreduces, cch.stats.halo_exchanges
        for _ in range(3):
            cch.step()
        per_step = (cch.stats.allreduces - a0) / 3.0
        halo_step = (cch.stats.halo_exchanges - h0) / 3.0
        # (peer-mapped arenas: one launch fewer carries a collective, and — round 5 — a block's second factorisation rides in the
        #  next block's scalar launch: 3 messages for the 4 Gram blocks — 6; the callback transport issues every one of the 8 itself)
        assert per_step == (6.0 if transport == "peer" else 8.0) and halo_step == 31.0, f"PERSTEP={per_step} HALOSTEP={halo_step}"
        cch.close()

        # ---------------- halo exchange overlapped with the interior row blocks (second stream + events): a grid large
        # enough to have both interior and boundary row blocks; every result is bitwise equal to the serial-exchange path
        ns2 = 64
        pb2, P2 = R.Bratu2D(ns2), nls.Bratu2D(ns2)
        b2, e2 = P2.row_begin, P2.row_begin + P2.n_local
        u2, v2 = 0.2 * rng.standard_normal(pb2.n), rng.standard_normal(pb2.n)
        u2l, v2l = torch.tensor(u2[b2:e2], device=dev), torch.tensor(v2[b2:e2], device=dev)
        J2 = P2.jac_csr()
        P2.jac_values(u2l, J2)
        rhs2 = torch.tensor(rng.standard_normal(pb2.n)[b2:e2], device=dev)
        outs = []
        for ov in (False, True):
            ctx.set_halo_overlap(ov)
            y = J2.matvec(v2l).clone()
            G2 = nls.GMRES(e2 - b2, restart=30).set_operator(J2)
            x2, gi2 = G2.solve(rhs2, reltol=1e-10, maxiters=2000)
            prob2 = nls.NonlinearProblem(P2, u0=torch.zeros(e2 - b2, dtype=torch.float64, device=dev))
            s2 = nls.solve(prob2, nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(), forcing=nls.EisenstatWalkerForcing2(),
                                                    concrete_jac=True), abstol=1e-9, maxiters=50)
            outs.append((y, x2.clone(), gi2["iters"], s2.u.clone(), s2.stats.nsteps, s2.stats.gmres_iters))
        ctx.set_halo_overlap(False)
        assert np.allclose(outs[0][0].cpu().numpy(), (pb2.jac(u2) @ v2)[b2:e2], rtol=1e-13, atol=1e-10)
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and outs[0][2] == outs[1][2]
        assert torch.equal(outs[0][3], outs[1][3]) and outs[0][4:] == outs[1][4:]

        # ---------------- Brusselator: periodic halo (with 2 ranks the same peer is both neighbours)
        N = 10
        rb_ = R.Brusselator2D(N)
        PB = nls.Brusselator2D(N)
        j0, j1 = nls.partition_range(N, 1, world, rank)
        nl = j1 - j0
        # local (i, j_local, species) layout ↔ reference ordering i + N j + N² s
        idx = np.array([i + N * (j0 + jl) + N * N * s for s in range(2) for jl in range(nl) for i in range(N)])
        assert PB.n_local == idx.size
        ub, vb = rb_.u0() + 0.1 * rng.standard_normal(rb_.n), rng.standard_normal(rb_.n)
        ubl, vbl = torch.tensor(ub[idx], device=dev), torch.tensor(vb[idx], device=dev)
        assert np.allclose(PB.initial_guess(device=True).cpu().numpy(), rb_.u0()[idx], rtol=1e-14)
        assert np.allclose(PB.residual(ubl).cpu().numpy(), rb_.f(ub)[idx], rtol=1e-12, atol=1e-10)
        assert np.allclose(PB.jvp(vbl, ubl).cpu().numpy(), rb_.jvp(vb, ub)[idx], rtol=1e-12, atol=1e-9)
        assert np.allclose(PB.vjp(vbl, ubl).cpu().numpy(), rb_.vjp(vb, ub)[idx], rtol=1e-12, atol=1e-9)
        JB = PB.jac_csr()
        PB.jac_values(ubl, JB)
        assert np.allclose(JB.matvec(vbl).cpu().numpy(), (rb_.jac(ub) @ vb)[idx], rtol=1e-12, atol=1e-8)
        # TrustRegion + GMRES on the partitioned Brusselator (matrix-free JVP/VJP) vs the serial oracle
        oc = R.init(rb_, R.TrustRegion(linsolve=R.KrylovJL_GMRES(gmres_restart=30, maxiters=4000)), abstol=1e-8,
                    maxiters=25)
        oc.lin_reltol, oc.lin_abstol = 1e-10, 0.0
        refb = oc.solve()
        probb = nls.NonlinearProblem(PB, u0=PB.initial_guess(device=True))
        solb = nls.solve(probb, nls.TrustRegion(linsolve=nls.KrylovJL_GMRES(gmres_restart=30, maxiters=4000,
                                                                            reltol=1e-10, abstol=0.0)),
                         abstol=1e-8, maxiters=25)
        assert solb.retcode == "Success" and solb.stats.nsteps == refb.stats.nsteps
        assert np.max(np.abs(solb.u.cpu().numpy() - refb.u[idx])) <= 1e-7
        note("brusselator_tr", solb.u.cpu().numpy())

        # ---------------- distributed transposed SpMV: the contributions to entries the other rank owns come back through
        # the halo plan in reverse (steepest.jl:75-77 / trust_region.jl:410 with a concrete J on several ranks)
        assert np.allclose(J.rmatvec(vl).cpu().numpy(), (pb.jac(u).T @ v)[b:e], rtol=1e-13, atol=1e-11)
        Jfull = rb_.jac(ub).tocsr()
        want = (Jfull.T @ vb)[idx]
        assert np.allclose(JB.rmatvec(vbl).cpu().numpy(), want, rtol=1e-12, atol=1e-8)
        assert torch.equal(JB.rmatvec(vbl), JB.rmatvec(vbl))           # fixed accumulation order: bitwise reproducible
        # ---------------- Julia's SparseMatrixCSC of the whole matrix, ingested slice by slice (collective)
        Jc = pb.jac(u).tocsc()
        Jcsc = nls.CSRMatrix.from_csc(Jc.indptr + 1, Jc.indices + 1, Jc.data, index_base=1, row_range=(b, e))
        assert Jcsc.info()["nrows_local"] == e - b and Jcsc.info()["n_halo"] == ns
        assert np.allclose(Jcsc.matvec(vl).cpu().numpy(), (pb.jac(u) @ v)[b:e], rtol=1e-13, atol=1e-11)
        assert np.allclose(Jcsc.rmatvec(vl).cpu().numpy(), (pb.jac(u).T @ v)[b:e], rtol=1e-13, atol=1e-11)
        # ---------------- colour-compressed assembly on two ranks (one global colouring, computed identically on every
        # rank) equals the closed-form fill; then config C5 as written: TrustRegion + GMRES on the concrete, coloured J
        JB2 = PB.jac_csr()
        ncol = PB.jac_values(ubl, JB2, colored=True)
        assert 6 <= ncol <= 14 and np.allclose(JB2.values(), JB.values(), rtol=1e-12, atol=1e-10)
        oc2 = R.init(rb_, R.TrustRegion(linsolve=R.KrylovJL_GMRES(gmres_restart=30, maxiters=4000), concrete_jac=True),
                     abstol=1e-8, maxiters=25)
        oc2.lin_reltol, oc2.lin_abstol = 1e-10, 0.0
        refc = oc2.solve()
        solc = nls.solve(nls.NonlinearProblem(PB, u0=PB.initial_guess(device=True)),
                         nls.TrustRegion(linsolve=nls.KrylovJL_GMRES(gmres_restart=30, maxiters=4000, reltol=1e-10, abstol=0.0),
                                         concrete_jac=True), abstol=1e-8, maxiters=25, store_trace=True)
        assert solc.retcode == "Success" == R.RETCODE_NAMES[refc.retcode] and solc.stats.nsteps == refc.stats.nsteps
        assert [t["accepted"] for t in solc.trace] == [t["accepted"] for t in refc.trace]
        assert np.max(np.abs(solc.u.cpu().numpy() - refc.u[idx])) <= 1e-7
        note("brusselator_tr_concrete", solc.u.cpu().numpy())
        note("spmv_t", JB.rmatvec(vbl).cpu().numpy())
        # ---------------- LevenbergMarquardt on two ranks: diag(JᵀJ) through the reverse halo exchange, the damped
        # normal-form operator (SpMV, distributed transposed SpMV, diagonal), geodesic acceleration — vs the serial oracle
        dJ = JB.colsumsq(like=vbl).cpu().numpy()
        assert np.allclose(dJ, np.asarray(Jfull.multiply(Jfull).sum(axis=0)).ravel()[idx], rtol=1e-13)
        kl_ = dict(gmres_restart=60, maxiters=600)
        Plm = nls.Bratu2D(12)    # (the size of tests/test_gpu_lm.py's single-rank case: 14 steps on the oracle)
        bl, el = Plm.row_begin, Plm.row_begin + Plm.n_local
        reflm = R.solve(R.Bratu2D(12), R.LevenbergMarquardt(linsolve=R.KrylovJL_GMRES(**kl_)), abstol=1e-8, maxiters=100)
        sollm = nls.solve(nls.NonlinearProblem(Plm, u0=torch.zeros(el - bl, dtype=torch.float64, device=dev)),
                          nls.LevenbergMarquardt(linsolve=nls.KrylovJL_GMRES(**kl_)), abstol=1e-8, maxiters=100, store_trace=True)
        assert sollm.retcode == "Success" == R.RETCODE_NAMES[reflm.retcode] and sollm.stats.nsteps == reflm.stats.nsteps
        assert np.allclose([t["trust_region"] for t in sollm.trace], [t["trust_region"] for t in reflm.trace], rtol=1e-12)
        assert np.max(np.abs(sollm.u.cpu().numpy() - reflm.u[bl:el])) <= 1e-7
        note("bratu_lm", sollm.u.cpu().numpy())
        # ---------------- PseudoTransient on two ranks: α⁻¹ from the global ‖f‖₂; the shift rides on the matrix-free
        # operator and on the diagonal of the distributed CSR (local column index of the owned diagonal)
        for cj in (None, True):
            refpt = R.solve(R.Bratu2D(12), R.PseudoTransient(linsolve=R.KrylovJL_GMRES(**kl_), alpha_initial=10.0, concrete_jac=cj),
                            abstol=1e-9, maxiters=100)
            solpt = nls.solve(nls.NonlinearProblem(Plm, u0=torch.zeros(el - bl, dtype=torch.float64, device=dev)),
                              nls.PseudoTransient(linsolve=nls.KrylovJL_GMRES(**kl_), alpha_initial=10.0, concrete_jac=cj),
                              abstol=1e-9, maxiters=100)
            assert solpt.retcode == "Success" == R.RETCODE_NAMES[refpt.retcode] and solpt.stats.nsteps == refpt.stats.nsteps
            assert np.max(np.abs(solpt.u.cpu().numpy() - refpt.u[bl:el])) <= 1e-8
        note("bratu_pt", solpt.u.cpu().numpy())
        # ---------------- s-step GMRES on two ranks: the block Gram partials are reduced locally, then ONE all-reduce of
        # (k + s)·s values per sweep (two per block of s columns instead of s)
        refss = R.solve(R.Bratu2D(12), R.NewtonRaphson(linsolve=R.KrylovJL_GMRES(**kl_)), abstol=1e-9, maxiters=50)
        for cj in (None, True):
            solss = nls.solve(nls.NonlinearProblem(Plm, u0=torch.zeros(el - bl, dtype=torch.float64, device=dev)),
                              nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(ortho="sstep", sstep=5, **kl_), concrete_jac=cj),
                              abstol=1e-9, maxiters=50)
            assert solss.retcode == "Success" == R.RETCODE_NAMES[refss.retcode] and solss.stats.nsteps == refss.stats.nsteps
            assert np.max(np.abs(solss.u.cpu().numpy() - refss.u[bl:el])) <= 1e-8
        note("bratu_sstep", solss.u.cpu().numpy())
        # the library default (Newton basis, 15 columns per block): the bounds of the spectrum — Gershgorin discs of the local
        # rows / the stencil's closed form — are all-reduced (max) over the ranks before the shifts are formed on every rank
        for cj in (None, True):
            sold = nls.solve(nls.NonlinearProblem(Plm, u0=torch.zeros(el - bl, dtype=torch.float64, device=dev)),
                             nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(**kl_), concrete_jac=cj), abstol=1e-9, maxiters=50)
            assert sold.retcode == "Success" and sold.stats.nsteps == refss.stats.nsteps
            assert np.max(np.abs(sold.u.cpu().numpy() - refss.u[bl:el])) <= 1e-8
        note("bratu_sstep_newton15", sold.u.cpu().numpy())
        # ---------------- ILU(0) / Jacobi objects on a row-partitioned matrix: the factorisation of the rank's LOCAL square block
        # (halo columns dropped — block-Jacobi ILU(0) across ranks, no communication in the apply), as Pl of the distributed
        # GMRES, against the oracle's GMRES with the block-diagonal preconditioner assembled from the same row ranges
        import scipy.sparse as sp_
        Ag = sp_.csr_matrix(pb.jac(u))
        rng_all = [nls.partition_range(ns * ns, ns, world, r_) for r_ in range(world)]
        blocks = [R.ilu0_preconditioner(Ag[b_:e_, b_:e_].tocsr(), "natural") for b_, e_ in rng_all]

        def Mblock(x_):
            return np.concatenate([blk(x_[b_:e_]) for blk, (b_, e_) in zip(blocks, rng_all)])
        Milu = nls.ILU0Preconditioner(J, ordering="natural")
        xt = rng.standard_normal(pb.n)
        assert np.max(np.abs(Milu.apply(torch.tensor(xt[b:e], device=dev)).cpu().numpy() - Mblock(xt)[b:e])) <= 1e-12 * np.max(np.abs(Mblock(xt)))
        xlo, ilo = R.gmres(lambda z: Ag @ z, rhs, rtol=1e-9, restart=30, itmax=3000, ortho="cgs2", Ml=Mblock)
        for ortho in ("sstep", "dcgs2"):
            Gl = nls.GMRES(e - b, restart=30, ortho=ortho).set_operator(J).set_preconditioner(Milu, side="left")
            xl_, gl_ = Gl.solve(torch.tensor(rhs[b:e], device=dev), reltol=1e-9, maxiters=3000)
            xlg = nls.dist.gather_vector(xl_, pb.n, b)
            assert gl_["converged"] and abs(gl_["iters"] - ilo.iters) <= 6 and ilo.iters < iref.iters / 2
            assert np.linalg.norm(xlg - xlo) <= 1e-6 * np.linalg.norm(xlo)
            assert abs(gl_["rnorm0"] - np.linalg.norm(Mblock(rhs))) <= 1e-9 * gl_["rnorm0"]    # the preconditioned norm, all-reduced
        note("gmres_left_ilu0", xlg)

        # ---------------- the Brusselator V-cycle on two ranks: every level split by lines (slab boundaries stay even, the
        # transfers use the problem's own one-line periodic halo), coarsest level gathered and solved redundantly — one
        # V-cycle equals the serial oracle's, GMRES and the TrustRegion solve take the oracle's counts
        Nm = 32
        rbm, PBm = R.Brusselator2D(Nm), nls.Brusselator2D(Nm)
        jm0, jm1 = nls.partition_range(Nm, 1, world, rank)
        idm = np.array([i + Nm * (jm0 + jl) + Nm * Nm * s_ for s_ in range(2) for jl in range(jm1 - jm0) for i in range(Nm)])
        ubm = rbm.u0() + 0.05 * np.random.default_rng(1).standard_normal(rbm.n)
        rhsb = np.random.default_rng(2).standard_normal(rbm.n)
        Mob = R.BrusselatorMultigrid(rbm, ubm, 2, 8)
        Jbm = rbm.jac(ubm)
        ubml = torch.tensor(ubm[idm], device=dev)
        opb = nls.StatefulJacobianOperator(nls.JacobianOperator(nls.NonlinearProblem(PBm)), ubml)
        Gb = nls.GMRES(idm.size, restart=30).set_operator(opb)
        Gb.set_multigrid_preconditioner(PBm, ubml, nu=2, coarse_max=8)
        xb1, _ = Gb.solve(torch.tensor(rhsb[idm], device=dev), fixed_iters=1)
        xob1, _ = R.gmres(lambda z: Jbm @ z, rhsb, restart=30, fixed_iters=1, M=Mob, ortho="cgs2")
        assert np.linalg.norm(xb1.cpu().numpy() - xob1[idm]) <= 1e-9 * np.linalg.norm(xob1[idm])
        xbf, gbf = Gb.solve(torch.tensor(rhsb[idm], device=dev), abstol=0.0, reltol=1e-9, maxiters=300)
        xob, iob = R.gmres(lambda z: Jbm @ z, rhsb, rtol=1e-9, restart=30, itmax=300, M=Mob, ortho="cgs2")
        assert gbf["converged"] and abs(gbf["iters"] - iob.iters) <= 1 and gbf["iters"] <= 9
        note("brus_mg_gmres", xbf.cpu().numpy())
        kwb = dict(gmres_restart=30, maxiters=300)
        refbm = R.solve(rbm, R.TrustRegion(linsolve=R.KrylovJL_GMRES(precs=R.MultigridPrecs(2, 8), **kwb), concrete_jac=True),
                        abstol=1e-8, maxiters=40)
        solbm = nls.solve(nls.NonlinearProblem(PBm, u0=PBm.initial_guess(device=True)),
                          nls.TrustRegion(linsolve=nls.KrylovJL_GMRES(precs=nls.MultigridPrecs(2, 8), **kwb), concrete_jac=True),
                          abstol=1e-8, maxiters=40, store_trace=True)
        assert solbm.retcode == "Success" == R.RETCODE_NAMES[refbm.retcode] and solbm.stats.nsteps == refbm.stats.nsteps
        assert [t["accepted"] for t in solbm.trace] == [t["accepted"] for t in refbm.trace]
        assert np.max(np.abs(solbm.u.cpu().numpy() - refbm.u[idm])) <= 1e-6 * np.max(np.abs(refbm.u))
        note("brus_mg_tr", solbm.u.cpu().numpy())

        # ---------------- the multigrid V-cycle behind the `precs` hook on a row-partitioned hierarchy: every level split by
        # grid lines, ghost lines of the neighbouring levels gathered for the transfers, coarsest level solved redundantly —
        # the arithmetic must not depend on the partition: one V-cycle equals the serial oracle's to rounding, and the
        # Newton–Krylov solve takes the oracle's step and iteration counts
        nsm = 96
        pbm, Pm = R.Bratu2D(nsm, 6.0), nls.Bratu2D(nsm, 6.0)
        bm, em = Pm.row_begin, Pm.row_begin + Pm.n_local
        xs = np.arange(1, nsm + 1) / (nsm + 1)
        X, Y = np.meshgrid(xs, xs)
        um = (0.8 * np.sin(np.pi * X) * np.sin(np.pi * Y)).ravel()
        rhs_m = np.random.default_rng(3).standard_normal(pbm.n)
        Mo = R.BratuMultigrid(pbm, um, 2, 15)
        Jm = pbm.jac(um)
        uml = torch.tensor(um[bm:em], device=dev)
        opm = nls.StatefulJacobianOperator(nls.JacobianOperator(nls.NonlinearProblem(Pm)), uml)
        Gm = nls.GMRES(em - bm, restart=30).set_operator(opm)
        Gm.set_multigrid_preconditioner(Pm, uml, nu=2, coarse_max=15)
        x1, _ = Gm.solve(torch.tensor(rhs_m[bm:em], device=dev), fixed_iters=1)
        xo1, _ = R.gmres(lambda z: Jm @ z, rhs_m, restart=30, fixed_iters=1, M=Mo, ortho="cgs2")
        x1g = nls.dist.gather_vector(x1, pbm.n, bm)
        assert np.linalg.norm(x1g - xo1) <= 1e-10 * np.linalg.norm(xo1)
        xf, gf = Gm.solve(torch.tensor(rhs_m[bm:em], device=dev), abstol=0.0, reltol=1e-9, maxiters=300)
        xo, io = R.gmres(lambda z: Jm @ z, rhs_m, rtol=1e-9, restart=30, itmax=300, M=Mo, ortho="cgs2")
        assert gf["converged"] and abs(gf["iters"] - io.iters) <= 1 and gf["iters"] <= 9
        note("mg_gmres", nls.dist.gather_vector(xf, pbm.n, bm))
        refm = R.solve(pbm, R.NewtonRaphson(linsolve=R.KrylovJL_GMRES(precs=R.MultigridPrecs(2, 15)), forcing=R.EisenstatWalkerForcing2()),
                       abstol=1e-9, maxiters=50)
        solm = nls.solve(nls.NonlinearProblem(Pm, u0=torch.zeros(em - bm, dtype=torch.float64, device=dev)),
                         nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(precs=nls.MultigridPrecs(2, 15)),
                                           forcing=nls.EisenstatWalkerForcing2()), abstol=1e-9, maxiters=50)
        assert solm.retcode == "Success" and solm.stats.nsteps == refm.stats.nsteps
        assert abs(solm.stats.gmres_iters - refm.stats.gmres_iters) <= 2
        assert np.max(np.abs(nls.dist.gather_vector(solm.u, pbm.n, bm) - refm.u)) <= 5e-7
        assert ctx.comm_peer_status()[1] == 0        # no device-side time-outs
        q.put((rank, "ok", digest))
    except Exception:
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc(), {}))
    finally:
        dist.destroy_process_group()


def _worker4(rank, world, port, q, transport="torch"):
    """Four ranks on one GPU: every rank has two DISTINCT neighbours (with two ranks the one peer is both), the all-reduce
    combines four contributions, the periodic Brusselator ring closes over four slabs. A compact pass over the partitioned
    operators and one solve per algorithm family, each against the serial oracle."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import nonlinearsolve_jl_amd as nls
        from oracle import reference_restatement as R
        torch.cuda.set_device(0)
        ctx = nls.Context(device=0)
        nls.set_default_context(ctx)
        nls.dist.init_comm(ctx, transport)
        assert ctx.comm_info()[1:] == (world, rank)
        dev = torch.device("cuda:0")
        rng = np.random.default_rng(11)
        # Bratu 32²: 8 lines per rank
        ns = 32
        pb, P = R.Bratu2D(ns), nls.Bratu2D(ns)
        b, e = P.row_begin, P.row_begin + P.n_local
        u, v = 0.2 * rng.standard_normal(pb.n), rng.standard_normal(pb.n)
        ul, vl = torch.tensor(u[b:e], device=dev), torch.tensor(v[b:e], device=dev)
        assert np.allclose(P.jvp(vl, ul).cpu().numpy(), pb.jvp(v, u)[b:e], rtol=1e-13, atol=1e-12)
        J = P.jac_csr()
        P.jac_values(ul, J)
        Jo = pb.jac(u)
        assert np.allclose(J.matvec(vl).cpu().numpy(), (Jo @ v)[b:e], rtol=1e-13, atol=1e-11)
        assert np.allclose(J.rmatvec(vl).cpu().numpy(), (Jo.T @ v)[b:e], rtol=1e-13, atol=1e-11)
        assert np.allclose(J.colsumsq(like=vl).cpu().numpy(), np.asarray(Jo.multiply(Jo).sum(axis=0)).ravel()[b:e], rtol=1e-13)
        rhs = rng.standard_normal(pb.n)
        xref, iref = R.gmres(lambda z: Jo @ z, rhs, rtol=1e-9, restart=30, itmax=3000)
        x, gi = nls.GMRES(e - b, restart=30).set_operator(J).solve(torch.tensor(rhs[b:e], device=dev), reltol=1e-9, maxiters=3000)
        xg = nls.dist.gather_vector(x, pb.n, b)
        assert gi["converged"] and np.linalg.norm(xg - xref) <= 1e-7 * np.linalg.norm(xref)
        ref = R.solve(pb, R.NewtonRaphson(linsolve=R.KrylovJL_GMRES(), forcing=R.EisenstatWalkerForcing2()), abstol=1e-9, maxiters=50)
        for concrete in (False, True):
            sol = nls.solve(nls.NonlinearProblem(P, u0=torch.zeros(e - b, dtype=torch.float64, device=dev)),
                            nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(), forcing=nls.EisenstatWalkerForcing2(),
                                              concrete_jac=concrete), abstol=1e-9, maxiters=50)
            assert sol.retcode == "Success" and abs(sol.stats.nsteps - ref.stats.nsteps) <= 1
            assert np.max(np.abs(nls.dist.gather_vector(sol.u, pb.n, b) - ref.u)) <= 1e-7
        # multigrid on four slabs
        refm = R.solve(pb, R.NewtonRaphson(linsolve=R.KrylovJL_GMRES(precs=R.MultigridPrecs(2, 8)), forcing=R.EisenstatWalkerForcing2()),
                       abstol=1e-9, maxiters=50)
        solm = nls.solve(nls.NonlinearProblem(P, u0=torch.zeros(e - b, dtype=torch.float64, device=dev)),
                         nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(precs=nls.MultigridPrecs(2, 8)),
                                           forcing=nls.EisenstatWalkerForcing2()), abstol=1e-9, maxiters=50)
        assert solm.retcode == "Success" and solm.stats.nsteps == refm.stats.nsteps
        assert np.max(np.abs(nls.dist.gather_vector(solm.u, pb.n, b) - refm.u)) <= 5e-7
        # LevenbergMarquardt (Krylov normal form) on four slabs of a 12² grid
        Pl = nls.Bratu2D(12)
        bl, el = Pl.row_begin, Pl.row_begin + Pl.n_local
        kl_ = dict(gmres_restart=60, maxiters=600)
        reflm = R.solve(R.Bratu2D(12), R.LevenbergMarquardt(linsolve=R.KrylovJL_GMRES(**kl_)), abstol=1e-8, maxiters=100)
        sollm = nls.solve(nls.NonlinearProblem(Pl, u0=torch.zeros(el - bl, dtype=torch.float64, device=dev)),
                          nls.LevenbergMarquardt(linsolve=nls.KrylovJL_GMRES(**kl_)), abstol=1e-8, maxiters=100)
        assert sollm.retcode == "Success" and sollm.stats.nsteps == reflm.stats.nsteps
        assert np.max(np.abs(sollm.u.cpu().numpy() - reflm.u[bl:el])) <= 1e-7
        # Brusselator 32²: periodic ring over four slabs; concrete (coloured) J, TrustRegion, multigrid precs
        N = 32
        rb, PB = R.Brusselator2D(N), nls.Brusselator2D(N)
        j0, j1 = nls.partition_range(N, 1, world, rank)
        idx = np.array([i + N * (j0 + jl) + N * N * s_ for s_ in range(2) for jl in range(j1 - j0) for i in range(N)])
        ub, vb = rb.u0() + 0.05 * rng.standard_normal(rb.n), rng.standard_normal(rb.n)
        ubl, vbl = torch.tensor(ub[idx], device=dev), torch.tensor(vb[idx], device=dev)
        assert np.allclose(PB.residual(ubl).cpu().numpy(), rb.f(ub)[idx], rtol=1e-12, atol=1e-9)
        assert np.allclose(PB.jvp(vbl, ubl).cpu().numpy(), rb.jvp(vb, ub)[idx], rtol=1e-12, atol=1e-8)
        assert np.allclose(PB.vjp(vbl, ubl).cpu().numpy(), rb.vjp(vb, ub)[idx], rtol=1e-12, atol=1e-8)
        JB = PB.jac_csr()
        ncol = PB.jac_values(ubl, JB, colored=True)
        Jbo = rb.jac(ub).tocsr()
        assert 6 <= ncol <= 14
        assert np.allclose(JB.matvec(vbl).cpu().numpy(), (Jbo @ vb)[idx], rtol=1e-12, atol=1e-7)
        assert np.allclose(JB.rmatvec(vbl).cpu().numpy(), (Jbo.T @ vb)[idx], rtol=1e-12, atol=1e-7)
        kwb = dict(gmres_restart=30, maxiters=300)
        refb = R.solve(rb, R.TrustRegion(linsolve=R.KrylovJL_GMRES(precs=R.MultigridPrecs(2, 8), **kwb), concrete_jac=True),
                       abstol=1e-8, maxiters=40)
        solb = nls.solve(nls.NonlinearProblem(PB, u0=PB.initial_guess(device=True)),
                         nls.TrustRegion(linsolve=nls.KrylovJL_GMRES(precs=nls.MultigridPrecs(2, 8), **kwb), concrete_jac=True,
                                         jac_colored=True), abstol=1e-8, maxiters=40, store_trace=True)
        assert solb.retcode == "Success" == R.RETCODE_NAMES[refb.retcode] and solb.stats.nsteps == refb.stats.nsteps
        assert [t["accepted"] for t in solb.trace] == [t["accepted"] for t in refb.trace]
        assert np.max(np.abs(solb.u.cpu().numpy() - refb.u[idx])) <= 1e-6 * np.max(np.abs(refb.u))
        assert ctx.comm_peer_status()[1] == 0
        q.put((rank, "ok", {}))
    except Exception:
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc(), {}))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,transport", [(4, "torch"), (4, "peer"), (8, "peer")])
def test_four_and_eight_ranks_on_one_gpu(world, transport):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker4, args=(r, world, port, q, transport)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=280) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res


def _run_two_ranks(transport):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, transport)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=280) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res
    return {r[0]: r[2] for r in res}


_DIGESTS = {}


def test_two_ranks_on_one_gpu_callback_comm():
    _DIGESTS["torch"] = _run_two_ranks("torch")


def test_two_ranks_on_one_gpu_peer_comm():
    """The same programme with the peer-mapped (hipIpc) all-reduce and halo exchange kernels — the two processes map each
    other's arenas on the shared GPU — must reproduce the callback transport bit for bit (fixed rank-order combination)."""
    _DIGESTS["peer"] = _run_two_ranks("peer")
    if "torch" in _DIGESTS:
        assert _DIGESTS["peer"] == _DIGESTS["torch"]


def _timeout_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", NK_PEER_TIMEOUT_MS="300")
    import time
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import nonlinearsolve_jl_amd as nls
        torch.cuda.set_device(0)
        ctx = nls.Context(device=0)
        nls.set_default_context(ctx)
        assert nls.dist.init_comm(ctx, "peer") == "peer+torch"
        dev = torch.device("cuda:0")
        P = nls.Bratu2D(24)
        nl = P.n_local
        J = P.jac_csr()
        P.jac_values(torch.zeros(nl, dtype=torch.float64, device=dev), J)
        G = nls.GMRES(nl, restart=10).set_operator(J)
        b = torch.ones(nl, dtype=torch.float64, device=dev)
        x, gi = G.solve(b, fixed_iters=10)                    # both ranks in step: fine
        assert ctx.comm_peer_status()[1] == 0
        dist.barrier()
        if rank == 1:
            time.sleep(2.5)                                   # a stalled rank: far beyond the 0.3 s bound
        msg = "no error"
        try:
            G.solve(b, fixed_iters=10)
        except nls.NKError as ex:
            msg = str(ex)
        # the rank that waited saw the time-out and its solve FAILED with NK_E_COMM instead of returning numbers built on a
        # missing contribution (round 2: counted, ignored, Success)
        q.put((rank, msg, ctx.comm_peer_status()[1]))
    except Exception:
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc(), -1))
    finally:
        dist.destroy_process_group()


def test_a_stalled_rank_fails_the_solve_instead_of_corrupting_it():
    """Peer-mapped collectives bound every device-side wait (NK_PEER_TIMEOUT_MS); a time-out is counted in the arena AND
    surfaces at the next host synchronisation point of the solve as NK_E_COMM (k_backsolve publishes the counter)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_timeout_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r[0]: r for r in (q.get(timeout=200) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
    assert "timed out" in res[0][1] and res[0][2] > 0, res


def _rccl_worker(q):
    sys.path.insert(0, ROOT)
    os.environ["NK_FORCE_COLLECTIVES"] = "1"
    try:
        import nonlinearsolve_jl_amd as nls
        from oracle import reference_restatement as R
        ctx = nls.Context(device=0)
        nls.set_default_context(ctx)
        ctx.comm_init_rccl(1, 0, nls.comm_unique_id())
        assert ctx.comm_info() == (1, 1, 0)
        prob = nls.NonlinearProblem(nls.Bratu2D(24))
        sol = nls.solve(prob, nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(), forcing=nls.EisenstatWalkerForcing2()),
                        abstol=1e-9, maxiters=50)
        ref = R.solve(R.Bratu2D(24), R.NewtonRaphson(linsolve=R.KrylovJL_GMRES(), forcing=R.EisenstatWalkerForcing2()),
                      abstol=1e-9, maxiters=50)
        assert sol.retcode == "Success" and np.max(np.abs(sol.u - ref.u)) <= 1e-7
        assert sol.stats.allreduces > 50  # ncclAllReduce really ran (sum and max) on the compute stream
        q.put("ok")
    except Exception:
        import traceback
        q.put("FAIL: " + traceback.format_exc())


def test_rccl_entry_points_world1():
    """ncclGetUniqueId / ncclCommInitRank / ncclAllReduce(sum,max) through the dlopen'ed librccl on one GPU."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(q,))
    p.start()
    res = q.get(timeout=280)
    p.join(timeout=60)
    assert res == "ok", res


# ----------------------------------------------------------------------------- the resident matrix-powers kernel on several ranks
def _powers_worker(rank, world, port, q, ns, pw_ranks):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", NK_PW_RANKS=pw_ranks,
                      NK_DEVICE_SHARED="0")   # (the ranks DO share this GPU: at these sizes all their workgroups fit on it together)
    import hashlib
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import nonlinearsolve_jl_amd as nls
        from oracle import reference_restatement as R
        from tests.test_gpu_powers import _powers_ref
        torch.cuda.set_device(0)
        ctx = nls.Context(device=0)
        nls.set_default_context(ctx)
        assert nls.dist.init_comm(ctx, "peer") == "peer+torch"
        dev = torch.device("cuda:0")
        rng = np.random.default_rng(ns)
        pb, P = R.Bratu2D(ns), nls.Bratu2D(ns)
        b, e = P.row_begin, P.row_begin + P.n_local
        u = 0.1 * rng.standard_normal(pb.n)
        Jg = pb.jac(u).tocsr()
        J = P.jac_csr()
        P.jac_values(torch.tensor(u[b:e], device=dev), J)
        x = rng.standard_normal(pb.n)
        lam = float(abs(Jg).sum(axis=1).max())
        s = 15
        theta = (0.5 + 0.4 * np.cos(np.arange(s))) * lam
        scale = 2.0 / lam
        ref = _powers_ref(Jg, x, s, theta, scale)[:, b:e]
        xl = torch.tensor(x[b:e], device=dev)
        for rep in range(3):   # (repeated launches: epochs and buffer pairs keep them apart)
            Y, resident = J.powers(xl, s, theta=theta, scale=scale)
            assert resident == (pw_ranks == "1"), (resident, pw_ranks)
            assert np.array_equal(Y.cpu().numpy(), ref), f"rep {rep}: first differing power " \
                f"{int(np.argmax((Y.cpu().numpy() != ref).any(axis=1)))}"
        # the headline protocol on the partitioned matrix: the blocks' operator applications are ONE launch with ONE exchange
        # protocol each — 2 per fixed-work step of two blocks, + the residual's exchange (31 with the streaming kernel)
        prob = nls.NonlinearProblem(P, u0=torch.zeros(e - b, dtype=torch.float64, device=dev))
        cch = nls.init(prob, nls.NewtonRaphson(linsolve=nls.KrylovJL_GMRES(gmres_restart=30, maxiters=30, fixed_iters=30),
                                               concrete_jac=True), abstol=1e-300, maxiters=10 ** 6)
        cch.step()
        h0 = cch.stats.halo_exchanges
        for _ in range(3):
            cch.step()
        halo_step = (cch.stats.halo_exchanges - h0) / 3.0
        assert halo_step == (3.0 if pw_ranks == "1" else 31.0), halo_step
        ug = nls.dist.gather_vector(cch.u, pb.n, b)
        digest = hashlib.sha1(np.ascontiguousarray(ug).tobytes()).hexdigest()
        cch.close()
        assert ctx.comm_peer_status()[1] == 0
        q.put((rank, "ok", digest))
    except Exception:
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc(), ""))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,ns", [(2, 64), (4, 64), (8, 128), (2, 512)])
def test_resident_matrix_powers_on_several_ranks(world, ns):
    """nk_powers.hip with the matrix row-partitioned over 2 / 4 / 8 ranks (processes on one GPU, peer-mapped arenas): a rank's
    first / last band hands its boundary slice to the rank next door through that rank's arena. Every column equals the serial
    oracle's bit for bit; four fixed-work Newton steps leave the SAME iterate as with the streaming kernel (NK_PW_RANKS=0);
    one exchange protocol per launch instead of one per operator application."""
    out = {}
    for pw_ranks in ("1", "0"):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_powers_worker, args=(r, world, port, q, ns, pw_ranks)) for r in range(world)]
        for p in procs:
            p.start()
        res = [q.get(timeout=280) for _ in range(world)]
        for p in procs:
            p.join(timeout=60)
        assert all(r[1] == "ok" for r in res), res
        out[pw_ranks] = {r[2] for r in res}
    assert out["1"] == out["0"] and len(out["1"]) == 1
