# Reference-generated golden vectors for the MI355X Newton–Krylov path: runs NonlinearSolve.jl ITSELF (the package under
# /root/reference, or the registered release) on the path's configurations and writes one JSON file per case into
# tests/golden/reference/. tests/test_reference_golden.py consumes them when present — oracle (CPU) and device (GPU) against the
# reference's own iterates, residual histories, statistics and return codes — and is skipped when they are absent.
#
# UNEXECUTED in the build container: Julia is not installed there and there is no network (SURVEY.md §0). This script is the
# route from "parity partial" (oracle pinned by the reference's known-answer tests) to "parity green" (oracle AND device pinned
# by the reference's own output): run it once in any environment that has Julia ≥ 1.10 and the packages below, commit the
# JSON files it writes.
#
#   julia --project=/path/with/NonlinearSolve tests/golden/make_reference_golden.jl [outdir]
#
# Needs: NonlinearSolve (≥ 4.27: the version under /root/reference), LinearSolve, SparseArrays, LinearAlgebra, SciMLBase,
# ADTypes, SparseConnectivityTracer + SparseMatrixColorings (case `brusselator32_sparse_ad` only; skipped if not installed).
# No JSON package: the writer below emits the few types it needs.
using NonlinearSolve, LinearSolve, SparseArrays, LinearAlgebra, SciMLBase
import NonlinearSolve: NonlinearSolveBase

const OUT = length(ARGS) >= 1 ? ARGS[1] : joinpath(@__DIR__, "reference")
mkpath(OUT)

# ------------------------------------------------------------------ a dependency-free JSON writer
json(x::Nothing) = "null"
json(x::Bool) = x ? "true" : "false"
json(x::Integer) = string(x)
json(x::AbstractFloat) = isfinite(x) ? repr(Float64(x)) : (isnan(x) ? "\"NaN\"" : (x > 0 ? "\"Infinity\"" : "\"-Infinity\""))
json(x::Union{AbstractString, Symbol}) = "\"" * escape_string(string(x)) * "\""
json(x::AbstractArray) = "[" * join((json(v) for v in vec(collect(x))), ",") * "]"
json(x::Union{NamedTuple, AbstractDict}) = "{" * join(("\"$(k)\":" * json(v) for (k, v) in pairs(x)), ",") * "}"
write_case(name, payload) = open(io -> write(io, json(payload)), joinpath(OUT, name * ".json"), "w")

# ------------------------------------------------------------------ run one configuration through the iterator interface
# `init` → `step!` until termination: the residual norm after every step is what the parity tests compare step by step
# (lib/NonlinearSolveFirstOrder/src/solve.jl:325-465; Base/src/solve.jl:360-387,835-859).
function run_case(name, prob, alg; kwargs...)
    meta = (; julia = string(VERSION), nonlinearsolve = string(pkgversion(NonlinearSolve)),
        linearsolve = string(pkgversion(LinearSolve)))
    try
        cache = init(prob, alg; kwargs...)
        fnorm_inf = Float64[norm(cache.fu, Inf)]
        fnorm_2 = Float64[norm(cache.fu, 2)]
        while NonlinearSolveBase.not_terminated(cache)
            step!(cache)
            push!(fnorm_inf, norm(NonlinearSolveBase.get_fu(cache), Inf))
            push!(fnorm_2, norm(NonlinearSolveBase.get_fu(cache), 2))
        end
        sol = solve!(cache)       # settles retcode / best-iterate rollback exactly as `solve` does
        st = sol.stats
        write_case(name, (; name, meta, retcode = string(sol.retcode), u = vec(sol.u), resid = vec(sol.resid),
            fnorm_inf, fnorm_2, nsteps = st.nsteps, nf = st.nf, njacs = st.njacs, nfactors = st.nfactors, nsolve = st.nsolve))
        println(rpad(name, 40), " ", sol.retcode, "  steps=", st.nsteps, "  |f|inf=", fnorm_inf[end])
    catch err
        write_case(name * ".FAILED", (; name, meta, error = sprint(showerror, err)))
        println(rpad(name, 40), " FAILED: ", sprint(showerror, err))
    end
end

# ------------------------------------------------------------------ C1: quadratic, N = 1000 (BASELINE.json configs[0])
quadratic!(du, u, p) = (du .= u .* u .- p; nothing)
run_case("c1_quadratic1000_newton", NonlinearProblem(quadratic!, ones(1000), 2.0), NewtonRaphson())
run_case("c1_quadratic1000_trustregion", NonlinearProblem(quadratic!, ones(1000), 2.0), TrustRegion())

# ------------------------------------------------------------------ Bratu 2-D (SURVEY.md §8d): F = (4u − Σ nbrs)/h² − λ eᵘ, scaled by h²
# (the library's Bratu2D(n, λ) with scale = 0: c_lap = 1, c_exp = λ h²), lexicographic k = (j − 1) n + i, zero Dirichlet data
struct Bratu
    n::Int
    λ::Float64
end
function bratu_lap(u, n, i, j)
    s = 4.0 * u[i, j]
    i > 1 && (s -= u[i - 1, j]); i < n && (s -= u[i + 1, j])
    j > 1 && (s -= u[i, j - 1]); j < n && (s -= u[i, j + 1])
    return s
end
function (b::Bratu)(du, u, p)
    n = b.n; h2 = 1.0 / (n + 1)^2
    U = reshape(u, n, n); D = reshape(du, n, n)
    for j in 1:n, i in 1:n
        D[i, j] = bratu_lap(U, n, i, j) - b.λ * h2 * exp(U[i, j])
    end
    return nothing
end
function bratu_jvp(b::Bratu)
    return function (Jv, v, u, p)
        n = b.n; h2 = 1.0 / (n + 1)^2
        U = reshape(u, n, n); V = reshape(v, n, n); W = reshape(Jv, n, n)
        for j in 1:n, i in 1:n
            W[i, j] = bratu_lap(V, n, i, j) - b.λ * h2 * exp(U[i, j]) * V[i, j]
        end
        return nothing
    end
end
function bratu_pattern(n)
    I = Int[]; J = Int[]
    k(i, j) = (j - 1) * n + i
    for j in 1:n, i in 1:n
        push!(I, k(i, j)); push!(J, k(i, j))
        i > 1 && (push!(I, k(i, j)); push!(J, k(i - 1, j)))
        i < n && (push!(I, k(i, j)); push!(J, k(i + 1, j)))
        j > 1 && (push!(I, k(i, j)); push!(J, k(i, j - 1)))
        j < n && (push!(I, k(i, j)); push!(J, k(i, j + 1)))
    end
    return sparse(I, J, ones(length(I)), n * n, n * n)
end
function bratu_jac(b::Bratu)
    return function (Jm, u, p)
        n = b.n; h2 = 1.0 / (n + 1)^2
        k(i, j) = (j - 1) * n + i
        fill!(nonzeros(Jm), 0.0)
        for j in 1:n, i in 1:n
            r = k(i, j)
            Jm[r, r] = 4.0 - b.λ * h2 * exp(u[r])
            i > 1 && (Jm[r, k(i - 1, j)] = -1.0); i < n && (Jm[r, k(i + 1, j)] = -1.0)
            j > 1 && (Jm[r, k(i, j - 1)] = -1.0); j < n && (Jm[r, k(i, j + 1)] = -1.0)
        end
        return nothing
    end
end
bratu_problem(n; jac = false, jvp = false, vjp = false) = begin
    b = Bratu(n, 6.0)
    f = NonlinearFunction{true}(b; (jac ? (; jac = bratu_jac(b), jac_prototype = bratu_pattern(n)) : (;))...,
        (jvp ? (; jvp = bratu_jvp(b)) : (;))..., (vjp ? (; vjp = bratu_jvp(b)) : (;))...)   # (J is symmetric: vjp = jvp)
    NonlinearProblem(f, zeros(n * n), nothing)
end

# C2 (configs[1]): 256², NewtonRaphson, sparse concrete J, LinearSolve's default sparse factorisation
run_case("c2_bratu256_newton_direct", bratu_problem(256; jac = true), NewtonRaphson(); abstol = 1e-8, maxiters = 50)
run_case("bratu64_newton_direct", bratu_problem(64; jac = true), NewtonRaphson(); abstol = 1e-8, maxiters = 50)
# C3's protocol at a size the Krylov solver converges at without a preconditioner: GMRES(30) with restarts + Eisenstat–Walker
# through the custom JVP (matrix-free JacobianOperator) and through the concrete J
for (tag, kw, cj) in (("matfree", (; jvp = true), false), ("concrete", (; jac = true), true))
    run_case("bratu64_newton_gmres30_ew_" * tag, bratu_problem(64; kw...),
        NewtonRaphson(linsolve = KrylovJL_GMRES(gmres_restart = 30), forcing = EisenstatWalkerForcing2(), concrete_jac = cj);
        abstol = 1e-8, maxiters = 50)
    run_case("bratu64_newton_gmres30_" * tag, bratu_problem(64; kw...),
        NewtonRaphson(linsolve = KrylovJL_GMRES(gmres_restart = 30), concrete_jac = cj); abstol = 1e-8, maxiters = 50)
end
run_case("bratu64_trustregion_gmres30_matfree", bratu_problem(64; jvp = true, vjp = true),
    TrustRegion(linsolve = KrylovJL_GMRES(gmres_restart = 30)); abstol = 1e-8, maxiters = 50)
run_case("bratu64_trustregion_direct", bratu_problem(64; jac = true), TrustRegion(); abstol = 1e-8, maxiters = 50)

# ------------------------------------------------------------------ tridiagonal residual through a MatrixOperator (operator_jacobian.jl:11-29)
let N = 40
    Wmat = sparse(Tridiagonal(fill(-1.0, N - 1), fill(4.0, N), fill(-1.0, N - 1)))
    bvec = collect(1.0:N)
    resid!(F, z, p) = (mul!(F, Wmat, z); F .-= bvec; nothing)
    try
        mop = NonlinearSolve.SciMLBase.SciMLOperators.MatrixOperator(copy(Wmat))
        prob = NonlinearProblem(NonlinearFunction(resid!; jac_prototype = mop), zeros(N))
        run_case("tridiagonal40_matrixoperator_gmres", prob, NewtonRaphson(linsolve = KrylovJL_GMRES()))
    catch err
        println("tridiagonal40: MatrixOperator not reachable here (", sprint(showerror, err), "); using the sparse prototype")
    end
    prob = NonlinearProblem(NonlinearFunction(resid!; jac = (J, u, p) -> (J .= Wmat; nothing), jac_prototype = copy(Wmat)), zeros(N))
    run_case("tridiagonal40_sparse_gmres", prob, NewtonRaphson(linsolve = KrylovJL_GMRES(), concrete_jac = true))
    write_case("tridiagonal40_xref", (; name = "tridiagonal40_xref", u = Wmat \ bvec))
end

# ------------------------------------------------------------------ Brusselator N = 32 (sparsity_tests__item1.jl:7-55)
let N = 32
    xyd = range(0, stop = 1, length = N)
    bf(x, y) = (((x - 0.3)^2 + (y - 0.6)^2) <= 0.1^2) * 5.0
    lim(a) = a == N + 1 ? 1 : a == 0 ? N : a
    function brus!(du, u, p)
        A, B, alpha, dx = p
        alpha = alpha / dx^2
        @inbounds for I in CartesianIndices((N, N))
            i, j = Tuple(I)
            x, y = xyd[i], xyd[j]
            ip1, im1, jp1, jm1 = lim(i + 1), lim(i - 1), lim(j + 1), lim(j - 1)
            du[i, j, 1] = alpha * (u[im1, j, 1] + u[ip1, j, 1] + u[i, jp1, 1] + u[i, jm1, 1] - 4u[i, j, 1]) + B +
                          u[i, j, 1]^2 * u[i, j, 2] - (A + 1) * u[i, j, 1] + bf(x, y)
            du[i, j, 2] = alpha * (u[im1, j, 2] + u[ip1, j, 2] + u[i, jp1, 2] + u[i, jm1, 2] - 4u[i, j, 2]) +
                          A * u[i, j, 1] - u[i, j, 1]^2 * u[i, j, 2]
        end
        return nothing
    end
    u0 = zeros(N, N, 2)
    for I in CartesianIndices((N, N))
        x = xyd[I[1]]; y = xyd[I[2]]
        u0[I, 1] = 22 * (y * (1 - y))^(3 / 2)
        u0[I, 2] = 27 * (x * (1 - x))^(3 / 2)
    end
    p = (3.4, 1.0, 10.0, step(xyd))
    run_case("brusselator32_newton_dense_ad", NonlinearProblem(brus!, u0, p), NewtonRaphson(); abstol = 1e-8)
    run_case("brusselator32_trustregion_dense_ad", NonlinearProblem(brus!, u0, p), TrustRegion(); abstol = 1e-8)
    try
        @eval using SparseConnectivityTracer, ADTypes, SparseMatrixColorings
        fs = NonlinearFunction(brus!; sparsity = Base.invokelatest(getfield(Main, :TracerSparsityDetector)))
        run_case("brusselator32_newton_sparse_ad", NonlinearProblem(fs, u0, p), NewtonRaphson(); abstol = 1e-8)
        run_case("brusselator32_newton_sparse_ad_gmres", NonlinearProblem(fs, u0, p),
            NewtonRaphson(linsolve = KrylovJL_GMRES(), concrete_jac = true); abstol = 1e-8, reltol = 1e-8)
    catch err
        println("brusselator32 sparse AD cases skipped: ", sprint(showerror, err))
    end
end

# ------------------------------------------------------------------ the `precs` call protocol (test/Core/core_tests__item21.jl:10-37)
mutable struct CountingPrecs
    i::Int
    ps::Vector{Float64}
end
(c::CountingPrecs)(W, p = nothing) = (c.i += 1; push!(c.ps, Float64(p.p)); (LinearAlgebra.I, LinearAlgebra.I))
let
    f(u, p) = -(u .- 0.1) .^ 3
    try
        precs = CountingPrecs(0, Float64[])
        it = init(NonlinearProblem(f, [0.0, 0.0], 0), NewtonRaphson(linsolve = KrylovJL_GMRES(precs = precs), concrete_jac = false))
        iinit = precs.i
        sol = solve!(it)
        ifirst = precs.i
        reinit!(it; u0 = [0.0, 0.0], p = 1)
        ireinit = precs.i
        solve!(it)
        isecond = precs.i
        reinit!(it; p = 2)
        ireinit2 = precs.i
        solve!(it)
        write_case("precs_protocol_counts", (; name = "precs_protocol_counts", calls_at_init = iinit, calls_first_solve = ifirst - iinit,
            calls_by_reinit_u0 = ireinit - ifirst, calls_second_solve = isecond - ireinit, calls_by_reinit_p = ireinit2 - isecond,
            calls_third_solve = precs.i - ireinit2, p_seen = precs.ps, nsteps_first = sol.stats.nsteps))
        println("precs protocol: ", precs.i, " calls")
    catch err
        println("precs protocol case failed: ", sprint(showerror, err))
    end
end
println("golden files in ", OUT)
