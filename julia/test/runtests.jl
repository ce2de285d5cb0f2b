# Smoke tests of the binding for a machine that has Julia, an MI355X and libmi355x_nk.so (MI355X_NK_LIB points at it):
#     julia --project=julia -e 'using Pkg; Pkg.test()'
# They restate, through the binding, fixtures of the reference's own test-suite — the ones the Python host mirror runs in
# tests/test_gpu_solvers.py against the oracle. NEVER executed in the build container (no Julia there).
using Test, LinearAlgebra, SparseArrays
using MI355XNewtonKrylov
import MI355XNewtonKrylov: DeviceCSR, same_pattern, update_values!
using NonlinearSolveFirstOrder, SciMLBase
import LinearSolve

@testset "DeviceCSR: CSC ingest, SpMV, pattern identity" begin
    A = sprand(200, 200, 0.05) + 4I
    x = randn(200)
    M = DeviceCSR(A)
    y = similar(x)
    mul!(y, M, x)
    @test y ≈ A * x
    @test same_pattern(M, A)
    B = copy(A)
    B.nzval .*= 2.0
    update_values!(M, B)
    mul!(y, M, x)
    @test y ≈ B * x
    # a structure change in place with the same nnz (arrays reused) is NOT the same pattern (ADVICE r03)
    C = copy(A)
    i = findfirst(j -> C.colptr[j + 1] - C.colptr[j] >= 2, 1:200)
    C.rowval[C.colptr[i]] = C.rowval[C.colptr[i]] == 1 ? 2 : 1
    @test !same_pattern(M, C) || C.rowval == A.rowval
end

@testset "seam 1: MI355XGMRES as linsolve (rootfind_tests__item1.jl / misc_tests__item6.jl)" begin
    f(u, p) = u .* u .- p
    prob = NonlinearProblem(f, [1.0, 1.0], 2.0)
    sol = solve(prob, NewtonRaphson(; linsolve = MI355XGMRES()); abstol = 1e-9)
    @test SciMLBase.successful_retcode(sol)
    @test sol.u ≈ [sqrt(2.0), sqrt(2.0)] atol = 1e-9
    cache = init(prob, NewtonRaphson(; linsolve = MI355XGMRES(), forcing = EisenstatWalkerForcing2()))
    fc = cache.forcing_cache
    @test SciMLBase.successful_retcode(solve!(cache))
    @test fc.η != fc.p.η₀
    reinit!(cache; p = 3.0)
    @test fc.η == fc.p.η₀
    @test solve!(cache).u ≈ [sqrt(3.0), sqrt(3.0)]
end

@testset "seam 3: the whole solver behind one ccall (Bratu 64²)" begin
    P = bratu2d(64)                                  # the built-in device problem (kernels: residual, JVP, Jacobian fill)
    prob = NonlinearProblem(mi355x_function(P), zeros(64 * 64), nothing)
    sol = solve(prob, MI355XNewtonKrylovAlg(; problem = P, forcing = true); abstol = 1e-8, maxiters = 50)
    @test SciMLBase.successful_retcode(sol)
    @test norm(sol.resid, Inf) <= 1e-8
end
