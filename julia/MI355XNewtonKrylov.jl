# MI355XNewtonKrylov.jl — the reference-side binding a NonlinearSolve.jl maintainer would add.
#
# Written against include/mi355x_nk.h. Julia is not installed in the build container, so this file has
# never been executed; every LinearSolve/SciMLBase internal it touches is marked [EXT] (to confirm against
# LinearSolve 5.x). The C ABI itself is exercised from Python (ctypes) in tests/.
#
# Three seams (SURVEY.md §8b), in the order a maintainer would adopt them:
#   1. MI355XGMRES           — a `linsolve` backend:  NewtonRaphson(linsolve = MI355XGMRES())
#   2. mi355x_function(...)  — `f`/`jvp`/`vjp`/`jac` callbacks that run the built-in device kernels
#   3. MI355XNewtonKrylov()  — whole-solver extension algorithm (SciMLBase.__solve), pattern of
#                              ext/NonlinearSolvePETScExt.jl:38-167
module MI355XNewtonKrylov

using LinearAlgebra, SparseArrays
using SciMLBase: SciMLBase, ReturnCode, NonlinearProblem, NonlinearFunction
using NonlinearSolveBase: NonlinearSolveBase, AbstractNonlinearSolveAlgorithm, NLStats
import LinearSolve                       # [EXT]

const libnk = get(ENV, "MI355X_NK_LIB", "libmi355x_nk.so")

# ------------------------------------------------------------------ error handling
function nkcheck(status::Cint)
    status == 0 && return nothing
    msg = unsafe_string(@ccall libnk.nk_last_error()::Cstring)
    error("libmi355x_nk status $status: $msg")
end

# ------------------------------------------------------------------ handles with finalizers
mutable struct Ctx
    ptr::Ptr{Cvoid}
    function Ctx(device::Integer = 0; stream::Ptr{Cvoid} = C_NULL)
        out = Ref{Ptr{Cvoid}}(C_NULL)
        nkcheck(@ccall libnk.nk_ctx_create(device::Cint, stream::Ptr{Cvoid}, out::Ptr{Ptr{Cvoid}})::Cint)
        c = new(out[])
        finalizer(x -> @ccall(libnk.nk_ctx_destroy(x.ptr::Ptr{Cvoid})::Cint), c)
        return c
    end
end
const DEFAULT_CTX = Ref{Union{Nothing, Ctx}}(nothing)
default_ctx() = something(DEFAULT_CTX[], (DEFAULT_CTX[] = Ctx(); DEFAULT_CTX[]))

mutable struct DeviceCSR
    ptr::Ptr{Cvoid}
    n::Int
end
"""Upload a `SparseMatrixCSC{Float64,Int}` as it is (1-based Int64 colptr/rowval): converted to CSR once."""
function DeviceCSR(A::SparseMatrixCSC{Float64, Int}; ctx = default_ctx())
    out = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve A nkcheck(@ccall libnk.nk_csr_create_from_csc(ctx.ptr::Ptr{Cvoid}, size(A, 1)::Int64,
        nnz(A)::Int64, 64::Cint, 1::Cint, A.colptr::Ptr{Int64}, A.rowval::Ptr{Int64}, A.nzval::Ptr{Float64},
        out::Ptr{Ptr{Cvoid}})::Cint)
    m = DeviceCSR(out[], size(A, 1))
    finalizer(x -> @ccall(libnk.nk_csr_destroy(x.ptr::Ptr{Cvoid})::Cint), m)
    return m
end
function LinearAlgebra.mul!(y::Vector{Float64}, A::DeviceCSR, x::Vector{Float64})
    GC.@preserve x y nkcheck(@ccall libnk.nk_spmv(A.ptr::Ptr{Cvoid}, x::Ptr{Float64}, y::Ptr{Float64}, 0::Cint)::Cint)
    return y
end

mutable struct DeviceProblem
    ptr::Ptr{Cvoid}
    n::Int
end
function DeviceProblem(kind::Integer, params::Vector{Float64}; ctx = default_ctx())
    out = Ref{Ptr{Cvoid}}(C_NULL)
    nkcheck(@ccall libnk.nk_problem_create(ctx.ptr::Ptr{Cvoid}, kind::Cint, params::Ptr{Float64},
        length(params)::Cint, out::Ptr{Ptr{Cvoid}})::Cint)
    nl = Ref{Int64}(0)
    nkcheck(@ccall libnk.nk_problem_size(out[]::Ptr{Cvoid}, nl::Ptr{Int64}, C_NULL::Ptr{Int64}, C_NULL::Ptr{Int64})::Cint)
    p = DeviceProblem(out[], nl[])
    finalizer(x -> @ccall(libnk.nk_problem_destroy(x.ptr::Ptr{Cvoid})::Cint), p)
    return p
end
bratu2d(n; λ = 6.0, scale = 0.0, kw...) = DeviceProblem(2, Float64[n, λ, scale]; kw...)
brusselator2d(N; A = 3.4, B = 1.0, α = 10.0, dx = 1 / (N - 1), kw...) = DeviceProblem(3, Float64[N, A, B, α, dx]; kw...)

# ------------------------------------------------------------------ seam 2: f / jvp / vjp callbacks
"""`NonlinearFunction` whose `f`, `jvp`, `vjp` run the built-in device kernels (host arrays in/out).
Signatures follow lib/SciMLJacobianOperators/test/core_tests__item2.jl:38-40,62-67."""
function mi355x_function(P::DeviceProblem)
    f!(du, u, p) = (GC.@preserve du u nkcheck(@ccall libnk.nk_residual(P.ptr::Ptr{Cvoid}, u::Ptr{Float64},
        du::Ptr{Float64}, 0::Cint)::Cint); nothing)
    jvp!(Jv, v, u, p) = (GC.@preserve Jv v u nkcheck(@ccall libnk.nk_jvp(P.ptr::Ptr{Cvoid}, u::Ptr{Float64},
        v::Ptr{Float64}, Jv::Ptr{Float64}, 0::Cint)::Cint); nothing)
    vjp!(vJ, v, u, p) = (GC.@preserve vJ v u nkcheck(@ccall libnk.nk_vjp(P.ptr::Ptr{Cvoid}, u::Ptr{Float64},
        v::Ptr{Float64}, vJ::Ptr{Float64}, 0::Cint)::Cint); nothing)
    return NonlinearFunction{true}(f!; jvp = jvp!, vjp = vjp!)
end

# ------------------------------------------------------------------ seam 1: linsolve backend
"""
    MI355XGMRES(; gmres_restart = 30, ortho = :dcgs2)

`NewtonRaphson(linsolve = MI355XGMRES())`. NonlinearSolveBase only needs what
ext/NonlinearSolveBaseLinearSolveExt.jl:16-32,60-115 uses: `needs_concrete_A == false`, a cache with settable
`A`, `b`, `u`, `solve!` returning `u` + `retcode`, and `update_tolerances!(; reltol)`.
"""
Base.@kwdef struct MI355XGMRES <: LinearSolve.AbstractKrylovSubspaceMethod   # [EXT]
    gmres_restart::Int = 30
    ortho::Symbol = :dcgs2
end
LinearSolve.needs_concrete_A(::MI355XGMRES) = false                           # [EXT]

mutable struct GMRESWorkspace
    ptr::Ptr{Cvoid}
    n::Int
    csr::Union{Nothing, DeviceCSR}
end
const ORTHO = Dict(:mgs => 0, :cgs2 => 1, :cgs => 2, :dcgs2 => 3)

function LinearSolve.init_cacheval(alg::MI355XGMRES, A, b, u, Pl, Pr, maxiters::Int, abstol, reltol,
        verbose, assumptions)                                                  # [EXT signature]
    out = Ref{Ptr{Cvoid}}(C_NULL)
    nkcheck(@ccall libnk.nk_gmres_create(default_ctx().ptr::Ptr{Cvoid}, length(b)::Int64,
        alg.gmres_restart::Cint, ORTHO[alg.ortho]::Cint, out::Ptr{Ptr{Cvoid}})::Cint)
    w = GMRESWorkspace(out[], length(b), nothing)
    finalizer(x -> @ccall(libnk.nk_gmres_destroy(x.ptr::Ptr{Cvoid})::Cint), w)
    return w
end

# operator plumbing: a concrete sparse J goes to the device once per `A` assignment; a matrix-free operator
# (StatefulJacobianOperator, FunctionOperator, …) is applied through a C callback that calls `mul!`.
function matvec_trampoline(user::Ptr{Cvoid}, x::Ptr{Float64}, y::Ptr{Float64}, stream::Ptr{Cvoid})::Cint
    A, n = unsafe_pointer_to_objref(user)::Tuple{Any, Int}
    # device pointers: wrap as ROCArray views (AMDGPU.jl) — or stage through host buffers when A is a host operator
    xv = unsafe_wrap(Array, x, n); yv = unsafe_wrap(Array, y, n)                 # host-operator variant
    try
        mul!(yv, A, xv)
        return 0
    catch
        return 1
    end
end

function SciMLBase.solve!(cache::LinearSolve.LinearCache, alg::MI355XGMRES; kwargs...)   # [EXT]
    w = cache.cacheval::GMRESWorkspace
    A = cache.A
    if cache.isfresh                                                            # [EXT] set by `cache.A = …`
        if A isa SparseMatrixCSC
            w.csr = DeviceCSR(A)
            nkcheck(@ccall libnk.nk_gmres_set_operator_csr(w.ptr::Ptr{Cvoid}, w.csr.ptr::Ptr{Cvoid})::Cint)
        else
            cb = @cfunction(matvec_trampoline, Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}))
            ref = Ref{Any}((A, w.n)); cache.cacheval_keepalive = (cb, ref)      # keep both alive
            nkcheck(@ccall libnk.nk_gmres_set_operator_fn(w.ptr::Ptr{Cvoid}, cb::Ptr{Cvoid},
                pointer_from_objref(ref[])::Ptr{Cvoid})::Cint)
        end
        cache.isfresh = false
    end
    info = Ref(ntuple(_ -> zero(UInt8), 32))        # nk_gmres_info (4×Int32 + 2×Float64)
    b, u = cache.b, cache.u
    GC.@preserve b u nkcheck(@ccall libnk.nk_gmres_solve(w.ptr::Ptr{Cvoid}, b::Ptr{Float64}, u::Ptr{Float64},
        0::Cint, 0::Cint, cache.abstol::Float64, cache.reltol::Float64, cache.maxiters::Cint, 0::Cint,
        info::Ptr{Cvoid})::Cint)
    iters, _, converged, failed = reinterpret(Int32, collect(info[][1:16]))
    rc = failed != 0 ? ReturnCode.Failure : (converged != 0 ? ReturnCode.Success : ReturnCode.MaxIters)
    return SciMLBase.build_linear_solution(alg, u, nothing, cache; retcode = rc, iters = Int(iters))
end

# ------------------------------------------------------------------ seam 3: whole-solver plugin
"""
    MI355XNewtonKrylov(; algorithm = :NewtonRaphson, linsolve = :matfree, gmres_restart = 30, forcing = true, …)

`solve(NonlinearProblem(P::DeviceProblem-backed function, u0, p), MI355XNewtonKrylov())`: the whole loop stays
on the device behind one ccall (nk_newton_solve); returns a regular NonlinearSolution.
"""
Base.@kwdef struct MI355XNewtonKrylovAlg <: AbstractNonlinearSolveAlgorithm
    problem::DeviceProblem
    trust_region::Bool = false
    concrete_jac::Bool = false
    gmres_restart::Int = 30
    gmres_maxiters::Int = 300
    forcing::Bool = false
    radius_update_scheme::Int = 0
end

# nk_options mirrors include/mi355x_nk.h field for field (isbits ⇒ passable by Ref)
Base.@kwdef mutable struct NKOptions
    algorithm::Int32 = 0; linsolve::Int32 = 0; maxiters::Int32 = 1000; termination_norm::Int32 = 0
    abstol::Float64 = 0.0; reltol::Float64 = 0.0; maxtime::Float64 = 0.0
    gmres_restart::Int32 = 30; gmres_maxiters::Int32 = 300; gmres_ortho::Int32 = 3; gmres_fixed_iters::Int32 = 0
    lin_abstol::Float64 = -1.0; lin_reltol::Float64 = -1.0
    forcing::Int32 = 0; ew_safeguard::Int32 = 1
    ew_eta0::Float64 = 0.5; ew_eta_max::Float64 = 0.9; ew_gamma::Float64 = 0.9; ew_alpha::Float64 = 2.0
    ew_safeguard_threshold::Float64 = 0.1
    radius_update_scheme::Int32 = 0; max_shrink_times::Int32 = 32
    max_trust_radius::Float64 = 0.0; initial_trust_radius::Float64 = 0.0; step_threshold::Float64 = 1e-4
    shrink_threshold::Float64 = 0.25; expand_threshold::Float64 = 0.75; shrink_factor::Float64 = 0.25
    expand_factor::Float64 = 2.0
    patience_steps::Int32 = 100; max_stalled_steps::Int32 = 32
    patience_objective_multiplier::Float64 = 3.0; min_max_factor::Float64 = 1.3; protective_threshold::Float64 = 0.0
    store_trace::Int32 = 0; termination_mode::Int32 = 0
    cheb_degree::Int32 = 0; linesearch::Int32 = 0; cheb_ratio::Float64 = 0.0
    ls_c1::Float64 = 1e-4; ls_rho_hi::Float64 = 0.5; ls_rho_lo::Float64 = 0.1; ls_order::Int32 = 3; ls_maxiters::Int32 = 1000
    mg_nu::Int32 = 0; mg_coarse::Int32 = 0
end

const RETCODES = (ReturnCode.Default, ReturnCode.Success, ReturnCode.MaxIters, ReturnCode.Unstable,
    ReturnCode.Stalled, ReturnCode.InternalLinearSolveFailed, ReturnCode.ShrinkThresholdExceeded,
    ReturnCode.MaxTime, ReturnCode.Failure, ReturnCode.InternalLineSearchFailed)

function SciMLBase.__solve(prob::NonlinearProblem, alg::MI355XNewtonKrylovAlg, args...;
        abstol = nothing, reltol = nothing, maxiters = 1000, kwargs...)
    o = NKOptions(; algorithm = alg.trust_region ? 1 : 0, linsolve = alg.concrete_jac ? 1 : 0,
        maxiters = maxiters, abstol = something(abstol, 0.0), reltol = something(reltol, 0.0),
        gmres_restart = alg.gmres_restart, gmres_maxiters = alg.gmres_maxiters,
        forcing = alg.forcing ? 1 : 0, radius_update_scheme = alg.radius_update_scheme)
    u0 = Vector{Float64}(vec(prob.u0)); u = similar(u0); resid = similar(u0)
    stats = zeros(Int64, 9); rc = Ref{Cint}(0)
    GC.@preserve u0 u resid stats nkcheck(@ccall libnk.nk_newton_solve(alg.problem.ptr::Ptr{Cvoid},
        u0::Ptr{Float64}, 0::Cint, Ref(o)::Ptr{Cvoid}, u::Ptr{Float64}, resid::Ptr{Float64},
        stats::Ptr{Int64}, rc::Ptr{Cint})::Cint)
    return SciMLBase.build_solution(prob, alg, reshape(u, size(prob.u0)), reshape(resid, size(prob.u0));
        retcode = RETCODES[rc[] + 1], stats = NLStats(stats[1], stats[2], stats[3], stats[4], stats[5]),
        original = (; gmres_iters = stats[6], op_applies = stats[7], allreduces = stats[8]))
end

# ------------------------------------------------------------------------------------------ seam 4: ensembles of small systems
# The tutorial's `vectorized_solve(prob, SimpleNewtonRaphson(); backend = ROCBackend())`
# (docs/src/tutorials/nonlinear_solve_gpus.md:106-114) behind one ccall: `f_source` is HIP C++ defining
# `template <typename T> __device__ void nk_f(const T *u, const double *p, T *f)`; hiprtc specialises it into the solver kernel
# the way GPUCompiler specialises a Julia `f` into the KernelAbstractions kernel. p: nparams × nbatch (one column per system).
mutable struct EnsembleKernel
    ptr::Ptr{Cvoid}
    n::Int
    nparams::Int
    function EnsembleKernel(ctx::Ctx, f_source::String, n::Integer, nparams::Integer; has_jac::Bool = false)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        nkcheck(@ccall libnk.nk_batch_create(ctx.ptr::Ptr{Cvoid}, f_source::Cstring, n::Cint, nparams::Cint,
            (has_jac ? 1 : 0)::Cint, h::Ptr{Ptr{Cvoid}})::Cint)
        k = new(h[], n, nparams)
        finalizer(x -> @ccall(libnk.nk_batch_destroy(x.ptr::Ptr{Cvoid})::Cint), k)
        return k
    end
end

function vectorized_solve(k::EnsembleKernel, u0::Vector{Float64}, p::Matrix{Float64}; alg = :SimpleNewtonRaphson,
        abstol = 0.0, maxiters = 1000)
    nb = size(p, 2)
    u = Matrix{Float64}(undef, k.n, nb); resid = similar(u)
    rc = Vector{Int32}(undef, nb); iters = Vector{Int32}(undef, nb)
    if alg === :SimpleTrustRegion   # reference defaults for thresholds / factors (negative = default)
        GC.@preserve u0 p u resid rc iters nkcheck(@ccall libnk.nk_batch_solve_trust_region(k.ptr::Ptr{Cvoid}, nb::Int64,
            u0::Ptr{Float64}, 0::Cint, p::Ptr{Float64}, 0::Cint, abstol::Float64, maxiters::Cint, (-1.0)::Float64,
            (-1.0)::Float64, (-1.0)::Float64, (-1.0)::Float64, (-1.0)::Float64, (-1)::Cint, u::Ptr{Float64},
            resid::Ptr{Float64}, rc::Ptr{Int32}, iters::Ptr{Int32})::Cint)
        return (; u, resid, retcode = [RETCODES[c + 1] for c in rc], iters)
    end
    GC.@preserve u0 p u resid rc iters nkcheck(@ccall libnk.nk_batch_solve(k.ptr::Ptr{Cvoid}, nb::Int64,
        u0::Ptr{Float64}, 0::Cint, p::Ptr{Float64}, 0::Cint, abstol::Float64, maxiters::Cint, u::Ptr{Float64},
        resid::Ptr{Float64}, rc::Ptr{Int32}, iters::Ptr{Int32})::Cint)
    return (; u, resid, retcode = [RETCODES[c + 1] for c in rc], iters)
end

export Ctx, DeviceCSR, DeviceProblem, bratu2d, brusselator2d, mi355x_function, MI355XGMRES, MI355XNewtonKrylovAlg,
    EnsembleKernel, vectorized_solve

end # module
