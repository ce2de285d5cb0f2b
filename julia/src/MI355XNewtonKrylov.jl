# MI355XNewtonKrylov.jl — the reference-side binding a NonlinearSolve.jl maintainer would add.
#
# Written against include/mi355x_nk.h. Julia is not installed in the build container, so this file has
# never been executed; every LinearSolve/SciMLBase internal it touches is marked [EXT] (to confirm against
# LinearSolve 5.x). The C ABI itself is exercised from Python (ctypes) in tests/, and the exact call sequence of
# seam 1 below (new A every step → update_tolerances! → solve!, device-pointer callback operator, resident vectors)
# is replayed by examples/linsolve_seam.c, which the GPU tests run against the oracle.
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
import SciMLJacobianOperators            # StatefulJacobianOperator (lib/SciMLJacobianOperators)

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

# ------------------------------------------------------------------ where a vector lives
# Every vector argument of the ABI carries a memory-space flag: 0 = host (copied per call), 1 = device (zero copy).
# `DeviceVector` keeps data resident without any GPU array package (nk_device_alloc / nk_device_copy); with AMDGPU.jl
# loaded, `ROCArray`s are passed as they are (see the extension block at the end of this file).
const NK_HOST, NK_DEVICE = Cint(0), Cint(1)
mutable struct DeviceVector <: AbstractVector{Float64}
    ptr::Ptr{Float64}
    n::Int
    ctx::Ctx
    function DeviceVector(n::Integer; ctx::Ctx = default_ctx())
        out = Ref{Ptr{Cvoid}}(C_NULL)
        nkcheck(@ccall libnk.nk_device_alloc(ctx.ptr::Ptr{Cvoid}, (8n)::Int64, out::Ptr{Ptr{Cvoid}})::Cint)
        v = new(Ptr{Float64}(out[]), n, ctx)
        finalizer(x -> @ccall(libnk.nk_device_free(x.ctx.ptr::Ptr{Cvoid}, x.ptr::Ptr{Cvoid})::Cint), v)
        return v
    end
end
Base.size(v::DeviceVector) = (v.n,)
Base.getindex(::DeviceVector, ::Int) = error("DeviceVector lives on the GPU: copy it with Array(v)")
function Base.copyto!(d::DeviceVector, h::Array{Float64})
    GC.@preserve h nkcheck(@ccall libnk.nk_device_copy(d.ctx.ptr::Ptr{Cvoid}, d.ptr::Ptr{Cvoid}, h::Ptr{Float64},
        (8 * d.n)::Int64, 0::Cint)::Cint)
    return d
end
function Base.copyto!(h::Array{Float64}, d::DeviceVector)
    GC.@preserve h nkcheck(@ccall libnk.nk_device_copy(d.ctx.ptr::Ptr{Cvoid}, h::Ptr{Float64}, d.ptr::Ptr{Cvoid},
        (8 * d.n)::Int64, 1::Cint)::Cint)
    return h
end
DeviceVector(h::Array{Float64}; kw...) = copyto!(DeviceVector(length(h); kw...), vec(h))
Base.Array(d::DeviceVector) = copyto!(Vector{Float64}(undef, d.n), d)
# The in-place vector algebra of the reference's step! on resident data, through nk_vec_axpby / nk_vec_fill / nk_dot / nk_nrm2 / nk_norm_inf (include/mi355x_nk.h):
# `copyto!`, `axpy!` / `axpby!` (`@bb axpy!(α, δu, u)`, FirstOrder/src/solve.jl:403,438,460), `rmul!`, `fill!`, `dot`, `norm`,
# `similar`. General broadcast expressions (`@. u = u + α * δu`) need a GPU array package: with AMDGPU.jl loaded, use
# ROCArrays (ext/MI355XNewtonKrylovAMDGPUExt.jl) — they go through the same ABI with memspace = NK_DEVICE.
Base.similar(d::DeviceVector) = DeviceVector(d.n; ctx = d.ctx)
Base.similar(d::DeviceVector, ::Type{Float64}) = DeviceVector(d.n; ctx = d.ctx)
function Base.copyto!(dst::DeviceVector, src::DeviceVector)
    @assert dst.n == src.n
    nkcheck(@ccall libnk.nk_device_copy(dst.ctx.ptr::Ptr{Cvoid}, dst.ptr::Ptr{Cvoid}, src.ptr::Ptr{Cvoid}, (8 * dst.n)::Int64, 2::Cint)::Cint)
    return dst
end
Base.copy(d::DeviceVector) = copyto!(similar(d), d)
function LinearAlgebra.axpby!(a::Real, x::DeviceVector, b::Real, y::DeviceVector)
    @assert x.n == y.n
    nkcheck(@ccall libnk.nk_vec_axpby(y.ctx.ptr::Ptr{Cvoid}, y.n::Int64, Float64(a)::Float64, x.ptr::Ptr{Float64}, Float64(b)::Float64, y.ptr::Ptr{Float64})::Cint)
    return y
end
LinearAlgebra.axpy!(a::Real, x::DeviceVector, y::DeviceVector) = axpby!(a, x, 1.0, y)
LinearAlgebra.rmul!(y::DeviceVector, a::Real) = axpby!(0.0, y, a, y)
function Base.fill!(y::DeviceVector, a::Real)
    nkcheck(@ccall libnk.nk_vec_fill(y.ctx.ptr::Ptr{Cvoid}, y.n::Int64, Float64(a)::Float64, y.ptr::Ptr{Float64})::Cint)
    return y
end
function LinearAlgebra.dot(x::DeviceVector, y::DeviceVector)
    r = Ref(0.0)
    nkcheck(@ccall libnk.nk_dot(x.ctx.ptr::Ptr{Cvoid}, x.n::Int64, x.ptr::Ptr{Float64}, y.ptr::Ptr{Float64}, r::Ptr{Float64})::Cint)
    return r[]
end
function LinearAlgebra.norm(x::DeviceVector, p::Real = 2)
    (p == 2 || p == Inf) || error("DeviceVector: norm(x, 2) and norm(x, Inf)")
    r = Ref(0.0)
    if p == 2
        nkcheck(@ccall libnk.nk_nrm2(x.ctx.ptr::Ptr{Cvoid}, x.n::Int64, x.ptr::Ptr{Float64}, r::Ptr{Float64})::Cint)
    else
        nkcheck(@ccall libnk.nk_norm_inf(x.ctx.ptr::Ptr{Cvoid}, x.n::Int64, x.ptr::Ptr{Float64}, r::Ptr{Float64})::Cint)
    end
    return r[]
end

memspace(::Array{Float64}) = NK_HOST
memspace(::DeviceVector) = NK_DEVICE
memspace(x::Base.ReshapedArray) = memspace(parent(x))           # `vec`/`reshape` views the reference hands around
rawptr(x::Array{Float64}) = pointer(x)
rawptr(x::DeviceVector) = x.ptr
rawptr(x::Base.ReshapedArray) = rawptr(parent(x))

mutable struct DeviceCSR
    ptr::Ptr{Cvoid}
    n::Int
    colptr::Vector{Int}      # the pattern this object was built from: a new Jacobian with the same pattern only refreshes values
    rowval::Vector{Int}
end
"""Upload a `SparseMatrixCSC{Float64,Int}` as it is (1-based Int64 colptr/rowval): converted to CSR once."""
function DeviceCSR(A::SparseMatrixCSC{Float64, Int}; ctx = default_ctx())
    out = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve A nkcheck(@ccall libnk.nk_csr_create_from_csc(ctx.ptr::Ptr{Cvoid}, size(A, 1)::Int64,
        nnz(A)::Int64, 64::Cint, 1::Cint, A.colptr::Ptr{Int64}, A.rowval::Ptr{Int64}, A.nzval::Ptr{Float64},
        out::Ptr{Ptr{Cvoid}})::Cint)
    # the pattern is COPIED: `same_pattern` must notice a Jacobian whose structure was changed in place with the same nnz
    # (arrays reused) — sharing them made every such matrix compare equal to itself and gather through a stale permutation
    m = DeviceCSR(out[], size(A, 1), copy(A.colptr), copy(A.rowval))
    finalizer(x -> @ccall(libnk.nk_csr_destroy(x.ptr::Ptr{Cvoid})::Cint), m)
    return m
end
same_pattern(m::DeviceCSR, A::SparseMatrixCSC{Float64, Int}) =
    size(A, 1) == m.n && A.colptr == m.colptr && A.rowval == m.rowval      # contents, always (two O(nnz) integer compares)
"""New values in `nonzeros(A)` order for the pattern the matrix was created with: one gather on the device
(nk_csr_set_values_csc) instead of a new CSC → CSR conversion and pattern upload."""
function update_values!(m::DeviceCSR, A::SparseMatrixCSC{Float64, Int})
    GC.@preserve A nkcheck(@ccall libnk.nk_csr_set_values_csc(m.ptr::Ptr{Cvoid}, A.nzval::Ptr{Float64}, nnz(A)::Int64, NK_HOST::Cint)::Cint)
    return m
end
function LinearAlgebra.mul!(y::AbstractVector{Float64}, A::DeviceCSR, x::AbstractVector{Float64})
    @assert memspace(x) == memspace(y)
    GC.@preserve x y nkcheck(@ccall libnk.nk_spmv(A.ptr::Ptr{Cvoid}, rawptr(x)::Ptr{Float64}, rawptr(y)::Ptr{Float64},
        memspace(x)::Cint)::Cint)
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

# ------------------------------------------------------------------ preconditioner objects (what `precs(A, p)` may return)
"""`DeviceILU0(A::DeviceCSR; ordering = :multicolor)` / `DeviceJacobi(A)` / `DeviceILUT(A; τ)` / `DeviceAMG(A; …)`: nk_precond objects — usable as `Pl` or `Pr` of
`MI355XGMRES` without a host round trip per application, and as `ldiv!(y, P, x)` on host or resident vectors."""
mutable struct DevicePreconditioner
    ptr::Ptr{Cvoid}
    A::DeviceCSR
end
function DeviceILU0(A::DeviceCSR; ordering::Symbol = :multicolor)
    out = Ref{Ptr{Cvoid}}(C_NULL)
    nkcheck(@ccall libnk.nk_precond_create_ilu0(A.ptr::Ptr{Cvoid}, (ordering === :natural ? 0 : 1)::Cint, out::Ptr{Ptr{Cvoid}})::Cint)
    p = DevicePreconditioner(out[], A)
    finalizer(x -> @ccall(libnk.nk_precond_destroy(x.ptr::Ptr{Cvoid})::Cint), p)
    return p
end
function DeviceJacobi(A::DeviceCSR)
    out = Ref{Ptr{Cvoid}}(C_NULL)
    nkcheck(@ccall libnk.nk_precond_create_jacobi(A.ptr::Ptr{Cvoid}, out::Ptr{Ptr{Cvoid}})::Cint)
    p = DevicePreconditioner(out[], A)
    finalizer(x -> @ccall(libnk.nk_precond_destroy(x.ptr::Ptr{Cvoid})::Cint), p)
    return p
end
"""`DeviceILUT(A::DeviceCSR; τ)`: threshold ILU with fill — the tutorial's `IncompleteLU.ilu(W, τ = 50.0)`
(docs/src/tutorials/large_systems.md:252-260): Crout ILU(τ), factorised on the host for every `update!` (the pattern depends on the
numbers), applied on the device by level-scheduled triangular solves."""
function DeviceILUT(A::DeviceCSR; τ::Real)
    out = Ref{Ptr{Cvoid}}(C_NULL)
    nkcheck(@ccall libnk.nk_precond_create_ilut(A.ptr::Ptr{Cvoid}, Float64(τ)::Cdouble, out::Ptr{Ptr{Cvoid}})::Cint)
    p = DevicePreconditioner(out[], A)
    finalizer(x -> @ccall(libnk.nk_precond_destroy(x.ptr::Ptr{Cvoid})::Cint), p)
    return p
end
# nk_amg_params (include/mi355x_nk.h): four 32-bit integers, three doubles; zero fields take the library's defaults
struct AMGParams
    nu::Int32
    passes::Int32
    coarse_max::Int32
    matching::Int32      # 0 automatic, 1 sequential pairwise pass (host set-up), 2 handshaking (device set-up)
    theta::Float64
    overcorrection::Float64
    cheb_ratio::Float64
end
"""`DeviceAMG(A::DeviceCSR; nu, passes, theta, overcorrection, cheb_ratio, coarse_max, matching)`: aggregation algebraic multigrid
built from the matrix alone, its hierarchy set up on the device — the slot of `aspreconditioner(ruge_stuben(W))`
(large_systems.md:276-316); `update!` refreshes every number for the matrix's current values and keeps the aggregates."""
function DeviceAMG(A::DeviceCSR; nu::Integer = 0, passes::Integer = 0, coarse_max::Integer = 0, matching::Integer = 0,
                   theta::Real = 0.0, overcorrection::Real = 0.0, cheb_ratio::Real = 0.0)
    prm = Ref(AMGParams(nu, passes, coarse_max, matching, theta, overcorrection, cheb_ratio))
    out = Ref{Ptr{Cvoid}}(C_NULL)
    nkcheck(@ccall libnk.nk_precond_create_amg(A.ptr::Ptr{Cvoid}, prm::Ptr{Cvoid}, out::Ptr{Ptr{Cvoid}})::Cint)
    p = DevicePreconditioner(out[], A)
    finalizer(x -> @ccall(libnk.nk_precond_destroy(x.ptr::Ptr{Cvoid})::Cint), p)
    return p
end
update!(P::DevicePreconditioner) = (nkcheck(@ccall libnk.nk_precond_update(P.ptr::Ptr{Cvoid})::Cint); P)
function LinearAlgebra.ldiv!(y::AbstractVector{Float64}, P::DevicePreconditioner, x::AbstractVector{Float64})
    @assert memspace(x) == memspace(y)
    GC.@preserve x y nkcheck(@ccall libnk.nk_precond_apply(P.ptr::Ptr{Cvoid}, rawptr(x)::Ptr{Float64}, rawptr(y)::Ptr{Float64}, memspace(x)::Cint)::Cint)
    return y
end

# ------------------------------------------------------------------ seam 2: f / jvp / vjp callbacks
"""`NonlinearFunction` whose `f`, `jvp`, `vjp` run the built-in device kernels. The arrays may be host `Array`s (copied
per call) or resident `DeviceVector`s / `ROCArray`s (zero copy): the memory-space flag follows the argument type.
Signatures follow lib/SciMLJacobianOperators/test/core_tests__item2.jl:38-40,62-67. The callbacks are callable structs,
not closures, so that `MI355XGMRES` can recognise the device problem behind `f.jvp` and bind the device JVP directly."""
struct DeviceResidual; P::DeviceProblem; end
struct DeviceJVP;      P::DeviceProblem; end
struct DeviceVJP;      P::DeviceProblem; end
function (r::DeviceResidual)(du, u, p)
    GC.@preserve du u nkcheck(@ccall libnk.nk_residual(r.P.ptr::Ptr{Cvoid}, rawptr(u)::Ptr{Float64},
        rawptr(du)::Ptr{Float64}, memspace(u)::Cint)::Cint)
    return nothing
end
function (j::DeviceJVP)(Jv, v, u, p)
    GC.@preserve Jv v u nkcheck(@ccall libnk.nk_jvp(j.P.ptr::Ptr{Cvoid}, rawptr(u)::Ptr{Float64}, rawptr(v)::Ptr{Float64},
        rawptr(Jv)::Ptr{Float64}, memspace(u)::Cint)::Cint)
    return nothing
end
function (j::DeviceVJP)(vJ, v, u, p)
    GC.@preserve vJ v u nkcheck(@ccall libnk.nk_vjp(j.P.ptr::Ptr{Cvoid}, rawptr(u)::Ptr{Float64}, rawptr(v)::Ptr{Float64},
        rawptr(vJ)::Ptr{Float64}, memspace(u)::Cint)::Cint)
    return nothing
end
mi355x_function(P::DeviceProblem) = NonlinearFunction{true}(DeviceResidual(P); jvp = DeviceJVP(P), vjp = DeviceVJP(P))

# ------------------------------------------------------------------ seam 1: linsolve backend
"""
    MI355XGMRES(; gmres_restart = 30, ortho = :dcgs2, sstep = 6)

`NewtonRaphson(linsolve = MI355XGMRES())`. NonlinearSolveBase only needs what
ext/NonlinearSolveBaseLinearSolveExt.jl:16-32,60-115 uses: `needs_concrete_A == false`, a cache with settable
`A`, `b`, `u`, `solve!` returning `u` + `retcode`, and `update_tolerances!(; reltol)`.
"""
Base.@kwdef struct MI355XGMRES <: LinearSolve.AbstractKrylovSubspaceMethod   # [EXT]
    gmres_restart::Int = 30
    ortho::Symbol = :sstep     # :mgs | :cgs2 | :cgs | :dcgs2 | :sstep (s columns per block, matrix-core Gram blocks; the default)
    sstep::Int = 0             # 0 = automatic: 15 with the Newton basis (spectrum bounds known), 6 with the monomial one
    sstep_basis::Symbol = :auto  # :auto | :monomial | :newton
    # `precs(A, p::LinearSolveParameters) -> (Pl, Pr)`, as on LinearSolve's Krylov algorithms: LinearSolve evaluates it when it
    # builds the cache (`hasproperty(alg, :precs)` in its `init` [EXT]) and this module's `solve!` re-evaluates it for every
    # fresh `A` (what KrylovJL's `solve!` does under `cache.precsisfresh` [EXT]); core_tests__item21.jl:10-37 pins the protocol
    precs::Any = nothing
end
LinearSolve.needs_concrete_A(::MI355XGMRES) = false                           # [EXT]

# nk_gmres_info, field for field (include/mi355x_nk.h)
struct GMRESInfo
    iters::Int32; restarts::Int32; converged::Int32; failed::Int32
    rnorm0::Float64; rnorm::Float64
end

# An operator the library calls back into: a mutable box (so that `pointer_from_objref` is legal) that the workspace
# keeps alive for as long as the device GMRES may call it.
mutable struct OperatorBox
    A::Any
    n::Int
end
mutable struct GMRESWorkspace
    ptr::Ptr{Cvoid}
    n::Int
    csr::Union{Nothing, DeviceCSR}
    box::Union{Nothing, OperatorBox}        # keep-alive of the operator behind the C callback
    precbox::Union{Nothing, OperatorBox}    # … of Pr
    lprecbox::Union{Nothing, OperatorBox}   # … of Pl
end
const ORTHO = Dict(:mgs => 0, :cgs2 => 1, :cgs => 2, :dcgs2 => 3, :dcgs2_1r => 4, :sstep => 5)
const SS_BASIS = Dict(:auto => 0, :monomial => 1, :newton => 2)

function LinearSolve.init_cacheval(alg::MI355XGMRES, A, b, u, Pl, Pr, maxiters::Int, abstol, reltol,
        verbose, assumptions)                                                  # [EXT signature]
    out = Ref{Ptr{Cvoid}}(C_NULL)
    nkcheck(@ccall libnk.nk_gmres_create(default_ctx().ptr::Ptr{Cvoid}, length(b)::Int64,
        alg.gmres_restart::Cint, ORTHO[alg.ortho]::Cint, out::Ptr{Ptr{Cvoid}})::Cint)
    if alg.ortho === :sstep
        nkcheck(@ccall libnk.nk_gmres_set_block_size(out[]::Ptr{Cvoid}, alg.sstep::Cint)::Cint)
        nkcheck(@ccall libnk.nk_gmres_set_sstep_basis(out[]::Ptr{Cvoid}, SS_BASIS[alg.sstep_basis]::Cint)::Cint)
    end
    w = GMRESWorkspace(out[], length(b), nothing, nothing, nothing, nothing)
    finalizer(x -> @ccall(libnk.nk_gmres_destroy(x.ptr::Ptr{Cvoid})::Cint), w)
    return w
end

# mul!(y, A, x) for an operator that lives in HOST memory (a Julia `mul!` on plain Arrays — e.g. a
# StatefulJacobianOperator around CPU closures): registered with nk_gmres_set_operator_fn_host, so the library hands the
# callback HOST pointers and stages the vectors itself. (The device-pointer variant of the contract, nk_matvec_fn proper,
# is what the AMDGPU extension below uses; wrapping device pointers in `Array`s would read invalid memory.)
function host_matvec_trampoline(user::Ptr{Cvoid}, x::Ptr{Float64}, y::Ptr{Float64}, ::Ptr{Cvoid})::Cint
    box = unsafe_pointer_to_objref(user)::OperatorBox
    xv = unsafe_wrap(Array, x, box.n); yv = unsafe_wrap(Array, y, box.n)        # host pointers: legal
    try
        mul!(yv, box.A, xv)
        return Cint(0)
    catch
        return Cint(1)
    end
end
# ldiv!-style right preconditioner from `precs(A, p) -> (Pl, Pr)` (test/Core/core_tests__item21.jl:10-18), host memory
function host_prec_trampoline(user::Ptr{Cvoid}, x::Ptr{Float64}, y::Ptr{Float64}, ::Ptr{Cvoid})::Cint
    box = unsafe_pointer_to_objref(user)::OperatorBox
    xv = unsafe_wrap(Array, x, box.n); yv = unsafe_wrap(Array, y, box.n)
    try
        ldiv!(yv, box.A, xv)
        return Cint(0)
    catch
        return Cint(1)
    end
end

# the device problem behind a StatefulJacobianOperator built from `mi355x_function` (seam 2), if any
device_jvp_of(A) = nothing
function device_jvp_of(A::SciMLJacobianOperators.StatefulJacobianOperator)      # fields: mode, jac_op, u, p  (:210-224)
    op = A.jac_op.jvp_op                                                         # JacobianOperator.jvp_op    (:86-96)
    return op isa DeviceJVP ? op.P : nothing
end

# Operators / preconditioners whose `mul!` / `ldiv!` work on resident (ROCArray) vectors declare it — `on_device(::MyOp) = true`
# — and are then called through the device-pointer contract by the AMDGPU extension (ext/MI355XNewtonKrylovAMDGPUExt.jl),
# which provides the two methods below; without the extension they error instead of reading device memory from the host.
on_device(A) = false
bind_device_operator!(w, A) = error("device-resident Julia operators need AMDGPU.jl (load it to activate the extension)")
bind_device_preconditioner!(w, side, P) = error("device-resident Julia preconditioners need AMDGPU.jl (load it to activate the extension)")

function set_operator!(w::GMRESWorkspace, A)
    if A isa SparseMatrixCSC
        # concrete J: Julia's CSC fields are ingested as they are (the CSC → CSR conversion is one O(nnz) host pass) — ONCE per
        # pattern: the Jacobian of the next Newton step has the same colptr / rowval and only refreshes the values (one gather
        # on the device, nk_csr_set_values_csc; round 2 re-ingested and re-uploaded the whole matrix every step)
        if w.csr !== nothing && same_pattern(w.csr, A)
            update_values!(w.csr, A)
        else
            w.csr = DeviceCSR(A)
        end
        nkcheck(@ccall libnk.nk_gmres_set_operator_csr(w.ptr::Ptr{Cvoid}, w.csr.ptr::Ptr{Cvoid})::Cint)
    elseif (P = device_jvp_of(A)) !== nothing
        # StatefulJacobianOperator whose f.jvp is the device kernel: bind it, no trampoline, nothing crosses PCIe
        u = A.u
        GC.@preserve u nkcheck(@ccall libnk.nk_gmres_set_operator_jvp(w.ptr::Ptr{Cvoid}, P.ptr::Ptr{Cvoid},
            rawptr(u)::Ptr{Float64}, memspace(u)::Cint)::Cint)
    elseif on_device(A)
        bind_device_operator!(w, A)
    else
        w.box = OperatorBox(A, w.n)
        cb = @cfunction(host_matvec_trampoline, Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}))
        nkcheck(@ccall libnk.nk_gmres_set_operator_fn_host(w.ptr::Ptr{Cvoid}, cb::Ptr{Cvoid},
            pointer_from_objref(w.box)::Ptr{Cvoid})::Cint)
    end
    return nothing
end

const NK_SIDE_RIGHT, NK_SIDE_LEFT = Cint(0), Cint(1)
is_identity(P) = P === nothing || P === LinearAlgebra.I || P isa LinearSolve.IdentityOperator ||      # [EXT]
    (P isa LinearAlgebra.UniformScaling && isone(P.λ))
function bind_preconditioner!(w::GMRESWorkspace, side::Cint, P)
    if is_identity(P)
        nkcheck(@ccall libnk.nk_gmres_set_preconditioner(w.ptr::Ptr{Cvoid}, side::Cint, C_NULL::Ptr{Cvoid})::Cint)
        side == NK_SIDE_LEFT ? (w.lprecbox = nothing) : (w.precbox = nothing)
    elseif P isa DevicePreconditioner
        box = OperatorBox(P, w.n)                                                # keep-alive only
        side == NK_SIDE_LEFT ? (w.lprecbox = box) : (w.precbox = box)
        nkcheck(@ccall libnk.nk_gmres_set_preconditioner(w.ptr::Ptr{Cvoid}, side::Cint, P.ptr::Ptr{Cvoid})::Cint)
    elseif on_device(P)
        bind_device_preconditioner!(w, side, P)
    else
        box = OperatorBox(P, w.n)
        side == NK_SIDE_LEFT ? (w.lprecbox = box) : (w.precbox = box)
        pcb = @cfunction(host_prec_trampoline, Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}))
        if side == NK_SIDE_LEFT
            nkcheck(@ccall libnk.nk_gmres_set_left_preconditioner_host(w.ptr::Ptr{Cvoid}, pcb::Ptr{Cvoid}, pointer_from_objref(box)::Ptr{Cvoid})::Cint)
        else
            nkcheck(@ccall libnk.nk_gmres_set_right_preconditioner_host(w.ptr::Ptr{Cvoid}, pcb::Ptr{Cvoid}, pointer_from_objref(box)::Ptr{Cvoid})::Cint)
        end
    end
    return nothing
end

function SciMLBase.solve!(cache::LinearSolve.LinearCache, alg::MI355XGMRES; kwargs...)   # [EXT]
    w = cache.cacheval::GMRESWorkspace
    if cache.isfresh                                                            # [EXT] set by `cache.A = …`
        set_operator!(w, cache.A)
        # `precs(A, p) -> (Pl, Pr)` [EXT: LinearSolve re-evaluates it for a fresh A and stores the pair in the cache]. BOTH sides
        # are bound — the reference's own documented precs return `(Pl, I)` (docs/src/tutorials/large_systems.md:257,284-287):
        # a DevicePreconditioner goes in as an object (no host round trip per application), anything else with `ldiv!` through
        # the host trampoline, identities remove that side.
        Pl, Pr = alg.precs === nothing ? (cache.Pl, cache.Pr) : alg.precs(cache.A, cache.p)   # a fresh A: precs is re-evaluated
        bind_preconditioner!(w, NK_SIDE_LEFT, Pl)                              # (`nothing` and `I` count as identities)
        bind_preconditioner!(w, NK_SIDE_RIGHT, Pr)
        cache.isfresh = false
    end
    info = Ref(GMRESInfo(0, 0, 0, 0, 0.0, 0.0))
    b, u = cache.b, cache.u
    @assert memspace(b) == memspace(u)
    # cache.reltol is what update_tolerances!(cache; reltol = η) set (eisenstat_walker.jl:50,77)
    GC.@preserve b u w nkcheck(@ccall libnk.nk_gmres_solve(w.ptr::Ptr{Cvoid}, rawptr(b)::Ptr{Float64},
        rawptr(u)::Ptr{Float64}, memspace(b)::Cint, 0::Cint, cache.abstol::Float64, cache.reltol::Float64,
        cache.maxiters::Cint, 0::Cint, info::Ptr{GMRESInfo})::Cint)
    i = info[]
    rc = i.failed != 0 ? ReturnCode.Failure : (i.converged != 0 ? ReturnCode.Success : ReturnCode.MaxIters)
    return SciMLBase.build_linear_solution(alg, u, nothing, cache; retcode = rc, iters = Int(i.iters))
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
    direct::Bool = false                  # linsolve = nothing: banded LU on the device (needs concrete_jac)
    gmres_restart::Int = 30
    gmres_maxiters::Int = 300
    ortho::Symbol = :sstep                # Arnoldi process: :mgs | :cgs2 | :cgs | :dcgs2 | :sstep
    sstep::Int = 0                        # ortho = :sstep: basis columns per block (0 = automatic)
    sstep_basis::Symbol = :auto           # :auto | :monomial | :newton
    forcing::Bool = false                 # EisenstatWalkerForcing2()
    radius_update_scheme::Int = 0         # RadiusUpdateSchemes.Simple … Fan (0…6)
    linesearch::Symbol = :none            # :none | :BackTracking | :Static | :StrongWolfe | :MoreThuente | :HagerZhang (LineSearchesJL methods)
    precs::Symbol = :none                 # :none | :chebyshev | :multigrid | :jacobi | :ilu0 | :ilu0_natural — the built-ins behind the `precs` hook
    precs_side::Symbol = :left            # :jacobi / :ilu0: which side (the reference's tutorial precs return (Pl, I))
    cheb_degree::Int = 32
    cheb_ratio::Float64 = 300.0
    mg_nu::Int = 2
    mg_coarse::Int = 31
    jac_colored::Bool = false             # concrete J by colour-compressed assembly (AutoSparse analogue) instead of f.jac
    # :auto (NewtonRaphson, or TrustRegion when trust_region = true) | :GaussNewton | :LevenbergMarquardt — the latter two
    # solve the (damped) normal equations through the device normal-form operator and need concrete_jac + a Krylov linsolve
    method::Symbol = :auto
    # LevenbergMarquardt(; damping_initial, damping_increase_factor, damping_decrease_factor, min_damping_D, α_geodesic,
    # finite_diff_step_geodesic, b_uphill, disable_geodesic) — levenberg_marquardt.jl:37-64
    lm_damping_initial::Float64 = 1.0
    lm_damping_increase_factor::Float64 = 2.0
    lm_damping_decrease_factor::Float64 = 3.0
    lm_min_damping_D::Float64 = 1e-8
    lm_alpha_geodesic::Float64 = 0.75
    lm_finite_diff_step_geodesic::Float64 = 0.1
    lm_b_uphill::Float64 = 1.0
    lm_disable_geodesic::Bool = false
    pt_alpha_initial::Float64 = 1e-3      # PseudoTransient(; alpha_initial) — method = :PseudoTransient
    pt_mass_diagonal::Union{Nothing, Vector{Float64}} = nothing   # PseudoTransient(; mass_matrix = Diagonal(m)); nothing = I
end

# termination_condition → nk_options.termination_mode / termination_norm and the mode struct's fields
# (lib/NonlinearSolveBase/src/termination_conditions.jl:243-376; order of include/mi355x_nk.h's nk_termination_mode)
const TERMINATION_MODES = Dict(:AbsNormSafeBestTerminationMode => 0, :NormTerminationMode => 1, :RelTerminationMode => 2,
    :RelNormTerminationMode => 3, :RelNormSafeTerminationMode => 4, :RelNormSafeBestTerminationMode => 5,
    :AbsTerminationMode => 6, :AbsNormTerminationMode => 7, :AbsNormSafeTerminationMode => 8)
function apply_termination!(o, tc)
    tc === nothing && return o                       # the solver's default: AbsNormSafeBest(maximum∘abs; max_stalled_steps = 32)
    o.termination_mode = TERMINATION_MODES[nameof(typeof(tc))]
    hasproperty(tc, :internalnorm) && (o.termination_norm = tc.internalnorm === LinearAlgebra.norm ? 1 : 0)
    # an explicitly constructed mode carries max_stalled_steps = nothing unless given
    o.max_stalled_steps = hasproperty(tc, :max_stalled_steps) && tc.max_stalled_steps !== nothing ? tc.max_stalled_steps : -1
    for f in (:patience_steps, :patience_objective_multiplier, :min_max_factor)
        hasproperty(tc, f) && setproperty!(o, f, getproperty(tc, f))
    end
    hasproperty(tc, :protective_threshold) && tc.protective_threshold !== nothing &&
        (o.protective_threshold = tc.protective_threshold)
    return o
end

# nk_options mirrors include/mi355x_nk.h field for field. (A mutable struct of plain bits: `Ref(o)` is a pointer to its fields,
# valid for the duration of the @ccall that receives it.)
Base.@kwdef mutable struct NKOptions
    algorithm::Int32 = 0; linsolve::Int32 = 0; maxiters::Int32 = 1000; termination_norm::Int32 = 0
    abstol::Float64 = 0.0; reltol::Float64 = 0.0; maxtime::Float64 = 0.0
    gmres_restart::Int32 = 30; gmres_maxiters::Int32 = 300; gmres_ortho::Int32 = 5; gmres_fixed_iters::Int32 = 0
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
    jac_colored::Int32 = 0; lm_disable_geodesic::Int32 = 0
    lm_damping_initial::Float64 = 1.0; lm_damping_increase_factor::Float64 = 2.0; lm_damping_decrease_factor::Float64 = 3.0
    lm_min_damping_D::Float64 = 1e-8; lm_alpha_geodesic::Float64 = 0.75; lm_finite_diff_step_geodesic::Float64 = 0.1
    lm_b_uphill::Float64 = 1.0
    pt_alpha_initial::Float64 = 1e-3
    gmres_sstep::Int32 = 0; gmres_sstep_basis::Int32 = 0
    precond_kind::Int32 = 0; precond_side::Int32 = 1
end

const RETCODES = (ReturnCode.Default, ReturnCode.Success, ReturnCode.MaxIters, ReturnCode.Unstable,
    ReturnCode.Stalled, ReturnCode.InternalLinearSolveFailed, ReturnCode.ShrinkThresholdExceeded,
    ReturnCode.MaxTime, ReturnCode.Failure, ReturnCode.InternalLineSearchFailed)

function SciMLBase.__solve(prob::NonlinearProblem, alg::MI355XNewtonKrylovAlg, args...;
        abstol = nothing, reltol = nothing, maxiters = 1000, maxtime = nothing, termination_condition = nothing,
        kwargs...)
    algorithm = alg.method === :PseudoTransient ? 4 : alg.method === :LevenbergMarquardt ? 3 : alg.method === :GaussNewton ? 2 : (alg.trust_region ? 1 : 0)
    o = NKOptions(; algorithm = algorithm,
        linsolve = alg.direct ? 2 : ((alg.concrete_jac || algorithm == 3) ? 1 : 0),
        lm_disable_geodesic = alg.lm_disable_geodesic ? 1 : 0, lm_damping_initial = alg.lm_damping_initial,
        lm_damping_increase_factor = alg.lm_damping_increase_factor,
        lm_damping_decrease_factor = alg.lm_damping_decrease_factor, lm_min_damping_D = alg.lm_min_damping_D,
        lm_alpha_geodesic = alg.lm_alpha_geodesic, lm_finite_diff_step_geodesic = alg.lm_finite_diff_step_geodesic,
        lm_b_uphill = alg.lm_b_uphill, pt_alpha_initial = alg.pt_alpha_initial,
        maxiters = maxiters, abstol = something(abstol, 0.0), reltol = something(reltol, 0.0),
        maxtime = something(maxtime, 0.0),
        gmres_restart = alg.gmres_restart, gmres_maxiters = alg.gmres_maxiters,
        gmres_ortho = ORTHO[alg.ortho], gmres_sstep = alg.sstep, gmres_sstep_basis = SS_BASIS[alg.sstep_basis],
        forcing = alg.forcing ? 1 : 0, radius_update_scheme = alg.radius_update_scheme,
        linesearch = get(Dict(:BackTracking => 1, :Static => 2, :StrongWolfe => 3, :MoreThuente => 4, :HagerZhang => 5), alg.linesearch, 0),
        cheb_degree = alg.precs === :chebyshev ? alg.cheb_degree : 0, cheb_ratio = alg.cheb_ratio,
        mg_nu = alg.precs === :multigrid ? alg.mg_nu : 0, mg_coarse = alg.mg_coarse,
        jac_colored = alg.jac_colored ? 1 : 0,
        precond_kind = get(Dict(:jacobi => 1, :ilu0_natural => 2, :ilu0 => 3), alg.precs, 0),
        precond_side = alg.precs_side === :right ? 0 : 1)
    apply_termination!(o, termination_condition)
    # u0 may be a host Array (copied in and out once) or already resident (DeviceVector / ROCArray): no PCIe traffic then
    u0 = prob.u0 isa Array ? Vector{Float64}(vec(prob.u0)) : vec(prob.u0)
    ms = memspace(u0)
    u = ms == NK_HOST ? similar(u0) : DeviceVector(length(u0))
    resid = ms == NK_HOST ? similar(u0) : DeviceVector(length(u0))
    stats = zeros(Int64, 9); rc = Ref{Cint}(0)
    if alg.pt_mass_diagonal === nothing
        GC.@preserve u0 u resid stats nkcheck(@ccall libnk.nk_newton_solve(alg.problem.ptr::Ptr{Cvoid},
            rawptr(u0)::Ptr{Float64}, ms::Cint, Ref(o)::Ptr{Cvoid}, rawptr(u)::Ptr{Float64}, rawptr(resid)::Ptr{Float64},
            stats::Ptr{Int64}, rc::Ptr{Cint})::Cint)
    else   # the cache interface, so that the mass matrix can be handed over between init and the first step
        m = alg.pt_mass_diagonal; h = Ref{Ptr{Cvoid}}(C_NULL)
        length(m) == length(u0) || throw(DimensionMismatch("mass matrix has $(length(m)) diagonal entries but the problem has $(length(u0)) unknowns"))
        GC.@preserve u0 nkcheck(@ccall libnk.nk_solver_init(alg.problem.ptr::Ptr{Cvoid}, rawptr(u0)::Ptr{Float64}, ms::Cint,
            Ref(o)::Ptr{Cvoid}, h::Ptr{Ptr{Cvoid}})::Cint)
        try
            GC.@preserve m nkcheck(@ccall libnk.nk_solver_set_mass_matrix_diagonal(h[]::Ptr{Cvoid}, m::Ptr{Float64}, NK_HOST::Cint)::Cint)
            nkcheck(@ccall libnk.nk_solver_solve(h[]::Ptr{Cvoid}, rc::Ptr{Cint})::Cint)
            GC.@preserve u resid stats begin
                nkcheck(@ccall libnk.nk_solver_get_u(h[]::Ptr{Cvoid}, rawptr(u)::Ptr{Float64}, ms::Cint)::Cint)
                nkcheck(@ccall libnk.nk_solver_get_resid(h[]::Ptr{Cvoid}, rawptr(resid)::Ptr{Float64}, ms::Cint)::Cint)
                nkcheck(@ccall libnk.nk_solver_get_stats(h[]::Ptr{Cvoid}, stats::Ptr{Int64})::Cint)
            end
        finally
            @ccall libnk.nk_solver_destroy(h[]::Ptr{Cvoid})::Cint
        end
    end
    shape(x) = x isa Array ? reshape(x, size(prob.u0)) : x
    return SciMLBase.build_solution(prob, alg, shape(u), shape(resid);
        retcode = RETCODES[rc[] + 1], stats = NLStats(stats[1], stats[2], stats[3], stats[4], stats[5]),
        original = (; gmres_iters = stats[6], op_applies = stats[7], allreduces = stats[8], halo_exchanges = stats[9]))
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

# ------------------------------------------------------------------------------------------ AMDGPU.jl arrays (optional)
# ext/MI355XNewtonKrylovAMDGPUExt.jl (a package extension, AMDGPU as a weak dependency): ROCArrays pass through every entry
# point with memspace = NK_DEVICE, and device-resident Julia operators / preconditioners serve through the device-pointer
# callback contract (nk_matvec_fn proper).

export Ctx, DeviceVector, DeviceCSR, DeviceProblem, bratu2d, brusselator2d, mi355x_function, MI355XGMRES,
    MI355XNewtonKrylovAlg, EnsembleKernel, vectorized_solve, DevicePreconditioner, DeviceILU0, DeviceILUT, DeviceAMG, DeviceJacobi, update!, update_values!

end # module
