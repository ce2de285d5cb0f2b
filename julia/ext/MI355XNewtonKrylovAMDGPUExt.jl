# Package extension of MI355XNewtonKrylov for AMDGPU.jl (weak dependency; Project.toml: [weakdeps] AMDGPU, [extensions]
# MI355XNewtonKrylovAMDGPUExt = "AMDGPU"). UNEXECUTED in the build container (no Julia there): written against the ABI in
# include/mi355x_nk.h, checked symbol by symbol and argument by argument by tests/test_abi.py; AMDGPU.jl calls are marked [EXT].
#
# What it adds:
#   * `ROCArray{Float64}` is a resident vector for every entry point (memspace = NK_DEVICE, zero copy). With ROCArrays as
#     `u0`, the reference's own step! runs unchanged around seams 1 + 2 — its broadcasts (`@bb axpy!`, `@. u = u + δu`,
#     lib/NonlinearSolveFirstOrder/src/solve.jl:195,403,438,460; docs/src/tutorials/nonlinear_solve_gpus.md:38-68) are
#     AMDGPU.jl's, the linear solve and f / jvp / vjp are this library's.
#   * Julia operators and preconditioners that work on ROCArrays (`mul!` / `ldiv!`) serve as `A`, `Pl`, `Pr` of MI355XGMRES
#     through the DEVICE-pointer callback contract (nk_matvec_fn proper): the trampolines wrap the pointers as ROCArrays —
#     never as Arrays — and order their work on the stream the library hands over.
module MI355XNewtonKrylovAMDGPUExt

using AMDGPU, LinearAlgebra
using MI355XNewtonKrylov
import MI355XNewtonKrylov: memspace, rawptr, NK_DEVICE, NK_SIDE_LEFT, OperatorBox, GMRESWorkspace, libnk, nkcheck,
    bind_device_operator!, bind_device_preconditioner!

memspace(::ROCArray{Float64}) = NK_DEVICE
rawptr(x::ROCArray{Float64}) = Ptr{Float64}(UInt(pointer(x)))                                  # [EXT] device address

# a non-owning ROCArray view of n doubles at a device address                                    [EXT AMDGPU.jl ≥ 1.0]
wrap(p::Ptr{Float64}, n::Int) = unsafe_wrap(ROCArray, p, (n,); lock = false)

function device_matvec_trampoline(user::Ptr{Cvoid}, x::Ptr{Float64}, y::Ptr{Float64}, stream::Ptr{Cvoid})::Cint
    box = unsafe_pointer_to_objref(user)::OperatorBox
    try
        mul!(wrap(y, box.n), box.A, wrap(x, box.n))
        AMDGPU.synchronize()        # the library's next kernel reads y on ITS stream: the Julia task's work must be complete [EXT]
        return Cint(0)
    catch
        return Cint(1)
    end
end
function device_prec_trampoline(user::Ptr{Cvoid}, x::Ptr{Float64}, y::Ptr{Float64}, stream::Ptr{Cvoid})::Cint
    box = unsafe_pointer_to_objref(user)::OperatorBox
    try
        ldiv!(wrap(y, box.n), box.A, wrap(x, box.n))
        AMDGPU.synchronize()
        return Cint(0)
    catch
        return Cint(1)
    end
end

# An operator whose `mul!` works on ROCArrays (declared by the user: `MI355XNewtonKrylov.on_device(A) = true`) is registered
# with nk_gmres_set_operator_fn — device pointers, no staging through pinned host memory.
function bind_device_operator!(w::GMRESWorkspace, A)
    w.box = OperatorBox(A, w.n)
    cb = @cfunction(device_matvec_trampoline, Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}))
    nkcheck(@ccall libnk.nk_gmres_set_operator_fn(w.ptr::Ptr{Cvoid}, cb::Ptr{Cvoid}, pointer_from_objref(w.box)::Ptr{Cvoid})::Cint)
    return nothing
end
function bind_device_preconditioner!(w::GMRESWorkspace, side::Cint, P)
    box = OperatorBox(P, w.n)
    side == NK_SIDE_LEFT ? (w.lprecbox = box) : (w.precbox = box)
    cb = @cfunction(device_prec_trampoline, Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}))
    if side == NK_SIDE_LEFT
        nkcheck(@ccall libnk.nk_gmres_set_left_preconditioner(w.ptr::Ptr{Cvoid}, cb::Ptr{Cvoid}, pointer_from_objref(box)::Ptr{Cvoid})::Cint)
    else
        nkcheck(@ccall libnk.nk_gmres_set_right_preconditioner(w.ptr::Ptr{Cvoid}, cb::Ptr{Cvoid}, pointer_from_objref(box)::Ptr{Cvoid})::Cint)
    end
    return nothing
end

end # module
