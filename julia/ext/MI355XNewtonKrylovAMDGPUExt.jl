# Package extension of MI355XNewtonKrylov for AMDGPU.jl (weak dependency; julia/Project.toml: [weakdeps] AMDGPU, [extensions]
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

# ---- stream ordering. The library calls device callbacks ASYNCHRONOUSLY on its own stream (`stream`, the callback's 4th
# argument): the kernels that produce x may still be queued there, and its next kernel reads y there. mul! / ldiv! run on
# AMDGPU.jl's task-local stream. The two are ordered with HIP events — no host synchronisation per application:
#     event ← record(library stream);  wait(julia stream, event);  mul!/ldiv!;  event ← record(julia stream);
#     wait(library stream, event)
# through the HIP C API itself (libamdhip64), so nothing here depends on AMDGPU.jl internals except the raw handle of the
# task-local stream (`AMDGPU.stream().stream` [EXT]); if that cannot be had, the trampolines fall back to synchronising the
# library's stream before and the device after the call (correct, slower).
const libhip = "libamdhip64"
const ORDER_EVENT = Ref{Ptr{Cvoid}}(C_NULL)
function order_event()
    if ORDER_EVENT[] == C_NULL
        ev = Ref{Ptr{Cvoid}}(C_NULL)
        rc = @ccall libhip.hipEventCreateWithFlags(ev::Ptr{Ptr{Cvoid}}, 0x2::Cuint)::Cint      # hipEventDisableTiming
        rc == 0 || error("hipEventCreateWithFlags failed ($rc)")
        ORDER_EVENT[] = ev[]
    end
    return ORDER_EVENT[]
end
# everything enqueued on `dst` from now on waits for what `src` holds now
function order_after(dst::Ptr{Cvoid}, src::Ptr{Cvoid})
    ev = order_event()
    rc = @ccall libhip.hipEventRecord(ev::Ptr{Cvoid}, src::Ptr{Cvoid})::Cint
    rc == 0 || error("hipEventRecord failed ($rc)")
    rc = @ccall libhip.hipStreamWaitEvent(dst::Ptr{Cvoid}, ev::Ptr{Cvoid}, 0::Cuint)::Cint
    rc == 0 || error("hipStreamWaitEvent failed ($rc)")
    return nothing
end
julia_stream_handle() = Ptr{Cvoid}(UInt(AMDGPU.stream().stream))                                # [EXT] hipStream_t of this task
function on_library_stream(f, stream::Ptr{Cvoid})
    js = try
        julia_stream_handle()
    catch
        nothing
    end
    if js === nothing                       # no handle: host-synchronise both sides
        rc = @ccall libhip.hipStreamSynchronize(stream::Ptr{Cvoid})::Cint
        rc == 0 || error("hipStreamSynchronize failed ($rc)")
        f()
        AMDGPU.synchronize()
    else
        order_after(js, stream)             # x is complete before mul! / ldiv! starts
        f()
        order_after(stream, js)             # the library's next kernel sees y
    end
    return nothing
end

function device_matvec_trampoline(user::Ptr{Cvoid}, x::Ptr{Float64}, y::Ptr{Float64}, stream::Ptr{Cvoid})::Cint
    box = unsafe_pointer_to_objref(user)::OperatorBox
    try
        on_library_stream(stream) do
            mul!(wrap(y, box.n), box.A, wrap(x, box.n))
        end
        return Cint(0)
    catch
        return Cint(1)
    end
end
function device_prec_trampoline(user::Ptr{Cvoid}, x::Ptr{Float64}, y::Ptr{Float64}, stream::Ptr{Cvoid})::Cint
    box = unsafe_pointer_to_objref(user)::OperatorBox
    try
        on_library_stream(stream) do
            ldiv!(wrap(y, box.n), box.A, wrap(x, box.n))
        end
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
