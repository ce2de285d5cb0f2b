"""ctypes binding of libmi355x_nk.so (include/mi355x_nk.h). Fails loudly when the library is missing:
there is no Python/NumPy/torch fallback for any compute entry point."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NK_LIB_PATH") or os.path.join(_HERE, "lib", "libmi355x_nk.so")   # (NK_LIB_PATH: development builds)
CSRC = os.path.join(_HERE, "csrc")


class NKError(RuntimeError):
    pass


# ---------------------------------------------------------------------------- enums (mirror the header)
HOST, DEVICE = 0, 1
RET_NAMES = ["Default", "Success", "MaxIters", "Unstable", "Stalled", "InternalLinearSolveFailed",
             "ShrinkThresholdExceeded", "MaxTime", "Failure", "InternalLineSearchFailed"]
PROBLEM_QUADRATIC, PROBLEM_BRATU2D, PROBLEM_BRUSSELATOR2D, PROBLEM_USER = 1, 2, 3, 100
ALG_NEWTON_RAPHSON, ALG_TRUST_REGION, ALG_GAUSS_NEWTON, ALG_LEVENBERG_MARQUARDT, ALG_PSEUDO_TRANSIENT = 0, 1, 2, 3, 4
LINSOLVE_GMRES_MATFREE, LINSOLVE_GMRES_CSR, LINSOLVE_BANDED_LU = 0, 1, 2
ORTHO_MGS, ORTHO_CGS2, ORTHO_CGS, ORTHO_DCGS2, ORTHO_DCGS2_1R, ORTHO_SSTEP = 0, 1, 2, 3, 4, 5
FORCING_NONE, FORCING_EW2 = 0, 1
COMM_NONE, COMM_RCCL, COMM_CALLBACKS = 0, 1, 2


class Stats(C.Structure):
    _fields_ = [(k, C.c_int64) for k in
                ("nf", "njacs", "nfactors", "nsolve", "nsteps", "gmres_iters", "op_applies", "allreduces",
                 "halo_exchanges")]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class AMGParams(C.Structure):
    _fields_ = [("nu", C.c_int32), ("passes", C.c_int32), ("coarse_max", C.c_int32), ("matching", C.c_int32),
                ("theta", C.c_double), ("overcorrection", C.c_double), ("cheb_ratio", C.c_double)]


class GmresInfo(C.Structure):
    _fields_ = [("iters", C.c_int32), ("restarts", C.c_int32), ("converged", C.c_int32), ("failed", C.c_int32),
                ("rnorm0", C.c_double), ("rnorm", C.c_double)]


class TraceEntry(C.Structure):
    _fields_ = [("iter", C.c_int32), ("gmres_iters", C.c_int32), ("accepted", C.c_int32), ("reserved", C.c_int32),
                ("fnorm_inf", C.c_double), ("step_norm2", C.c_double), ("eta", C.c_double),
                ("trust_region", C.c_double), ("rho", C.c_double)]


class Options(C.Structure):
    _fields_ = [
        ("algorithm", C.c_int32), ("linsolve", C.c_int32), ("maxiters", C.c_int32), ("termination_norm", C.c_int32),
        ("abstol", C.c_double), ("reltol", C.c_double), ("maxtime", C.c_double),
        ("gmres_restart", C.c_int32), ("gmres_maxiters", C.c_int32), ("gmres_ortho", C.c_int32),
        ("gmres_fixed_iters", C.c_int32), ("lin_abstol", C.c_double), ("lin_reltol", C.c_double),
        ("forcing", C.c_int32), ("ew_safeguard", C.c_int32),
        ("ew_eta0", C.c_double), ("ew_eta_max", C.c_double), ("ew_gamma", C.c_double), ("ew_alpha", C.c_double),
        ("ew_safeguard_threshold", C.c_double),
        ("radius_update_scheme", C.c_int32), ("max_shrink_times", C.c_int32),
        ("max_trust_radius", C.c_double), ("initial_trust_radius", C.c_double), ("step_threshold", C.c_double),
        ("shrink_threshold", C.c_double), ("expand_threshold", C.c_double), ("shrink_factor", C.c_double),
        ("expand_factor", C.c_double),
        ("patience_steps", C.c_int32), ("max_stalled_steps", C.c_int32),
        ("patience_objective_multiplier", C.c_double), ("min_max_factor", C.c_double),
        ("protective_threshold", C.c_double),
        ("store_trace", C.c_int32), ("termination_mode", C.c_int32),
        ("cheb_degree", C.c_int32), ("linesearch", C.c_int32), ("cheb_ratio", C.c_double),
        ("ls_c1", C.c_double), ("ls_rho_hi", C.c_double), ("ls_rho_lo", C.c_double),
        ("ls_order", C.c_int32), ("ls_maxiters", C.c_int32), ("mg_nu", C.c_int32), ("mg_coarse", C.c_int32),
        ("jac_colored", C.c_int32), ("lm_disable_geodesic", C.c_int32),
        ("lm_damping_initial", C.c_double), ("lm_damping_increase_factor", C.c_double),
        ("lm_damping_decrease_factor", C.c_double), ("lm_min_damping_D", C.c_double),
        ("lm_alpha_geodesic", C.c_double), ("lm_finite_diff_step_geodesic", C.c_double), ("lm_b_uphill", C.c_double),
        ("pt_alpha_initial", C.c_double),
        ("gmres_sstep", C.c_int32),
        ("gmres_sstep_basis", C.c_int32),
        ("precond_kind", C.c_int32),
        ("precond_side", C.c_int32),
    ]


RESIDUAL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
JVP_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
JACVALS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
MATVEC_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
PRECS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)   # (user, nk_gmres*, nk_csr*, u)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p)
ALLTOALLV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                           C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p)


class UserCallbacks(C.Structure):
    _fields_ = [("residual", RESIDUAL_FN), ("jvp", JVP_FN), ("vjp", JVP_FN), ("jac_values", JACVALS_FN)]


class CommCallbacks(C.Structure):
    _fields_ = [("allreduce", ALLREDUCE_FN), ("alltoallv", ALLTOALLV_FN), ("user", C.c_void_p)]


# every symbol include/mi355x_nk.h declares: (restype, argtypes)
_P, _I, _L, _D = C.c_void_p, C.c_int, C.c_int64, C.c_double
_PP = C.POINTER(C.c_void_p)
SIGNATURES = {
    "nk_version": (C.c_char_p, []),
    "nk_last_error": (C.c_char_p, []),
    "nk_device_count": (_I, [C.POINTER(_I)]),
    "nk_ctx_create": (_I, [_I, _P, _PP]),
    "nk_ctx_destroy": (_I, [_P]),
    "nk_ctx_set_stream": (_I, [_P, _P]),
    "nk_ctx_synchronize": (_I, [_P]),
    "nk_ctx_set_halo_overlap": (_I, [_P, _I]),
    "nk_ctx_set_deterministic": (_I, [_P, _I]),
    "nk_ctx_profile_enable": (_I, [_P, _I]),
    "nk_ctx_profile_kernel_count": (_I, []),
    "nk_ctx_profile_query": (_I, [_P, _I, C.POINTER(C.c_char_p), C.POINTER(_L), C.POINTER(_D), C.POINTER(_D)]),
    "nk_comm_unique_id": (_I, [C.c_char_p]),
    "nk_ctx_comm_init_rccl": (_I, [_P, _I, _I, C.c_char_p]),
    "nk_ctx_comm_init_callbacks": (_I, [_P, _I, _I, C.POINTER(CommCallbacks)]),
    "nk_ctx_comm_info": (_I, [_P, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    "nk_ctx_comm_device_shared": (_I, [_P, C.POINTER(_I)]),
    "nk_ctx_comm_peer_handle": (_I, [_P, _L, C.c_char_p]),
    "nk_ctx_comm_enable_peer": (_I, [_P, C.c_char_p]),
    "nk_ctx_comm_peer_status": (_I, [_P, C.POINTER(_I), C.POINTER(_L)]),
    "nk_ctx_comm_peer_selftest": (_I, [_P, C.POINTER(_I)]),
    "nk_ctx_comm_peer_disable": (_I, [_P]),
    "nk_ctx_comm_allreduce": (_I, [_P, _P, _I, _I]),
    "nk_vec_axpby": (_I, [_P, _L, _D, _P, _D, _P]),
    "nk_vec_fill": (_I, [_P, _L, _D, _P]),
    "nk_partition_range": (_I, [_L, _L, _I, _I, C.POINTER(_L), C.POINTER(_L)]),
    "nk_csr_create": (_I, [_P, _L, _L, _L, _L, _I, _I, _P, _P, _P, _I, _PP]),
    "nk_csr_create_from_csc": (_I, [_P, _L, _L, _I, _I, _P, _P, _P, _PP]),
    "nk_csr_create_from_csc_rows": (_I, [_P, _L, _L, _I, _I, _P, _P, _P, _L, _L, _PP]),
    "nk_csr_destroy": (_I, [_P]),
    "nk_csr_set_values": (_I, [_P, _P, _I]),
    "nk_csr_get_values": (_I, [_P, _P, _I]),
    "nk_csr_set_values_csc": (_I, [_P, _P, _L, _I]),
    "nk_csr_info": (_I, [_P, C.POINTER(_L), C.POINTER(_L), C.POINTER(_L), C.POINTER(_L)]),
    "nk_csr_values_device": (_P, [_P]),
    "nk_spmv": (_I, [_P, _P, _P, _I]),
    "nk_spmv_t": (_I, [_P, _P, _P, _I]),
    "nk_csr_powers": (_I, [_P, _P, _P, _L, _I, _P, _D, _I, _P]),
    "nk_csr_powers_layout": (_I, [_L, _P, _P, _I, _P]),
    "nk_csr_colsumsq": (_I, [_P, _P, _I]),
    "nk_gmres_set_normal_form_damping": (_I, [_P, _P, C.c_double]),
    "nk_gmres_set_shift": (_I, [_P, C.c_double]),
    "nk_gmres_set_block_size": (_I, [_P, _I]),
    "nk_gmres_set_sstep_basis": (_I, [_P, _I]),
    "nk_gmres_set_spectrum_interval": (_I, [_P, C.c_double, C.c_double]),
    "nk_gmres_get_sstep_state": (_I, [_P, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    "nk_gmres_set_shift_weights": (_I, [_P, _P]),
    "nk_problem_create": (_I, [_P, _I, C.POINTER(_D), _I, _PP]),
    "nk_problem_create_user": (_I, [_P, _L, _L, _L, C.POINTER(UserCallbacks), _P, _P, _PP]),
    "nk_problem_destroy": (_I, [_P]),
    "nk_problem_size": (_I, [_P, C.POINTER(_L), C.POINTER(_L), C.POINTER(_L)]),
    "nk_problem_set_params": (_I, [_P, C.POINTER(_D), _I]),
    "nk_problem_initial_guess": (_I, [_P, _P, _I]),
    "nk_residual": (_I, [_P, _P, _P, _I]),
    "nk_jvp": (_I, [_P, _P, _P, _P, _I]),
    "nk_vjp": (_I, [_P, _P, _P, _P, _I]),
    "nk_problem_jac_csr": (_I, [_P, _PP]),
    "nk_jac_values": (_I, [_P, _P, _I, _P]),
    "nk_jac_values_colored": (_I, [_P, _P, _I, _P, C.POINTER(_I)]),
    "nk_gmres_create": (_I, [_P, _L, _I, _I, _PP]),
    "nk_gmres_destroy": (_I, [_P]),
    "nk_gmres_set_operator_csr": (_I, [_P, _P]),
    "nk_gmres_set_operator_jvp": (_I, [_P, _P, _P, _I]),
    "nk_gmres_set_operator_fn": (_I, [_P, MATVEC_FN, _P]),
    "nk_gmres_set_normal_form": (_I, [_P, _I]),
    "nk_gmres_set_right_preconditioner": (_I, [_P, MATVEC_FN, _P]),
    "nk_gmres_set_operator_fn_host": (_I, [_P, MATVEC_FN, _P]),
    "nk_gmres_set_right_preconditioner_host": (_I, [_P, MATVEC_FN, _P]),
    "nk_gmres_set_left_preconditioner": (_I, [_P, MATVEC_FN, _P]),
    "nk_gmres_set_left_preconditioner_host": (_I, [_P, MATVEC_FN, _P]),
    "nk_gmres_set_preconditioner": (_I, [_P, _I, _P]),
    "nk_precond_create_jacobi": (_I, [_P, _PP]),
    "nk_precond_create_ilu0": (_I, [_P, _I, _PP]),
    "nk_precond_create_ilut": (_I, [_P, _D, _PP]),
    "nk_amg_params_default": (_I, [_P]),
    "nk_precond_create_amg": (_I, [_P, _P, _PP]),
    "nk_precond_amg_info": (_I, [_P, C.POINTER(_I), _I, _P, _P, _P]),
    "nk_precond_amg_aggregates": (_I, [_P, _I, _P, _L]),
    "nk_precond_amg_matching": (_I, [_P, C.POINTER(_I)]),
    "nk_precond_update": (_I, [_P]),
    "nk_precond_apply": (_I, [_P, _P, _P, _I]),
    "nk_precond_info": (_I, [_P, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    "nk_precond_ilu0_factors": (_I, [_P, C.POINTER(_L), _P, _P, _P, _P]),
    "nk_precond_destroy": (_I, [_P]),
    "nk_solver_set_precs": (_I, [_P, PRECS_FN, _P]),
    "nk_solver_gmres": (_P, [_P]),
    "nk_solver_jacobian": (_P, [_P]),
    "nk_device_alloc": (_I, [_P, _L, C.POINTER(C.c_void_p)]),
    "nk_device_free": (_I, [_P, _P]),
    "nk_device_copy": (_I, [_P, _P, _P, _L, _I]),
    "nk_gmres_set_chebyshev_preconditioner": (_I, [_P, _I, _D, _D, _D]),
    "nk_gmres_set_multigrid_preconditioner": (_I, [_P, _P, _P, _I, _I, _I]),
    "nk_gmres_get_chebyshev_interval": (_I, [_P, C.POINTER(_D), C.POINTER(_D)]),
    "nk_gmres_solve": (_I, [_P, _P, _P, _I, _I, _D, _D, _I, _I, C.POINTER(GmresInfo)]),
    "nk_lu_create": (_I, [_P, _PP]),
    "nk_lu_destroy": (_I, [_P]),
    "nk_lu_factor": (_I, [_P, _P, C.POINTER(_I)]),
    "nk_lu_solve": (_I, [_P, _P, _P, _I]),
    "nk_lu_info": (_I, [_P, C.POINTER(_I), C.POINTER(_I), C.POINTER(_L)]),
    "nk_lu_engine": (_I, [_P, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    "nk_batch_compile_check": (_I, [C.c_char_p, _I, _I, _I, C.POINTER(_L)]),
    "nk_batch_create": (_I, [_P, C.c_char_p, _I, _I, _I, _PP]),
    "nk_batch_destroy": (_I, [_P]),
    "nk_batch_solve": (_I, [_P, _L, _P, _I, _P, _I, _D, _I, _P, _P, _P, _P]),
    "nk_batch_solve_trust_region": (_I, [_P, _L, _P, _I, _P, _I, _D, _I, _D, _D, _D, _D, _D, _I, _P, _P, _P, _P]),
    "nk_options_default": (_I, [C.POINTER(Options)]),
    "nk_solver_init": (_I, [_P, _P, _I, C.POINTER(Options), _PP]),
    "nk_solver_destroy": (_I, [_P]),
    "nk_solver_step": (_I, [_P]),
    "nk_solver_step_ex": (_I, [_P, _I, _I]),
    "nk_solver_supports_deferred_residual": (_I, [_P, C.POINTER(_I)]),
    "nk_solver_refresh_residual": (_I, [_P]),
    "nk_solver_solve": (_I, [_P, C.POINTER(_I)]),
    "nk_solver_reinit": (_I, [_P, _P, _I, C.POINTER(_D), _I]),
    "nk_solver_set_mass_matrix_diagonal": (_I, [_P, _P, _I]),
    "nk_solver_get_u": (_I, [_P, _P, _I]),
    "nk_solver_get_resid": (_I, [_P, _P, _I]),
    "nk_solver_get_stats": (_I, [_P, C.POINTER(Stats)]),
    "nk_solver_get_retcode": (_I, [_P, C.POINTER(_I), C.POINTER(_I), C.POINTER(_I)]),
    "nk_solver_get_scalars": (_I, [_P, C.POINTER(_D), C.POINTER(_D), C.POINTER(_D)]),
    "nk_solver_get_trace": (_I, [_P, C.POINTER(TraceEntry), _I, C.POINTER(_I)]),
    "nk_newton_solve": (_I, [_P, _P, _I, C.POINTER(Options), _P, _P, C.POINTER(Stats), C.POINTER(_I)]),
    "nk_dot": (_I, [_P, _L, _P, _P, C.POINTER(_D)]),
    "nk_nrm2": (_I, [_P, _L, _P, C.POINTER(_D)]),
    "nk_norm_inf": (_I, [_P, _L, _P, C.POINTER(_D)]),
    "nk_axpy": (_I, [_P, _L, _D, _P, _P]),
    "nk_multidot": (_I, [_P, _L, _I, _P, _L, _P, C.POINTER(_D)]),
    "nk_multiaxpy": (_I, [_P, _L, _I, _P, _L, C.POINTER(_D), _P, C.POINTER(_D)]),
    "nk_fused_axpy_dot": (_I, [_P, _L, _I, _P, _L, C.POINTER(_D), C.POINTER(_D), _P, C.POINTER(_D)]),
}

_lib = None


def build(force: bool = False) -> str:
    """Compile libmi355x_nk.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", CSRC, "-j8"], stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NKError(f"{LIB_PATH} is missing — run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status: int):
    if status != 0:
        raise NKError(f"libmi355x_nk status {status}: {lib().nk_last_error().decode(errors='replace')}")
