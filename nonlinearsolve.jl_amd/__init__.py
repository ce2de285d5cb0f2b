"""nonlinearsolve.jl_amd — MI355X-native Newton–Krylov inner loop behind NonlinearSolve.jl's first-order
step path. Import as `nonlinearsolve_jl_amd` (root-level shim; the directory name has a dot in it).

Product code: csrc/ (HIP kernels + C ABI → lib/libmi355x_nk.so), _lib.py (ctypes), core.py (host mirror of
the reference interface). Nothing here imports oracle/."""
from ._lib import NKError, LIB_PATH, build  # noqa: F401
from . import dist  # noqa: F401
from .core import (  # noqa: F401
    Context, default_context, set_default_context, partition_range, comm_unique_id,
    CSRMatrix, DeviceProblem, Quadratic, Bratu2D, Brusselator2D,
    NonlinearFunction, NonlinearProblem,
    KrylovJL_GMRES, ChebyshevPrecs, MultigridPrecs, ObjectPrecs, LinearSolveParameters, Preconditioner, JacobiPreconditioner,
    ILU0Preconditioner, ILUTPreconditioner, AMGPreconditioner, IDENTITY, EisenstatWalkerForcing2, RadiusUpdateSchemes, BackTracking, LineSearchesJL, NewtonRaphson, TrustRegion, GaussNewton, LevenbergMarquardt, PseudoTransient,
    NonlinearLeastSquaresProblem,
    AbsNormSafeBestTerminationMode, NormTerminationMode, RelTerminationMode, RelNormTerminationMode,
    RelNormSafeTerminationMode, RelNormSafeBestTerminationMode, AbsTerminationMode, AbsNormTerminationMode,
    AbsNormSafeTerminationMode, TERMINATION_CONDITIONS,
    NLStats, NonlinearSolution, FirstOrderCache, init, solve, step_, solve_, reinit_,
    supports_deferred_residual, refresh_residual,
    SimpleNewtonRaphson, SimpleTrustRegion, ImmutableNonlinearProblem, EnsembleSolution, vectorized_solve,
    GMRES, BandedLU, JacobianOperator, JacVecOperator, VecJacOperator, StatefulJacobianOperator,
    StatefulJacobianNormalFormOperator,
)
from .polyalg import NonlinearSolvePolyAlgorithm, RobustMultiNewton, FastShortcutNLLSPolyalg, PolyAlgorithmCache  # noqa: F401
