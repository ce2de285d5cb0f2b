"""HIP C++ residual sources for the ensemble (kernel-generation) tests, and their numpy twins for the oracle."""
import numpy as np

# f(u, p) = u .* u .- p  (docs/src/tutorials/nonlinear_solve_gpus.md:47-62 and common_rootfind_testing.jl:15-17)
QUADRATIC = """
template <typename T> __device__ void nk_f(const T *u, const double *p, T *f) {
  for (int i = 0; i < NK_N; ++i) f[i] = u[i] * u[i] - p[i];
}
"""

# p2_f of the tutorial (nonlinear_solve_gpus.md:120-127)
P2 = """
template <typename T> __device__ void nk_f(const T *x, const double *p, T *out) {
  out[0] = x[0] + p[0] * x[1];
  out[1] = sqrt(p[1]) * (x[2] - x[3]);
  out[2] = (x[1] - p[2] * x[2]) * (x[1] - p[2] * x[2]);
  out[3] = sqrt(p[3]) * (x[0] - x[3]) * (x[0] - x[3]);
}
"""

# a coupled transcendental system with a user-supplied analytic Jacobian (SciMLBase.has_jac path)
TRIG_WITH_JAC = """
template <typename T> __device__ void nk_f(const T *u, const double *p, T *f) {
  f[0] = exp(u[0]) + u[1] * u[2] - p[0];
  f[1] = sin(u[1]) + u[0] * u[0] - p[1];
  f[2] = u[2] * u[2] * u[2] + tanh(u[0]) - p[2];
}
__device__ void nk_jac(const double *u, const double *p, double *J) {
  J[0] = exp(u[0]);            J[1] = u[2];      J[2] = u[1];
  J[3] = 2.0 * u[0];           J[4] = cos(u[1]); J[5] = 0.0;
  const double t = tanh(u[0]);
  J[6] = 1.0 - t * t;          J[7] = 0.0;       J[8] = 3.0 * u[2] * u[2];
}
"""


def quadratic_f(u, p):
    return u * u - p


def quadratic_jac(u, p):
    return np.diag(2.0 * u)


def p2_f(x, p):
    return np.array([x[0] + p[0] * x[1], np.sqrt(p[1]) * (x[2] - x[3]), (x[1] - p[2] * x[2]) ** 2,
                     np.sqrt(p[3]) * (x[0] - x[3]) ** 2])


def p2_jac(x, p):
    s1, s3 = np.sqrt(p[1]), np.sqrt(p[3])
    d = x[1] - p[2] * x[2]
    e = x[0] - x[3]
    return np.array([[1.0, p[0], 0.0, 0.0], [0.0, 0.0, s1, -s1], [0.0, 2 * d, -2 * p[2] * d, 0.0],
                     [2 * s3 * e, 0.0, 0.0, -2 * s3 * e]])


def trig_f(u, p):
    return np.array([np.exp(u[0]) + u[1] * u[2] - p[0], np.sin(u[1]) + u[0] ** 2 - p[1], u[2] ** 3 + np.tanh(u[0]) - p[2]])


def trig_jac(u, p):
    t = np.tanh(u[0])
    return np.array([[np.exp(u[0]), u[2], u[1]], [2 * u[0], np.cos(u[1]), 0.0], [1 - t * t, 0.0, 3 * u[2] ** 2]])
