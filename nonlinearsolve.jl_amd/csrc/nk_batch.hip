// Ensembles of small dense nonlinear systems, one system per GPU thread — the "kernel generation" use case of the
// reference (docs/src/tutorials/nonlinear_solve_gpus.md:70-176): `SimpleNewtonRaphson` (lib/SimpleNonlinearSolve/src/
// raphson.jl:39-83) called from inside a KernelAbstractions kernel over a vector of parameters. There the residual is
// Julia code that GPUCompiler specialises into the kernel; here it is HIP C++ source handed over as a string and
// compiled at run time with hiprtc together with the solver kernel for the given sizes (`-DNK_N`, `-DNK_NP`), so the
// residual, the forward-mode dual-number Jacobian (AutoForwardDiff is the reference's default, raphson.jl:28-33) and the
// pivoted LU are one straight-line register program per thread when N ≤ 8.
//
// Reference semantics kept (raphson.jl:50-82): `iszero(fx)` short cut; J evaluated at the current iterate; per iteration
// δx = J \ fx, x −= δx, THEN the termination check on the residual of the previous iterate (`check_termination` precedes
// `evaluate_f!!`), default mode AbsNormTerminationMode(maximum∘abs) (termination_conditions.jl:376-380), default abstol
// eps^(4/5) (common_defaults.jl:39-48), maxiters 1000; retcodes Success / MaxIters; a NaN residual never terminates.
#include <dlfcn.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "nk_internal.h"

// ----------------------------------------------------------------------------- device source (compiled by hiprtc)
static const char *k_prelude = R"NKSRC(
// ---- forward-mode dual numbers with NK_CH partials (ForwardDiff.Dual analogue)
struct Dual {
  double v;
  double d[NK_CH];
  __device__ Dual() {}
  __device__ Dual(double x) : v(x) {
#pragma unroll
    for (int k = 0; k < NK_CH; ++k) d[k] = 0.0;
  }
};
#define NK_DUAL_LOOP _Pragma("unroll") for (int k = 0; k < NK_CH; ++k)
__device__ inline Dual operator+(const Dual &a, const Dual &b) { Dual r; r.v = a.v + b.v; NK_DUAL_LOOP r.d[k] = a.d[k] + b.d[k]; return r; }
__device__ inline Dual operator-(const Dual &a, const Dual &b) { Dual r; r.v = a.v - b.v; NK_DUAL_LOOP r.d[k] = a.d[k] - b.d[k]; return r; }
__device__ inline Dual operator-(const Dual &a) { Dual r; r.v = -a.v; NK_DUAL_LOOP r.d[k] = -a.d[k]; return r; }
__device__ inline Dual operator*(const Dual &a, const Dual &b) { Dual r; r.v = a.v * b.v; NK_DUAL_LOOP r.d[k] = a.d[k] * b.v + a.v * b.d[k]; return r; }
__device__ inline Dual operator/(const Dual &a, const Dual &b) {
  Dual r; const double ib = 1.0 / b.v; r.v = a.v * ib;
  NK_DUAL_LOOP r.d[k] = (a.d[k] - r.v * b.d[k]) * ib;
  return r;
}
__device__ inline Dual operator+(const Dual &a, double b) { Dual r = a; r.v += b; return r; }
__device__ inline Dual operator+(double b, const Dual &a) { Dual r = a; r.v += b; return r; }
__device__ inline Dual operator-(const Dual &a, double b) { Dual r = a; r.v -= b; return r; }
__device__ inline Dual operator-(double b, const Dual &a) { Dual r = -a; r.v += b; return r; }
__device__ inline Dual operator*(const Dual &a, double b) { Dual r; r.v = a.v * b; NK_DUAL_LOOP r.d[k] = a.d[k] * b; return r; }
__device__ inline Dual operator*(double b, const Dual &a) { return a * b; }
__device__ inline Dual operator/(const Dual &a, double b) { return a * (1.0 / b); }
__device__ inline Dual operator/(double b, const Dual &a) { return Dual(b) / a; }
__device__ inline Dual &operator+=(Dual &a, const Dual &b) { a = a + b; return a; }
__device__ inline Dual &operator-=(Dual &a, const Dual &b) { a = a - b; return a; }
__device__ inline Dual &operator*=(Dual &a, const Dual &b) { a = a * b; return a; }
__device__ inline Dual &operator/=(Dual &a, const Dual &b) { a = a / b; return a; }
__device__ inline bool operator<(const Dual &a, const Dual &b) { return a.v < b.v; }
__device__ inline bool operator>(const Dual &a, const Dual &b) { return a.v > b.v; }
__device__ inline bool operator<=(const Dual &a, const Dual &b) { return a.v <= b.v; }
__device__ inline bool operator>=(const Dual &a, const Dual &b) { return a.v >= b.v; }
__device__ inline Dual nk_chain(const Dual &a, double fv, double dfv) { Dual r; r.v = fv; NK_DUAL_LOOP r.d[k] = dfv * a.d[k]; return r; }
__device__ inline Dual sqrt(const Dual &a) { const double s = sqrt(a.v); return nk_chain(a, s, 0.5 / s); }
__device__ inline Dual exp(const Dual &a) { const double e = exp(a.v); return nk_chain(a, e, e); }
__device__ inline Dual log(const Dual &a) { return nk_chain(a, log(a.v), 1.0 / a.v); }
__device__ inline Dual sin(const Dual &a) { return nk_chain(a, sin(a.v), cos(a.v)); }
__device__ inline Dual cos(const Dual &a) { return nk_chain(a, cos(a.v), -sin(a.v)); }
__device__ inline Dual tan(const Dual &a) { const double t = tan(a.v); return nk_chain(a, t, 1.0 + t * t); }
__device__ inline Dual tanh(const Dual &a) { const double t = tanh(a.v); return nk_chain(a, t, 1.0 - t * t); }
__device__ inline Dual atan(const Dual &a) { return nk_chain(a, atan(a.v), 1.0 / (1.0 + a.v * a.v)); }
__device__ inline Dual fabs(const Dual &a) { return nk_chain(a, fabs(a.v), a.v < 0.0 ? -1.0 : 1.0); }
__device__ inline Dual pow(const Dual &a, double e) { const double pw = pow(a.v, e - 1.0); return nk_chain(a, pw * a.v, e * pw); }
__device__ inline Dual pow(const Dual &a, int e) { return pow(a, (double)e); }
__device__ inline Dual pow(const Dual &a, const Dual &b) { return exp(b * log(a)); }
)NKSRC";

static const char *k_kernel = R"NKSRC(
#if NK_N <= 8
#define NK_UNROLL _Pragma("unroll")
#else
#define NK_UNROLL _Pragma("nounroll")
#endif

__device__ inline void nk_jacobian(const double *x, const double *p, double (*J)[NK_N]) {
#ifdef NK_HAS_JAC
  nk_jac(x, p, &J[0][0]);  // user-supplied analytic Jacobian, row-major N×N (SciMLBase.has_jac, utils.jl:98-99)
#else
  // AutoForwardDiff: NK_CH directions per sweep of the residual on dual numbers
  NK_UNROLL for (int c0 = 0; c0 < NK_N; c0 += NK_CH) {
    Dual xd[NK_N], fd[NK_N];
    NK_UNROLL for (int i = 0; i < NK_N; ++i) {
      xd[i].v = x[i];
      NK_DUAL_LOOP xd[i].d[k] = (i == c0 + k) ? 1.0 : 0.0;
    }
    nk_f<Dual>(xd, p, fd);
    NK_UNROLL for (int i = 0; i < NK_N; ++i) {
      NK_DUAL_LOOP if (c0 + k < NK_N) J[i][c0 + k] = fd[i].d[k];
    }
  }
#endif
}

// dx = A \ b by Gaussian elimination with partial pivoting (A and b are destroyed)
__device__ inline void nk_lu_solve(double (*A)[NK_N], double *b, double *dx) {
  NK_UNROLL for (int c = 0; c < NK_N; ++c) {
    int piv = c;
    double best = fabs(A[c][c]);
    NK_UNROLL for (int r = c + 1; r < NK_N; ++r) {
      const double v = fabs(A[r][c]);
      if (v > best) { best = v; piv = r; }
    }
#if NK_N <= 8
    // row exchange by selects, so that every index stays a compile-time constant and the matrix lives in registers
    NK_UNROLL for (int r = c + 1; r < NK_N; ++r) {
      const bool s = (r == piv);
      NK_UNROLL for (int k = c; k < NK_N; ++k) {
        const double t1 = A[c][k], t2 = A[r][k];
        A[c][k] = s ? t2 : t1;
        A[r][k] = s ? t1 : t2;
      }
      const double b1 = b[c], b2 = b[r];
      b[c] = s ? b2 : b1;
      b[r] = s ? b1 : b2;
    }
#else
    if (piv != c) {
      for (int k = c; k < NK_N; ++k) { const double t = A[c][k]; A[c][k] = A[piv][k]; A[piv][k] = t; }
      const double t = b[c]; b[c] = b[piv]; b[piv] = t;
    }
#endif
    const double inv = 1.0 / A[c][c];
    NK_UNROLL for (int r = c + 1; r < NK_N; ++r) {
      const double l = A[r][c] * inv;
      NK_UNROLL for (int k = c + 1; k < NK_N; ++k) A[r][k] -= l * A[c][k];
      b[r] -= l * b[c];
    }
  }
  NK_UNROLL for (int r = NK_N - 1; r >= 0; --r) {
    double s = b[r];
    NK_UNROLL for (int k = r + 1; k < NK_N; ++k) s -= A[r][k] * dx[k];
    dx[r] = s / A[r][r];
  }
}

extern "C" __global__ void __launch_bounds__(NK_BLOCK_T)
nk_batch_newton(long nbatch, const double *__restrict__ u0, int u0_per_system, const double *__restrict__ p, double abstol,
                int maxiters, double *__restrict__ u_out, double *__restrict__ r_out, int *__restrict__ retcode,
                int *__restrict__ iters) {
  const long b = (long)blockIdx.x * NK_BLOCK_T + threadIdx.x;
  if (b >= nbatch) return;
  double x[NK_N], fx[NK_N], dx[NK_N], pp[NK_NP > 0 ? NK_NP : 1];
  NK_UNROLL for (int i = 0; i < NK_N; ++i) x[i] = u0[(u0_per_system ? b * NK_N : 0) + i];
  NK_UNROLL for (int i = 0; i < NK_NP; ++i) pp[i] = p[b * NK_NP + i];
  nk_f<double>(x, pp, fx);
  bool allzero = true;
  NK_UNROLL for (int i = 0; i < NK_N; ++i) allzero = allzero && (fx[i] == 0.0);
  int rc = 2 /* MaxIters */, it = 0;
  if (allzero) {
    rc = 1;  // Success (raphson.jl:55-56)
  } else {
    double J[NK_N][NK_N], A[NK_N][NK_N], rhs[NK_N];
    nk_jacobian(x, pp, J);
    for (it = 1; it <= maxiters; ++it) {
      NK_UNROLL for (int i = 0; i < NK_N; ++i) {
        rhs[i] = fx[i];
        NK_UNROLL for (int k = 0; k < NK_N; ++k) A[i][k] = J[i][k];
      }
      nk_lu_solve(A, rhs, dx);
      NK_UNROLL for (int i = 0; i < NK_N; ++i) x[i] -= dx[i];
      // AbsNormTerminationMode(maximum∘abs) on the residual of the PREVIOUS iterate (the check precedes evaluate_f!!);
      // maximum propagates NaN, and NaN <= abstol is false
      double nrm = 0.0;
      bool nan = false;
      NK_UNROLL for (int i = 0; i < NK_N; ++i) { const double a = fabs(fx[i]); nan = nan || (a != a); nrm = a > nrm ? a : nrm; }
      if (!nan && nrm <= abstol) { rc = 1; break; }
      nk_f<double>(x, pp, fx);
      nk_jacobian(x, pp, J);
    }
    if (it > maxiters) it = maxiters;
  }
  NK_UNROLL for (int i = 0; i < NK_N; ++i) { u_out[b * NK_N + i] = x[i]; r_out[b * NK_N + i] = fx[i]; }
  retcode[b] = rc;
  iters[b] = it;
}
// ---- SimpleTrustRegion (lib/SimpleNonlinearSolve/src/trust_region.jl:57-229, default update rule): dogleg step
// (Newton step if inside the region, else −g clipped to Δ, else the boundary point of the segment), ratio
// r = (f_{k+1} − f_k)/(δ·g + δ·Hδ/2) with H = JᵀJ, g = Jᵀf; shrink by t₁ when r < η₂ (ShrinkThresholdExceeded after
// max_shrink consecutive shrinks), accept when r ≥ η₁ (termination test on the NEW residual, then J, g at the new point,
// expand by t₂ up to Δmax when r > η₃). Δmax = max(‖f(u0)‖₂, max(u0) − min(u0)), Δ0 = Δmax/11.
__device__ inline double nk_norm2(const double *v) {
  double s = 0.0;
  NK_UNROLL for (int i = 0; i < NK_N; ++i) s += v[i] * v[i];
  return sqrt(s);
}
extern "C" __global__ void __launch_bounds__(NK_BLOCK_T)
nk_batch_trust_region(long nbatch, const double *__restrict__ u0, int u0_per_system, const double *__restrict__ p, double abstol,
                      int maxiters, double eta1, double eta2, double eta3, double t1, double t2, int max_shrink,
                      double *__restrict__ u_out, double *__restrict__ r_out, int *__restrict__ retcode, int *__restrict__ iters) {
  const long b = (long)blockIdx.x * NK_BLOCK_T + threadIdx.x;
  if (b >= nbatch) return;
  double x[NK_N], xo[NK_N], fx[NK_N], g[NK_N], dl[NK_N], dN[NK_N], dsd[NK_N], tmp[NK_N], pp[NK_NP > 0 ? NK_NP : 1];
  double J[NK_N][NK_N], A[NK_N][NK_N];
  NK_UNROLL for (int i = 0; i < NK_N; ++i) { x[i] = u0[(u0_per_system ? b * NK_N : 0) + i]; xo[i] = x[i]; }
  NK_UNROLL for (int i = 0; i < NK_NP; ++i) pp[i] = p[b * NK_NP + i];
  nk_f<double>(x, pp, fx);
  const double norm_fx = nk_norm2(fx);
  nk_jacobian(x, pp, J);
  double xmax = x[0], xmin = x[0];
  NK_UNROLL for (int i = 1; i < NK_N; ++i) { xmax = x[i] > xmax ? x[i] : xmax; xmin = x[i] < xmin ? x[i] : xmin; }
  const double dmax = norm_fx > xmax - xmin ? norm_fx : xmax - xmin;
  double delta = dmax / 11.0;
  double fk = 0.5 * norm_fx * norm_fx;
  NK_UNROLL for (int i = 0; i < NK_N; ++i) { double s = 0.0; NK_UNROLL for (int k = 0; k < NK_N; ++k) s += J[k][i] * fx[k]; g[i] = s; }
  int shrink = 0, rc = 2, it = 0;
  auto absmax_ok = [&](const double *f) {
    double nrm = 0.0; bool nan = false;
    NK_UNROLL for (int i = 0; i < NK_N; ++i) { const double a = fabs(f[i]); nan = nan || (a != a); nrm = a > nrm ? a : nrm; }
    return !nan && nrm <= abstol;
  };
  if (absmax_ok(fx)) rc = 1;
  else {
    for (it = 1; it <= maxiters; ++it) {
      // dogleg
      NK_UNROLL for (int i = 0; i < NK_N; ++i) { tmp[i] = fx[i]; NK_UNROLL for (int k = 0; k < NK_N; ++k) A[i][k] = J[i][k]; }
      nk_lu_solve(A, tmp, dN);
      NK_UNROLL for (int i = 0; i < NK_N; ++i) dN[i] = -dN[i];
      if (nk_norm2(dN) <= delta) {
        NK_UNROLL for (int i = 0; i < NK_N; ++i) dl[i] = dN[i];
      } else {
        NK_UNROLL for (int i = 0; i < NK_N; ++i) dsd[i] = -g[i];
        const double nsd = nk_norm2(dsd);
        if (nsd >= delta) {
          NK_UNROLL for (int i = 0; i < NK_N; ++i) dl[i] = dsd[i] * (delta / nsd);
        } else {
          double dNN = 0.0, dSN = 0.0, dSS = 0.0;
          NK_UNROLL for (int i = 0; i < NK_N; ++i) { const double q = dN[i] - dsd[i]; dNN += q * q; dSN += dsd[i] * q; dSS += dsd[i] * dsd[i]; }
          const double fact = dSN * dSN - dNN * (dSS - delta * delta);
          const double tau = (-dSN + sqrt(fact)) / dNN;
          NK_UNROLL for (int i = 0; i < NK_N; ++i) dl[i] = dsd[i] + tau * (dN[i] - dsd[i]);
        }
      }
      NK_UNROLL for (int i = 0; i < NK_N; ++i) x[i] = xo[i] + dl[i];
      nk_f<double>(x, pp, fx);
      const double nf = nk_norm2(fx);
      const double fk1 = nf * nf / 2.0;
      // Hδ = Jᵀ(Jδ)
      NK_UNROLL for (int i = 0; i < NK_N; ++i) { double s = 0.0; NK_UNROLL for (int k = 0; k < NK_N; ++k) s += J[i][k] * dl[k]; tmp[i] = s; }
      double dg = 0.0, dHd = 0.0;
      NK_UNROLL for (int i = 0; i < NK_N; ++i) {
        double s = 0.0;
        NK_UNROLL for (int k = 0; k < NK_N; ++k) s += J[k][i] * tmp[k];
        dHd += dl[i] * s;
        dg += dl[i] * g[i];
      }
      const double r = (fk1 - fk) / (dg + dHd / 2.0);
      if (r >= eta2) shrink = 0;
      else {
        delta = t1 * delta;
        if (++shrink > max_shrink) { rc = 6; break; }   // ShrinkThresholdExceeded
      }
      if (r >= eta1) {
        if (absmax_ok(fx)) { rc = 1; break; }
        NK_UNROLL for (int i = 0; i < NK_N; ++i) xo[i] = x[i];
        nk_jacobian(x, pp, J);
        if (r > eta3) delta = t2 * delta < dmax ? t2 * delta : dmax;
        fk = fk1;
        NK_UNROLL for (int i = 0; i < NK_N; ++i) { double s = 0.0; NK_UNROLL for (int k = 0; k < NK_N; ++k) s += J[k][i] * fx[k]; g[i] = s; }
      }
    }
    if (it > maxiters) it = maxiters;
  }
  NK_UNROLL for (int i = 0; i < NK_N; ++i) { u_out[b * NK_N + i] = x[i]; r_out[b * NK_N + i] = fx[i]; }
  retcode[b] = rc;
  iters[b] = it;
}
)NKSRC";


// ----------------------------------------------------------------------------- one system per WAVEFRONT (8 < n ≤ 64)
// The per-thread kernel keeps the n×n Jacobian in registers only while n ≤ 8; beyond that it runs from scratch memory
// (measured ≈ 1 TFLOP/s against 16–27 TFLOP/s for n ≤ 8). Here a wavefront owns one system and lane j owns COLUMN j of the
// Jacobian: lane j evaluates the residual on dual numbers seeded with e_j (one partial), which gives it its column; the LU
// uses COLUMN pivoting — the pivot of row c is the largest entry of that row among the columns not used yet, found with a
// wave reduction, so no register ever moves; the multipliers l_r = a[r][p]/a[c][p] live in lane p and reach the others through
// v_readlane with a scalar lane index; every lane then updates its own column with register indices that are compile-time
// constants. The right-hand side, x and f(x) are replicated in all lanes. (nk_jac, if supplied, is not used by this kernel.)
static const char *k_kernel_wave = R"NKSRC(
__device__ inline double nk_rl(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
__device__ inline double nk_wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const double w = __shfl_xor(v, o, 64); v = w > v ? w : v; }
  return v;
}
extern "C" __global__ void __launch_bounds__(NK_BLOCK_T)
nk_batch_newton_wave(long nbatch, const double *__restrict__ u0, int u0_per_system, const double *__restrict__ p, double abstol,
                     int maxiters, double *__restrict__ u_out, double *__restrict__ r_out, int *__restrict__ retcode,
                     int *__restrict__ iters) {
  const int lane = threadIdx.x & 63;
  const long b = (long)blockIdx.x * (NK_BLOCK_T / 64) + (threadIdx.x >> 6);
  if (b >= nbatch) return;   // whole wavefronts leave together
  double x[NK_N], fx[NK_N], col[NK_N], pp[NK_NP > 0 ? NK_NP : 1];  // fx doubles as the right-hand side of the solve
#pragma unroll
  for (int i = 0; i < NK_N; ++i) x[i] = u0[(u0_per_system ? b * NK_N : 0) + i];
#pragma unroll
  for (int i = 0; i < NK_NP; ++i) pp[i] = p[b * NK_NP + i];
  nk_f<double>(x, pp, fx);
  bool allzero = true;
#pragma unroll
  for (int i = 0; i < NK_N; ++i) allzero = allzero && (fx[i] == 0.0);
  int rc = 2, it = 0;
  if (allzero) rc = 1;
  else {
    for (it = 1; it <= maxiters; ++it) {
      {  // column `lane` of J by one dual-number sweep (AutoForwardDiff with a single partial per lane)
        Dual xd[NK_N], fd[NK_N];
#pragma unroll
        for (int i = 0; i < NK_N; ++i) { xd[i].v = x[i]; xd[i].d[0] = (i == lane) ? 1.0 : 0.0; }
        nk_f<Dual>(xd, pp, fd);
#pragma unroll
        for (int i = 0; i < NK_N; ++i) col[i] = (lane < NK_N) ? fd[i].d[0] : 0.0;
      }
      // AbsNormTerminationMode(maximum∘abs) on the residual of the iterate the step starts from — the quantity the reference
      // tests AFTER the update (raphson.jl:72-75); taken here because the solve below overwrites fx
      double nrm = 0.0;
      bool nan = false;
#pragma unroll
      for (int i = 0; i < NK_N; ++i) { const double a = fabs(fx[i]); nan = nan || (a != a); nrm = a > nrm ? a : nrm; }
      const bool converged = !nan && nrm <= abstol;
      double *rhs = fx;
      // ---- LU with column pivoting, applied to the right-hand side on the fly
      bool used = lane >= NK_N;   // lanes without a column never pivot
      int perm[NK_N];
#pragma unroll
      for (int c = 0; c < NK_N; ++c) {
        const double mine = used ? -1.0 : fabs(col[c]);
        const double best = nk_wave_max(mine);
        const unsigned long long m = __ballot(!used && mine == best);
        const int pl = m ? (int)__ffsll((long long)m) - 1 : 0;   // (a NaN row: every compare fails — take lane 0, NaNs propagate)
        const int pv = __builtin_amdgcn_readfirstlane(pl);
        perm[c] = pv;
        const double inv = 1.0 / nk_rl(col[c], pv);
        if (lane == pv) used = true;
        const double cc = col[c];
#pragma unroll
        for (int r = c + 1; r < NK_N; ++r) {
          const double l = nk_rl(col[r], pv) * inv;
          if (!used) col[r] -= l * cc;
          rhs[r] -= l * rhs[c];
        }
      }
      // ---- back substitution: the unknown of step c belongs to lane perm[c]
      double mydx = 0.0;
#pragma unroll
      for (int c = NK_N - 1; c >= 0; --c) {
        const int pv = perm[c];
        const double xc = rhs[c] / nk_rl(col[c], pv);
        if (lane == pv) mydx = xc;
#pragma unroll
        for (int r = 0; r < c; ++r) rhs[r] -= nk_rl(col[r], pv) * xc;
      }
#pragma unroll
      for (int i = 0; i < NK_N; ++i) x[i] -= nk_rl(mydx, i);
      if (converged) {  // the reference returns the residual of the iterate the last step started from: rebuild it (x + δ)
        rc = 1;
        double xp[NK_N];
#pragma unroll
        for (int i = 0; i < NK_N; ++i) xp[i] = x[i] + nk_rl(mydx, i);
        nk_f<double>(xp, pp, fx);
        break;
      }
      nk_f<double>(x, pp, fx);
    }
    if (it > maxiters) it = maxiters;
  }
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NK_N; ++i) { u_out[b * NK_N + i] = x[i]; r_out[b * NK_N + i] = fx[i]; }
    retcode[b] = rc;
    iters[b] = it;
  }
}
)NKSRC";

// ----------------------------------------------------------------------------- hiprtc through dlopen
typedef void *rtc_program;
static struct {
  void *h = nullptr;
  int (*Create)(rtc_program *, const char *, const char *, int, const char **, const char **) = nullptr;
  int (*Compile)(rtc_program, int, const char **) = nullptr;
  int (*LogSize)(rtc_program, size_t *) = nullptr;
  int (*Log)(rtc_program, char *) = nullptr;
  int (*CodeSize)(rtc_program, size_t *) = nullptr;
  int (*Code)(rtc_program, char *) = nullptr;
  int (*Destroy)(rtc_program *) = nullptr;
} RTC;

static int rtc_load() {
  if (RTC.h) return NK_OK;
  const char *names[] = {"libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"};
  for (const char *nm : names) {
    RTC.h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    if (RTC.h) break;
  }
  if (!RTC.h) NK_FAIL(NK_E_UNSUPPORTED, "cannot dlopen libhiprtc (needed to compile the residual): %s", dlerror());
#define RTC_SYM(field, name)                                                          \
  RTC.field = (decltype(RTC.field))dlsym(RTC.h, name);                                \
  if (!RTC.field) { RTC.h = nullptr; NK_FAIL(NK_E_UNSUPPORTED, "libhiprtc lacks symbol %s", name); }
  RTC_SYM(Create, "hiprtcCreateProgram");
  RTC_SYM(Compile, "hiprtcCompileProgram");
  RTC_SYM(LogSize, "hiprtcGetProgramLogSize");
  RTC_SYM(Log, "hiprtcGetProgramLog");
  RTC_SYM(CodeSize, "hiprtcGetCodeSize");
  RTC_SYM(Code, "hiprtcGetCode");
  RTC_SYM(Destroy, "hiprtcDestroyProgram");
#undef RTC_SYM
  return NK_OK;
}

struct nk_batch {
  nk_ctx *ctx = nullptr;
  int n = 0, np = 0, block = 64;
  hipModule_t mod = nullptr, mod_wave = nullptr;
  hipFunction_t fn = nullptr, fn_tr = nullptr, fn_wave = nullptr;  // fn_wave: one system per wavefront (8 < n ≤ 64)
  // staging for host-memspace calls
  double *d_u0 = nullptr, *d_p = nullptr, *d_u = nullptr, *d_r = nullptr;
  int *d_rc = nullptr, *d_it = nullptr;
  int64_t cap = 0;
};

// compile `source` (+ prelude + solver kernel) for n unknowns / np parameters; code object into `code`, log into `log`
static int batch_compile(const char *source, int n, int np, int flags, std::vector<char> *code, std::string *log,
                         bool wave = false) {
  NK_REQUIRE(source, "NULL source");
  NK_REQUIRE(n >= 1 && n <= 64, "n = %d outside 1..64 (one system per thread)", n);
  NK_REQUIRE(np >= 0 && np <= 256, "nparams = %d outside 0..256", np);
  NK_TRY(rtc_load());
  std::string full = std::string(k_prelude) + "\n// ---- user source\n" + source + "\n" + (wave ? k_kernel_wave : k_kernel);
  rtc_program prog = nullptr;
  if (RTC.Create(&prog, full.c_str(), "nk_batch_user.hip", 0, nullptr, nullptr) != 0) NK_FAIL(NK_E_HIP, "hiprtcCreateProgram failed");
  const int ch = wave ? 1 : (n < 8 ? n : 8);  // dual-number partials per residual sweep (wave kernel: one per lane)
  const std::string dn = "-DNK_N=" + std::to_string(n), dp = "-DNK_NP=" + std::to_string(np), dc = "-DNK_CH=" + std::to_string(ch),
                    db = wave ? "-DNK_BLOCK_T=256" : "-DNK_BLOCK_T=64";
  std::vector<const char *> opts = {"--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", dn.c_str(), dp.c_str(), dc.c_str(),
                                    db.c_str()};
  if ((flags & 1) && !wave) opts.push_back("-DNK_HAS_JAC=1");
  const int rc = RTC.Compile(prog, (int)opts.size(), opts.data());
  size_t ls = 0;
  RTC.LogSize(prog, &ls);
  if (ls > 1 && log) { log->resize(ls); RTC.Log(prog, &(*log)[0]); }
  if (rc != 0) {
    RTC.Destroy(&prog);
    NK_FAIL(NK_E_INVALID, "residual source does not compile: %.900s", log && !log->empty() ? log->c_str() : "(no log)");
  }
  size_t cs = 0;
  RTC.CodeSize(prog, &cs);
  code->resize(cs);
  RTC.Code(prog, code->data());
  RTC.Destroy(&prog);
  return NK_OK;
}

// compile only (no device needed): the "does the user's residual build for gfx950" check
extern "C" int nk_batch_compile_check(const char *source, int n, int nparams, int flags, int64_t *code_bytes) {
  std::vector<char> code;
  std::string log;
  NK_TRY(batch_compile(source, n, nparams, flags, &code, &log));
  if (code_bytes) *code_bytes = (int64_t)code.size();
  if (n > 8) {  // medium systems also get the per-wavefront Newton kernel
    std::vector<char> wcode;
    std::string wlog;
    NK_TRY(batch_compile(source, n, nparams, flags, &wcode, &wlog, true));
    if (code_bytes) *code_bytes += (int64_t)wcode.size();
  }
  return NK_OK;
}

extern "C" int nk_batch_create(nk_ctx *ctx, const char *source, int n, int nparams, int flags, nk_batch **out) {
  NK_REQUIRE(ctx && out, "NULL argument");
  NK_HIP(hipSetDevice(ctx->device));
  std::vector<char> code;
  std::string log;
  NK_TRY(batch_compile(source, n, nparams, flags, &code, &log));
  nk_batch *B = new nk_batch();
  B->ctx = ctx;
  B->n = n;
  B->np = nparams;
  if (hipModuleLoadData(&B->mod, code.data()) != hipSuccess) { delete B; NK_FAIL(NK_E_HIP, "hipModuleLoadData failed"); }
  if (hipModuleGetFunction(&B->fn, B->mod, "nk_batch_newton") != hipSuccess) {
    hipModuleUnload(B->mod);
    delete B;
    NK_FAIL(NK_E_HIP, "kernel nk_batch_newton not found in the compiled module");
  }
  if (hipModuleGetFunction(&B->fn_tr, B->mod, "nk_batch_trust_region") != hipSuccess) B->fn_tr = nullptr;
  if (n > 8) {  // Newton for medium systems: the per-wavefront kernel
    std::vector<char> wcode;
    std::string wlog;
    if (batch_compile(source, n, nparams, flags, &wcode, &wlog, true) == NK_OK &&
        hipModuleLoadData(&B->mod_wave, wcode.data()) == hipSuccess) {
      if (hipModuleGetFunction(&B->fn_wave, B->mod_wave, "nk_batch_newton_wave") != hipSuccess) B->fn_wave = nullptr;
    }
    static const bool no_wave = getenv("NK_BATCH_NO_WAVE") != nullptr;  // A/B switch
    if (no_wave) B->fn_wave = nullptr;
  }
  *out = B;
  return NK_OK;
}

extern "C" int nk_batch_destroy(nk_batch *B) {
  if (!B) return NK_OK;
  hipFree(B->d_u0); hipFree(B->d_p); hipFree(B->d_u); hipFree(B->d_r); hipFree(B->d_rc); hipFree(B->d_it);
  if (B->mod) hipModuleUnload(B->mod);
  if (B->mod_wave) hipModuleUnload(B->mod_wave);
  delete B;
  return NK_OK;
}

// Solve all systems. u0: n doubles shared by every system (u0_per_system = 0, the tutorial's case) or nbatch×n;
// p: nbatch×nparams. Outputs (nbatch×n, nbatch×n, nbatch, nbatch); retcode/iters may be NULL.
// tr == nullptr: SimpleNewtonRaphson; else SimpleTrustRegion with tr = {η₁, η₂, η₃, t₁, t₂, max_shrink_times}.
static int batch_run(nk_batch *B, int64_t nbatch, const double *u0, int u0_per_system, const double *p, int memspace,
                     double abstol, int maxiters, const double *tr, double *u_out, double *resid_out, int32_t *retcode_out,
                     int32_t *iters_out) {
  NK_REQUIRE(B && u0 && u_out && resid_out, "NULL argument");
  NK_REQUIRE(nbatch >= 0, "negative batch size");
  NK_REQUIRE(B->np == 0 || p, "parameters are required (nparams = %d)", B->np);
  NK_REQUIRE(!tr || B->fn_tr, "the compiled module lacks the trust-region kernel");
  nk_ctx *ctx = B->ctx;
  NK_HIP(hipSetDevice(ctx->device));
  if (nbatch == 0) return NK_OK;
  if (!(abstol > 0.0)) abstol = pow(2.220446049250313e-16, 0.8);  // common_defaults.jl:39-48
  if (maxiters <= 0) maxiters = 1000;                             // raphson.jl:42 / trust_region.jl:60
  const int n = B->n, np = B->np;
  if (B->cap < nbatch) {
    hipFree(B->d_u0); hipFree(B->d_p); hipFree(B->d_u); hipFree(B->d_r); hipFree(B->d_rc); hipFree(B->d_it);
    B->d_u0 = B->d_p = B->d_u = B->d_r = nullptr;
    B->d_rc = B->d_it = nullptr;
    NK_TRY(nk_dev_alloc(&B->d_u0, (size_t)nbatch * n));
    NK_TRY(nk_dev_alloc(&B->d_p, (size_t)nbatch * (np > 0 ? np : 1)));
    NK_TRY(nk_dev_alloc(&B->d_u, (size_t)nbatch * n));
    NK_TRY(nk_dev_alloc(&B->d_r, (size_t)nbatch * n));
    NK_TRY(nk_dev_alloc(&B->d_rc, (size_t)nbatch));
    NK_TRY(nk_dev_alloc(&B->d_it, (size_t)nbatch));
    B->cap = nbatch;
  }
  const double *du0 = u0, *dp = p;
  double *du = u_out, *dr = resid_out;
  const size_t nu0 = (size_t)(u0_per_system ? nbatch : 1) * n;
  if (memspace != NK_DEVICE) {
    NK_HIP(hipMemcpyAsync(B->d_u0, u0, nu0 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (np > 0) NK_HIP(hipMemcpyAsync(B->d_p, p, (size_t)nbatch * np * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    du0 = B->d_u0;
    dp = B->d_p;
    du = B->d_u;
    dr = B->d_r;
  }
  long nb = (long)nbatch;
  int ups = u0_per_system ? 1 : 0;
  int *drc = B->d_rc, *dit = B->d_it;
  const unsigned grid = (unsigned)((nbatch + B->block - 1) / B->block);
  if (!tr && B->fn_wave) {
    void *args[] = {&nb, &du0, &ups, &dp, &abstol, &maxiters, &du, &dr, &drc, &dit};
    const unsigned wgrid = (unsigned)((nbatch + 3) / 4);  // 4 wavefronts = 4 systems per 256-thread workgroup
    if (hipModuleLaunchKernel(B->fn_wave, wgrid, 1, 1, 256, 1, 1, 0, ctx->stream, args, nullptr) != hipSuccess)
      NK_FAIL(NK_E_HIP, "launch of nk_batch_newton_wave failed");
  } else if (!tr) {
    void *args[] = {&nb, &du0, &ups, &dp, &abstol, &maxiters, &du, &dr, &drc, &dit};
    if (hipModuleLaunchKernel(B->fn, grid, 1, 1, B->block, 1, 1, 0, ctx->stream, args, nullptr) != hipSuccess)
      NK_FAIL(NK_E_HIP, "launch of nk_batch_newton failed");
  } else {
    double e1 = tr[0], e2 = tr[1], e3 = tr[2], t1 = tr[3], t2 = tr[4];
    int ms = (int)tr[5];
    void *args[] = {&nb, &du0, &ups, &dp, &abstol, &maxiters, &e1, &e2, &e3, &t1, &t2, &ms, &du, &dr, &drc, &dit};
    if (hipModuleLaunchKernel(B->fn_tr, grid, 1, 1, B->block, 1, 1, 0, ctx->stream, args, nullptr) != hipSuccess)
      NK_FAIL(NK_E_HIP, "launch of nk_batch_trust_region failed");
  }
  if (memspace != NK_DEVICE) {
    NK_HIP(hipMemcpyAsync(u_out, du, (size_t)nbatch * n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    NK_HIP(hipMemcpyAsync(resid_out, dr, (size_t)nbatch * n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  }
  // retcode / iteration outputs follow the memory space of the other arrays
  if (retcode_out)
    NK_HIP(hipMemcpyAsync(retcode_out, drc, (size_t)nbatch * sizeof(int32_t),
                          memspace == NK_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
  if (iters_out)
    NK_HIP(hipMemcpyAsync(iters_out, dit, (size_t)nbatch * sizeof(int32_t),
                          memspace == NK_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
  if (memspace != NK_DEVICE) NK_HIP(hipStreamSynchronize(ctx->stream));
  return NK_OK;
}

extern "C" int nk_batch_solve(nk_batch *B, int64_t nbatch, const double *u0, int u0_per_system, const double *p, int memspace,
                              double abstol, int maxiters, double *u_out, double *resid_out, int32_t *retcode_out,
                              int32_t *iters_out) {
  return batch_run(B, nbatch, u0, u0_per_system, p, memspace, abstol, maxiters, nullptr, u_out, resid_out, retcode_out, iters_out);
}

// SimpleTrustRegion (lib/SimpleNonlinearSolve/src/trust_region.jl); thresholds/factors ≤ 0 and max_shrink_times < 0 select
// the reference defaults η₁ = 1e-4, η₂ = 0.25, η₃ = 0.75, t₁ = 0.25, t₂ = 2, 32. Retcodes: Success, MaxIters,
// ShrinkThresholdExceeded.
extern "C" int nk_batch_solve_trust_region(nk_batch *B, int64_t nbatch, const double *u0, int u0_per_system, const double *p,
                                           int memspace, double abstol, int maxiters, double step_threshold,
                                           double shrink_threshold, double expand_threshold, double shrink_factor,
                                           double expand_factor, int max_shrink_times, double *u_out, double *resid_out,
                                           int32_t *retcode_out, int32_t *iters_out) {
  const double tr[6] = {step_threshold > 0 ? step_threshold : 1e-4, shrink_threshold > 0 ? shrink_threshold : 0.25,
                        expand_threshold > 0 ? expand_threshold : 0.75, shrink_factor > 0 ? shrink_factor : 0.25,
                        expand_factor > 0 ? expand_factor : 2.0, (double)(max_shrink_times >= 0 ? max_shrink_times : 32)};
  return batch_run(B, nbatch, u0, u0_per_system, p, memspace, abstol, maxiters, tr, u_out, resid_out, retcode_out, iters_out);
}
