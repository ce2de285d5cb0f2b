// Device-resident restarted GMRES(m) (seam 1: replaces solve!(cache.lincache) at
// lib/NonlinearSolveBase/ext/NonlinearSolveBaseLinearSolveExt.jl:26, i.e. LinearSolve.KrylovJL_GMRES →
// Krylov.gmres! [EXT]).
//
// Structure
//  * The Krylov basis (n × (m+1), column-major), the Hessenberg factor, the Givens rotations and the
//    least-squares right-hand side all live in device memory. A whole restart cycle is enqueued without host
//    synchronisation: the one-wave `k_givens` kernel decides convergence on the device and raises `ctl.done`,
//    which every later kernel of the cycle reads first and returns on. The host reads the 128-byte control
//    block once per cycle.
//  * Lagged normalisation: column j holds the UN-normalised vector ṽ_j and a device scalar s_j = 1/‖ṽ_j‖;
//    v_j = s_j ṽ_j is never materialised. The operator kernel writes s_k·A ṽ_k straight into column k+1, inner
//    products are scaled in the stage-2 reducer, and axpy coefficients are h_j s_j. This removes the separate
//    "v = w/‖w‖" pass (16 n bytes per Arnoldi step) and the w→V copy.
//  * CGS2 in three passes over V instead of four: multidot, then one fused kernel that applies the first
//    projection and accumulates the second projection's inner products while ṽ_j is still in registers, then
//    the second update (+‖·‖²). MGS (Krylov.jl's structure) and CGS+DGKS are the other two variants.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "nk_internal.h"

// ----------------------------------------------------------------------------- small device kernels
// progress word in coherent pinned host memory (nk_gmres_pub): the host polls it to bound its run-ahead and to stop
// enqueueing once the cycle has converged — no copy, no stream synchronisation
__device__ __forceinline__ void pub_progress(nk_gmres_pub *pub, uint64_t seq, int k, int done) {
  if (pub != nullptr)
    __hip_atomic_store(&pub->progress, (seq << 16) | ((uint64_t)k << 1) | (uint64_t)(done ? 1 : 0), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_gmres_begin(nk_gmres_ctl *ctl, const double *d_ss, double atol, double rtol, int fixed,
                              int first, double *g, double *s, int m, nk_gmres_pub *pub, uint64_t seq) {
  if (threadIdx.x != 0) return;
  nk_gmres_begin_body(ctl, *d_ss, atol, rtol, fixed, first, g, s, m, pub, seq);
}

// DGKS test after the first projection: re-orthogonalise iff ‖w'‖² < ½‖w‖². pad0 is the "skip pass 2" flag.
// h[k+1] = ‖w‖² (self slot of the first multidot), h2[k+1] = ‖w'‖² (from the fused pass).
__global__ void k_dgks(nk_gmres_ctl *ctl, const double *h, double *h2, double *d_ss, double inv_nranks) {
  if (threadIdx.x != 0) return;
  if (ctl->done) { ctl->pad0 = 1; return; }
  const int k = ctl->k;
  const double before = h[k + 1], after = h2[k + 1];
  const int need = (after < 0.5 * before) ? 1 : 0;
  ctl->need_reorth = need;
  ctl->pad0 = need ? 0 : 1;
  if (!need) {
    for (int i = 0; i <= k; ++i) h2[i] = 0.0;
    *d_ss = after * inv_nranks;  // the unconditional all-reduce inside the skipped pass restores `after`
  }
}

// DCGS2, start of step k ≥ 1: r = h2 of step k−1 (the pending re-orthogonalisation of column k), c = H̄_{k−1} r.
//   a_j = r_j s_j (j < k)                      : ṽ_k ← p − Σ a_j ṽ_j
//   b_j = s_k c_j s_j (j < k), b_k = s_k² c_k  : A v_k = s_k A p − s_k Σ_{j≤k} c_j v_j   (v_j = s_j ṽ_j)
__global__ __launch_bounds__(64) void k_dcgs2_coef(const nk_gmres_ctl *ctl, int k, const double *__restrict__ Hraw, int m,
                                                   const double *__restrict__ r, const double *__restrict__ s,
                                                   double *__restrict__ a, double *__restrict__ b) {
  if (ctl->done) return;
  __shared__ double sr[NK_MAX_NV];
  const int t = threadIdx.x;
  if (t < k) sr[t] = r[t];
  __syncthreads();
  const double sk = s[k];
  if (t <= k) {
    double c = 0.0;
    for (int j = (t > 0 ? t - 1 : 0); j < k; ++j) c += Hraw[(size_t)t * m + j] * sr[j];  // H is upper Hessenberg
    b[t] = sk * c * s[t];
    if (t < k) a[t] = sr[t] * s[t];
  }
}

// Single-rank DCGS2 tail of an Arnoldi step in ONE launch (1024 threads): (1) reduce pass B's per-block partials —
// 32 lanes per slot, fixed order — into h2[0..k] (scaled by s_j) and ‖w′‖²; (2) Hessenberg column h1 + h2, ‖w″‖ by
// Pythagoras, Givens update, convergence flags (as k_givens); (3) the pass-A coefficients of the NEXT step
// (a_j = h2_j s_j, b = s_{k+1}·(H̄ h2)·s). Replaces k_reduce_sum + k_givens + k_dcgs2_coef.
__global__ __launch_bounds__(1024) void k_givens_dcgs2(nk_gmres_ctl *ctl, const double *__restrict__ h1,
                                                       const double *__restrict__ partials, int nblk, double *R,
                                                       double *cs, double *sn, double *g, double *s, int m,
                                                       double *__restrict__ Hraw, double *__restrict__ a_out,
                                                       double *__restrict__ b_out, double *__restrict__ h2_out,
                                                       nk_gmres_pub *pub, uint64_t seq) {
  if (ctl->done) return;
  constexpr int LH = NK_MAX_NV + 1;
  __shared__ double sh[NK_MAX_NV + 2], sh2[NK_MAX_NV + 2], sc[NK_MAX_NV + 2], ss_[NK_MAX_NV + 2], ssc[NK_MAX_NV + 2];
  __shared__ double sH[(NK_MAX_NV / 2 + 2) * LH];  // H̄ rows 0..k+1, columns 0..k (restart ≤ 31 ⇒ 33 × 32 entries)
  __shared__ double s_next, s_gk, s_tol;
  const int k = ctl->k, t = threadIdx.x;
  const int nslots = k + 2;  // k+1 inner products and ‖w′‖²
  // every global read of the kernel is issued here, in one round trip: partials, h1, rotations, scales, g_k, H̄
  const int slot = t >> 5, lane = t & 31;
  double v = 0.0;
  if (slot < nslots) {
    const double *p = partials + (size_t)slot * nblk;
    if ((nblk & 1) == 0) {  // 16-byte loads (the slot base stays 16-byte aligned when nblk is even)
      const double2 *p2 = reinterpret_cast<const double2 *>(p);
      double v1 = 0.0;
      for (int i = lane; i < (nblk >> 1); i += 32) { const double2 q = p2[i]; v += q.x; v1 += q.y; }
      v += v1;
    } else {
      for (int i = lane; i < nblk; i += 32) v += p[i];
    }
  }
  if (t <= k) {
    sh[t] = h1[t];
    ssc[t] = s[t];
    if (t < k) { sc[t] = cs[t]; ss_[t] = sn[t]; }
  }
  if (t == 0) { s_gk = g[k]; s_tol = ctl->tol; }
  for (int e = t; e < (k + 2) * k; e += 1024) {  // previous columns j < k of H̄ (upper Hessenberg: zero below the subdiagonal)
    const int i = e / k, j = e - i * k;
    sH[i * LH + j] = (i <= j + 1) ? Hraw[(size_t)i * m + j] : 0.0;
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  __syncthreads();
  if (lane == 0 && slot < nslots) sh2[slot] = (slot <= k) ? v * ssc[slot] : v;
  __syncthreads();
  if (t <= k) {
    const double hv = sh[t] + sh2[t];
    sh[t] = hv;
    sH[t * LH + k] = hv;
    Hraw[(size_t)t * m + k] = hv;
    h2_out[t] = sh2[t];
  }
  __syncthreads();
  if (t == 0) {
    double s2 = 0.0;
    for (int i = 0; i <= k; ++i) s2 += sh2[i] * sh2[i];
    double ssq = sh2[k + 1] - s2;  // ‖w″‖² = ‖w′‖² − ‖h₂‖²
    if (ssq < 0.0) ssq = 0.0;
    const double hn = sqrt(ssq);
    sH[(k + 1) * LH + k] = hn;
    Hraw[(size_t)(k + 1) * m + k] = hn;
    double hk = sh[0];
    for (int i = 0; i < k; ++i) {
      const double a = hk, b = sh[i + 1];
      R[(size_t)i * m + k] = sc[i] * a + ss_[i] * b;
      hk = -ss_[i] * a + sc[i] * b;
    }
    const double d = hypot(hk, hn);
    double c, sgn;
    if (d == 0.0) { c = 1.0; sgn = 0.0; } else { c = hk / d; sgn = hn / d; }
    cs[k] = c;
    sn[k] = sgn;
    R[(size_t)k * m + k] = d;
    const double gk = s_gk;
    g[k + 1] = -sgn * gk;
    g[k] = c * gk;
    const double rn = fabs(sgn * gk);
    const double inv = (hn > 0.0) ? 1.0 / hn : 0.0;
    ctl->k = k + 1;
    ctl->rnorm = rn;
    ctl->hn = hn;
    ctl->inv_hn = inv;
    s[k + 1] = inv;
    s_next = inv;
    int dn = 0;
    if (!(rn == rn) || isinf(rn) || !(hn == hn)) { ctl->failed = 1; ctl->done = 1; dn = 1; }
    else if (s_tol >= 0.0 && rn <= s_tol) { ctl->converged = 1; ctl->done = 1; dn = 1; }
    else if (hn == 0.0) { ctl->converged = 1; ctl->done = 1; dn = 1; }
    pub_progress(pub, seq, k + 1, dn);
  }
  __syncthreads();
  // coefficients for the next step's pass A (pending column k+1, r = h2): c_i = Σ_{j ≤ k} H̄[i][j] r_j, i ≤ k+1
  if (t <= k + 1) {
    double c = 0.0;
    for (int j = (t > 0 ? t - 1 : 0); j <= k; ++j) c += sH[t * LH + j] * sh2[j];
    const double st = (t <= k) ? ssc[t] : s_next;
    b_out[t] = s_next * c * st;
    if (t <= k) a_out[t] = sh2[t] * st;
  }
}

// DCGS2 with one reduction per step — the scalar work between the dot sweep and the axpy sweep of step k (one wave does
// the serial part; k < NK_MAX_NV). red = [R_j = ṽ_j·u (k), a = u·u, G_j = ṽ_j·z (k), d = u·z], all-reduced, unscaled.
//   r = s∘R, g = s∘G, β² = a − rᵀr, s_k = 1/β;  H[0:k, k−1] = t_prev + r, H[k, k−1] = β  → Givens on column k−1 (one step
//   late), stopping test;  c = H̄_{k−1} r;  t = [(g − c_{0:k})/β ; (d − rᵀg − β c_k)/β²]  (first projection of A v_k)
//   axpy coefficients: a_j = r_j s_j;  b_j = (s_k c_j + t_j) s_j (j<k), b_k = (s_k c_k + t_k) s_k, b_{k+1} = s_k (scale of z)
// k = 0: column 0 is final, only t_0 = s_0² d. `last`: the flush after the cycle's last step — no z, no coefficients.
// Latency matters here, not bandwidth (one workgroup between two sweeps): every global operand is requested up front with
// clamped, unconditional addresses (a load behind a lane predicate costs a full round trip each — five of them in the first
// version of this kernel), the two inner products run as wave reductions, and H̄ r is split four ways.
// tprev is double-buffered by the parity of k (tprev_in = buffer k & 1, tprev_out = the other one): in the merged form
// below every workgroup reads the old values while workgroup 0 writes the new ones.
__global__ __launch_bounds__(256) void k_dcgs2r_tail(nk_gmres_ctl *ctl, int k, int last, const double *__restrict__ red,
                                                     double *s, double *__restrict__ Hraw, int m,
                                                     const double *__restrict__ tprev_in, double *__restrict__ tprev,
                                                     double *R, double *cs, double *sn, double *g,
                                                     double *__restrict__ a_out, double *__restrict__ b_out,
                                                     nk_gmres_pub *pub, uint64_t seq) {
  const int done = ctl->done;
  constexpr int NH = NK_MAX_NV + 2, LH = NK_MAX_NV + 1;  // any restart the GMRES object accepts (m < NK_MAX_NV)
  __shared__ double sr[NH], sg[NH], sc_[NH], sh[NH], scs[NH], ssn[NH], ssc[NH];
  __shared__ double sH[NH * LH];  // H̄ rows 0..k, columns 0..k−1, staged in one round trip
  __shared__ double spart[4 * 64];
  const int t = threadIdx.x;
  if (k == 0) {
    if (t == 0 && !done) {
      const double s0 = s[0], tl = s0 * s0 * red[1];
      tprev[0] = tl;
      b_out[0] = tl * s0;
      b_out[1] = s0;
    }
    return;
  }
  // ---- every global read, back to back
  const int tc = t < k ? t : k - 1;
  const int tr = t < k - 1 ? t : (k > 1 ? k - 2 : 0);
  const double st = s[tc];
  const double r_raw = (last == 2) ? 0.0 : red[tc];  // last = 2: the pending column closes the cycle un-re-orthogonalised
  const double g_raw = red[last ? tc : k + 1 + tc];
  const double tp = tprev_in[tc];
  const double csv = cs[tr], snv = sn[tr];
  const double gj = g[k - 1], ra = red[k], rd = red[last ? k : 2 * k + 1], tol = ctl->tol;
  const int km1 = k > 1 ? k - 1 : 1;
  const int tot = last ? 0 : (k + 1) * (k - 1);  // columns 0..k−2 (column k−1 is completed below)
  for (int e0 = 0; e0 < tot; e0 += 1024) {
    double hv[4];
    int at[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = e0 + t + 256 * q, ec = e < tot ? e : 0;
      const int i = ec / km1, j = ec - i * km1;
      hv[q] = Hraw[(size_t)i * m + j];
      at[q] = (e < tot) ? (i * LH + j) : -1;
      if (i > j + 1) hv[q] = 0.0;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (at[q] >= 0) sH[at[q]] = hv[q];
  }
  const double rt = (t < k) ? st * r_raw : 0.0;
  const double gt = (t < k && !last) ? st * g_raw : 0.0;
  const double ht = tp + rt;  // entry t of the Hessenberg column that completes now (t < k)
  double rr = rt * rt, rg = rt * gt;  // wave 0 holds every t < k ≤ 62
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    rr += __shfl_xor(rr, o, 64);
    rg += __shfl_xor(rg, o, 64);
  }
  if (done) return;  // uniform
  const int jc = k - 1;
  if (t < k) {
    ssc[t] = st;
    sr[t] = rt;
    sg[t] = gt;
    sh[t] = ht;
    if (t < k - 1) { scs[t] = csv; ssn[t] = snv; }
    Hraw[(size_t)t * m + jc] = ht;
    sH[t * LH + jc] = ht;
  }
  double b2 = ra - rr;  // ‖u − V r‖² by Pythagoras
  if (b2 < 0.0) b2 = 0.0;
  const double beta = sqrt(b2), sk = (beta > 0.0) ? 1.0 / beta : 0.0;
  if (t == 0) {
    s[k] = sk;
    Hraw[(size_t)k * m + jc] = beta;
    sH[k * LH + jc] = beta;
  }
  __syncthreads();
  if (t == 0) {
    double hk = sh[0];
#pragma unroll 4
    for (int i = 0; i < jc; ++i) {
      const double a = hk, b = sh[i + 1];
      R[(size_t)i * m + jc] = scs[i] * a + ssn[i] * b;
      hk = -ssn[i] * a + scs[i] * b;
    }
    const double d = hypot(hk, beta);
    double c, sgn;
    if (d == 0.0) { c = 1.0; sgn = 0.0; } else { c = hk / d; sgn = beta / d; }
    cs[jc] = c;
    sn[jc] = sgn;
    R[(size_t)jc * m + jc] = d;
    g[jc + 1] = -sgn * gj;
    g[jc] = c * gj;
    const double rn = fabs(sgn * gj);
    ctl->k = k;
    ctl->rnorm = rn;
    ctl->hn = beta;
    ctl->inv_hn = sk;
    int dn = 0;
    if (!(rn == rn) || isinf(rn) || !(beta == beta)) { ctl->failed = 1; ctl->done = 1; dn = 1; }
    else if (tol >= 0.0 && rn <= tol) { ctl->converged = 1; ctl->done = 1; dn = 1; }
    else if (beta == 0.0) { ctl->converged = 1; ctl->done = 1; dn = 1; }
    pub_progress(pub, seq, k, dn);
  }
  if (last) return;
  {  // c = H̄_{k−1} r: rows 0..k, columns 0..k−1; the entries below the sub-diagonal are stored zeros
    const int row = t & 63, part = t >> 6;
    double c = 0.0;
    if (row <= k) {
#pragma unroll 4
      for (int j = part; j < k; j += 4) c += sH[row * LH + j] * sr[j];
    }
    spart[part * 64 + row] = c;
  }
  __syncthreads();
  if (t <= k) sc_[t] = (spart[t] + spart[64 + t]) + (spart[128 + t] + spart[192 + t]);
  __syncthreads();
  if (t < k) {
    const double tt = (sg[t] - sc_[t]) * sk;
    tprev[t] = tt;
    a_out[t] = sr[t] * ssc[t];
    b_out[t] = (sk * sc_[t] + tt) * ssc[t];
  }
  if (t == 0) {
    const double tl = (rd - rg - beta * sc_[k]) * sk * sk;
    tprev[k] = tl;
    b_out[k] = (sk * sc_[k] + tl) * sk;
    b_out[k + 1] = sk;
  }
}


// The same scalar work as k_dcgs2r_tail (step k, not the flush) done in the PROLOGUE of the axpy sweep: every workgroup
// derives the axpy coefficients from the reduced inner products itself (≈ 8 KB of L2-resident operands, one round trip,
// issued behind the loads of its own row tile), so the one-workgroup launch between the reduction and the sweep —
// ≈ 5 µs of a 69 µs Arnoldi step at the launch-latency floor — disappears. Workgroup 0 also carries the state forward
// (s_k, the Hessenberg column, tprev, Givens rotation, residual estimate, stopping test, progress word); the other
// workgroups only read values that workgroup 0 does not write in this launch (tprev is double-buffered; `done` may flip
// while the launch is in flight, which only decides whether columns k, k+1 — unused once the cycle is done — get updated).
constexpr int DRT = 8;  // rows per thread (the tile shape of nk_blas.hip's sweeps)
template <int MAXM>
__global__ __launch_bounds__(NK_BLOCK) void k_dcgs2r_axpy_tail(int64_t n, int k, double *__restrict__ V, int64_t ldv,
                                                               nk_gmres_ctl *ctl, const double *__restrict__ red, double *s,
                                                               double *__restrict__ Hraw, int m,
                                                               const double *__restrict__ tprev_in,
                                                               double *__restrict__ tprev_out, double *R, double *cs,
                                                               double *sn, double *g, double *__restrict__ ss_partials,
                                                               nk_gmres_pub *pub, uint64_t seq, int desc) {
  constexpr int NH = MAXM + 2, LH = MAXM + 1;
  __shared__ double sr[NH], sg[NH], sc_[NH], sh[NH], scs[NH], ssn[NH], ssc[NH], ca[NH + 1], cb[NH + 1];
  __shared__ double sH[NH * LH];
  __shared__ double spart[4 * 64];
  __shared__ double s_bk, s_sz, s_beta, s_gj, s_tol, s_w[4];
  __shared__ int s_done;
  const int t = threadIdx.x;
  const bool writer = blockIdx.x == 0;
  if (t == 0) s_done = ctl->done;
  // ---- this workgroup's row tile of the pending column p and of z = A p: independent of the coefficients, requested first
  double *__restrict__ pk = V + (size_t)k * ldv;
  double *__restrict__ zk = V + (size_t)(k + 1) * ldv;
  const unsigned nn = (unsigned)n, base = blockIdx.x * (NK_BLOCK * DRT) + threadIdx.x;
  unsigned idx[DRT];
  double pv[DRT], zv[DRT];
#pragma unroll
  for (int i = 0; i < DRT; ++i) {
    const unsigned r = base + NK_BLOCK * i;
    idx[i] = r < nn ? r : 0;
  }
#pragma unroll
  for (int i = 0; i < DRT; ++i) {
    pv[i] = pk[idx[i]];
    zv[i] = zk[idx[i]];
  }
  if (k == 0) {  // column 0 is final: only t_0 = s_0² (v_0·z)
    const double s0 = s[0], tl = s0 * s0 * red[1];
    if (t == 0) { s_bk = tl * s0; s_sz = s0; ca[0] = 0.0; cb[0] = 0.0; }
    __syncthreads();
    if (s_done) return;
    if (writer && t == 0) tprev_out[0] = tl;
  } else {
    // ---- every global operand of the scalar work, back to back (clamped, unconditional addresses)
    const int tc = t < k ? t : k - 1;
    const int tr = t < k - 1 ? t : (k > 1 ? k - 2 : 0);
    const double st = s[tc], r_raw = red[tc], g_raw = red[k + 1 + tc], tp = tprev_in[tc];
    const double ra = red[k], rd = red[2 * k + 1];
    const double csv = cs[tr], snv = sn[tr];
    if (t == 0) { s_gj = g[k - 1]; s_tol = ctl->tol; }
    const int km1 = k > 1 ? k - 1 : 1;
    const int tot = (k + 1) * (k - 1);  // rows 0..k, columns 0..k−2 of H̄ (column k−1 is completed below)
    for (int e0 = 0; e0 < tot; e0 += 4 * NK_BLOCK) {
      double hv[4];
      int at[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int e = e0 + t + NK_BLOCK * q, ec = e < tot ? e : 0;
        const int i = ec / km1, j = ec - i * km1;
        hv[q] = Hraw[(size_t)i * m + j];
        at[q] = (e < tot) ? (i * LH + j) : -1;
        if (i > j + 1) hv[q] = 0.0;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (at[q] >= 0) sH[at[q]] = hv[q];
    }
    const double rt = (t < k) ? st * r_raw : 0.0;
    const double gt = (t < k) ? st * g_raw : 0.0;
    const double ht = tp + rt;  // entry t of the Hessenberg column that completes now (t < k)
    double rr = rt * rt, rg = rt * gt;  // wave 0 holds every t < k ≤ 62
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      rr += __shfl_xor(rr, o, 64);
      rg += __shfl_xor(rg, o, 64);
    }
    const int jc = k - 1;
    if (t < k) {
      ssc[t] = st;
      sr[t] = rt;
      sg[t] = gt;
      sh[t] = ht;
      sH[t * LH + jc] = ht;
      if (t < k - 1) { scs[t] = csv; ssn[t] = snv; }
    }
    double b2 = ra - rr;  // ‖u − V r‖² by Pythagoras (valid in wave 0)
    if (b2 < 0.0) b2 = 0.0;
    const double beta = sqrt(b2), sk = (beta > 0.0) ? 1.0 / beta : 0.0;
    if (t == 0) {
      sH[k * LH + jc] = beta;
      s_beta = beta;
      s_sz = sk;
    }
    __syncthreads();
    if (s_done) return;  // uniform
    if (writer) {  // the state other kernels (and the next step) read
      if (t < k) Hraw[(size_t)t * m + jc] = ht;
      if (t == 0) { s[k] = sk; Hraw[(size_t)k * m + jc] = beta; }
    }
    {  // c = H̄_{k−1} r: rows 0..k, columns 0..k−1; the entries below the sub-diagonal are stored zeros
      const int row = t & 63, part = t >> 6;
      double c = 0.0;
      if (row <= k) {
#pragma unroll 4
        for (int j = part; j < k; j += 4) c += sH[row * LH + j] * sr[j];
      }
      spart[part * 64 + row] = c;
    }
    __syncthreads();
    if (t <= k) sc_[t] = (spart[t] + spart[64 + t]) + (spart[128 + t] + spart[192 + t]);
    __syncthreads();
    const double skk = s_sz, bet = s_beta;
    if (t < k) {
      const double tt = (sg[t] - sc_[t]) * skk;
      if (writer) tprev_out[t] = tt;
      ca[t] = sr[t] * ssc[t];
      cb[t] = (skk * sc_[t] + tt) * ssc[t];
    }
    if (t == 0) {
      const double tl = (rd - rg - bet * sc_[k]) * skk * skk;
      if (writer) tprev_out[k] = tl;
      s_bk = (skk * sc_[k] + tl) * skk;
      ca[k] = 0.0;  // one zero entry past the end pads an odd column count
      cb[k] = 0.0;
    }
    __syncthreads();
  }
  // ---- the sweep: p ← p − Σ a_j ṽ_j ; z ← s_k z − Σ b_j ṽ_j − b_k p   (as k_dcgs2r_axpy)
  const double sz = s_sz, bk = s_bk;
#pragma unroll
  for (int i = 0; i < DRT; ++i) zv[i] *= sz;
  // column pairs (j, j+1), j even. `desc`: from the last pair down to the first — the dot sweep that ran just before
  // this launch read the columns in ascending order, so the highest columns are the ones most recently brought into the
  // Infinity Cache (256 MiB against a basis of up to 260 MB + the 74 MB matrix at 1024²: a cyclic ascending/ascending
  // order evicts every line before it is read again, the zig-zag re-reads the freshest ≈ 150 MB).
  const int npair = (k + 1) >> 1;
  for (int q = 0; q < npair; ++q) {
    const int j = 2 * (desc ? npair - 1 - q : q);
    const int j1 = min(j + 1, k - 1);
    const double *__restrict__ c0 = V + (size_t)j * ldv;
    const double *__restrict__ c1 = V + (size_t)j1 * ldv;
    const double a0 = ca[j], a1 = ca[j + 1], b0 = cb[j], b1 = cb[j + 1];
    double v0[DRT], v1[DRT];
#pragma unroll
    for (int i = 0; i < DRT; ++i) { v0[i] = c0[idx[i]]; v1[i] = c1[idx[i]]; }
#pragma unroll
    for (int i = 0; i < DRT; ++i) {
      pv[i] -= a0 * v0[i];
      pv[i] -= a1 * v1[i];
      zv[i] -= b0 * v0[i];
      zv[i] -= b1 * v1[i];
    }
  }
  double ss = 0.0;
#pragma unroll
  for (int i = 0; i < DRT; ++i) {
    const unsigned r = base + NK_BLOCK * i;
    if (r < nn) {
      if (k > 0) pk[r] = pv[i];
      const double zn = zv[i] - bk * pv[i];
      zk[r] = zn;
      ss += zn * zn;
    }
  }
  if (ss_partials != nullptr) {  // uniform: the cycle's last step also needs ‖z_new‖²
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = ss;
    __syncthreads();
    if (threadIdx.x == 0) ss_partials[blockIdx.x] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
  }
  // ---- workgroup 0, after its share of the sweep is on its way: Givens rotation of the column that closed, residual
  // estimate, stopping test (nothing in this launch waits for it)
  if (writer && t == 0 && k > 0) {
    const int jc = k - 1;
    const double beta = s_beta, gj = s_gj, tol = s_tol;
    double hk = sh[0];
#pragma unroll 4
    for (int i = 0; i < jc; ++i) {
      const double a = hk, b = sh[i + 1];
      R[(size_t)i * m + jc] = scs[i] * a + ssn[i] * b;
      hk = -ssn[i] * a + scs[i] * b;
    }
    const double d = hypot(hk, beta);
    double c, sgn;
    if (d == 0.0) { c = 1.0; sgn = 0.0; } else { c = hk / d; sgn = beta / d; }
    cs[jc] = c;
    sn[jc] = sgn;
    R[(size_t)jc * m + jc] = d;
    g[jc + 1] = -sgn * gj;
    g[jc] = c * gj;
    const double rn = fabs(sgn * gj);
    ctl->k = k;
    ctl->rnorm = rn;
    ctl->hn = beta;
    ctl->inv_hn = s_sz;
    int dn = 0;
    if (!(rn == rn) || isinf(rn) || !(beta == beta)) { ctl->failed = 1; ctl->done = 1; dn = 1; }
    else if (tol >= 0.0 && rn <= tol) { ctl->converged = 1; ctl->done = 1; dn = 1; }
    else if (beta == 0.0) { ctl->converged = 1; ctl->done = 1; dn = 1; }
    pub_progress(pub, seq, k, dn);
  }
}

// new Hessenberg column → apply old rotations, create the new one, update g and the residual estimate
// ss_partials != nullptr (single rank): ‖w‖² arrives as nblk per-block partials and is reduced here, saving a launch
__global__ __launch_bounds__(64) void k_givens(nk_gmres_ctl *ctl, double *h, const double *h2, const double *d_ss,
                                               double *R, double *cs, double *sn, double *g, double *s, int m,
                                               const double *__restrict__ ss_partials, int nblk, int pythag,
                                               double *__restrict__ Hraw, nk_gmres_pub *pub, uint64_t seq) {
  if (ctl->done) return;
  __shared__ double sh[NK_MAX_NV + 2], sc[NK_MAX_NV + 2], ss_[NK_MAX_NV + 2];
  const int k = ctl->k, t = threadIdx.x;
  double ssq_red = 0.0;
  if (ss_partials != nullptr) {
    for (int i = t; i < nblk; i += 64) ssq_red += ss_partials[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ssq_red += __shfl_xor(ssq_red, o, 64);
  }
  if (t <= k) {
    double v = h[t];
    if (h2) v += h2[t];
    sh[t] = v;
    if (t < k) { sc[t] = cs[t]; ss_[t] = sn[t]; }
  }
  __syncthreads();
  if (t != 0) return;
  double ssq = (ss_partials != nullptr) ? ssq_red : *d_ss;
  if (pythag) {
    // multi-GPU CGS2: ‖w₂‖² = ‖w₁‖² − ‖h₂‖² (w₂ = w₁ − V h₂ with VᵀV = I and h₂ = Vᵀw₁). h₂ is the tiny
    // re-orthogonalisation correction, so there is no cancellation; this saves the third all-reduce of the step.
    double s2 = 0.0;
    for (int i = 0; i <= k; ++i) s2 += h2[i] * h2[i];
    ssq = h2[k + 1] - s2;
  }
  if (ssq < 0.0) ssq = 0.0;
  const double hn = sqrt(ssq);
  if (Hraw) {  // DCGS2 keeps the un-rotated Hessenberg column ((m+1) × m, row-major) for the Arnoldi-relation update
    for (int i = 0; i <= k; ++i) Hraw[(size_t)i * m + k] = sh[i];
    Hraw[(size_t)(k + 1) * m + k] = hn;
  }
  double hk = sh[0];
  for (int i = 0; i < k; ++i) {  // rotation i acts on (h[i], h[i+1])
    const double a = hk, b = sh[i + 1];
    R[(size_t)i * m + k] = sc[i] * a + ss_[i] * b;
    hk = -ss_[i] * a + sc[i] * b;
  }
  const double d = hypot(hk, hn);
  double c, sgn;
  if (d == 0.0) { c = 1.0; sgn = 0.0; } else { c = hk / d; sgn = hn / d; }
  cs[k] = c;
  sn[k] = sgn;
  R[(size_t)k * m + k] = d;
  const double gk = g[k];
  g[k + 1] = -sgn * gk;
  g[k] = c * gk;
  const double rn = fabs(g[k + 1]);
  ctl->k = k + 1;
  ctl->rnorm = rn;
  ctl->hn = hn;
  ctl->inv_hn = (hn > 0.0) ? 1.0 / hn : 0.0;
  s[k + 1] = ctl->inv_hn;
  int dn = 0;
  if (!(rn == rn) || isinf(rn) || !(hn == hn)) { ctl->failed = 1; ctl->done = 1; dn = 1; }
  else if (ctl->tol >= 0.0 && rn <= ctl->tol) { ctl->converged = 1; ctl->done = 1; dn = 1; }
  else if (hn == 0.0) { ctl->converged = 1; ctl->done = 1; dn = 1; }  // happy breakdown
  pub_progress(pub, seq, k + 1, dn);
}

// y = R(0:k,0:k)^{-1} g(0:k), k = ctl->k. R is staged in LDS with batched loads; wave 0 runs the column-oriented
// recurrence (lane t owns g_t: after y_i is known every lane t < i takes R_ti y_i off its entry) — k dependent steps instead
// of k²/2 on one lane.
// `fx` (s-step form, sb > 0): the cycle's last block was left at its first pass (nk_sstep.hip: no third sweep) — its columns
// Q = (Q₁ − V C₂) R₂⁻¹ enter x += [V Q] y through the columns as they are: [V Q] y = V (y_k − C₂ b) + Q₁ b, b = R₂⁻¹ y_Q. The
// same wavefront turns y into those coefficients (an sb × sb triangular solve and a k0 × sb product; a launch of its own took
// 9 µs on one lane).
__global__ __launch_bounds__(256) void k_backsolve(const nk_gmres_ctl *ctl, const double *__restrict__ R,
                                                   const double *__restrict__ g, double *__restrict__ y, int m,
                                                   nk_gmres_pub *pub, uint64_t seq, const uint64_t *peer_err, const nk_ss_fix fx) {
  constexpr int LK = NK_MAX_NV + 1;  // odd stride: the column reads below are conflict-free
  __shared__ double sR[NK_MAX_NV * LK];
  __shared__ double sC2[NK_SS_NFIX][NK_MAX_NV * 16], sR2[NK_SS_NFIX][16 * 16];
  const int k = ctl->k, failed = ctl->failed;
  const int t = threadIdx.x;
  if (pub != nullptr && t == 0) {  // what the host needs of the control block, then the release of the sequence word
    pub->k = k;
    pub->converged = ctl->converged;
    pub->failed = failed;
    pub->rnorm0 = ctl->rnorm0;
    pub->rnorm = ctl->rnorm;
    // time-outs of the peer-mapped collectives so far (a rank stalled for longer than the bound: its contribution to an
    // all-reduce or a halo was missing) — the host fails the solve with NK_E_COMM instead of returning numbers built on them
    pub->pad = peer_err ? (int)(__hip_atomic_load(peer_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0x7fffffff) : 0;
    __threadfence_system();
    __hip_atomic_store(&pub->end_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  const int tot = k * k, kd = k > 0 ? k : 1;
  double gv = g[t < k ? t : 0];
  for (int e0 = 0; e0 < tot; e0 += 1024) {
    double rv[4];
    int at[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = e0 + t + 256 * q, ec = e < tot ? e : 0;
      const int i = ec / kd, j = ec - i * kd;
      rv[q] = R[(size_t)i * m + j];
      at[q] = (e < tot) ? i * LK + j : -1;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (at[q] >= 0) sR[at[q]] = rv[q];
  }
  for (int bq = 0; bq < fx.n; ++bq) {   // (a cycle that ended before a block: nothing of it in y)
    if (failed || k <= fx.k0[bq]) continue;
    for (int e = t; e < fx.k0[bq] * fx.sb[bq]; e += 256) sC2[bq][e] = fx.C2[bq][e];
    if (t < fx.sb[bq] * fx.sb[bq]) sR2[bq][t] = fx.R2[bq][t];
  }
  __syncthreads();
  // The arithmetic of nk_sstep.hip's ss_backsolve, operation for operation (the s-step cycle's last scalar launch back-substitutes
  // itself where it can; the transports are compared bit for bit): reciprocal diagonals, every triangular factor scaled by them
  // with its diagonal and lower part zeroed, then chains of multiply-adds.
  __shared__ double rdv[NK_MAX_NV + 16 * NK_SS_NFIX];
  if (!failed) {
    if (t < k) rdv[t] = 1.0 / sR[t * LK + t];
    for (int bq = 0; bq < fx.n; ++bq) {
      const int u = t - 64 - 16 * bq;
      if (k > fx.k0[bq] && u >= 0 && u < fx.sb[bq]) rdv[NK_MAX_NV + 16 * bq + u] = 1.0 / sR2[bq][u * fx.sb[bq] + u];
    }
  }
  __syncthreads();
  if (!failed) {
    for (int r = t >> 5; r < k; r += 8) {
      const double rd = rdv[r];
      for (int c = t & 31; c < k; c += 32) {
        const double v = sR[r * LK + c];
        sR[r * LK + c] = c > r ? v * rd : 0.0;
      }
    }
    for (int bq = 0; bq < fx.n; ++bq) {
      const int fsb = fx.sb[bq], a = t >> 4, c = t & 15;
      if (k > fx.k0[bq] && a < fsb && c < fsb) {
        const double v = sR2[bq][a * fsb + c];
        sR2[bq][a * fsb + c] = c > a ? v * rdv[NK_MAX_NV + 16 * bq + a] : 0.0;
      }
    }
  }
  __syncthreads();
  if (t >= 64) return;
  if (t >= k) gv = 0.0;
  if (!failed) {
    if (t < k) gv *= rdv[t];
    const double *row = sR + (t < k ? t : (k > 0 ? k - 1 : 0)) * LK;   // (lanes ≥ k: the last row — all zero now)
    for (int i = k - 1; i >= 1; --i) gv = __builtin_fma(-row[i], __shfl(gv, i, 64), gv);
    // the blocks left at their first pass, last first: coefficients on [V_true Q] → on V_true and the block's stored columns
    for (int bq = fx.n - 1; bq >= 0; --bq) {
      const int fk0 = fx.k0[bq], fsb = fx.sb[bq];
      if (k <= fk0) continue;
      const int cc = t - fk0;
      const bool inb = cc >= 0 && cc < fsb;
      const double *r2row = sR2[bq] + (inb ? cc : fsb - 1) * fsb, *c2row = sC2[bq] + (t < fk0 ? t : 0) * fsb;
      if (inb) gv *= rdv[NK_MAX_NV + 16 * bq + cc];
      for (int c = fsb - 1; c >= 1; --c) gv = __builtin_fma(-r2row[c], __shfl(gv, fk0 + c, 64), gv);   // b = R₂⁻¹ y_Q
      double acc = 0.0;
      for (int c = 0; c < fsb; ++c) acc = __builtin_fma(c2row[c], __shfl(gv, fk0 + c, 64), acc);
      if (t < fk0) gv -= acc;
    }
  } else {
    gv = 0.0;
  }
  if (t < m) y[t] = gv;
}

// ----------------------------------------------------------------------------- create / destroy / operators
extern "C" int nk_gmres_create(nk_ctx *ctx, int64_t n_local, int restart_m, int ortho, nk_gmres **out) {
  NK_REQUIRE(ctx && out, "NULL argument");
  NK_REQUIRE(n_local >= 0, "negative size");
  if (restart_m <= 0) restart_m = 30;
  NK_REQUIRE(restart_m < NK_MAX_NV, "restart m=%d too large (max %d)", restart_m, NK_MAX_NV - 1);
  NK_REQUIRE(ortho == NK_ORTHO_MGS || ortho == NK_ORTHO_CGS2 || ortho == NK_ORTHO_CGS || ortho == NK_ORTHO_DCGS2 ||
                 ortho == NK_ORTHO_DCGS2_1R || ortho == NK_ORTHO_SSTEP,
             "bad ortho %d", ortho);
  NK_HIP(hipSetDevice(ctx->device));
  nk_gmres *G = new nk_gmres();
  auto guard = nk_make_guard(G, [](nk_gmres *g) { nk_gmres_destroy(g); });
  G->ctx = ctx;
  G->n = n_local;
  G->m = restart_m;
  G->ortho = ortho;
  G->ldv = (n_local + 31) & ~(int64_t)31;  // 256-byte aligned columns
  if (G->ldv == 0) G->ldv = 32;
  {  // de-phase the columns: a power-of-two column stride puts all nv concurrent streams on the same HBM channel
    const char *e = getenv("NK_LDV_PAD");
    const int64_t pad = e ? atoll(e) : NK_LDV_PAD_DEFAULT;
    if (pad > 0 && n_local >= 4096) G->ldv += (pad + 31) & ~(int64_t)31;
  }
  const int m = restart_m;
  NK_TRY(nk_dev_alloc(&G->V, (size_t)G->ldv * (m + 1)));
  NK_TRY(nk_dev_alloc(&G->w, (size_t)G->ldv));
  NK_TRY(nk_dev_alloc(&G->r, (size_t)G->ldv));
  NK_TRY(nk_dev_alloc(&G->d_h, (size_t)NK_MAX_NV + 2));
  NK_TRY(nk_dev_alloc(&G->d_h2, (size_t)NK_MAX_NV + 2));
  NK_TRY(nk_dev_alloc(&G->d_Hraw, (size_t)(NK_MAX_NV + 1) * NK_MAX_NV));
  NK_TRY(nk_dev_alloc(&G->d_ca, (size_t)NK_MAX_NV + 2));
  NK_TRY(nk_dev_alloc(&G->d_tprev, (size_t)2 * (NK_MAX_NV + 2)));  // double-buffered by the parity of the step
  NK_TRY(nk_dev_alloc(&G->d_red, (size_t)2 * NK_MAX_NV + 4));
  NK_TRY(nk_dev_alloc(&G->d_cb, (size_t)NK_MAX_NV + 2));
  NK_TRY(nk_dev_alloc(&G->d_s, (size_t)NK_MAX_NV + 2));
  NK_TRY(nk_dev_alloc(&G->d_R, (size_t)m * m));
  NK_TRY(nk_dev_alloc(&G->d_cs, (size_t)m + 1));
  NK_TRY(nk_dev_alloc(&G->d_sn, (size_t)m + 1));
  NK_TRY(nk_dev_alloc(&G->d_g, (size_t)m + 2));
  NK_TRY(nk_dev_alloc(&G->d_y, (size_t)m + 1));
  NK_TRY(nk_dev_alloc(&G->d_ss, (size_t)4));
  NK_TRY(nk_dev_alloc(&G->d_ctl, (size_t)1));
  NK_HIP(hipHostMalloc((void **)&G->h_ctl, sizeof(nk_gmres_ctl), hipHostMallocDefault));
  NK_HIP(hipHostMalloc((void **)&G->h_pub, sizeof(nk_gmres_pub), hipHostMallocCoherent | hipHostMallocMapped));
  memset(G->h_pub, 0, sizeof(nk_gmres_pub));
  NK_HIP(hipHostGetDevicePointer((void **)&G->h_pub_dev, G->h_pub, 0));
  {
    const char *e = getenv("NK_GMRES_RUN_AHEAD");
    G->run_ahead = e ? atoi(e) : 4;
  }
  NK_HIP(nk_memset(ctx, G->d_ctl, 0, sizeof(nk_gmres_ctl)));
  NK_HIP(nk_memset(ctx, G->V, 0, (size_t)G->ldv * (m + 1) * sizeof(double)));
  NK_HIP(nk_memset(ctx, G->d_s, 0, (NK_MAX_NV + 2) * sizeof(double)));
  *out = guard.release();
  return NK_OK;
}
extern "C" int nk_gmres_destroy(nk_gmres *G) {
  if (!G) return NK_OK;
  hipFree(G->V); hipFree(G->w); hipFree(G->z); hipFree(G->r); hipFree(G->lz);
  hipFree(G->d_Hraw); hipFree(G->d_ca); hipFree(G->d_cb); hipFree(G->d_tprev); hipFree(G->d_red);
  nk_mg_destroy(G->mg);
  nk_ss_destroy(G->ss);
  hipFree(G->x0_keep);
  if (G->gexec) hipGraphExecDestroy(G->gexec);
  if (G->cap_stream) hipStreamDestroy(G->cap_stream);
  hipFree(G->d_h); hipFree(G->d_h2); hipFree(G->d_s); hipFree(G->d_R); hipFree(G->d_cs); hipFree(G->d_sn);
  hipFree(G->d_g); hipFree(G->d_y); hipFree(G->d_ss); hipFree(G->d_ctl);
  hipFree(G->d_u_own); hipFree(G->d_b); hipFree(G->d_x);
  hipFree(G->cr); hipFree(G->cd); hipFree(G->ct); hipFree(G->cd2); hipFree(G->nrm_tmp);
  hipHostFree(G->h_ctl);
  hipHostFree(G->h_pub);
  if (G->h_stage) hipHostFree(G->h_stage);
  delete G;
  return NK_OK;
}
extern "C" int nk_gmres_set_block_size(nk_gmres *G, int s) {
  if (G) G->ahead.valid = false;   // (a cycle begin run ahead of the next solve belongs to the old setting)
  NK_REQUIRE(G, "NULL argument");
  NK_REQUIRE(s >= 0 && s <= 16, "s-step block size %d outside 0..16 (0 = automatic)", s);
  G->ss_s = s;
  return NK_OK;
}
extern "C" int nk_gmres_set_operator_csr(nk_gmres *G, nk_csr *A) {
  if (G) G->ahead.valid = false;   // (a cycle begin run ahead of the next solve belongs to the old setting)
  NK_REQUIRE(G && A, "NULL argument");
  NK_REQUIRE(A->nrows == G->n, "operator size %lld != GMRES size %lld", (long long)A->nrows, (long long)G->n);
  G->op_kind = 1;
  G->A = A;
  return NK_OK;
}
extern "C" int nk_gmres_set_operator_jvp(nk_gmres *G, nk_problem *P, const double *u, int memspace) {
  if (G) G->ahead.valid = false;   // (a cycle begin run ahead of the next solve belongs to the old setting)
  NK_REQUIRE(G && P && u, "NULL argument");
  NK_REQUIRE(P->n_local == G->n, "problem size %lld != GMRES size %lld", (long long)P->n_local, (long long)G->n);
  NK_HIP(hipSetDevice(G->ctx->device));
  G->op_kind = 2;
  G->P = P;
  if (memspace == NK_DEVICE) {
    G->d_u = u;
  } else {
    if (!G->d_u_own) NK_TRY(nk_dev_alloc(&G->d_u_own, (size_t)G->n + 1));
    NK_HIP(hipMemcpyAsync(G->d_u_own, u, G->n * sizeof(double), hipMemcpyHostToDevice, G->ctx->stream));
    G->d_u = G->d_u_own;
  }
  return nk_problem_jvp_prepare(P, G->d_u);
}
extern "C" int nk_gmres_set_operator_fn(nk_gmres *G, nk_matvec_fn fn, void *user) {
  if (G) G->ahead.valid = false;   // (a cycle begin run ahead of the next solve belongs to the old setting)
  NK_REQUIRE(G && fn, "NULL argument");
  G->op_kind = 3;
  G->fn = fn;
  G->fn_user = user;
  G->fn_host = false;
  return NK_OK;
}
// callbacks on host memory (a host language's `mul!` on plain arrays): x down, the call, y up — all on the stream's order
static int host_callback(nk_gmres *G, nk_matvec_fn fn, void *user, const double *d_x, double *d_y, const char *what) {
  nk_ctx *ctx = G->ctx;
  const size_t bytes = (size_t)G->n * sizeof(double);
  if (!G->h_stage) NK_HIP(hipHostMalloc((void **)&G->h_stage, 2 * bytes + 16, hipHostMallocDefault));
  NK_HIP(hipMemcpyAsync(G->h_stage, d_x, bytes, hipMemcpyDeviceToHost, ctx->stream));
  NK_HIP(hipStreamSynchronize(ctx->stream));
  if (fn(user, G->h_stage, G->h_stage + G->n, nullptr) != 0) NK_FAIL(NK_E_CALLBACK, "%s callback failed", what);
  NK_HIP(hipMemcpyAsync(d_y, G->h_stage + G->n, bytes, hipMemcpyHostToDevice, ctx->stream));
  NK_HIP(hipStreamSynchronize(ctx->stream));  // the staging buffer is reused by the next call
  return NK_OK;
}
extern "C" int nk_gmres_set_operator_fn_host(nk_gmres *G, nk_matvec_fn fn, void *user) {
  if (G) G->ahead.valid = false;   // (a cycle begin run ahead of the next solve belongs to the old setting)
  NK_TRY(nk_gmres_set_operator_fn(G, fn, user));
  G->fn_host = true;
  return NK_OK;
}
extern "C" int nk_gmres_set_right_preconditioner_host(nk_gmres *G, nk_matvec_fn fn, void *user) {
  if (G) G->ahead.valid = false;   // (a cycle begin run ahead of the next solve belongs to the old setting)
  NK_TRY(nk_gmres_set_right_preconditioner(G, fn, user));
  G->prec_host = fn != nullptr;
  return NK_OK;
}
extern "C" int nk_gmres_set_right_preconditioner(nk_gmres *G, nk_matvec_fn fn, void *user) {
  if (G) G->ahead.valid = false;   // (a cycle begin run ahead of the next solve belongs to the old setting)
  NK_REQUIRE(G, "NULL argument");
  G->prec = fn;
  G->prec_user = user;
  G->prec_host = false;
  G->rprec_obj = nullptr;
  G->prec_kind = fn ? 1 : 0;
  if (fn && !G->z) NK_TRY(nk_dev_alloc(&G->z, (size_t)G->ldv));
  return NK_OK;
}

extern "C" int nk_gmres_set_left_preconditioner(nk_gmres *G, nk_matvec_fn fn, void *user) {
  if (G) G->ahead.valid = false;   // (a cycle begin run ahead of the next solve belongs to the old setting)
  NK_REQUIRE(G, "NULL argument");
  NK_HIP(hipSetDevice(G->ctx->device));
  G->lprec = fn;
  G->lprec_user = user;
  G->lprec_host = false;
  G->lprec_obj = nullptr;
  G->lprec_kind = fn ? 1 : 0;
  if (fn && !G->lz) NK_TRY(nk_dev_alloc(&G->lz, (size_t)G->ldv));
  return NK_OK;
}
extern "C" int nk_gmres_set_left_preconditioner_host(nk_gmres *G, nk_matvec_fn fn, void *user) {
  if (G) G->ahead.valid = false;   // (a cycle begin run ahead of the next solve belongs to the old setting)
  NK_TRY(nk_gmres_set_left_preconditioner(G, fn, user));
  G->lprec_host = fn != nullptr;
  return NK_OK;
}
// a built-in preconditioner object (nk_precond.hip) on either side; NULL removes what sits on that side
extern "C" int nk_gmres_set_preconditioner(nk_gmres *G, int side, nk_precond *P) {
  if (G) G->ahead.valid = false;   // (a cycle begin run ahead of the next solve belongs to the old setting)
  NK_REQUIRE(G, "NULL argument");
  NK_REQUIRE(side == NK_SIDE_LEFT || side == NK_SIDE_RIGHT, "bad preconditioner side %d", side);
  NK_REQUIRE(!P || nk_precond_size(P) == G->n, "preconditioner size %lld != GMRES size %lld",
             (long long)(P ? nk_precond_size(P) : 0), (long long)G->n);
  NK_HIP(hipSetDevice(G->ctx->device));
  if (side == NK_SIDE_LEFT) {
    G->lprec = nullptr;
    G->lprec_host = false;
    G->lprec_obj = P;
    G->lprec_kind = P ? 4 : 0;
    if (P && !G->lz) NK_TRY(nk_dev_alloc(&G->lz, (size_t)G->ldv));
  } else {
    G->prec = nullptr;
    G->prec_host = false;
    G->rprec_obj = P;
    G->prec_kind = P ? 4 : 0;
    if (P && !G->z) NK_TRY(nk_dev_alloc(&G->z, (size_t)G->ldv));
  }
  return NK_OK;
}

extern "C" int nk_gmres_set_normal_form(nk_gmres *G, int on) {
  if (G) G->ahead.valid = false;   // (a cycle begin run ahead of the next solve belongs to the old setting)
  NK_REQUIRE(G, "NULL argument");
  NK_HIP(hipSetDevice(G->ctx->device));
  if (on && !G->nrm_tmp) NK_TRY(nk_dev_alloc(&G->nrm_tmp, (size_t)G->ldv));
  G->normal = on != 0;
  return NK_OK;
}

extern "C" int nk_gmres_set_normal_form_damping(nk_gmres *G, const double *d_diag, double lambda) {
  if (G) G->ahead.valid = false;   // (a cycle begin run ahead of the next solve belongs to the old setting)
  NK_REQUIRE(G, "NULL argument");
  G->nrm_diag = d_diag;
  G->nrm_lambda = d_diag ? lambda : 0.0;
  return NK_OK;
}
// y = (y + λ d∘x) · scale — the diagonal of the damped normal form, fused with the lagged-normalisation scale
__global__ __launch_bounds__(NK_BLOCK) void k_add_diag_scale(int64_t n, double lambda, const double *__restrict__ d,
                                                             const double *__restrict__ x, double *__restrict__ y,
                                                             const double *d_scale, const int *d_skip) {
  if (d_skip != nullptr && *d_skip != 0) return;
  const double sc = d_scale ? *d_scale : 1.0;
  const int64_t stride = (int64_t)gridDim.x * NK_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x; i < n; i += stride)
    y[i] = (y[i] + lambda * d[i] * x[i]) * sc;
}

extern "C" int nk_gmres_set_shift(nk_gmres *G, double sigma) {
  if (G) G->ahead.valid = false;   // (a cycle begin run ahead of the next solve belongs to the old setting)
  NK_REQUIRE(G, "NULL argument");
  G->shift = sigma;
  return NK_OK;
}
extern "C" int nk_gmres_set_shift_weights(nk_gmres *G, const double *d_m) {
  if (G) G->ahead.valid = false;   // (a cycle begin run ahead of the next solve belongs to the old setting)
  NK_REQUIRE(G, "NULL argument");
  G->d_shift_w = d_m;  // borrowed device vector (local rows); NULL = identity
  return NK_OK;
}
// y += scale·σ·(m ⊙ x) — the mass-matrix part of a shifted operator (y already carries scale·A x); m = NULL: identity
__global__ __launch_bounds__(NK_BLOCK) void k_add_shift(int64_t n, double sigma, const double *__restrict__ m,
                                                        const double *__restrict__ x, double *__restrict__ y,
                                                        const double *d_scale, const int *d_skip) {
  if (d_skip != nullptr && *d_skip != 0) return;
  const double c = sigma * (d_scale ? *d_scale : 1.0);
  const int64_t stride = (int64_t)gridDim.x * NK_BLOCK;
  if (m == nullptr)
    for (int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x; i < n; i += stride) y[i] += c * x[i];
  else
    for (int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x; i < n; i += stride) y[i] += c * m[i] * x[i];
}

// raw operator: y = scale · A x (no preconditioner)
static int op_apply_raw_unshifted(nk_gmres *G, const double *src, double *d_y, const int *d_skip, const double *oscale);
static int op_apply_raw(nk_gmres *G, const double *src, double *d_y, const int *d_skip, const double *oscale) {
  NK_TRY(op_apply_raw_unshifted(G, src, d_y, d_skip, oscale));
  if (G->shift != 0.0) {
    const int grid = nk_grid_for(G->n, NK_BLOCK * 4, 2048);
    NK_LAUNCH(G->ctx, k_add_shift, dim3(grid), dim3(NK_BLOCK), G->n, G->shift, G->d_shift_w, src, d_y, oscale, d_skip);
    NK_HIP(hipGetLastError());
  }
  return NK_OK;
}
static int op_apply_raw_unshifted(nk_gmres *G, const double *src, double *d_y, const int *d_skip, const double *oscale) {
  nk_ctx *ctx = G->ctx;
  if (G->normal) {  // AᵀA x: the plain half into a work vector, the transposed half into y (scaled afterwards if asked)
    NK_REQUIRE(G->op_kind == 1 || G->op_kind == 2, "the normal form needs a CSR or a problem operator");
    if (G->op_kind == 1) {
      NK_TRY(nk_csr_spmv_dev(G->A, src, G->nrm_tmp, d_skip));
      NK_TRY(nk_csr_spmv_t_dev(G->A, G->nrm_tmp, d_y));
    } else {
      NK_TRY(nk_problem_jvp_dev(G->P, G->d_u, src, G->nrm_tmp, d_skip));
      NK_TRY(nk_problem_vjp_dev(G->P, G->d_u, G->nrm_tmp, d_y));
    }
    if (G->nrm_diag) {
      const int grid = nk_grid_for(G->n, NK_BLOCK * 4, 2048);
      NK_LAUNCH(ctx, k_add_diag_scale, dim3(grid), dim3(NK_BLOCK), G->n, G->nrm_lambda, G->nrm_diag, src, d_y, oscale,
                d_skip);
      NK_HIP(hipGetLastError());
    } else if (oscale) {
      NK_TRY(nk_blas_scale_to(ctx, G->n, oscale, d_y, d_y, d_skip));
    }
    return NK_OK;
  }
  switch (G->op_kind) {
    case 1: return nk_csr_spmv_dev(G->A, src, d_y, d_skip, oscale);
    case 2: return nk_problem_jvp_dev(G->P, G->d_u, src, d_y, d_skip, oscale);
    case 3:
      ctx->stats.op_applies++;
      if (oscale) NK_FAIL(NK_E_INVALID, "internal: output scale with a callback operator");
      if (G->fn_host) return host_callback(G, G->fn, G->fn_user, src, d_y, "operator");
      if (G->fn(G->fn_user, src, d_y, (void *)ctx->stream) != 0) NK_FAIL(NK_E_CALLBACK, "operator callback failed");
      return NK_OK;
    default: NK_FAIL(NK_E_INVALID, "GMRES has no operator");
  }
}

// ----------------------------------------------------------------------------- Chebyshev polynomial preconditioner
// M⁻¹ v = p_d(A) v: d steps of the Chebyshev iteration for A y = v from y = 0 on the interval [lmin, lmax]
// (Saad, Iterative Methods, Alg. 12.1). Operator applications only — no inner products, hence no all-reduce on
// multi-GPU — and it reuses the fastest kernel of the library (SpMV / fused JVP). The `precs` hook of the
// reference (test/Core/core_tests__item21.jl) is where a user would plug such a preconditioner in.
__global__ __launch_bounds__(NK_BLOCK) void k_cheb_init(int64_t n, const double *__restrict__ v, double inv_theta,
                                                        double *__restrict__ r, double *__restrict__ d,
                                                        double *__restrict__ y, const int *d_skip) {
  if (d_skip != nullptr && *d_skip != 0) return;
  const int64_t stride = (int64_t)gridDim.x * NK_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x; i < n; i += stride) {
    const double x = v[i], dd = x * inv_theta;
    r[i] = x;
    d[i] = dd;
    y[i] = dd;
  }
}
// r −= A d ; d = c1 d + c2 r ; y += d      (56 n bytes)
__global__ __launch_bounds__(NK_BLOCK) void k_cheb_update(int64_t n, const double *__restrict__ Ad, double c1, double c2,
                                                          double *__restrict__ r, double *__restrict__ d,
                                                          double *__restrict__ y, const int *d_skip) {
  if (d_skip != nullptr && *d_skip != 0) return;
  const int64_t npair = n >> 1, stride = (int64_t)gridDim.x * NK_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x; i < npair; i += stride) {
    const double2 t = reinterpret_cast<const double2 *>(Ad)[i];
    double2 rr = reinterpret_cast<double2 *>(r)[i], dd = reinterpret_cast<double2 *>(d)[i],
            yy = reinterpret_cast<double2 *>(y)[i];
    rr.x -= t.x; rr.y -= t.y;
    dd.x = c1 * dd.x + c2 * rr.x; dd.y = c1 * dd.y + c2 * rr.y;
    yy.x += dd.x; yy.y += dd.y;
    reinterpret_cast<double2 *>(r)[i] = rr;
    reinterpret_cast<double2 *>(d)[i] = dd;
    reinterpret_cast<double2 *>(y)[i] = yy;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t i = n - 1;
    r[i] -= Ad[i];
    d[i] = c1 * d[i] + c2 * r[i];
    y[i] += d[i];
  }
}

static int cheb_apply(nk_gmres *G, const double *src, double *dst, const int *d_skip) {
  nk_ctx *ctx = G->ctx;
  const int64_t n = G->n;
  const double theta = 0.5 * (G->cheb_lmax + G->cheb_lmin), delta = 0.5 * (G->cheb_lmax - G->cheb_lmin);
  const double sigma1 = theta / delta;
  double rho = 1.0 / sigma1;
  const int grid = nk_grid_for(n >> 1, NK_BLOCK * 2, 4096);
  {
    nk_prof_scope prof_(ctx, NK_K_OTHER, 32.0 * (double)n);
    NK_LAUNCH(ctx, k_cheb_init, dim3(grid), dim3(NK_BLOCK), n, src, 1.0 / theta, G->cr, G->cd, dst, d_skip);
  }
  // concrete CSR operator: the vector update rides in the SpMV's row epilogue (d ping-pongs between two buffers)
  const bool fuse = !G->normal && G->shift == 0.0 && ((G->op_kind == 1) || (G->op_kind == 2 && G->P->kind == NK_PROBLEM_BRATU2D));
  double *dcur = G->cd, *dnext = G->cd2;
  for (int k = 1; k < G->cheb_degree; ++k) {
    const double rho_new = 1.0 / (2.0 * sigma1 - rho);
    if (fuse) {
      nk_spmv_epi ep;
      ep.mode = 1;
      ep.c1 = rho_new * rho;
      ep.c2 = 2.0 * rho_new / delta;
      ep.r = G->cr;
      ep.dnew = dnext;
      ep.yacc = dst;
      if (G->op_kind == 1) NK_TRY(nk_csr_spmv_dev(G->A, dcur, nullptr, d_skip, nullptr, &ep));
      else NK_TRY(nk_problem_jvp_dev(G->P, G->d_u, dcur, nullptr, d_skip, nullptr, &ep));
      double *tmp = dcur; dcur = dnext; dnext = tmp;
    } else {
      NK_TRY(op_apply_raw(G, G->cd, G->ct, d_skip, nullptr));
      nk_prof_scope prof_(ctx, NK_K_OTHER, 56.0 * (double)n);
      NK_LAUNCH(ctx, k_cheb_update, dim3(grid), dim3(NK_BLOCK), n, (const double *)G->ct, rho_new * rho,
                2.0 * rho_new / delta, G->cr, G->cd, dst, d_skip);
    }
    rho = rho_new;
  }
  NK_HIP(hipGetLastError());
  return NK_OK;
}

// start vector with energy in every mode (a constant vector has none in the oscillatory ones)
__global__ __launch_bounds__(NK_BLOCK) void k_hash_fill(int64_t n, double *__restrict__ x) {
  const int64_t stride = (int64_t)gridDim.x * NK_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x; i < n; i += stride) {
    const uint32_t hsh = (uint32_t)i * 2654435761u;
    x[i] = 0.5 + (double)((hsh >> 8) & 0xffff) / 65536.0 * ((hsh & 1) ? 1.0 : -1.0);
  }
}
// Gershgorin: per-block max of Σ_j |a_ij| and of the signed diagonal-dominant centre (for the sign of the spectrum)
__global__ __launch_bounds__(NK_BLOCK) void k_gershgorin(int64_t nrows, const int32_t *__restrict__ rowptr,
                                                         const int32_t *__restrict__ col, const double *__restrict__ val,
                                                         double *__restrict__ partials) {
  __shared__ double sm[8];
  double mx = 0.0, dsum = 0.0;
  const int64_t stride = (int64_t)gridDim.x * NK_BLOCK;
  for (int64_t r = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x; r < nrows; r += stride) {
    double s = 0.0;
    for (int32_t p = rowptr[r]; p < rowptr[r + 1]; ++p) {
      s += fabs(val[p]);
      if (col[p] == r) dsum += val[p];
    }
    mx = s > mx ? s : mx;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double m2 = __shfl_xor(mx, o, 64);
    mx = m2 > mx ? m2 : mx;
    dsum += __shfl_xor(dsum, o, 64);
  }
  if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6] = mx; sm[4 + (threadIdx.x >> 6)] = dsum; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = sm[0];
    for (int k = 1; k < 4; ++k) a = sm[k] > a ? sm[k] : a;
    partials[blockIdx.x] = a;
    partials[gridDim.x + blockIdx.x] = sm[4] + sm[5] + sm[6] + sm[7];
  }
}

// bound / estimate of the dominant eigenvalue (with its sign):
//   concrete CSR: Gershgorin max_i Σ_j|a_ij| — a guaranteed bound; sign from the trace
//   otherwise   : 30 power iterations, Rayleigh quotient, widened by 15 %
static int estimate_lambda(nk_gmres *G, double *lambda) {
  nk_ctx *ctx = G->ctx;
  const int64_t n = G->n;
  if (G->op_kind == 1 && !G->normal) {
    nk_csr *A = G->A;
    const int grid = nk_grid_for(A->nrows, NK_BLOCK, NK_MAX_RED_BLOCKS);
    NK_LAUNCH(ctx, k_gershgorin, dim3(grid), dim3(NK_BLOCK), A->nrows, A->d_rowptr, A->d_col, A->d_val, ctx->d_partials);
    // reduce on the host side of a tiny copy (≤ 2048 doubles)
    std::vector<double> hp(2 * (size_t)grid);
    NK_HIP(hipMemcpyAsync(hp.data(), ctx->d_partials, hp.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    NK_HIP(hipStreamSynchronize(ctx->stream));
    double mx = 0.0, tr = 0.0;
    for (int b = 0; b < grid; ++b) { mx = hp[b] > mx ? hp[b] : mx; tr += hp[grid + b]; }
    if (ctx->nranks > 1) {
      // both scalars go up in ONE copy from two distinct pinned slots (re-using a slot between two async copies raced
      // with the first copy when the collective is asynchronous, i.e. under RCCL)
      ctx->h_pinned[0] = mx;
      ctx->h_pinned[1] = tr;
      NK_HIP(hipMemcpyAsync(ctx->d_scal, ctx->h_pinned, 2 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
      NK_TRY(nk_comm_allreduce(ctx, ctx->d_scal, 1, 1));
      NK_TRY(nk_comm_allreduce(ctx, ctx->d_scal + 1, 1, 0));
      double v[2];
      NK_TRY(nk_scalars_to_host(ctx, ctx->d_scal, 2, v));
      mx = v[0];
      tr = v[1];
    }
    *lambda = (tr < 0.0) ? -mx : mx;
    return NK_OK;
  }
  NK_LAUNCH(ctx, k_hash_fill, dim3(nk_grid_for(n, NK_BLOCK * 4, 4096)), dim3(NK_BLOCK), n, G->cd);
  double lam = 0.0;
  for (int it = 0; it < 30; ++it) {
    NK_TRY(op_apply_raw(G, G->cd, G->ct, nullptr, nullptr));
    NK_TRY(nk_blas_dot(ctx, n, G->cd, G->ct, ctx->d_scal));
    NK_TRY(nk_blas_dot(ctx, n, G->cd, G->cd, ctx->d_scal + 1));
    NK_TRY(nk_blas_dot(ctx, n, G->ct, G->ct, ctx->d_scal + 2));
    double v[3];
    NK_TRY(nk_scalars_to_host(ctx, ctx->d_scal, 3, v));
    if (!(v[1] > 0.0) || !(v[2] > 0.0)) break;
    lam = v[0] / v[1];  // Rayleigh quotient of the current vector
    NK_TRY(nk_blas_lincomb(ctx, n, 1.0 / sqrt(v[2]), G->ct, 0.0, G->ct, G->cd));
  }
  *lambda = 1.15 * lam;
  return NK_OK;
}

extern "C" int nk_gmres_set_chebyshev_preconditioner(nk_gmres *G, int degree, double lambda_min, double lambda_max,
                                                     double ratio) {
  if (G) G->ahead.valid = false;   // (a cycle begin run ahead of the next solve belongs to the old setting)
  NK_REQUIRE(G, "NULL argument");
  NK_REQUIRE(G->op_kind != 0, "set the operator before the preconditioner");
  if (degree <= 0) {
    G->prec_kind = G->rprec_obj ? 4 : (G->prec ? 1 : 0);
    return NK_OK;
  }
  NK_HIP(hipSetDevice(G->ctx->device));
  if (!G->z) NK_TRY(nk_dev_alloc(&G->z, (size_t)G->ldv));
  if (!G->cr) NK_TRY(nk_dev_alloc(&G->cr, (size_t)G->ldv));
  if (!G->cd) NK_TRY(nk_dev_alloc(&G->cd, (size_t)G->ldv));
  if (!G->ct) NK_TRY(nk_dev_alloc(&G->ct, (size_t)G->ldv));
  if (!G->cd2) NK_TRY(nk_dev_alloc(&G->cd2, (size_t)G->ldv));
  if (lambda_max == 0.0) {  // estimate the dominant eigenvalue, widen by 10 %, and take lmin = lmax / ratio
    double lam = 0.0;
    NK_TRY(estimate_lambda(G, &lam));
    NK_REQUIRE(lam != 0.0 && lam == lam, "could not estimate the spectrum of the operator");
    if (ratio <= 1.0) ratio = 30.0;
    lambda_max = lam;
    lambda_min = lambda_max / ratio;
  }
  NK_REQUIRE(lambda_min * lambda_max > 0.0 && lambda_min != lambda_max, "Chebyshev interval must not contain 0");
  G->cheb_degree = degree;
  G->cheb_lmin = lambda_min;
  G->cheb_lmax = lambda_max;
  G->prec_kind = 2;
  return NK_OK;
}
extern "C" int nk_gmres_get_chebyshev_interval(nk_gmres *G, double *lambda_min, double *lambda_max) {
  NK_REQUIRE(G, "NULL argument");
  if (lambda_min) *lambda_min = G->cheb_lmin;
  if (lambda_max) *lambda_max = G->cheb_lmax;
  return NK_OK;
}

// Built-in multigrid V-cycle as the right preconditioner (Bratu problems, single rank): builds the hierarchy on first use,
// afterwards only re-linearises it at `u` — call it again for every new Jacobian, like `precs(A, p)`.
extern "C" int nk_gmres_set_multigrid_preconditioner(nk_gmres *G, nk_problem *P, const double *u, int memspace, int nu,
                                                     int coarse_max) {
  if (G) G->ahead.valid = false;   // (a cycle begin run ahead of the next solve belongs to the old setting)
  NK_REQUIRE(G, "NULL argument");
  if (nu <= 0 || !P) {  // remove
    G->prec_kind = G->rprec_obj ? 4 : (G->prec ? 1 : 0);
    return NK_OK;
  }
  NK_REQUIRE(u, "NULL argument");
  NK_REQUIRE(G->op_kind != 0, "set the operator before the preconditioner");
  NK_REQUIRE(P->n_local == G->n, "problem size %lld != GMRES size %lld", (long long)P->n_local, (long long)G->n);
  NK_HIP(hipSetDevice(G->ctx->device));
  const double *du = u;
  if (memspace != NK_DEVICE) {
    if (!G->d_u_own) NK_TRY(nk_dev_alloc(&G->d_u_own, (size_t)G->n + 1));
    NK_HIP(hipMemcpyAsync(G->d_u_own, u, G->n * sizeof(double), hipMemcpyHostToDevice, G->ctx->stream));
    du = G->d_u_own;
  }
  if (!G->mg) NK_TRY(nk_mg_create(P, nu, coarse_max, &G->mg));
  NK_TRY(nk_problem_jvp_prepare(P, du));
  NK_TRY(nk_mg_update(G->mg, du));
  if (!G->z) NK_TRY(nk_dev_alloc(&G->z, (size_t)G->ldv));
  G->prec_kind = 3;
  return NK_OK;
}

static int lprec_apply(nk_gmres *G, const double *src, double *dst, const int *d_skip) {
  if (G->lprec_kind == 4) return nk_precond_apply_dev(G->lprec_obj, src, dst, d_skip);
  if (G->lprec_host) return host_callback(G, G->lprec, G->lprec_user, src, dst, "left preconditioner");
  if (G->lprec(G->lprec_user, src, dst, (void *)G->ctx->stream) != 0) NK_FAIL(NK_E_CALLBACK, "left preconditioner failed");
  return NK_OK;
}
static int prec_apply(nk_gmres *G, const double *src, double *dst, const int *d_skip) {
  if (G->prec_kind == 2) return cheb_apply(G, src, dst, d_skip);
  if (G->prec_kind == 3) return nk_mg_apply(G->mg, src, dst, d_skip);
  if (G->prec_kind == 4) return nk_precond_apply_dev(G->rprec_obj, src, dst, d_skip);
  if (G->prec_host) return host_callback(G, G->prec, G->prec_user, src, dst, "preconditioner");
  if (G->prec(G->prec_user, src, dst, (void *)G->ctx->stream) != 0) NK_FAIL(NK_E_CALLBACK, "preconditioner failed");
  return NK_OK;
}

// can the operator kernel apply the output scale itself (built-in kernels) or do we have to scale the input?
static bool op_fuses_scale(const nk_gmres *G) {
  if (G->prec_kind) return false;
  if (G->op_kind == 1) return true;
  if (G->op_kind == 2) return G->P->kind != NK_PROBLEM_USER;
  return false;
}

// y −= scale·θ·x — the shift of a Newton-basis step for operators whose kernels cannot take it in their row epilogue
__global__ __launch_bounds__(NK_BLOCK) void k_sub_theta(int64_t n, const double *__restrict__ theta,
                                                        const double *__restrict__ x, double *__restrict__ y,
                                                        const double *d_scale, const int *d_skip) {
  if (d_skip != nullptr && *d_skip != 0) return;
  const double c = (*theta) * (d_scale ? *d_scale : 1.0);
  const int64_t stride = (int64_t)gridDim.x * NK_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x; i < n; i += stride) y[i] -= c * x[i];
}

// y = scale · (A x − θ x) (right-preconditioned: A M⁻¹ x); scale / theta may be nullptr (= 1 / 0)
static int op_apply(nk_gmres *G, const double *d_x, double *d_y, const int *d_skip, const double *d_scale,
                    const double *d_theta = nullptr) {
  nk_ctx *ctx = G->ctx;
  const double *src = d_x;
  const double *oscale = d_scale;
  if (d_scale && !op_fuses_scale(G)) {  // callback operators / preconditioners see the normalised vector
    NK_TRY(nk_blas_scale_to(ctx, G->n, d_scale, d_x, G->w, d_skip));
    src = G->w;
    oscale = nullptr;
  }
  const double *src0 = src;
  if (G->prec_kind) {
    NK_TRY(prec_apply(G, src, G->z, d_skip));
    src = G->z;
  }
  if (d_theta) {
    // the built-in kernels subtract θ·x[row] in their row epilogue (the diagonal gather they have just made)
    const bool epi_ok = !G->prec_kind && !G->lprec_kind && !G->normal && G->shift == 0.0 &&
                        (G->op_kind == 1 || (G->op_kind == 2 && G->P->kind == NK_PROBLEM_BRATU2D));
    if (epi_ok) {
      nk_spmv_epi ep;
      ep.mode = 3;
      ep.theta = d_theta;
      if (G->op_kind == 1) return nk_csr_spmv_dev(G->A, src, d_y, d_skip, oscale, &ep);
      return nk_problem_jvp_dev(G->P, G->d_u, src, d_y, d_skip, oscale, &ep);
    }
    double *dst = G->lprec_kind ? G->lz : d_y;
    NK_TRY(op_apply_raw(G, src, dst, d_skip, oscale));
    if (G->lprec_kind) NK_TRY(lprec_apply(G, G->lz, d_y, d_skip));
    const int grid = nk_grid_for(G->n, NK_BLOCK * 4, 2048);
    NK_LAUNCH(ctx, k_sub_theta, dim3(grid), dim3(NK_BLOCK), G->n, d_theta, src0, d_y, oscale, d_skip);
    NK_HIP(hipGetLastError());
    return NK_OK;
  }
  if (G->lprec_kind) {   // Pl⁻¹ (A Pr⁻¹ x): the raw product into a work vector, the left solve into the basis column
    NK_TRY(op_apply_raw(G, src, G->lz, d_skip, oscale));
    return lprec_apply(G, G->lz, d_y, d_skip);
  }
  return op_apply_raw(G, src, d_y, d_skip, oscale);
}

int nk_gmres_op_apply(nk_gmres *G, const double *d_x, double *d_y, const int *d_skip, const double *d_scale,
                      const double *d_theta) {
  return op_apply(G, d_x, d_y, d_skip, d_scale, d_theta);
}
// s applications of the shifted, scaled operator in a row — Y[:, j] = scale_j·(A − θ_j I) Y[:, j−1], Y[:, −1] = x; scale_0 =
// d_scale[0], later ones d_scale[1] — as ONE launch of the resident matrix-powers kernel (nk_powers.hip) when the operator is a
// plain CSR matrix that kernel can hold on the chip; *done = false: the caller applies the operator column by column
int nk_gmres_op_powers(nk_gmres *G, const double *d_x, double *d_Y, int64_t ldy, int s, const int *d_skip, const double *d_scale,
                       const double *d_theta, bool *done) {
  *done = false;
  if (s < 2 || G->prec_kind || G->lprec_kind || G->normal || G->shift != 0.0) return NK_OK;
  if (G->op_kind == 1) {
    if (!nk_csr_powers_ready(G->A)) return NK_OK;
    NK_TRY(nk_csr_powers_dev(G->A, d_x, d_Y, ldy, s, d_scale, d_scale + 1, d_theta, d_skip));
  } else if (G->op_kind == 2 && G->P->kind == NK_PROBLEM_BRATU2D) {   // the matrix-free stencil operator: rows generated
    if (!nk_problem_powers_ready(G->P)) return NK_OK;
    NK_TRY(nk_problem_powers_dev(G->P, G->d_u, d_x, d_Y, ldy, s, d_scale, d_scale + 1, d_theta, d_skip));
  } else {
    return NK_OK;
  }
  *done = true;
  return NK_OK;
}

// Real bounds of the operator's spectrum for the s-step Newton basis, left on the device as {−lo, hi}: Gershgorin discs of
// a concrete CSR operator, the closed-form discs of the Bratu stencil (diagonal 4c − d_k, radius ≤ 4c), or bounds the caller
// supplied (nk_gmres_set_spectrum_interval). None for callback operators, preconditioned or normal-form operators.
__global__ void k_store2(double a, double b, double *out2) {
  if (threadIdx.x == 0) { out2[0] = a; out2[1] = b; }
}
int nk_gmres_spectrum_interval_dev(nk_gmres *G, double *d_out2, const double **where, bool *have) {
  nk_ctx *ctx = G->ctx;
  *have = false;
  *where = d_out2;
  if (G->ss_ival_user) {
    NK_LAUNCH(ctx, k_store2, dim3(1), dim3(64), -G->ss_ival[0], G->ss_ival[1], d_out2);
    *have = true;
    return NK_OK;
  }
  if (G->prec_kind || G->lprec_kind || G->normal || G->shift != 0.0) return NK_OK;
  if (G->op_kind == 1 && G->A->nblocks > 0) {
    NK_TRY(nk_csr_gershgorin_dev(G->A, d_out2, where));
    if (*where != d_out2) {   // the Jacobian fill left them (one rank): nothing to compute, nothing to reduce
      *have = true;
      return NK_OK;
    }
  } else if (G->op_kind == 2 && G->P->kind == NK_PROBLEM_BRATU2D && G->n > 0) {
    NK_TRY(nk_problem_spectrum_interval_dev(G->P, G->d_u, d_out2));
  } else {
    return NK_OK;
  }
  if (ctx->nranks > 1) NK_TRY(nk_comm_allreduce(ctx, d_out2, 2, 1));
  *have = true;
  return NK_OK;
}
extern "C" int nk_gmres_set_spectrum_interval(nk_gmres *G, double lo, double hi) {
  if (G) G->ahead.valid = false;   // (a cycle begin run ahead of the next solve belongs to the old setting)
  NK_REQUIRE(G, "NULL argument");
  if (lo == 0.0 && hi == 0.0) { G->ss_ival_user = false; return NK_OK; }
  NK_REQUIRE(lo < hi && lo == lo && hi == hi && !std::isinf(lo) && !std::isinf(hi), "spectrum interval needs finite lo < hi");
  G->ss_ival_user = true;
  G->ss_ival[0] = lo;
  G->ss_ival[1] = hi;
  return NK_OK;
}
extern "C" int nk_gmres_set_sstep_basis(nk_gmres *G, int basis) {
  if (G) G->ahead.valid = false;   // (a cycle begin run ahead of the next solve belongs to the old setting)
  NK_REQUIRE(G, "NULL argument");
  NK_REQUIRE(basis == NK_SS_BASIS_AUTO || basis == NK_SS_BASIS_MONOMIAL || basis == NK_SS_BASIS_NEWTON, "bad s-step basis %d", basis);
  G->ss_basis = basis;
  return NK_OK;
}

// ----------------------------------------------------------------------------- DCGS2, one reduction per step
// eligible: built-in linear operators (they take the un-normalised pending column as it is) and no callback preconditioner
static bool dcgs2r_eligible(const nk_gmres *G) {
  const bool op_ok = (G->op_kind == 1) || (G->op_kind == 2 && G->P->kind != NK_PROBLEM_USER);
  return op_ok && (G->prec_kind == 0 || G->prec_kind == 2 || G->prec_kind == 3 || G->prec_kind == 4) &&
         (G->lprec_kind == 0 || G->lprec_kind == 4) && G->m <= NK_MAX_NV - 2 &&
         G->n <= (int64_t)NK_MAX_ROW_TILES * NK_BLOCK * 8;
}
static bool use_dcgs2r(const nk_gmres *G) {
  if (!dcgs2r_eligible(G)) return false;
  static const bool two = getenv("NK_DCGS2_TWO_REDUCTIONS") != nullptr;  // A/B switch: keep the two-reduction form
  return G->ortho == NK_ORTHO_DCGS2_1R || (G->ortho == NK_ORTHO_DCGS2 && !two);
}
// The cycle's last Hessenberg column needs the norm of the pending column that the last step leaves behind. That column is
// never used as a basis vector, so it is closed WITHOUT its second projection: β = ‖u‖ comes out of the last axpy sweep for
// free and the column is the first projection alone (differs from the re-orthogonalised one by the re-orthogonalisation
// correction, i.e. at rounding level: 1e-15 relative on the solutions, oracle/reference_restatement.py::gmres_dcgs2_1r).
// NK_DCGS2_FULL_FLUSH=1 keeps the full form (one more dot sweep over all columns per cycle) for A/B runs.
static bool dcgs2r_full_flush() {
  static const bool full = getenv("NK_DCGS2_FULL_FLUSH") != nullptr;
  return full;
}
static double *tprev_buf(nk_gmres *G, int k) { return G->d_tprev + (size_t)(k & 1) * (NK_MAX_NV + 2); }
static bool dcgs2r_split_tail() {
  static const bool split = getenv("NK_DCGS2_SPLIT_TAIL") != nullptr;  // A/B switch: the scalar work as its own launch
  return split;
}
static int arnoldi_step_1r(nk_gmres *G, int k, bool last_of_cycle) {
  nk_ctx *ctx = G->ctx;
  const int64_t n = G->n, ldv = G->ldv;
  const int *skip = &G->d_ctl->done;
  double *zk = G->V + (size_t)(k + 1) * ldv;
  NK_TRY(op_apply(G, G->V + (size_t)k * ldv, zk, skip, nullptr));  // z = A u on the pending (un-normalised) column
  NK_TRY(nk_blas_dcgs2r_dots(ctx, n, k, false, G->V, ldv, G->d_red, skip));
  double *ss_out = (last_of_cycle && !dcgs2r_full_flush()) ? G->d_red + (k + 1) : nullptr;  // slot of u·u at the flush
  if (dcgs2r_split_tail()) {
    NK_LAUNCH(ctx, k_dcgs2r_tail, dim3(1), dim3(256), G->d_ctl, k, 0, (const double *)G->d_red, G->d_s, G->d_Hraw, G->m,
              (const double *)tprev_buf(G, k), tprev_buf(G, k + 1), G->d_R, G->d_cs, G->d_sn, G->d_g, G->d_ca, G->d_cb,
              G->h_pub_dev, G->cycle_seq);
    NK_TRY(nk_blas_dcgs2r_axpy(ctx, n, k, G->V, ldv, G->d_ca, G->d_cb, skip, ss_out));
    return NK_OK;
  }
  // scalar work in the prologue of the sweep (k_dcgs2r_axpy_tail): one launch less per Arnoldi step
  const int64_t tile = (int64_t)NK_BLOCK * DRT;
  const int grid = (int)((n + tile - 1) / tile > 0 ? (n + tile - 1) / tile : 1);
  {
    nk_prof_scope prof_(ctx, NK_K_MULTIAXPY, 8.0 * (double)n * (k + 4));
    double *ssp = ss_out ? ctx->d_partials : (double *)nullptr;
    static const int desc = getenv("NK_AXPY_DESC") ? atoi(getenv("NK_AXPY_DESC")) : 1;  // NK_AXPY_DESC=0: ascending (A/B)
    if (G->m <= 31)
      NK_LAUNCH(ctx, k_dcgs2r_axpy_tail<32>, dim3(grid), dim3(NK_BLOCK), n, k, G->V, ldv, G->d_ctl, (const double *)G->d_red,
                G->d_s, G->d_Hraw, G->m, (const double *)tprev_buf(G, k), tprev_buf(G, k + 1), G->d_R, G->d_cs, G->d_sn,
                G->d_g, ssp, G->h_pub_dev, G->cycle_seq, desc);
    else
      NK_LAUNCH(ctx, k_dcgs2r_axpy_tail<NK_MAX_NV>, dim3(grid), dim3(NK_BLOCK), n, k, G->V, ldv, G->d_ctl,
                (const double *)G->d_red, G->d_s, G->d_Hraw, G->m, (const double *)tprev_buf(G, k), tprev_buf(G, k + 1),
                G->d_R, G->d_cs, G->d_sn, G->d_g, ssp, G->h_pub_dev, G->cycle_seq, desc);
  }
  NK_HIP(hipGetLastError());
  if (ss_out) NK_TRY(nk_blas_reduce_one(ctx, ctx->d_partials, grid, ss_out, skip));
  return NK_OK;
}
// after the last step of a cycle: the pending column's reduction completes the last Hessenberg column
static int arnoldi_flush_1r(nk_gmres *G, int steps) {
  nk_ctx *ctx = G->ctx;
  const int *skip = &G->d_ctl->done;
  if (steps <= 0) return NK_OK;
  const bool full = dcgs2r_full_flush();
  if (full) NK_TRY(nk_blas_dcgs2r_dots(ctx, G->n, steps, true, G->V, G->ldv, G->d_red, skip));
  NK_LAUNCH(ctx, k_dcgs2r_tail, dim3(1), dim3(256), G->d_ctl, steps, full ? 1 : 2, (const double *)G->d_red, G->d_s,
            G->d_Hraw, G->m, (const double *)tprev_buf(G, steps), tprev_buf(G, steps + 1), G->d_R, G->d_cs, G->d_sn, G->d_g,
            G->d_ca, G->d_cb, G->h_pub_dev, G->cycle_seq);
  NK_HIP(hipGetLastError());
  return NK_OK;
}

// ----------------------------------------------------------------------------- one Arnoldi step (enqueue only)
static int arnoldi_step(nk_gmres *G, int k) {
  nk_ctx *ctx = G->ctx;
  const int64_t n = G->n, ldv = G->ldv;
  const int *skip = &G->d_ctl->done;
  const int nv = k + 1;
  double *wk = G->V + (size_t)(k + 1) * ldv;  // the new (un-normalised) column is built in place
  NK_TRY(op_apply(G, G->V + (size_t)k * ldv, wk, skip, G->d_s + k));
  // (operators reached through callbacks cannot take the one-reduction form; their two-reduction form keeps ≤ 32 columns in
  //  registers, so longer restarts fall through to plain CGS2 below)
  const bool dcgs2_family = (G->ortho == NK_ORTHO_DCGS2 || G->ortho == NK_ORTHO_DCGS2_1R);
  if (dcgs2_family && G->m <= 31) {
    // CGS2 with delayed re-orthogonalisation: column k holds p (first projection only) when k ≥ 1; the operator above
    // was applied to it. Pass A applies the pending correction to column k, rebuilds A v_k from A p through the Arnoldi
    // relation and takes the first projection of the new vector — one sweep over the basis; pass B is the usual fused
    // axpy + second projection, whose correction stays pending. ‖w″‖ comes from Pythagoras in k_givens. Two sweeps and
    // two reductions per step instead of three (oracle/reference_restatement.py::gmres(ortho="dcgs2")).
    const bool single = nk_ctx_is_single(ctx);
    if (k == 0) {
      NK_TRY(nk_blas_multidot(ctx, n, 1, G->V, ldv, wk, G->d_h, false, skip, G->d_s));
    } else {
      if (!single)  // (single rank: the coefficients were produced by the previous step's k_givens_dcgs2)
        NK_LAUNCH(ctx, k_dcgs2_coef, dim3(1), dim3(64), (const nk_gmres_ctl *)G->d_ctl, k, (const double *)G->d_Hraw, G->m,
                  (const double *)G->d_h2, (const double *)G->d_s, G->d_ca, G->d_cb);
      NK_TRY(nk_blas_dcgs2_pass_a(ctx, n, k, G->V, ldv, G->d_ca, G->d_cb, G->d_s, G->d_h, skip));
    }
    if (single) {
      NK_TRY(nk_blas_fused_axpy_dot(ctx, n, nv, G->V, ldv, G->d_h, G->d_s, wk, NK_SUMSQ_PARTIALS_ONLY, skip));
      nk_prof_scope prof_(ctx, NK_K_REDUCE_SMALL, 8.0 * (nv + 1) * ctx->last_red_grid);
      NK_LAUNCH(ctx, k_givens_dcgs2, dim3(1), dim3(1024), G->d_ctl, (const double *)G->d_h, (const double *)ctx->d_partials,
                ctx->last_red_grid, G->d_R, G->d_cs, G->d_sn, G->d_g, G->d_s, G->m, G->d_Hraw, G->d_ca, G->d_cb, G->d_h2,
                G->h_pub_dev, G->cycle_seq);
    } else {
      NK_TRY(nk_blas_fused_axpy_dot(ctx, n, nv, G->V, ldv, G->d_h, G->d_s, wk, G->d_h2, skip));
      NK_LAUNCH(ctx, k_givens, dim3(1), dim3(64), G->d_ctl, G->d_h, (const double *)G->d_h2, G->d_ss, G->d_R, G->d_cs,
                G->d_sn, G->d_g, G->d_s, G->m, (const double *)nullptr, 0, 1, G->d_Hraw, G->h_pub_dev, G->cycle_seq);
    }
  } else if (G->ortho == NK_ORTHO_MGS) {
    for (int i = 0; i <= k; ++i) {  // h_i = v_i·w ; w -= h_i v_i ; the last axpy also yields ‖w‖²
      NK_TRY(nk_blas_multidot(ctx, n, 1, G->V + (size_t)i * ldv, ldv, wk, G->d_h + i, false, skip, G->d_s + i));
      NK_TRY(nk_blas_multiaxpy(ctx, n, 1, G->V + (size_t)i * ldv, ldv, G->d_h + i, -1.0, wk,
                               i == k ? G->d_ss : nullptr, skip, nullptr, G->d_s + i));
    }
    NK_LAUNCH(ctx, k_givens, dim3(1), dim3(64), G->d_ctl, G->d_h, (const double *)nullptr, G->d_ss,
                       G->d_R, G->d_cs, G->d_sn, G->d_g, G->d_s, G->m, (const double *)nullptr, 0, 0, (double *)nullptr,
                       G->h_pub_dev, G->cycle_seq);
  } else {
    const bool dgks = (G->ortho == NK_ORTHO_CGS);
    const int *skip2 = dgks ? &G->d_ctl->pad0 : skip;
    // (Folding the stage-2 reductions into the consumers' prologues was tried and measured slower — 253 vs 282 steps/s:
    //  every one of the 512 consumer blocks re-reads all nv × 512 partials, ≈130 MB of extra L2 traffic per kernel.)
    // pass 1: h = Vᵀw (DGKS also needs ‖w‖² → self slot h[k+1])
    NK_TRY(nk_blas_multidot(ctx, n, nv, G->V, ldv, wk, G->d_h, dgks, skip, G->d_s));
    if (nv <= 32) {
      // pass 2 (fused): w ← w − V h ; h2 = Vᵀw ; h2[nv] = ‖w‖²
      NK_TRY(nk_blas_fused_axpy_dot(ctx, n, nv, G->V, ldv, G->d_h, G->d_s, wk, G->d_h2, skip));
    } else {
      NK_TRY(nk_blas_multiaxpy(ctx, n, nv, G->V, ldv, G->d_h, -1.0, wk, G->d_h2 + nv, skip, nullptr, G->d_s));
      NK_TRY(nk_blas_multidot(ctx, n, nv, G->V, ldv, wk, G->d_h2, false, skip, G->d_s));
    }
    if (dgks)
      NK_LAUNCH(ctx, k_dgks, dim3(1), dim3(64), G->d_ctl, G->d_h, G->d_h2, G->d_ss,
                         1.0 / (double)ctx->nranks);
    // pass 3: w ← w − V h2 ; ‖w‖²   (CGS2: always; CGS+DGKS: only when the test asked for it).
    // Single rank + CGS2: the ‖w‖² partials are reduced inside k_givens (one launch less per Arnoldi step).
    const bool fold = (!dgks && nk_ctx_is_single(ctx));
    // several ranks + CGS2 + fused pass: ‖w‖² by Pythagoras inside k_givens → 2 all-reduces per step instead of 3
    const bool pythag = (!dgks && !fold && nv <= 32);
    double *ss_dst = fold ? NK_SUMSQ_PARTIALS_ONLY : (pythag ? nullptr : G->d_ss);
    NK_TRY(nk_blas_multiaxpy(ctx, n, nv, G->V, ldv, G->d_h2, -1.0, wk, ss_dst, skip2, nullptr, G->d_s));
    NK_LAUNCH(ctx, k_givens, dim3(1), dim3(64), G->d_ctl, G->d_h, (const double *)G->d_h2, G->d_ss, G->d_R, G->d_cs,
              G->d_sn, G->d_g, G->d_s, G->m, fold ? (const double *)ctx->d_partials_ss : (const double *)nullptr,
              fold ? ctx->last_red_grid : 0, pythag ? 1 : 0, (double *)nullptr, G->h_pub_dev, G->cycle_seq);
  }
  NK_HIP(hipGetLastError());
  return NK_OK;
}

// A/B switch NK_GMRES_GRAPH=1: one fixed-work restart cycle (b → column 0, ≤ m Arnoldi steps of four launches, flush,
// back-substitution, x = V y: ≈ 125 launches) captured ONCE into a HIP graph and replayed per linear solve, for operators whose
// arguments do not change between solves (the concrete CSR Jacobian). The kernels' by-value cycle number is the captured one,
// so the host waits with a stream synchronisation instead of polling the published sequence word. Measured on MI355X
// (Bratu 1024², 30 Arnoldi steps per Newton step): see DESIGN.md §8 — the eager path keeps the queue full (a launch costs
// the host 2–4 µs, a step's four kernels run for 64 µs), so the graph has nothing to remove; it stays an opt-in.
static int gmres_solve_graph(nk_gmres *G, const double *d_b, double *d_x, double atol, double rtol, int steps,
                             nk_gmres_info *info, bool *used) {
  nk_ctx *ctx = G->ctx;
  const int64_t n = G->n, ldv = G->ldv;
  const int m = G->m;
  *used = false;
  const void *key[5] = {d_b, d_x, G->A, G->P, G->d_u};
  const bool same = G->gexec && G->gsteps == steps && !memcmp(key, G->gkey, sizeof(key));
  if (!same) {
    if (G->gexec) { hipGraphExecDestroy(G->gexec); G->gexec = nullptr; }
    if (!G->cap_stream && hipStreamCreateWithFlags(&G->cap_stream, hipStreamNonBlocking) != hipSuccess) {
      G->graph_broken = true;
      return NK_OK;
    }
    NK_HIP(hipStreamSynchronize(ctx->stream));
    hipStream_t user_stream = ctx->stream;
    ctx->stream = G->cap_stream;
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamBeginCapture(G->cap_stream, hipStreamCaptureModeThreadLocal);
    int st = (e == hipSuccess) ? NK_OK : NK_E_HIP;
    const uint64_t seq = ++G->cycle_seq;
    auto body = [&]() -> int {
      NK_TRY(nk_blas_copy_sumsq(ctx, n, d_b, G->V, G->d_ss));
      NK_LAUNCH(ctx, k_gmres_begin, dim3(1), dim3(64), G->d_ctl, G->d_ss, atol, rtol, 1, 1, G->d_g, G->d_s, m, G->h_pub_dev, seq);
      for (int k = 0; k < steps; ++k) NK_TRY(arnoldi_step_1r(G, k, k == steps - 1));
      NK_TRY(arnoldi_flush_1r(G, steps));
      NK_LAUNCH(ctx, k_backsolve, dim3(1), dim3(256), G->d_ctl, G->d_R, G->d_g, G->d_y, m, G->h_pub_dev, seq,
              (const uint64_t *)(ctx->peer.on ? nk_peer_err_ptr(ctx) : nullptr), nk_ss_fix{});
      NK_TRY(nk_blas_multiaxpy(ctx, n, m, G->V, ldv, G->d_y, 1.0, d_x, nullptr, nullptr, &G->d_ctl->k, G->d_s, true));
      return NK_OK;
    };
    if (st == NK_OK) st = body();
    if (e == hipSuccess) e = hipStreamEndCapture(G->cap_stream, &graph);
    ctx->stream = user_stream;
    if (st == NK_OK && e == hipSuccess && graph) e = hipGraphInstantiate(&G->gexec, graph, nullptr, nullptr, 0);
    if (graph) hipGraphDestroy(graph);
    if (st != NK_OK || e != hipSuccess || !G->gexec) {
      G->gexec = nullptr;
      G->graph_broken = true;
      (void)hipGetLastError();
      return NK_OK;  // the caller falls through to plain launches
    }
    memcpy(G->gkey, key, sizeof(key));
    G->gsteps = steps;
  }
  NK_HIP(hipGraphLaunch(G->gexec, ctx->stream));
  NK_HIP(hipStreamSynchronize(ctx->stream));
  volatile nk_gmres_pub *pub = G->h_pub;
  nk_gmres_info inf;
  memset(&inf, 0, sizeof(inf));
  inf.iters = pub->k;
  inf.rnorm0 = pub->rnorm0;
  inf.rnorm = pub->rnorm;
  inf.converged = pub->converged;
  inf.failed = pub->failed;
  ctx->stats.gmres_iters += inf.iters;
  if (info) *info = inf;
  *used = true;
  return NK_OK;
}

static int gmres_solve_once(nk_gmres *G, const double *d_b, double *d_x, int use_x0, double atol, double rtol, int maxiter,
                            int fixed_iters, nk_gmres_info *info);
// The resident matrix-powers kernel needs all its workgroups on the chip at once; when something else holds compute units for
// longer than its time-out (another process, a long kernel of the caller's on a second stream) a launch gives up, the columns of
// that block are garbage and the solve reports NK_E_HIP. The plan is then off for good — and a solve that started from x = 0 (every
// Newton step's) is simply run again on the streaming kernel instead of handing the failure to the caller.
int nk_gmres_solve_dev(nk_gmres *G, const double *d_b, double *d_x, int use_x0, double atol, double rtol,
                       int maxiter, int fixed_iters, nk_gmres_info *info) {
  // (a plan that timed out in an earlier solve comes back with this one: nk_powers.hip)
  if (G->op_kind == 1 && G->A) nk_csr_powers_rearm(G->A);
  if (G->op_kind == 2 && G->P) nk_problem_powers_rearm(G->P);
  const bool had_plan = (G->op_kind == 1 && G->A && nk_csr_powers_ready(G->A)) ||
                        (G->op_kind == 2 && G->P && G->P->kind == NK_PROBLEM_BRATU2D && nk_problem_powers_ready(G->P));
  // a warm start is kept aside while a resident plan is in play: a torn launch leaves garbage in x
  if (had_plan && use_x0) {
    if (!G->x0_keep) NK_TRY(nk_dev_alloc(&G->x0_keep, (size_t)G->n + 1));
    NK_HIP(hipMemcpyAsync(G->x0_keep, d_x, G->n * sizeof(double), hipMemcpyDeviceToDevice, G->ctx->stream));
  }
  // (a preloaded right-hand side — column 0 and its ‖b‖² partials, nk_gmres_preloaded_rhs — survives a torn attempt: nothing writes
  //  column 0. The rerun takes it again, so that it sums ‖b‖² in the order an untorn solve does: k_copy_sumsq pairs the entries
  //  differently, a last-bit difference in β now and then — found in round 6 when a trajectory changed and
  //  tests/test_gpu_powers.py's three-strikes case, which compares hashes, stopped being lucky)
  const auto pre_keep = G->pre;
  const int rc = gmres_solve_once(G, d_b, d_x, use_x0, atol, rtol, maxiter, fixed_iters, info);
  if (rc == NK_E_HIP && had_plan) {
    const bool now = (G->op_kind == 1 && nk_csr_powers_ready(G->A)) || (G->op_kind == 2 && nk_problem_powers_ready(G->P));
    if (!now && hipStreamQuery(G->ctx->stream) != hipErrorUnknown) {   // the plan broke in this solve; the stream is alive
      hipStreamSynchronize(G->ctx->stream);
      if (use_x0) NK_HIP(hipMemcpyAsync(d_x, G->x0_keep, G->n * sizeof(double), hipMemcpyDeviceToDevice, G->ctx->stream));
      G->pre = pre_keep;
      return gmres_solve_once(G, d_b, d_x, use_x0, atol, rtol, maxiter, fixed_iters, info);
    }
  }
  return rc;
}
static bool gmres_can_preload(const nk_gmres *G) {
  static const bool off = (getenv("NK_PRELOADED_RHS") && atoi(getenv("NK_PRELOADED_RHS")) == 0) ||
                          (getenv("NK_SS_FUSED_BEGIN") && atoi(getenv("NK_SS_FUSED_BEGIN")) == 0) || getenv("NK_GMRES_GRAPH");
  return !off && nk_ctx_is_single(G->ctx) && G->ortho == NK_ORTHO_SSTEP && !G->lprec_kind && !G->normal && G->V != nullptr;
}
double *nk_gmres_rhs_column(nk_gmres *G) { return gmres_can_preload(G) ? G->V : nullptr; }
void nk_gmres_preloaded_rhs(nk_gmres *G, const double *d_b, const double *ss_partials, int grid) {
  G->pre.b = d_b; G->pre.ss = ss_partials; G->pre.grid = grid;
}
void nk_gmres_arm_fused_update(nk_gmres *G, const double *u_old, double *u_new, double usign, double *partials) {
  G->fu = nk_fused_update{};
  G->fu.u_old = u_old; G->fu.u_new = u_new; G->fu.usign = usign; G->fu.partials = partials;
  G->fu.armed = true;
}
bool nk_gmres_take_fused_update(nk_gmres *G, int *grid) {
  const bool done = G->fu.armed && G->fu.done;
  if (grid) *grid = G->fu.grid;
  G->fu.armed = false;
  G->fu.done = false;
  return done;
}
void nk_gmres_drop_ahead(nk_gmres *G) { G->ahead.valid = false; }
static bool gmres_graph_wanted() {
  static const bool want_graph = getenv("NK_GMRES_GRAPH") && atoi(getenv("NK_GMRES_GRAPH")) != 0;
  return want_graph;
}
int nk_gmres_begin_ahead(nk_gmres *G, const double *d_b, const double *ss_partials, int ss_grid, const double *bpart, int bnblk,
                         double atol, double rtol, int maxiter, int fixed_iters, nk_ss_begin_args *out, bool *done) {
  static const bool off = getenv("NK_BEGIN_AHEAD") && atoi(getenv("NK_BEGIN_AHEAD")) == 0;   // A/B switch
  nk_ctx *ctx = G->ctx;
  *done = false;
  G->ahead.valid = false;
  // exactly the solve whose begin needs nothing the host learns in between: fixed work in ONE cycle from a zero guess, the s-step
  // form on one rank with b in column 0, a CSR operator with the Newton basis on its own Gershgorin bounds, no preconditioner
  if (off || !(fixed_iters > 0 && fixed_iters <= G->m) || G->op_kind != 1 || !G->A || G->prec_kind || G->lprec_kind || G->normal ||
      G->shift != 0.0 || G->ortho != NK_ORTHO_SSTEP || !nk_ss_eligible(G) || !gmres_can_preload(G) || ss_partials == nullptr ||
      ss_grid <= 0 || bpart == nullptr || bnblk <= 0 || G->ss_basis == NK_SS_BASIS_MONOMIAL || G->ss_ival_user ||
      G->ss_force_break_cycle >= 0 || ctx->audit.on || gmres_graph_wanted())
    return NK_OK;
  double *dst = nk_csr_bounds_word(G->A);
  if (dst == nullptr) return NK_OK;
  NK_TRY(nk_ss_prepare_ahead(G, bpart, bnblk, dst));
  const uint64_t seq = ++G->cycle_seq;
  NK_TRY(nk_ss_begin_args_for(G, atol, rtol, 1, 1, seq, ss_partials, ss_grid, out));
  G->ahead.valid = true;
  G->ahead.b = d_b; G->ahead.val = nullptr;   // (the caller names the value array once its fill is enqueued: nk_gmres_ahead_values)
  G->ahead.atol = atol; G->ahead.rtol = rtol; G->ahead.maxiter = maxiter; G->ahead.fixed_iters = fixed_iters; G->ahead.seq = seq;
  *done = true;
  return NK_OK;
}
void nk_gmres_ahead_values(nk_gmres *G, const double *d_val) { G->ahead.val = d_val; }
static int gmres_solve_once(nk_gmres *G, const double *d_b, double *d_x, int use_x0, double atol, double rtol, int maxiter,
                            int fixed_iters, nk_gmres_info *info) {
  nk_ctx *ctx = G->ctx;
  const int64_t n = G->n, ldv = G->ldv;
  const int m = G->m;
  NK_REQUIRE(G->op_kind != 0, "GMRES has no operator");
  // this very solve's cycle begin has run already (nk_gmres_begin_ahead): same right-hand side, tolerances and work, same values
  const bool ahead = G->ahead.valid && !use_x0 && G->ahead.b == d_b && G->ahead.atol == atol && G->ahead.rtol == rtol &&
                     G->ahead.maxiter == maxiter && G->ahead.fixed_iters == fixed_iters && G->op_kind == 1 && G->A &&
                     G->ahead.val != nullptr && nk_csr_get_valstate(G->A).d_val == G->ahead.val && G->ortho == NK_ORTHO_SSTEP &&
                     !G->prec_kind && !G->lprec_kind && !G->normal && G->ahead.seq == G->cycle_seq && G->pre.b == d_b;
  G->ahead.valid = false;
  // the arena's time-out counter is cumulative per CONTEXT: a solver object created after a time-out (or used after another
  // object's solve saw one) starts from what the context has already reported — not from zero
  if (ctx->peer.on && ctx->peer_err_reported > G->peer_err_seen) G->peer_err_seen = ctx->peer_err_reported;
  {
    const bool want_graph = gmres_graph_wanted();
    if (want_graph && fixed_iters > 0 && fixed_iters <= m && !use_x0 && nk_ctx_is_single(ctx) && !ctx->prof.on &&
        !G->prec_kind && !G->lprec_kind && !G->normal && G->op_kind != 3 && use_dcgs2r(G) && !G->graph_broken) {
      bool used = false;
      NK_TRY(gmres_solve_graph(G, d_b, d_x, atol, rtol, fixed_iters, info, &used));
      if (used) return NK_OK;
    }
  }
  if (maxiter <= 0) maxiter = 300;
  const int cap = fixed_iters > 0 ? fixed_iters : maxiter;
  // a breakdown of the s-step form switches THIS solve to delayed CGS2; the object's choice comes back on every exit
  struct ortho_guard_t { nk_gmres *g; int o; ~ortho_guard_t() { g->ortho = o; } } ortho_guard{G, G->ortho};
  // Where the automatic choice (block size 0) does NOT take the s-step form, although the object asks for it:
  //  * a solve that stops on a tolerance under a preconditioner — such solves need a handful of iterations, every operator
  //    application past the column that meets the tolerance is a wasted V-cycle / triangular solve, and the sweeps the blocks
  //    save are small change next to the preconditioner (measured: the multigrid solves of configs C3 / C4 / C5 took 2× as long
  //    in blocks of ≥ 2) — the column-by-column form with its one-step run-ahead serves them;
  //  * the normal-form operator JᵀJ (+ λDᵀD): its monomial blocks square an already squared condition number.
  // An explicit block size is honoured everywhere; the fixed-work protocol always builds full blocks.
  if (G->ortho == NK_ORTHO_SSTEP && G->ss_s == 0 &&
      ((fixed_iters <= 0 && (G->prec_kind || G->lprec_kind)) || G->normal))
    G->ortho = NK_ORTHO_DCGS2;
  if (G->ortho == NK_ORTHO_SSTEP && nk_ss_eligible(G) && !ahead) NK_TRY(nk_ss_prepare(G));
  nk_gmres_info inf;
  memset(&inf, 0, sizeof(inf));
  // r0 = b − A x0, written straight into column 0 of the basis (un-normalised); zero initial guess: b → column 0 and
  // ‖b‖² in one pass, and the first solution update WRITES x = V y (no memset of x)
  bool have_ss = false, x_is_zero = false;
  static const bool fused_begin_off = getenv("NK_SS_FUSED_BEGIN") && atoi(getenv("NK_SS_FUSED_BEGIN")) == 0;   // A/B switch
  bool ss_from_partials = false;   // ‖b‖² is still per-workgroup partial sums (in ctx->d_partials, or where the producer of b left them)
  const double *ss_partials = ctx->d_partials;
  int ss_grid = 0;
  const auto pre = G->pre;
  G->pre = {};   // (one solve)
  // with a left preconditioner every residual that enters the basis is Pl⁻¹(b − A x): the norms of the solve are preconditioned
  auto residual_to_v0 = [&]() -> int {   // column 0 ← Pl⁻¹ (b − A x), x in the original space
    NK_TRY(op_apply_raw(G, d_x, G->r, nullptr, nullptr));
    if (G->lprec_kind) {
      NK_TRY(nk_blas_lincomb(ctx, n, 1.0, d_b, -1.0, G->r, G->r));
      return lprec_apply(G, G->r, G->V, nullptr);
    }
    return nk_blas_lincomb(ctx, n, 1.0, d_b, -1.0, G->r, G->V);
  };
  auto rhs_to_v0 = [&]() -> int {        // column 0 ← Pl⁻¹ b and its squared norm (zero initial guess)
    if (G->lprec_kind) {
      NK_TRY(lprec_apply(G, d_b, G->V, nullptr));
      return nk_blas_sumsq(ctx, n, G->V, G->d_ss);
    }
    if (!fused_begin_off && nk_ctx_is_single(ctx) && G->ortho == NK_ORTHO_SSTEP && nk_ss_eligible(G)) {
      // the s-step form's begin kernel reduces the partial sums itself (one rank)
      ss_from_partials = true;
      if (pre.b == d_b && pre.grid > 0 && gmres_can_preload(G)) {   // b is in column 0 already (nk_gmres_preloaded_rhs)
        ss_partials = pre.ss;
        ss_grid = pre.grid;
        return NK_OK;
      }
      return nk_blas_copy_sumsq_stage1(ctx, n, d_b, G->V, &ss_grid);
    }
    return nk_blas_copy_sumsq(ctx, n, d_b, G->V, G->d_ss);
  };
  if (ahead) {           // column 0, ‖b‖² and the cycle's begin: done (nk_gmres_begin_ahead)
    have_ss = true;
    x_is_zero = true;
  } else if (!use_x0) {
    NK_TRY(rhs_to_v0());
    have_ss = true;
    x_is_zero = true;
  } else {
    if (G->prec_kind) NK_FAIL(NK_E_UNSUPPORTED, "use_x0 with a right preconditioner is not supported");
    NK_TRY(residual_to_v0());
  }
  int first = 1;
  int total_iters = 0;
  volatile nk_gmres_pub *pub = G->h_pub;
  static const bool fused_update_off = getenv("NK_FUSED_UPDATE") && atoi(getenv("NK_FUSED_UPDATE")) == 0;   // A/B switch
  G->fu.done = false;
  for (;;) {
    G->fu.done = false;   // (a further cycle moves x again: an update fused into an earlier one is stale)
    if (!have_ss) NK_TRY(nk_blas_sumsq(ctx, n, G->V, G->d_ss));
    have_ss = false;
    const bool begun = ahead && first == 1;   // this cycle's begin has run ahead of the solve
    const uint64_t seq = begun ? G->cycle_seq : ++G->cycle_seq;
    if (begun) {
    } else if (G->ortho == NK_ORTHO_SSTEP && nk_ss_eligible(G)) {
      NK_TRY(nk_ss_begin_cycle(G, atol, rtol, fixed_iters > 0 ? 1 : 0, first, seq,
                               ss_from_partials ? ss_partials : nullptr, ss_grid));
    } else {
      if (ss_from_partials)   // (a breakdown switched the solve to the column form between rhs → column 0 and this cycle)
        NK_TRY(nk_blas_reduce_slots_allreduce(ctx, ss_partials, ss_grid, 1, G->d_ss, nullptr));
      NK_LAUNCH(ctx, k_gmres_begin, dim3(1), dim3(64), G->d_ctl, G->d_ss, atol, rtol,
                fixed_iters > 0 ? 1 : 0, first, G->d_g, G->d_s, m, G->h_pub_dev, seq);
    }
    ss_from_partials = false;
    first = 0;
    const bool x_is_zero_before = x_is_zero;
    bool ss_backsolved = false;
    const int steps = (cap - total_iters) < m ? (cap - total_iters) : m;
    // The cycle is enqueued without a host synchronisation; kernels after convergence return at once on the device flag.
    // The device also publishes a progress word (cycle sequence, columns closed, done) in coherent pinned memory, which the
    // host merely reads: when the solve can stop early (a tolerance is set) the host keeps at most `run_ahead` Arnoldi steps
    // in the queue ahead of the device and stops enqueueing as soon as `done` shows — otherwise every step after
    // convergence would still cost its no-op launches (≈ 70 per step with a multigrid V-cycle inside).
    if (G->ortho == NK_ORTHO_SSTEP && nk_ss_eligible(G)) {
      // s columns per block (nk_sstep.hip). With a tolerance set and one rank, the host stays one block ahead of the device
      // and stops enqueueing once the cycle is done; several ranks enqueue the whole cycle (every rank must issue the same
      // collectives), the kernels past convergence return on the device flag.
      std::function<bool(int)> wait_progress;
      if (fixed_iters <= 0 && G->run_ahead > 0 && nk_ctx_is_single(ctx))
        wait_progress = [&](int need) -> bool {
          bool is_done = false;
          auto ready = [&] {
            const uint64_t w = __atomic_load_n(&pub->progress, __ATOMIC_ACQUIRE);
            if ((w >> 16) != seq) return false;
            if (w & 1) { is_done = true; return true; }
            return (int)((w >> 1) & 0x7fff) >= need;
          };
          if (nk_spin_wait(ctx, ready, "GMRES progress") != NK_OK) return false;
          return !is_done;
        };
      G->ss_cycle_idx = inf.restarts;
      G->ss_grow = fixed_iters <= 0;
      NK_TRY(nk_ss_cycle(G, steps, wait_progress, &ss_backsolved));
    } else {
      const bool one_red = use_dcgs2r(G);
      const int ahead = (fixed_iters > 0 || G->run_ahead <= 0) ? 0 : (G->prec_kind == 3 ? 1 : G->run_ahead);
      // Every rank must enqueue the SAME number of steps (each carries collectives): the stop is therefore not "when this
      // host happens to see `done`" but a function of K, the number of columns closed when the device raised it — a value
      // all ranks compute identically. A host is never more than ahead + one_red steps past the step that closes column K,
      // so "enqueue exactly the steps below K + ahead + one_red" is reachable by all of them.
      bool stopped = false;
      int stop_at = steps;
      for (int k = 0; k < stop_at; ++k) {
        if (ahead > 0 && !stopped) {
          // column j closes in step j (plain forms) or in step j+1 (one-reduction form): wait until step k−ahead is done
          const int need = k - ahead - (one_red ? 1 : 0);
          auto ready = [&] {
            const uint64_t w = __atomic_load_n(&pub->progress, __ATOMIC_ACQUIRE);
            if ((w >> 16) != seq) return false;  // k_gmres_begin of this cycle has not run yet
            if (w & 1) {
              stopped = true;
              const int K = (int)((w >> 1) & 0x7fff);
              stop_at = K + ahead + (one_red ? 1 : 0) < steps ? K + ahead + (one_red ? 1 : 0) : steps;
              return true;
            }
            return (int)((w >> 1) & 0x7fff) >= need;
          };
          if (need >= 0) NK_TRY(nk_spin_wait(ctx, ready, "GMRES progress"));
          else (void)ready();  // nothing to wait for yet, but a raised `done` fixes the stop
          if (k >= stop_at) break;
        }
        NK_TRY(one_red ? arnoldi_step_1r(G, k, k == steps - 1) : arnoldi_step(G, k));
      }
      // (the flush is a no-op on the device once `done` is up; whether it is ENQUEUED must not depend on what this host saw)
      if (one_red && stop_at == steps) NK_TRY(arnoldi_flush_1r(G, steps));
    }
    // x += M⁻¹ V y  (coefficients y_j s_j on the un-normalised columns); the back-substitution also publishes the control
    // block's outcome to the host
    // (s-step form: the cycle's last scalar launch may have back-substituted already — nk_sstep.hip)
    if (!ss_backsolved)
      NK_LAUNCH(ctx, k_backsolve, dim3(1), dim3(256), G->d_ctl, G->d_R, G->d_g, G->d_y, m, G->h_pub_dev, seq,
                (const uint64_t *)(ctx->peer.on ? nk_peer_err_ptr(ctx) : nullptr),
                G->ortho == NK_ORTHO_SSTEP ? nk_ss_take_last_block(G) : nk_ss_fix{});
    if (!G->prec_kind) {
      // the Newton driver's update rides in this pass when the solve is ONE cycle from a zero guess (whatever its outcome: the
      // update is out of place, a failed solve's is never looked at)
      nk_fused_update *fu = (G->fu.armed && x_is_zero && cap <= m && !fused_update_off) ? &G->fu : nullptr;
      NK_TRY(nk_blas_multiaxpy(ctx, n, m, G->V, ldv, G->d_y, 1.0, d_x, nullptr, nullptr, &G->d_ctl->k, G->d_s, x_is_zero, fu));
    } else {
      NK_TRY(nk_blas_multiaxpy(ctx, n, m, G->V, ldv, G->d_y, 1.0, G->r, nullptr, nullptr, &G->d_ctl->k, G->d_s, true));
      NK_TRY(prec_apply(G, G->r, G->z, nullptr));
      if (x_is_zero) NK_TRY(nk_blas_copy(ctx, n, G->z, d_x));
      else NK_TRY(nk_blas_axpby(ctx, n, 1.0, G->z, 1.0, d_x));
    }
    x_is_zero = false;
    NK_TRY(nk_spin_wait(ctx, [&] { return __atomic_load_n(&pub->end_seq, __ATOMIC_ACQUIRE) == seq; }, "the end of a GMRES cycle"));
    if (G->op_kind == 1 && G->A) NK_TRY(nk_csr_powers_check(G->A));
    if (G->op_kind == 2 && G->P) NK_TRY(nk_problem_powers_check(G->P));
    if (pub->pad != G->peer_err_seen) {
      // the arena's counter is cumulative and sticky: a time-out fails THIS solve (its reductions / halos are not valid);
      // the next one starts from the value seen here, so one transient stall does not condemn the context for good
      const int fresh = (int)pub->pad - G->peer_err_seen;
      G->peer_err_seen = (int)pub->pad;
      ctx->peer_err_reported = G->peer_err_seen;
      NK_FAIL(NK_E_COMM, "%d peer-mapped collective(s) timed out (a rank stalled beyond NK_PEER_TIMEOUT_MS or died): the "
                         "reductions / halos of this solve are not valid", fresh);
    }
    nk_gmres_ctl c;
    memset(&c, 0, sizeof(c));
    c.k = pub->k;
    c.converged = pub->converged;
    c.failed = pub->failed;
    c.rnorm0 = pub->rnorm0;
    c.rnorm = pub->rnorm;
    c.done = (c.converged || c.failed) ? 1 : 0;
    *G->h_ctl = c;
    total_iters += c.k;
    inf.rnorm0 = c.rnorm0;
    inf.rnorm = c.rnorm;
    inf.converged = c.converged;
    inf.failed = c.failed;
    if (c.failed == 2 && G->ortho == NK_ORTHO_SSTEP) {
      // a block lost rank numerically (Cholesky breakdown): this cycle left x as it was (y = 0) and its columns do not
      // count. The REST OF THIS SOLVE runs column by column from the same restart loop — it forms r = b − A x through the raw
      // operator, so right preconditioners are fine, and the tolerance stays the one fixed by the first cycle; the next solve
      // tries the s-step form again (ortho_guard restores it).
      G->ss_breakdowns++;
      {  // automatic block size: retry with a narrower block (15 → 8 → 4; the object keeps it); otherwise, or once that is
         // exhausted, column by column
        const int s_now = nk_ss_block_size(G);
        if (G->ss_s == 0 && s_now > 4 && G->ss_force_break_cycle < 0) G->ss_s_cap = s_now > 8 ? 8 : 4;
        else G->ortho = NK_ORTHO_DCGS2;
      }
      total_iters -= c.k;
      inf.failed = 0;
      first = 2;
      if (x_is_zero_before) {
        NK_TRY(rhs_to_v0());
        have_ss = true;
        x_is_zero = true;
      } else {
        NK_TRY(residual_to_v0());
      }
      continue;
    }
    if (c.failed || c.converged || total_iters >= cap || steps == 0) break;
    inf.restarts++;
    // restart: r = Pl⁻¹(b − A x) into column 0 (A x directly: x lives in the original space)
    NK_TRY(residual_to_v0());
  }
  inf.iters = total_iters;
  ctx->stats.gmres_iters += total_iters;
  if (info) *info = inf;
  return NK_OK;
}

extern "C" int nk_gmres_solve(nk_gmres *G, const double *b, double *x, int memspace, int use_x0, double atol,
                              double rtol, int maxiter, int fixed_iters, nk_gmres_info *info) {
  NK_REQUIRE(G && b && x, "NULL argument");
  NK_HIP(hipSetDevice(G->ctx->device));
  if (memspace == NK_DEVICE) return nk_gmres_solve_dev(G, b, x, use_x0, atol, rtol, maxiter, fixed_iters, info);
  nk_ctx *ctx = G->ctx;
  if (!G->d_b) NK_TRY(nk_dev_alloc(&G->d_b, (size_t)G->n + 1));
  if (!G->d_x) NK_TRY(nk_dev_alloc(&G->d_x, (size_t)G->n + 1));
  NK_HIP(hipMemcpyAsync(G->d_b, b, G->n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  if (use_x0) NK_HIP(hipMemcpyAsync(G->d_x, x, G->n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  NK_TRY(nk_gmres_solve_dev(G, G->d_b, G->d_x, use_x0, atol, rtol, maxiter, fixed_iters, info));
  NK_HIP(hipMemcpyAsync(x, G->d_x, G->n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  NK_HIP(hipStreamSynchronize(ctx->stream));
  return NK_OK;
}
