// s-step (communication-avoiding) Arnoldi for GMRES(m): the basis grows s columns at a time.
//
//   matrix powers   W_j = (A − θ_j I) W_{j−1} / σ, W_{−1} = v_k    (s operator applications; σ a power of two)
//                   Newton basis: θ_j = Leja-ordered Chebyshev points of a real interval that bounds the operator's spectrum
//                   (Gershgorin discs of a CSR operator, the stencil's closed form, or the caller's bounds) — the block stays
//                   well conditioned up to s = 16 (κ ≈ 1e4–1e5 at s = 15 where the monomial basis, θ = 0, breaks down at
//                   s ≈ 10: Bai, Hu, Reichel, "A Newton basis GMRES implementation"; Hoemmen's thesis §7); the shift rides
//                   in the operator kernel's row epilogue. Without bounds: monomial basis, s = 6.
//   block CGS, Pythagorean form, twice (BCGS-PIP2):
//     sweep A   [V_k W]ᵀ W                       → C₁ = V_kᵀW, G₁ = WᵀW          one pass over k + s columns
//     tail 1    R₁ᵀR₁ = G₁ − C₁ᵀC₁ (Cholesky)
//     sweep B   Q₁ = (W − V_k C₁) R₁⁻¹ in place, and [V_k Q₁]ᵀ Q₁ of the result   one pass (read k + s, write s)
//     tail 2    R₂ᵀR₂ = G₂ − C₂ᵀC₂ ; C = C₁ + C₂R₁ ; R = R₂R₁ ; the s new Hessenberg columns from (C, R) and the old ones ;
//               Givens rotations, residual norms, stopping test
//     sweep C   Q = (Q₁ − V_k C₂) R₂⁻¹ in place                                   one pass
//
// Three sweeps over the basis per s columns instead of two per column (delayed CGS2, nk_blas.hip): the basis traffic of a
// restart cycle falls by 2s/3. The sweeps are HBM-bound; the skinny Gram products [V_k W]ᵀW — (k+s)·s accumulators, far more
// than a thread can hold — run on the FP64 matrix cores (v_mfma_f64_16x16x4: rows are the reduction dimension, so a wavefront
// keeps a 16×16 tile of sums in 4 VGPR pairs per lane and needs no cross-lane reduction until the very end).
//
// Data movement of a sweep: a workgroup takes tiles of 256 rows (one row per thread, coalesced 2 KB per column per
// workgroup), applies the update in registers while the columns stream past, parks the tile in LDS (column-major, odd pitch:
// the MFMA operand reads — 16 columns × 4 rows per instruction — are conflict-free), and the four wavefronts accumulate the
// Gram tile from LDS. Workgroups are persistent (2 per CU), so the partial sums leave the chip once per launch.
#include "nk_internal.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

constexpr int SS_R = 256;        // rows per tile = threads per workgroup
constexpr int SS_P = SS_R + 1;   // LDS pitch in doubles (odd)
constexpr int SS_MTMAX = 5;      // Gram tiles of 16 rows: k + s ≤ 80
constexpr int SS_SMAX = NK_SS_SMAX;   // one 16-wide matrix-core tile of new columns
constexpr int SS_TH = NK_SS_TH;   // scal[SS_TH + j] = θ_j; scal[0..5): first-application scale, 1/σ, σ, carried σ estimate, Newton flag
constexpr int SS_MAX_WG_PER_CU = 4;
typedef double ss_d4 __attribute__((ext_vector_type(4)));
typedef unsigned int ss_u2 __attribute__((ext_vector_type(2)));
// development: phase time stamps of the scalar work (nk_ss_debug_stamps, tools/ss_stamps.py) — in builds with -DNK_SS_STAMPS only
// (make STAMPS=1): even switched off at run time a stamp is a load of the pointer, i.e. a memory round trip of ≈ 0.7 µs in the
// middle of a latency-bound workgroup — thirteen of them sat in the cycle's last launch through round 4
__device__ unsigned long long *g_ss_stamp = nullptr;
__device__ int g_ss_stamp_bank = 0;   // 16 stamps per kind of scalar launch (0: first factorisation only, 1: both, 2: the cycle's last)
#ifdef NK_SS_STAMPS
// (into LDS: a stamp is one s_memrealtime and one DS write; k_ss_job's last workgroup copies them out behind its work)
__shared__ unsigned long long s_ss_stamps[16];
#define SS_STAMP(i) do { if (threadIdx.x == 0) s_ss_stamps[i] = wall_clock64(); } while (0)
#else
#define SS_STAMP(i) do { } while (0)
#endif

// ============================================================================= the scalar work of a block (one workgroup)
__device__ __forceinline__ void ss_pub_progress(nk_gmres_pub *pub, uint64_t seq, int k, int done) {
  if (pub != nullptr)
    __hip_atomic_store(&pub->progress, (seq << 16) | ((uint64_t)k << 1) | (uint64_t)(done ? 1 : 0), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}
constexpr int SS_KMAX = 16 * SS_MTMAX;  // k + s ≤ 80

// what the scalar work reads and writes in global memory
struct ss_tail_args {
  nk_gmres_ctl *ctl;
  const double *red;   // the reduced block [V_kᵀX ; XᵀX], (k + s) × s
  double *sc;          // scales of un-normalised columns (only column 0: 1/β)
  double *C1, *R1;     // pass 1's factors, kept for pass 2
  double *C2, *R2;     // pass 2's factors, left for the workgroup that derives the Hessenberg columns inside sweep C
  double *H;           // un-rotated Hessenberg columns, row-major with pitch m
  int m;
  double *Rg, *cs, *sn, *g, *scal;
  nk_gmres_pub *pub;
  uint64_t seq;
  // implicit second pass: the earlier blocks of this cycle whose stored columns were left at their first pass (their C₂, R₂
  // carry stored inner products into true coordinates and true coefficients back onto the stored columns), and the block
  // whose last STORED column started this block's matrix powers (usb = 0: the powers started from a true basis vector)
  nk_ss_fix fix;
  int uk0, usb;
  const double *uC2, *uR2;
  double *Wi, *D;      // this block's R₂⁻¹ and C₂R₂⁻¹, written next to its Hessenberg columns when it is left at its first pass
  double ptol;         // rank-loss bar of the factorisation: a pivot ≤ ptol·(XᵀX)_aa is a breakdown (1e-12 on every path)
};
// LDS arrays of the scalar work, carved from one dynamic block
struct ss_ws {
  double *Ct, *U, *Ri, *Rm, *Sm, *R1s, *Gd, *Fr, *Fx, *F, *NC, *Hs, *scs, *ssn, *sg, *uu;
  int *ok;
  int o0;   // offset of the workspace (= of Ct) in the LDS block it was carved from, in doubles (ss_ws_off)
};
// offsets of the arrays that are addressed as LDS offsets (asynchronous loads, the back-substitution), relative to ss_ws::o0
struct ss_ws_offs { int Rm, R1s, Fx, F, Hs, scs, ssn, sg, uu; };
__host__ __device__ inline ss_ws_offs ss_ws_off(int k, int s) {
  ss_ws_offs o;
  int b = 2 * k * s + SS_SMAX * SS_SMAX;   // Ct, U, Ri
  o.Rm = b; b += 2 * SS_SMAX * SS_SMAX;    // Rm, Sm
  o.R1s = b; b += SS_SMAX * SS_SMAX + SS_SMAX + SS_SMAX * (SS_SMAX + 1);   // R1s, Gd, Fr
  o.Fx = b; b += SS_SMAX * (SS_SMAX + 1) + 2;   // Fx, ok
  o.F = b; b += 2 * (k + s) * s;           // F, NC
  o.Hs = b; b += k * (k > 1 ? k - 1 : 1);
  o.scs = b; b += k + s;
  o.ssn = b; b += k + s;
  o.sg = b; b += k + s + 1;
  o.uu = b;
  return o;
}
constexpr int SS_SS = SS_SMAX * SS_SMAX;
__host__ __device__ inline size_t ss_ws_doubles(int k, int s, bool hess) {
  size_t d = (size_t)2 * k * s + 4 * SS_SS + SS_SMAX + 2 + 2 * SS_SMAX * (SS_SMAX + 1);
  if (hess) d += (size_t)2 * (k + s) * s + (size_t)k * (k > 1 ? k - 1 : 1) + 4 * (size_t)(k + s) + 1;
  return d;
}
__device__ inline ss_ws ss_ws_carve(double *b, int k, int s, bool hess, int o0 = 0) {
  ss_ws w;
  w.o0 = o0;
  w.Ct = b; b += k * s;
  w.U = b; b += k * s;
  w.Ri = b; b += SS_SS;
  w.Rm = b; b += SS_SS;
  w.Sm = b; b += SS_SS;
  w.R1s = b; b += SS_SS;
  w.Gd = b; b += SS_SMAX;
  w.Fr = b; b += SS_SMAX * (SS_SMAX + 1);
  w.Fx = b; b += SS_SMAX * (SS_SMAX + 1);
  w.ok = reinterpret_cast<int *>(b); b += 2;
  w.F = w.NC = w.Hs = w.scs = w.ssn = w.sg = w.uu = nullptr;
  if (hess) {
    w.F = b; b += (k + s) * s;
    w.NC = b; b += (k + s) * s;
    w.Hs = b; b += k * (k > 1 ? k - 1 : 1);
    w.scs = b; b += k + s;
    w.ssn = b; b += k + s;
    w.sg = b; b += k + s + 1;
    w.uu = b;
  }
  return w;
}

// From the reduced block [V_kᵀX ; XᵀX] (X = the s columns behind V_k; `sc` un-normalised-column scales): the true
// coefficients Ct = diag(sc)·V_kᵀX, the Cholesky factor R of XᵀX − CtᵀCt (Pythagorean form of ‖X − V Ct‖), R⁻¹, and the
// coefficients U = diag(sc)·Ct the update takes off the stored columns. Returns false when the block lost rank numerically:
// a pivot that is not positive RELATIVE to the column's own squared norm — d ≤ ptol·(XᵀX)_aa, ptol = 1e-12: the column lies in the span of
// the others to 1e-6, the block's condition number is beyond 1e6, and the Hessenberg columns recovered through R would carry
// errors above 1e-10 (a pivot near ε (XᵀX)_aa is rounding noise altogether) — or is not finite.
// Every workgroup that runs it on the same `red` reaches the same verdict.
// The s × s factorisation and inverse are latency, not work: a dependent FP64 operation costs a lone wavefront ≈ 10 cycles, and
// the first version (one wavefront, lane b owning column b, v_readlane broadcasts: ≈ 1300 serial operations) took 12 µs. Here
// the 256 threads own one entry (a, b) of a fixed 16 × 16 frame each (identity beyond s): right-looking S = Uᵀ D U — 16 steps
// of three LDS reads, one refined v_rcp and one FMA per thread —, R = D^½ U, then R⁻¹ row by row from the bottom (row a: 16
// threads, one ≤ 15-term dot product each). ≈ 35 dependent steps instead of 1300.
constexpr int SS_FP = SS_SMAX + 1;   // pitch of the frame in LDS
__device__ __forceinline__ double ss_rcp(double d) {   // 1/d to the last bit or two: v_rcp_f64 + two Newton steps
  double r = __builtin_amdgcn_rcp(d);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  return r;
}
// Implicit second pass. A block (k0, sp) whose stored columns S_b were left at their first pass relates to the true basis by
// S_b = V_true[:k0] C₂ + Q_b R₂. With Wi = R₂⁻¹ and D = C₂ Wi (left in global memory by the workgroup that derived the block's
// Hessenberg columns: ss_fix_prepare) both directions are plain products — no dependent steps in the reductions' critical path:
//   stored inner products → true:   Q_bᵀX = Wiᵀ (S_bᵀX) − Dᵀ (V_true[:k0]ᵀX)       (rows above the block already true: blocks in order)
//   true coefficients → stored:     W[:k0] −= D W_b ;  W_b ← Wi W_b                 (blocks last first)
// P / Wm: k × sb, row-major, rows = basis columns. tmp: sp × sb scratch.
// (thread maps: (t >> 4, t & 15) — a division by a run-time block width is ≈ 40 instructions of a single wavefront's issue time)
__device__ void ss_fix_to_true(int sb, int k0, int sp, const double *sD, const double *sWi, double *P) {
  const int t = threadIdx.x;
  const int a = t >> 4, c = t & 15;
  const bool own = a < sp && c < sb;
  double val = 0.0;
  if (own) {
    for (int p = 0; p <= a; ++p) val = __builtin_fma(sWi[p * sp + a], P[(k0 + p) * sb + c], val);     // (Wiᵀ P_b)[a][c]
    for (int i = 0; i < k0; ++i) val = __builtin_fma(-sD[i * sp + a], P[i * sb + c], val);
  }
  __syncthreads();
  if (own) P[(k0 + a) * sb + c] = val;
  __syncthreads();
}
__device__ void ss_fix_to_stored(int sb, int k0, int sp, const double *sD, const double *sWi, double *Wm) {
  const int t = threadIdx.x;
  const int a = t >> 4, c = t & 15;
  if (c < sb) {
    for (int i = a; i < k0; i += 16) {      // W[:k0] −= D W_b (the OLD W_b)
      double v = Wm[i * sb + c];
      for (int q = 0; q < sp; ++q) v = __builtin_fma(-sD[i * sp + q], Wm[(k0 + q) * sb + c], v);
      Wm[i * sb + c] = v;
    }
  }
  const bool own = a < sp && c < sb;
  double val = 0.0;
  if (own)
    for (int p = a; p < sp; ++p) val = __builtin_fma(sWi[a * sp + p], Wm[(k0 + p) * sb + c], val);    // (Wi W_b)[a][c]
  __syncthreads();
  if (own) Wm[(k0 + a) * sb + c] = val;
  __syncthreads();
}
// The factors of the blocks left at their first pass as the scalar work reads them: in LDS, one slot per block (the struct lives in
// LDS as well — indexing a kernel argument by a run-time block number would put it in scratch memory). A slot is filled from global
// memory (ss_fixc_request: all slots in ONE round trip, together with the reduced block) or, for the block whose second
// factorisation has just been done by this workgroup, by ss_fix_prepare.
// (offsets from the workgroup's dynamic LDS block, not pointers: a pointer read back from LDS is a generic one, and every access
//  through it a FLAT instruction — measured 5 µs in the back-substitution's 60 steps)
struct ss_fixc {
  int n;
  int k0[NK_SS_NFIX], sb[NK_SS_NFIX];
  int oD[NK_SS_NFIX], oWi[NK_SS_NFIX];
};
__host__ __device__ inline size_t ss_fixc_doubles(const nk_ss_fix &fix) {
  size_t d = 0;
#pragma unroll
  for (int q = 0; q < NK_SS_NFIX; ++q)
    if (q < fix.n) d += (size_t)fix.k0[q] * fix.sb[q] + (size_t)fix.sb[q] * fix.sb[q];
  return d;
}
// carve the slots out of `b` (uniform; thread 0 writes the struct, the caller's next barrier publishes it) and request the
// slots' contents — all but the last `skip_last` blocks' (those are computed here)
__device__ inline void ss_fixc_request(ss_fixc *fc, const nk_ss_fix &fix, double *lds, int off, int skip_last, bool c2r2 = false) {
  const int t = threadIdx.x;
#pragma unroll
  for (int q = 0; q < NK_SS_NFIX; ++q) {
    if (q < fix.n) {
      const int oD = off; off += fix.k0[q] * fix.sb[q];
      const int oWi = off; off += fix.sb[q] * fix.sb[q];
      double *sD = lds + oD, *sWi = lds + oWi;
      if (t == 0) { fc->k0[q] = fix.k0[q]; fc->sb[q] = fix.sb[q]; fc->oD[q] = oD; fc->oWi[q] = oWi; }
      if (q < fix.n - skip_last) {
        const double *gD = c2r2 ? fix.C2[q] : fix.D[q], *gWi = c2r2 ? fix.R2[q] : fix.Wi[q];
        for (int e = t; e < fix.k0[q] * fix.sb[q]; e += SS_R) sD[e] = gD[e];
        if (t < fix.sb[q] * fix.sb[q]) sWi[t] = gWi[t];
      }
    }
  }
  if (t == 0) fc->n = fix.n;
}
// Wi = R₂⁻¹ and D = C₂ Wi of a block left at its first pass (k0 rows above it, sp columns), from its pass-2 factors in LDS
// (Ct: C₂, k0 × sp; Rm: R₂, sp × sp upper) into global memory. Column j of Wi by back substitution, one lane per column.
__device__ void ss_fix_prepare(int k0, int sp, const double *Ct, const double *Rm, double *scratch /* sp × sp */, double *Wig,
                               double *Dg, bool to_lds = false, double *sWiout = nullptr, double *sDout = nullptr) {
  const int t = threadIdx.x;
  if (t < sp) {
    const int j = t;
    for (int a = sp - 1; a >= 0; --a) {
      double v = (a == j) ? 1.0 : 0.0;
      if (a <= j) {
        for (int p = a + 1; p <= j; ++p) v = __builtin_fma(-Rm[a * sp + p], scratch[p * sp + j], v);
        v /= Rm[a * sp + a];
      } else {
        v = 0.0;
      }
      scratch[a * sp + j] = v;
    }
  }
  __syncthreads();
  if (t < sp * sp) {
    Wig[t] = scratch[t];
    if (to_lds) sWiout[t] = scratch[t];
  }
  for (int e = t; e < k0 * sp; e += SS_R) {
    const int i = e / sp, a = e % sp;
    double v = 0.0;
    for (int p = 0; p <= a; ++p) v = __builtin_fma(Ct[i * sp + p], scratch[p * sp + a], v);
    Dg[e] = v;
    if (to_lds) sDout[e] = v;
  }
  __syncthreads();
}
__device__ bool ss_factor(int k, int sb, const double *__restrict__ red, const double *__restrict__ sc, const ss_ws &w,
                          const ss_fixc &fix, int nfix, double ptol, double *lds) {
  const int t = threadIdx.x;
  double *Ct = w.Ct, *Rm = w.Rm, *Ri = w.Ri, *Sm = w.Sm;
  double *F = w.Fr;   // 16 × 17 frame: the factor in progress
  const int ta4 = t >> 4, tc = t & 15;
  if (tc < sb) {
    for (int r = ta4; r < k; r += 16) {
      const int e = r * sb + tc;
      const double scj = sc[r], c = scj * red[e];
      Ct[e] = c;
      if (nfix == 0) w.U[e] = scj * c;
    }
  }
  if (t == 0) *w.ok = 1;
  __syncthreads();
  if (nfix > 0) {   // (uniform) stored → true coordinates for the factorisation; true → stored coefficients for the update
    for (int bq = 0; bq < nfix; ++bq) ss_fix_to_true(sb, fix.k0[bq], fix.sb[bq], lds + fix.oD[bq], lds + fix.oWi[bq], Ct);
    for (int e = t; e < k * sb; e += SS_R) w.U[e] = Ct[e];
    __syncthreads();
    for (int bq = nfix - 1; bq >= 0; --bq) ss_fix_to_stored(sb, fix.k0[bq], fix.sb[bq], lds + fix.oD[bq], lds + fix.oWi[bq], w.U);
    if (tc < sb)
      for (int r = ta4; r < k; r += 16) w.U[r * sb + tc] *= sc[r];
    __syncthreads();
  }
  SS_STAMP(5);
  if (ta4 < sb && tc < sb) {
    const int a = ta4, b = tc;
    double s0 = red[(size_t)(k + a) * sb + b], s1 = 0.0;
    if (a == b) w.Gd[a] = s0;
    int j = 0;
    for (; j + 1 < k; j += 2) {   // two independent chains
      s0 = __builtin_fma(-Ct[j * sb + a], Ct[j * sb + b], s0);
      s1 = __builtin_fma(-Ct[(j + 1) * sb + a], Ct[(j + 1) * sb + b], s1);
    }
    if (j < k) s0 = __builtin_fma(-Ct[j * sb + a], Ct[j * sb + b], s0);
    Sm[a * sb + b] = s0 + s1;
  }
  __syncthreads();
  const int a = (t >> 4) & (SS_SMAX - 1), b = t & (SS_SMAX - 1);   // (SS_R = 256: one entry per thread)
  const bool in = a < sb && b < sb;
  double val = in ? 0.5 * (Sm[in ? a * sb + b : 0] + Sm[in ? b * sb + a : 0]) : ((a == b) ? 1.0 : 0.0);
  F[a * SS_FP + b] = val;
  __syncthreads();
  SS_STAMP(6);
  // Two pivots per barrier: every thread re-derives what step p would have left in row p + 1 (its diagonal and the two entries
  // it needs: the same expressions the owning threads evaluate — bit-identical to one pivot per step) and applies both rank-1
  // updates behind one round of LDS reads. Eight barrier-separated steps instead of sixteen (≈ 0.24 µs each).
#pragma unroll 1
  for (int p = 0; p < SS_SMAX; p += 2) {   // after a step: rows p, p + 1 hold T_pb = D_p U_pb (b ≥ p), the trailing block is updated
    const double d0 = F[p * SS_FP + p], f0a = F[p * SS_FP + a], f0b = F[p * SS_FP + b], f01 = F[p * SS_FP + p + 1];
    const double d1r = F[(p + 1) * SS_FP + p + 1], f1ar = F[(p + 1) * SS_FP + a], f1br = F[(p + 1) * SS_FP + b];
    const double l01 = f01 * ss_rcp(d0);
    const double d1 = __builtin_fma(-l01, f01, d1r);
    const double f1a = __builtin_fma(-l01, f0a, f1ar), f1b = __builtin_fma(-l01, f0b, f1br);
    if (t == 0) {
      if (p < sb && (!(d0 > ptol * w.Gd[p]) || isinf(d0))) *w.ok = 0;
      if (p + 1 < sb && (!(d1 > ptol * w.Gd[p + 1]) || isinf(d1))) *w.ok = 0;
    }
    if (a > p && b >= a) {
      val = __builtin_fma(-(f0a * ss_rcp(d0)), f0b, val);
      if (a > p + 1) {   // rows ≥ p + 2: nobody reads them before the barrier
        val = __builtin_fma(-(f1a * ss_rcp(d1)), f1b, val);
        F[a * SS_FP + b] = val;
      }
    }
    __syncthreads();
    // Row p + 1 — final after pivot p — is READ by every thread in the interval above (d1r, f1ar, f1br: its values BEFORE pivot p),
    // so its owners store it BEHIND the barrier: the next interval reads rows p + 2 and p + 3 only. (Round 5 stored it in front of
    // the barrier: a write-after-read race between wavefronts — harmless while the four wavefronts run in step, i.e. alone on
    // the device; beside another process's kernels a delayed wavefront read the updated row and applied pivot p twice: the
    // "Cholesky breakdowns a deterministic algorithm cannot hit" and the silently wrong Gram factors of
    // profiles/r05_m_shared_device.txt, found by tools/shared_device_probe.py --audit, profiles/r06_a_shared_device_root_cause.md.)
    if (a == p + 1 && b >= a) F[a * SS_FP + b] = val;
  }
  __syncthreads();   // (row 15's store in front of the diagonal reads below)
  // R = D^½ U (upper; zero below the diagonal). The sweeps apply R⁻¹ by forward substitution, row by row of the tile
  // (q_c = (w_c − Σ_{cc<c} q_cc R_cc,c) / R_cc: the same 120 multiply-adds as a product with an explicit inverse, whose
  // 16 dependent steps were the longest phase of this routine): `Ri` carries R in that form — reciprocal diagonal.
  const double da = F[a * SS_FP + a];
  const double isq = 1.0 / sqrt(da);
  const double rab = (b >= a) ? val * isq : 0.0;     // val = T_ab for b ≥ a (row a was final after step a − 1)
  if (in) {
    Rm[a * sb + b] = rab;
    Ri[a * sb + b] = (b == a) ? isq : rab;
  }
  __syncthreads();
  return *w.ok != 0;
}

// Implicit second pass: how far pass 1 left the block from orthonormal — max(|C₂|, |R₂ − I|) over pass 2's factors (in LDS:
// w.Ct k × sb, w.Rm sb × sb). Uniform verdict: true = the block may stay at its first pass.
__device__ bool ss_first_pass_departure_ok(int k, int sb, const ss_ws &w, double bar) {
  const int t = threadIdx.x;
  double m = 0.0;
  for (int e = t; e < k * sb; e += SS_R) m = fmax(m, fabs(w.Ct[e]));
  {
    const int a = t >> 4, c = t & 15;
    if (a < sb && c < sb) m = fmax(m, fabs(w.Rm[a * sb + c] - (a == c ? 1.0 : 0.0)));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
  __syncthreads();
  if ((t & 63) == 0) w.Gd[t >> 6] = m;
  __syncthreads();
  m = fmax(fmax(w.Gd[0], w.Gd[1]), fmax(w.Gd[2], w.Gd[3]));
  __syncthreads();
  return m <= bar && m == m;
}
// after pass 1 (one workgroup): C₁ and R₁ are kept for pass 2; σ estimate for the next cycle from ‖A v‖ of the first block
__device__ void ss_keep_pass1(int k, int sb, const ss_ws &w, const ss_tail_args &ta, const double *red) {
  const int t = threadIdx.x;
  for (int e = t; e < k * sb; e += SS_R) ta.C1[e] = w.Ct[e];
  if (t < sb * sb) ta.R1[t] = w.Rm[t];
  if (t == 0 && k == 1 && ta.scal[4] == 0.0) {  // ‖A v₁‖ = σ·√(XᵀX)₀₀: the scale of the next cycle's monomial basis, rounded to a power of two
    const double est = ta.scal[2] * sqrt(red[(size_t)k * sb]);
    if (est > 0.0 && !isinf(est)) ta.scal[3] = exp2(rint(log2(est)));
  }
}
__device__ void ss_fail(nk_gmres_ctl *ctl, nk_gmres_pub *pub, uint64_t seq) {
  if (threadIdx.x == 0) {
    ctl->failed = 2;
    ctl->done = 1;
    ctl->pad1 = 1;
    ss_pub_progress(pub, seq, ctl->k, 1);
  }
}
__device__ void ss_fail(const ss_tail_args &ta) { ss_fail(ta.ctl, ta.pub, ta.seq); }

// after pass 2 (one workgroup; ss_factor has run on pass 2's block): C = C₁ + C₂R₁, R = R₂R₁; the s new Hessenberg columns;
// Givens rotations, residual norms, stopping test.
// Coordinates in the basis [V_k Q] (K = k + s): X_j (column j of the block, j = 0..s−1) = F[:, j] = [C_j ; R_j], and
//   A v_k = σ X_0 + θ_0 v_k ;  A X_{j−1} = σ X_j + θ_j X_{j−1} ;
//   q_j = (X_{j−1} − V_k C_{j−1} − Σ_{i<j} q_i R_{i,j−1}) / R_{j−1,j−1}   (j = 1..s−1)     (θ = 0: the monomial basis)
// so the coordinates of A q_j follow from those of A v_1..A v_{k−1} (the old Hessenberg columns), A v_k and A q_1..A q_{j−1}.
// everything the serial parts of ss_hessenberg read from global memory, requested up front by all threads (no barrier behind it:
// the caller's next barrier publishes it)
__device__ void ss_hess_load(int k, int sb, const ss_ws &w, const ss_tail_args &ta) {
  const int t = threadIdx.x, nt = SS_R;
  const int ko = k - 1, m = ta.m;
  for (int e = t; e < k * ko; e += nt) w.Hs[e] = ta.H[(size_t)(e / ko) * m + (e % ko)];
  for (int e = t; e < ko; e += nt) { w.scs[e] = ta.cs[e]; w.ssn[e] = ta.sn[e]; }
  for (int e = t; e <= ko; e += nt) w.sg[e] = ta.g[e];
  if (t < sb * sb) w.R1s[t] = ta.R1[t];
  if (t < k) w.uu[t] = ta.usb > 0 ? (t < ta.uk0 ? ta.uC2[t * ta.usb + ta.usb - 1] : ta.uR2[(t - ta.uk0) * ta.usb + ta.usb - 1])
                                  : (t == k - 1 ? 1.0 : 0.0);
  for (int e = t; e < k * sb; e += nt) w.F[e] = ta.C1[e];
  // the shifts, σ and the tolerance (w.Fx is free during the Hessenberg work): a lone thread's loads behind the serial parts
  // cost a memory round trip each
  if (t < SS_SMAX) w.Fx[t] = ta.scal[SS_TH + t];
  if (t == SS_SMAX) w.Fx[SS_SMAX] = ta.scal[2];
  if (t == SS_SMAX + 1) w.Fx[SS_SMAX + 1] = ta.ctl->tol;
}
// sR, verdict (LK > 0 only; pitch LK): an LDS copy of the rotated factor's new columns, for a back-substitution in the same workgroup.
// raise_pad1: a verdict that ends the cycle also voids the block whose sweeps are in flight (deferred second pass: this block's
// Hessenberg columns are derived while the NEXT block is under way).
__device__ __forceinline__ double ss_readlane(double v, int lane) {
  lane = __builtin_amdgcn_readfirstlane(lane);   // (uniform by construction; the compiler cannot always see it)
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u & 0xffffffffull), lane);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), lane);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// verdict (optional, LDS): {columns closed so far, converged, failed, residual norm} as thread 0 has just stored them in the
// control block — for the same workgroup's back-substitution, which must not read them back through the cache.
__device__ void ss_hessenberg(int k, int sb, const ss_ws &w, const ss_tail_args &ta, bool loaded = false, double *sR = nullptr,
                              int LK = 0, bool raise_pad1 = false, double *verdict = nullptr) {
  const int t = threadIdx.x;
  const int K = k + sb, ko = k - 1, m = ta.m;                // ko old Hessenberg columns / rotations
  double *F = w.F, *NC = w.NC, *Hs = w.Hs, *scs = w.scs, *ssn = w.ssn, *sg = w.sg;
  if (!loaded) ss_hess_load(k, sb, w, ta);
  SS_STAMP(8);
  __syncthreads();
  SS_STAMP(9);
  const double sigma = w.Fx[SS_SMAX];
  const double *th = w.Fx;
  // (thread maps without a division by a run-time width: (t >> 4, t & 15) over sb ≤ 16 columns, (t >> 5, t & 31) over rows)
  const int t4 = t >> 4, c4 = t & 15;
  if (c4 < sb) {
    for (int j = t4; j < k; j += 16) {  // C = C₁ + C₂ R₁
      double v = F[j * sb + c4];
      for (int a = 0; a <= c4; ++a) v = __builtin_fma(w.Ct[j * sb + a], w.R1s[a * sb + c4], v);
      F[j * sb + c4] = v;
    }
    if (t4 < sb) {  // R = R₂ R₁ (upper)
      const int a = t4, c = c4;
      double v = 0.0;
      for (int p = a; p <= c; ++p) v = __builtin_fma(w.Rm[a * sb + p], w.R1s[p * sb + c], v);
      F[(k + a) * sb + c] = (c >= a) ? v : 0.0;
    }
  }
  __syncthreads();
  // A·(the column the powers started from) = σ X_0 + θ_0·(that column). It is the true basis vector v_k (u = e_k) — or, after a
  // block left at its first pass, that block's last STORED column = V_true u, u = [C₂ ; R₂][:, last]: then
  // A v_k = (σ F_0 + θ_0 u − Σ_{i<k} u_i A v_i) / u_k, with A v_i the old Hessenberg columns.
  if (t < K) {   // (u was requested with the first loads of this routine)
    double a = sigma * F[t * sb];
    if (t < k) {
      a = __builtin_fma(th[0], w.uu[t], a);
      if (ta.usb > 0)
        for (int tt = (t > 0 ? t - 1 : 0); tt < ko; ++tt) a = __builtin_fma(-Hs[t * ko + tt], w.uu[tt], a);
    }
    NC[t] = ta.usb > 0 ? a / w.uu[k - 1] : a;
  }
  __syncthreads();
  SS_STAMP(10);
  // Coordinates of A q_j, j = 1..sb−1: NC_j = (σF_j + θ_j F_{j−1} − H_old F_{j−1}[:k] − Σ_{q<j} NC_q · Fb(q, j−1)) / R_{j−1,j−1}
  // with Fb(0, c) = F[k−1][c] and Fb(q, c) = R[q−1][c]. The part that does not involve other NC columns is a (K × k)(k × sb)
  // product — all (j, i) entries at once —, the rest a triangular solve with K right-hand sides: right-looking, NC_q leaves
  // every later column as soon as it is final. Per entry the multiply-adds run in the order of the column-by-column
  // recurrence this replaces (one entry per thread and step: 14 dependent steps of one multiply-add instead of 14 steps of
  // ≤ 30: measured 14.6 → ≈ 2 µs at k = 16, s = 15).
  const int t5 = t >> 5, r5 = t & 31;
  for (int j = 1 + t5; j < sb; j += 8) {
    for (int i = r5; i < K; i += 32) {
      double a = sigma * F[i * sb + j] + th[j] * F[i * sb + (j - 1)];
      if (i < k)
        for (int tt = (i > 0 ? i - 1 : 0); tt < ko; ++tt) a = __builtin_fma(-Hs[i * ko + tt], F[tt * sb + (j - 1)], a);
      NC[j * K + i] = a;
    }
  }
  // (a thread's entries keep their (column offset, row) through all steps: the divisions by K are done once; the final division of
  //  a column is a multiplication by a reciprocal formed up front — a division inside the step is ≈ 300 cycles of its ≈ 500)
  double *rdiag = w.Gd;   // (free until the rotations)
  if (t < sb - 1) rdiag[t] = 1.0 / F[(k + t) * sb + t];
  __syncthreads();
#pragma unroll 1
  for (int q = 0; q + 1 < sb; ++q) {
    const int frow = (q == 0) ? (k - 1) : (k + q - 1);
    const double rdq = rdiag[q];
    for (int jj = t5; jj < sb - 1 - q; jj += 8) {   // (a thread's entries keep their (column offset, row) through all steps)
      const int j = q + 1 + jj;
      const double fj = F[frow * sb + (j - 1)];
      for (int i = r5; i < K; i += 32) {
        double a = __builtin_fma(-NC[q * K + i], fj, NC[j * K + i]);
        if (jj == 0) a *= rdq;    // its last term: the column is final
        NC[j * K + i] = a;
      }
    }
    __syncthreads();
  }
  SS_STAMP(11);
  for (int j = t5; j < sb; j += 8) {  // the un-rotated columns (rows ≤ column + 1; the rest is rounding noise)
    const int jc = ko + j;
    for (int i = r5; i < K; i += 32)
      if (i <= jc + 1 && jc < m) ta.H[(size_t)i * m + jc] = NC[j * K + i];
  }
  __syncthreads();   // (the rotations below overwrite an entry of every column)
  if (t < sb) {  // the rotations of earlier blocks: every new column on its own lane, the entry under way in a register
    const int jc = ko + t;
    double *h = &NC[t * K];
    double a = h[0];
    for (int i = 0; i < ko; ++i) {
      const double b = h[i + 1], c = scs[i], sn = ssn[i];
      const double rij = c * a + sn * b;
      ta.Rg[(size_t)i * m + jc] = rij;
      if (LK > 0) sR[i * LK + jc] = rij;
      a = -sn * a + c * b;
    }
    h[ko] = a;
  }
  __syncthreads();
  SS_STAMP(12);
  // The rotations this block creates: rotation i comes from new column i after the rotations before it — a chain of sb steps —,
  // but every rotation is applied to all LATER columns at once (lane j owns column j): sb steps of one hypot + one rotation
  // instead of thread 0 walking sb²/2 rotations (measured 16.3 → ≈ 3 µs at s = 15). All sb rotations are formed; which columns
  // count (the first that meets the tolerance closes the cycle) is decided by the scalar pass behind it.
  if (t < 64) {
    // one wavefront, lane j owns column j: the entry the previous rotation left in row ko + i (`carry`, a register) and the
    // column's next row as the recurrence left it (requested from LDS one step ahead: off the chain). Rotation i is formed by
    // EVERY lane from lane i's pair (two v_readlane: the branch on the pair's magnitude is uniform) — c = h·r, s = β·r, d = x·r
    // with r = x^(−½) from v_rsq_f64 and two Newton steps, x = h² + β²: a dozen dependent operations where sqrt and two
    // divisions are ≈ 70 (round 4: lane i alone, through LDS with four fences per rotation: 11 µs for 15; formed from each lane's
    // own pair the finished columns' stale entries sent the wavefront through the scaled hypot on every step). A ROLLED loop: the
    // code runs once from a cold instruction cache, the unrolled form was fetched byte by byte.
    const int col = t < sb ? t : sb - 1;   // (lanes ≥ sb shadow the last column; nothing of theirs is stored)
    const double *h = &NC[col * K + ko];
    double carry = h[0], bnext = h[1];
    // the residual recurrence g_{j+1} = −s_j g_j, g_j ← c_j g_j and the stopping test ride in the same loop: the rotation is the
    // same number in every lane, so every lane carries g and the verdict (round 4: thread 0 walked the columns afterwards, eight LDS
    // round trips and five global stores each: 4.2 µs)
    const double tol = w.Fx[SS_SMAX + 1];
    double gcur = sg[ko], rn = 0.0, beta_l = 0.0;
    int closed = 0, dn = 0, cv = 0, fl = 0;
#pragma unroll 1
    for (int i = 0; i < sb; ++i) {
      const double b = bnext;
      if (i + 2 <= sb) bnext = h[i + 2];
      const double hk = ss_readlane(carry, i), beta = ss_readlane(b, i);
      const double x = __builtin_fma(hk, hk, beta * beta);
      double ci, si, d;
      if (x > 1e-280 && x < 1e280) {
        double r = __builtin_amdgcn_rsq(x);
        r = __builtin_fma(0.5 * r, __builtin_fma(-(x * r), r, 1.0), r);
        r = __builtin_fma(0.5 * r, __builtin_fma(-(x * r), r, 1.0), r);
        ci = hk * r; si = beta * r; d = x * r;
      } else {   // zero, tiny, huge or not a number: the scaled evaluation
        d = hypot(hk, beta);
        if (d == 0.0) { ci = 1.0; si = 0.0; } else { ci = hk / d; si = beta / d; }
      }
      double outv = d;
      if (t != i) {
        outv = ci * carry + si * b;
        carry = -si * carry + ci * b;
      }
      if (t >= i && t < sb) {   // R entries of the block's own rows: rotated values above the diagonal, d on it
        ta.Rg[(size_t)(ko + i) * m + (ko + t)] = outv;
        if (LK > 0) sR[(ko + i) * LK + (ko + t)] = outv;
      }
      if (!dn) {   // (uniform) column ko + i counts
        const double gnext = -si * gcur;
        if (t == i) {
          scs[ko + i] = ci;
          ssn[ko + i] = si;
          sg[ko + i] = ci * gcur;
        }
        rn = fabs(gnext);
        beta_l = beta;
        closed = i + 1;
        if (!(rn == rn) || isinf(rn) || !(beta == beta)) { fl = 1; dn = 1; }
        else if (tol >= 0.0 && rn <= tol) { cv = 1; dn = 1; }
        else if (beta == 0.0) { cv = 1; dn = 1; }
        gcur = gnext;
      }
    }
    if (t == 0) sg[ko + closed] = gcur;
    // what the next blocks, the back-substitution and the host read — the closed columns' rotations and residuals by their lanes
    if (t < closed) { ta.cs[ko + t] = scs[ko + t]; ta.sn[ko + t] = ssn[ko + t]; ta.g[ko + t] = sg[ko + t]; }
    if (t < sb) ta.sc[k + t] = 1.0;   // the new columns are normalised
    if (t == 0) {
      nk_gmres_ctl *ctl = ta.ctl;
      if (LK > 0) { verdict[0] = (double)(ko + closed); verdict[1] = (double)cv; verdict[2] = (double)fl; verdict[3] = rn; }
      ta.g[ko + closed] = gcur;
      if (fl) ctl->failed = 1;
      if (cv) ctl->converged = 1;
      ctl->rnorm = rn;
      ctl->hn = beta_l;
      ctl->k = ko + closed;
      if (dn) {
        ctl->done = 1;
        if (raise_pad1) ctl->pad1 = 1;
      }
      ta.scal[0] = 1.0 / sigma;   // the next block starts from a normalised column
      ss_pub_progress(ta.pub, ta.seq, ko + closed, dn);
    }
  }
  SS_STAMP(7);
  __syncthreads();
  SS_STAMP(13);
}

// the scalar work as launches of their own (the streaming size class k + s > 48, and NK_SS_FUSED=0)
__global__ __launch_bounds__(256) void k_ss_tail1(int k, int sb, double *__restrict__ coef, ss_tail_args ta) {
  extern __shared__ double s_tail[];
  __shared__ ss_fixc s_fc;
  if (ta.ctl->done) return;
  const ss_ws w = ss_ws_carve(s_tail, k, sb, false);
  ss_fixc_request(&s_fc, ta.fix, s_tail, (int)ss_ws_doubles(k, sb, false), 0);
  __syncthreads();
  if (!ss_factor(k, sb, ta.red, ta.sc, w, s_fc, ta.fix.n, ta.ptol, s_tail)) { ss_fail(ta); return; }
  const int t = threadIdx.x;
  for (int e = t; e < k * sb; e += 256) coef[e] = w.U[e];
  if (t < sb * sb) coef[(size_t)k * sb + t] = w.Ri[t];
  ss_keep_pass1(k, sb, w, ta, ta.red);
}
__global__ __launch_bounds__(256) void k_ss_tail2(int k, int sb, double *__restrict__ coef, ss_tail_args ta) {
  extern __shared__ double s_tail[];
  __shared__ ss_fixc s_fc;
  if (ta.ctl->pad1) return;  // (pad1: the cycle was done when this block started, or its first pass failed)
  const ss_ws w = ss_ws_carve(s_tail, k, sb, true);
  ss_fixc_request(&s_fc, ta.fix, s_tail, (int)ss_ws_doubles(k, sb, true), 0);
  __syncthreads();
  if (!ss_factor(k, sb, ta.red, ta.sc, w, s_fc, ta.fix.n, ta.ptol, s_tail)) { ss_fail(ta); return; }
  if (ta.Wi != nullptr && !ss_first_pass_departure_ok(k, sb, w, 0.1)) { ss_fail(ta); return; }
  const int t = threadIdx.x;
  for (int e = t; e < k * sb; e += 256) { coef[e] = w.U[e]; ta.C2[e] = w.Ct[e]; }
  if (t < sb * sb) { coef[(size_t)k * sb + t] = w.Ri[t]; ta.R2[t] = w.Rm[t]; }
  __syncthreads();
  if (ta.Wi != nullptr) ss_fix_prepare(k, sb, w.Ct, w.Rm, w.Sm, ta.Wi, ta.D);   // the block is left at its first pass
  ss_hessenberg(k, sb, w, ta);
}
// the Hessenberg columns of a block as a launch of its own: the LAST block of a cycle, whose third sweep is never run (below)
// (shared by k_ss_hess and the sweeps that host this work in their workgroup 0)
__device__ void ss_hess_block(int k, int sb, double *lds, const ss_tail_args &ta, bool raise_pad1 = false) {
  const ss_ws w = ss_ws_carve(lds, k, sb, true);
  const int t = threadIdx.x;
  for (int e = t; e < k * sb; e += SS_R) w.Ct[e] = ta.C2[e];      // pass 2's factors, left by the reduction's last workgroup
  if (t < sb * sb) w.Rm[t] = ta.R2[t];
  ss_hess_load(k, sb, w, ta);   // (the same round trip: hosted beside streaming workgroups a round trip is ≈ 4 µs)
  __syncthreads();
  if (ta.Wi != nullptr) ss_fix_prepare(k, sb, w.Ct, w.Rm, w.Sm, ta.Wi, ta.D);   // (Sm is free: ss_factor is not run here)
  ss_hessenberg(k, sb, w, ta, true, nullptr, 0, raise_pad1);
}
__global__ __launch_bounds__(256) void k_ss_hess(int k, int sb, ss_tail_args ta) {
  extern __shared__ double s_tail[];
  if (ta.ctl->pad1) return;
  ss_hess_block(k, sb, s_tail, ta);
}
// The last block of a cycle never gets its second update (sweep C): its columns Q = (Q₁ − V_k C₂) R₂⁻¹ are used once more only —
// in x += [V_k Q] y —, and that product can be taken from the columns as pass 1 left them:
//     [V_k Q] y = V_k (y_k − C₂ b) + Q₁ b,   b = R₂⁻¹ y_Q
// (a restart forms r = b − A x afresh and starts a new basis). One sweep over k + 2s columns less per cycle — 386 MB of the
// 1.7 GB the sweeps of a 1024² cycle moved — for an s × s triangular solve and a k × s product inside the back-substitution
// kernel (k_backsolve's `fx`, nk_gmres.hip).

// The scalar work of the block scheme as ONE kind of launch (k_ss_job, behind the sweeps): what a job does
constexpr int SSJ_F1 = 1, SSJ_F2 = 2, SSJ_HESS = 4, SSJ_BACK = 8, SSJ_COEF2 = 16, SSJ_PREP = 32;
struct ss_job {
  const double *part0, *part1;
  int nblk0, nblk1, nslots0, nslots1;
  int k0, sb0, k1, sb1;   // [0]: this block (first pass); [1]: the pending block (second pass)
  int mode, m;
  double *red, *coef;
  int raw_last;               // SSJ_BACK: the pending block's sweep B stored nothing — its columns are the matrix powers' X (below)
  const int *d_skip;
  unsigned int *ticket;
  nk_gmres_ctl *ctl;          // (what both blocks' argument sets share)
  const double *sc;
  nk_gmres_pub *pub;
  uint64_t seq;
  nk_ss_fix cfix;             // the earlier blocks whose factors the factorisations apply: this block's list (SSJ_F1), else the pending block's
  double *y;                  // SSJ_BACK
  const double *Rg, *g;
  const uint64_t *peer_err;
  nk_ss_fix bfx;              // SSJ_BACK: the blocks left at their first pass (the pending block is the last of them)
  int host_wgs;               // the job rides in the LAST host_wgs workgroups of a sweep A (k_ss_block): 0 = a launch of its own
};
struct ss_job_lds { size_t sc, fix, w1, w0, sR, sg, verdict, bfix, rdv, total; int LK; };
__host__ __device__ inline ss_job_lds ss_job_layout(const ss_job &j, int mode);
// (defined behind the sweeps; a sweep A may host it)
template <bool PEER, int MODE>
__device__ __forceinline__ void ss_job_body(const ss_job &j, const ss_tail_args &ta0, const ss_tail_args &ta1, const nk_peer_ar_view &pv,
                                            int slot, int nwg, double *s_rf);

// coef (UPDATE): U (k × S, row-major: the coefficients the update takes off, scales of un-normalised columns folded in),
// then R (S × S, row-major, upper triangular, its diagonal replaced by the reciprocals: the form the substitution takes)
// MTC = 1, 2, 3: k + S ≤ 16·MTC. The k + S values of a thread's row live in registers and the NEXT tile's loads are issued
// before the matrix-core phase of the current one — the LDS tile bounds the occupancy at 2 workgroups per CU, which then keep
// ≈ 2 × (k+S) × 2 KB of loads in flight per CU through both phases; the Gram block is exactly MTC tiles of 16 rows.
// MTC = 0: any k ≤ 80 − S, the columns stream past eight at a time (no prefetch), five Gram tiles.
// FUSE: the scalar work of the block runs inside the sweep that consumes it — EVERY workgroup factors the reduced block itself
// (≈ 2 KB of L2-resident operands, a few µs once per persistent workgroup) and takes its update coefficients from LDS, so the
// one-workgroup launches between reduction and sweep disappear; workgroup 0 also keeps pass 1's factors (sweep B) or, after
// its share of sweep C is on its way, derives the Hessenberg columns, rotations and the stopping test. `mark`: workgroup 0
// records the skip flag it saw (sweep A stamps "the cycle was done when this block started" for sweep C, which must not look at
// a flag that workgroup 0 of its own launch may raise).
// KC > 0: k is the compile-time constant KC (the shapes of the default cycle, 15-column blocks behind 1 and 16 columns). With a
// run-time k every basis column's load and its S multiply-adds sit behind a uniform branch `j < k`, and the wait-count pass
// then cannot tell how many loads are outstanding: it waits for vmcnt(0) in front of the first use of the CURRENT register set
// — i.e. for the NEXT tile's loads it has just issued (round 4, read in the ISA: the double buffering never overlapped anything
// inside a workgroup), and every column's coefficients are a scalar-load round trip of their own. Straight-line code lets it
// count (vmcnt(k + S) …) and batch the scalar loads.
template <int S, bool UPDATE, bool GRAM, int MTC, bool FUSE, int KC = 0>
__global__ __launch_bounds__(SS_R) void k_ss_block(int64_t n, int k_rt, double *__restrict__ V, int64_t ldv,
                                                   const double *__restrict__ coef_in, double *__restrict__ partials,
                                                   const int *d_skip, int ntiles, ss_tail_args ta, int *mark, int ws_off, int hk, int hs,
                                                   const ss_job hj) {
  const int k = KC > 0 ? KC : k_rt;
  // JOBHOST (sweep A of the default cycle's second block): the LAST hj.host_wgs workgroups stream nothing — they are the
  // scalar launch that closes the PREVIOUS block (reduction of its sweep B's partial Gram block, second factorisation, Wi / D,
  // Hessenberg columns and stopping test: ss_job_body), hidden behind this sweep; `ta` is that block's argument set
  constexpr bool JOBHOST = FUSE && GRAM && !UPDATE && KC == 16;
  const int njob = JOBHOST ? hj.host_wgs : 0;
  const bool bar = (ws_off & 1) != 0;   // development switch NK_SS_BARRIERS=1: the per-tile workgroup barriers of rounds 2–3
  ws_off = 0;
  {
    const int dskip = (d_skip != nullptr) ? *d_skip : 0;
    if (mark != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *mark = dskip;
    if (dskip) return;
  }
  extern __shared__ double sX[];
  if constexpr (JOBHOST) {
    if (njob > 0 && (int)blockIdx.x >= (int)gridDim.x - njob) {
      ss_job_body<false, SSJ_F2 | SSJ_PREP | SSJ_HESS>(hj, ta, ta, nk_peer_ar_view{nullptr, 0, 0, 0, nullptr},
                                                       (int)blockIdx.x - ((int)gridDim.x - njob), njob, sX);
      return;
    }
  }
  const int gstream = (int)gridDim.x - njob;           // workgroups that stream (= the pitch of the partial blocks)
  constexpr int NT = MTC > 0 ? MTC : SS_MTMAX;         // Gram tiles this instantiation accumulates
  constexpr int NVR = MTC > 0 ? 16 * MTC - S : 1;      // basis values a thread holds (k ≤ NVR)
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int K = k + S;
  ss_d4 acc[NT];
#pragma unroll
  for (int mt = 0; mt < NT; ++mt) acc[mt] = ss_d4{0.0, 0.0, 0.0, 0.0};
  double *__restrict__ Wc = V + (size_t)k * ldv;
  // DB: two register sets — the NEXT tile's k + S loads are issued before the current tile is touched, so they are in flight
  // through the update, the LDS traffic AND the matrix-core phase (one set: only through the matrix-core phase; with 15-column
  // blocks the multiply-adds of a tile are ≈ 40 % of its memory time and did not overlap with it). The LDS tile bounds these
  // kernels at 2–3 workgroups per CU, so the extra VGPRs are free.
  constexpr bool DB = MTC > 0 && S > 8;
  double vr[NVR], w[S], vr2[NVR], w2[S];   // (the second set is dead code without DB)
  auto prefetch = [&](double (&vrx)[NVR], double (&wx)[S], int tile) {
    const int64_t r = (int64_t)tile * SS_R + t;
    const int64_t rc = r < n ? r : n - 1;
#pragma unroll
    for (int c = 0; c < S; ++c) wx[c] = Wc[(size_t)c * ldv + rc];
    if (MTC > 0) {
#pragma unroll
      for (int j = 0; j < NVR; ++j)
        if (j < k) vrx[j] = V[(size_t)j * ldv + rc];
    }
  };
  // operand addresses of the matrix-core phase. Lane (i, q4) supplies A[i][q4] = X[row + q4][16·mt + i] and
  // B[q4][i] = X[row + q4][k + i]; columns past the end are clamped to the last one: they only feed rows ≥ K / columns ≥ S of
  // the accumulator tiles, which are never stored — no masks, no branches.
  const int li = lane & 15, q4 = lane >> 4;
  const double *pb = sX + (k + (li < S ? li : S - 1)) * SS_P + wv * 64 + q4;
  const double *pa[NT];
#pragma unroll
  for (int mt = 0; mt < NT; ++mt) {
    const int col = mt * 16 + li;
    pa[mt] = sX + (col < K ? col : K - 1) * SS_P + wv * 64 + q4;
  }
  // a workgroup walks CONTIGUOUS tiles: each of its k + S column streams then advances through adjacent 2 KB pieces
  // (sweep C with the Hessenberg duty: workgroup 0 streams nothing — its scalar work, ≈ 20 µs of dependent chains, then
  // hides behind the others' share instead of extending the launch)
  const bool hw = FUSE && gstream > 1 && njob == 0;
  const int nwk = gstream - (hw ? 1 : 0), me = (int)blockIdx.x - (hw ? 1 : 0);
  const int tpw = (ntiles + nwk - 1) / nwk;
  int tile0 = me >= 0 ? me * tpw : 0, tile1 = me >= 0 ? min(tile0 + tpw, ntiles) : 0;
  if (njob > 0) {   // workgroups given up to a hosted job: ⌊tiles/nwk⌋ or one more each, so that none of the rest idles
    const int base = ntiles / nwk, rem = ntiles % nwk;
    tile0 = me * base + min(me, rem);
    tile1 = tile0 + base + (me < rem ? 1 : 0);
  }
  auto process = [&](double (&vr)[NVR], double (&w)[S], int tile, bool valid, auto &&mid) {
    const int64_t r = (int64_t)tile * SS_R + t;
    const bool ok = valid && r < n;
    const double *__restrict__ coef = coef_in;
    const double *__restrict__ Rinv = coef + (size_t)k * S;
    if (MTC > 0) {
#pragma unroll
      for (int j = 0; j < NVR; ++j) {
        if (j < k) {
          if (GRAM) sX[j * SS_P + t] = ok ? vr[j] : 0.0;
          if (UPDATE) {
#pragma unroll
            for (int c = 0; c < S; ++c) w[c] = __builtin_fma(-vr[j], coef[j * S + c], w[c]);
          }
        }
      }
    } else {
      const int64_t rc = ok ? r : n - 1;
      constexpr int U = 8;
      for (int j0 = 0; j0 < k; j0 += U) {
        double v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int j = j0 + u < k ? j0 + u : k - 1;
          v[u] = V[(size_t)j * ldv + rc];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (j0 + u < k) {
            const int j = j0 + u;
            if (GRAM) sX[j * SS_P + t] = ok ? v[u] : 0.0;
            if (UPDATE) {
#pragma unroll
              for (int c = 0; c < S; ++c) w[c] = __builtin_fma(-v[u], coef[j * S + c], w[c]);
            }
          }
        }
      }
    }
    if (UPDATE) {
      // q = w R⁻¹ by forward substitution (Rinv: R with a reciprocal diagonal): as soon as q_cc is known it leaves every later
      // column — S(S−1)/2 independent multiply-adds behind a dependent chain of S multiplications
#pragma unroll
      for (int cc = 0; cc < S; ++cc) {
        const double q = w[cc] * Rinv[cc * S + cc];
        w[cc] = q;
#pragma unroll
        for (int c = cc + 1; c < S; ++c) w[c] = __builtin_fma(-q, Rinv[cc * S + c], w[c]);
      }
#pragma unroll
      for (int c = 0; c < S; ++c)
        if (ok) Wc[(size_t)c * ldv + r] = w[c];
    }
    if (GRAM) {
#pragma unroll
      for (int c = 0; c < S; ++c) sX[(k + c) * SS_P + t] = ok ? w[c] : 0.0;
      // (no workgroup barrier: a wavefront's matrix-core operands are the 64 rows it has just written itself — LDS executes a
      //  wavefront's instructions in order — so the four wavefronts of a workgroup run through their tiles independently)
      if (bar) __syncthreads();
    }
    mid();   // (one register set: the next tile's loads go out here, in flight through the matrix-core phase)
    if (GRAM) {
      // 64 rows per wavefront, 4 per instruction; operands of four instructions are requested together
#pragma unroll
      for (int kk = 0; kk < 16; kk += 4) {
        double bb[4], aa[NT][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          bb[u] = pb[(kk + u) * 4];
#pragma unroll
          for (int mt = 0; mt < NT; ++mt) aa[mt][u] = pa[mt][(kk + u) * 4];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
          for (int mt = 0; mt < NT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(aa[mt][u], bb[u], acc[mt], 0, 0, 0);
        }
      }
      if (bar) __syncthreads();
    }
  };
  if (tile0 < tile1) prefetch(vr, w, tile0);
  if constexpr (DB && KC > 0) {
    // straight-line pipeline: every prefetch is issued unconditionally (past the end: the workgroup's last tile again; the update sweep below reads 2 KB of column 0 instead — measured neutral to
    // 0.6 µs slower here) and an odd tile count runs its phantom tile with all rows masked (zeros into the Gram block, no stores) — no branch
    // around a batch of loads, so the wait in front of a register set's first use is vmcnt(loads issued behind it), not 0
    // (requesting a set again right after it is staged — two tiles ahead — measured 1–2 µs SLOWER for the read-only sweep on
    //  the same box: 27.1 / 48.3 against 26.0 / 46.1 µs behind 1 / 16 columns; the update sweep below keeps that order)
    for (int tile = tile0; tile < tile1; tile += 2) {
      prefetch(vr2, w2, tile + 1 < tile1 ? tile + 1 : tile1 - 1);
      process(vr, w, tile, true, [] {});
      prefetch(vr, w, tile + 2 < tile1 ? tile + 2 : tile1 - 1);
      process(vr2, w2, tile + 1, tile + 1 < tile1, [] {});
    }
  } else if constexpr (DB) {
    for (int tile = tile0; tile < tile1; tile += 2) {
      if (tile + 1 < tile1) prefetch(vr2, w2, tile + 1);
      process(vr, w, tile, true, [] {});
      if (tile + 1 < tile1) {
        if (tile + 2 < tile1) prefetch(vr, w, tile + 2);
        process(vr2, w2, tile + 1, true, [] {});
      }
    }
  } else {
    for (int tile = tile0; tile < tile1; ++tile)
      process(vr, w, tile, true, [&] { if (tile + 1 < tile1) prefetch(vr, w, tile + 1); });
  }
  if (GRAM) {
    // the four wavefronts' tiles → one partial per (basis column, new column) and workgroup; fixed order
    __syncthreads();   // (the scratch overlays rows the other wavefronts may still be reading)
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) sX[((wv * NT + mt) * 4 + rr) * 64 + lane] = acc[mt][rr];
    }
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
      const int rr = t >> 6, ln = t & 63;
      const int mrow = mt * 16 + (ln >> 4) + 4 * rr, ncol = ln & 15;
      if (mrow < K && ncol < S) {
        const int e = (mt * 4 + rr) * 64 + ln;
        const double sum = (sX[e] + sX[NT * 256 + e]) + (sX[2 * NT * 256 + e] + sX[3 * NT * 256 + e]);
        partials[(size_t)(mrow * S + ncol) * gstream + blockIdx.x] = sum;
      }
    }
  }
  // workgroup 0, its share of the sweep issued (none when there are others): the Hessenberg columns of the block (hk, hs) —
  // sweep C: this block's; sweep A: those of the PREVIOUS block, which was left at its first pass (implicit second pass)
  if (FUSE && blockIdx.x == 0 && njob == 0) {
    if (GRAM) __syncthreads();
    ss_hess_block(hk, hs, sX + ws_off, ta);
  }
}

// T = [−U N ; N], N = R⁻¹ (upper triangular; `coef` = U (k × S) then R with a reciprocal diagonal), formed in LDS scratch at sX by
// the whole workgroup; returns each lane's share as matrix-core B operands: tb[ks] = T[4 ks + q4][li] (rows ≥ k + S and column
// S … 15 zero). Ends with a barrier (the scratch overlays the tile). (c0, c1) = coef[t], coef[t + 256] — requested by the caller
// BEFORE its first tiles: vector memory returns in order, and behind two tiles per workgroup of a grid that has just started
// (63 MB on the chip) the coefficients arrived ≈ 6 µs late, the matrix pipe idle until T was formed
// (profiles/r06_u_sweep_b_prologue.txt).
template <int S, int KC>
__device__ __forceinline__ void ss_mm_form_t(double c0, double c1, double *sX, double (&tb)[(KC + S + 3) / 4]) {
  constexpr int k = KC, K = KC + S, NKS = (K + 3) / 4;
  static_assert(k * S + S * S <= 2 * SS_R, "two coefficients per thread");
  const int t = threadIdx.x, lane = t & 63, li = lane & 15, q4 = lane >> 4;
  double *sU = sX, *sR = sU + k * S, *sN = sR + S * S, *sT = sN + 256;
  sX[t] = c0;
  if (t + SS_R < k * S + S * S) sX[t + SS_R] = c1;
  __syncthreads();
  if (t < S) {   // row t of N: n R = e_t by forward substitution (R's diagonal arrives as reciprocals)
    double nr[S];
#pragma unroll
    for (int c = 0; c < S; ++c) {
      double a = (c == t) ? 1.0 : 0.0;
#pragma unroll
      for (int c2 = 0; c2 < c; ++c2) a = __builtin_fma(-nr[c2], sR[c2 * S + c], a);
      nr[c] = (c < t) ? 0.0 : a * sR[c * S + c];
    }
#pragma unroll
    for (int c = 0; c < S; ++c) sN[t * 16 + c] = nr[c];
  }
  __syncthreads();
  for (int e = t; e < 4 * NKS * 16; e += SS_R) {
    const int j = e >> 4, c = e & 15;
    double v = 0.0;
    if (c < S) {
      if (j < k) {
        double a = 0.0;   // (all S terms, unrolled — N is upper triangular, the terms past c are exact zeros: one LDS latency, not c)
#pragma unroll
        for (int c2 = 0; c2 < S; ++c2) a = __builtin_fma(sU[j * S + c2], sN[c2 * 16 + c], a);
        v = -a;
      } else if (j < K) {
        v = sN[(j - k) * 16 + c];
      }
    }
    sT[e] = v;
  }
  __syncthreads();
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) tb[ks] = sT[(4 * ks + q4) * 16 + li];
  __syncthreads();
}

// Sweep B of the default cycle's shapes with the UPDATE on the matrix cores as well (round 4). X ← (X − V U) R⁻¹ is one
// product of the tile [V X] (256 rows × (k + S) columns, already staged in LDS for the Gram block) with the (k + S) × S matrix
//     T = [ −U N ; N ],  N = R⁻¹ (upper triangular, formed once per workgroup from the reduction's U and R),
// i.e. (k + S)/4 v_mfma_f64_16x16x4 per 16 rows instead of k·S + S(S+1)/2 multiply-adds per row whose k·S + S² coefficients
// had to come through scalar loads (≈ 135 round trips per tile with a run-time k; hoisted, spilled into VGPR lanes and read back
// with ≈ 9 v_readlane per multiply-add with a compile-time k). The loop body has no scalar memory traffic and no branch around
// its loads: both register sets' waits are exact (vmcnt(k + 2S) …), so the next tile's k + S loads are in flight through the
// whole of the current tile. A wavefront's operands are the 64 rows it staged itself: no workgroup barrier inside the loop.
// Same contract as k_ss_block<S, true, true, …>: `coef` = U (k × S) then R (S × S, reciprocal diagonal); the updated columns go
// back to V[:, k..k+S) and the Gram block [V Q]ᵀQ to `partials`. Rounding differs from the substitution form in the last bits
// (explicit inverse: the same ε κ(R) bound; pass 2 of the block scheme repairs both alike).
// HOST: workgroup 0 streams no tiles — it derives the Hessenberg columns, rotations and stopping test of the PREVIOUS block
// (hk, hs), whose second factorisation was deferred into the launch in front of this sweep (its workspace overlays the tile);
// the other workgroups share the tiles evenly (⌊tiles/(grid − 1)⌋ or one more).
// NOSTORE (round 6): the cycle's LAST block. Its updated columns are read once more only — by x = [V Q] y — and that product can
// be taken from the columns as the matrix powers left them: Q₁ = (X − V U) N, so Q₁ b = X (N b) − V (U N b) — one more entry
// (U, R₁) in the back-substitution's list of blocks (nk_ss_cycle). The sweep then writes nothing: the update lives in the LDS tile
// for the Gram block and is gone with the tile — 8 n S bytes less per cycle, and a read-only stream.
// development (tools/gpu_sweep_decomposition.sh): NK_SS_EXP = 1 — the loop without its matrix instructions (loads and staging only);
// 2 — without its loads (the first two tiles' registers staged again and again); 3 — without loads and staging; 4 — also without
// the operands' LDS reads (the matrix instructions and their accumulator copies alone). Timing builds: the results are meaningless.
#ifndef NK_SS_EXP
#define NK_SS_EXP 0
#endif
template <int S, int KC, bool HOST, bool NOSTORE = false>
__global__ __launch_bounds__(SS_R) void k_ss_block_mm(int64_t n, double *__restrict__ V, int64_t ldv,
                                                      const double *__restrict__ coef, double *__restrict__ partials,
                                                      const int *d_skip, int ntiles, int *mark, ss_tail_args hta, int hk, int hs) {
  {
    const int dskip = (d_skip != nullptr) ? *d_skip : 0;
    if (mark != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *mark = dskip;
    if (dskip) return;
  }
  extern __shared__ double sX[];
  constexpr int k = KC, K = KC + S, NT = (K + 15) / 16, NKS = (K + 3) / 4;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, li = lane & 15, q4 = lane >> 4;
  if (HOST && blockIdx.x == 0) {
    for (int e = t; e < K * S; e += SS_R) partials[(size_t)e * gridDim.x] = 0.0;
    ss_hess_block(hk, hs, sX, hta, true);
    return;
  }
  double *__restrict__ Wc = V + (size_t)k * ldv;
  const int nwk = (int)gridDim.x - (HOST ? 1 : 0), me = (int)blockIdx.x - (HOST ? 1 : 0);
  const int tpw = (ntiles + nwk - 1) / nwk;
  const int tq = ntiles / nwk, tr = ntiles - tq * nwk;
  const int tile0 = HOST ? me * tq + min(me, tr) : me * tpw;
  const int tile1 = HOST ? tile0 + tq + (me < tr ? 1 : 0) : min(tile0 + tpw, ntiles);
  double vr[KC], w[S], vr2[KC], w2[S];
  // (tile < 0: a prefetch past the workgroup's last tile — issued all the same, so that the waits stay exact — reads 2 KB of
  //  column 0 k + S times over: its own last tile again was 62 KB per workgroup, 32 MB over the grid = the whole L2, +9 % HBM
  //  reads by the counters)
  auto prefetch = [&](double (&vrx)[KC], double (&wx)[S], int tile) __attribute__((always_inline)) {
    const int64_t r = (int64_t)(tile < 0 ? tile0 : tile) * SS_R + t;
    const int64_t rc = r < n ? r : n - 1;
    const size_t cs = tile < 0 ? 0 : (size_t)ldv;
    const double *__restrict__ Wr = tile < 0 ? V : Wc;
#pragma unroll
    for (int c = 0; c < S; ++c) wx[c] = Wr[(size_t)c * cs + rc];
#pragma unroll
    for (int j = 0; j < KC; ++j) vrx[j] = V[(size_t)j * cs + rc];
  };
  // (the coefficients are requested in front of the tiles: vector memory returns in order — ss_mm_form_t)
  constexpr int NCOEF = KC * S + S * S;
  static_assert(NCOEF <= 2 * SS_R, "two coefficients per thread");
  const double cf0 = coef[t < NCOEF ? t : 0], cf1 = coef[t + SS_R < NCOEF ? t + SS_R : 0];
  __builtin_amdgcn_sched_barrier(0);
  if (tile0 < tile1) {   // both register sets are in flight while T is formed
    prefetch(vr, w, tile0);
    prefetch(vr2, w2, tile0 + 1 < tile1 ? tile0 + 1 : -1);
  }
  __builtin_amdgcn_sched_barrier(0);
  // ---- T = [−U N ; N] in LDS (the tile is not in use yet), then each lane's share of it as matrix-core B operands
  double tb[NKS];
  ss_mm_form_t<S, KC>(cf0, cf1, sX, tb);
  ss_d4 acc[NT];
#pragma unroll
  for (int mt = 0; mt < NT; ++mt) acc[mt] = ss_d4{0.0, 0.0, 0.0, 0.0};
  // operand addresses. Update: lane (i, q4) supplies A[i][q4] = [V X][row0 + i][4·ks + q4]. Gram: as in k_ss_block.
  const double *pu[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const int col = 4 * ks + q4;
    pu[ks] = sX + (col < K ? col : K - 1) * SS_P + wv * 64 + li;   // (T's rows ≥ K are zero: any finite operand will do)
  }
  const double *pb = sX + (k + (li < S ? li : S - 1)) * SS_P + wv * 64 + q4;
  const double *pa[NT];
#pragma unroll
  for (int mt = 0; mt < NT; ++mt) {
    const int col = mt * 16 + li;
    pa[mt] = sX + (col < K ? col : K - 1) * SS_P + wv * 64 + q4;
  }
  // where lane (i, q4) puts Q[row0 + q4 + 4·rr][i] (i = 15, the padding column of the 16-wide product: a spare column K of the
  // tile — no branch)
  double *pq = sX + (k + li) * SS_P + wv * 64 + q4;
  // the block's columns as ONE raw buffer (the launcher checks that S·ldv·8 fits 32 bits)
  const unsigned ldvb = (unsigned)ldv * 8u;
  const __amdgpu_buffer_rsrc_t wrs =
      __builtin_amdgcn_make_buffer_rsrc((void *)Wc, 0, (int)(unsigned)(((int64_t)(S - 1) * ldv + n) * 8), 0x00020000);
  auto process = [&](double (&vr)[KC], double (&w)[S], int tile, bool valid, int next) __attribute__((always_inline)) {
    const int64_t r = (int64_t)tile * SS_R + t;
    const bool ok = valid && r < n;
#if NK_SS_EXP < 3
#pragma unroll
    for (int j = 0; j < KC; ++j) sX[j * SS_P + t] = ok ? vr[j] : 0.0;
#pragma unroll
    for (int c = 0; c < S; ++c) sX[(k + c) * SS_P + t] = ok ? w[c] : 0.0;
#endif
    // the set is staged: request it again for the tile after next (past the end: the last tile once more, a cache hit) — every
    // load then has two tiles' time to land. (sched_barrier: the machine scheduler otherwise sinks the loads to their uses.)
    __builtin_amdgcn_sched_barrier(0);
#if NK_SS_EXP < 2
    prefetch(vr, w, next);
#endif
    __builtin_amdgcn_sched_barrier(0);
#if NK_SS_EXP == 1
    return;
#endif
    // the update: 16 rows per instruction group, the result back into the X columns of the same rows
    // REGGRAM (k = 16, round 6): the Gram block takes the updated rows straight from the update's accumulator. The product leaves
    // lane (q4, li) with Q[row0 + q4 + 4 r][li], r = 0 … 3 — which is exactly the B operand of a Gram instruction whose four rows
    // are {q4 + 4 r}, and (the A and B lane maps of 16x16x4 coincide) the A operand of the QᵀQ tile for the same rows; the sum
    // over rows does not care how the rows are grouped four at a time. Only the V tile's A operand comes from the LDS tile. No
    // write-back of Q for the Gram phase (its transposed ds_writes were the sweep's only bank conflicts: 12 % of its LDS cycles,
    // profiles/r06_p_sweeps_sq_counters.md), no re-read of Q: 48 LDS instructions per wavefront and tile less, and the 64 matrix
    // instructions of a tile in one run.
    constexpr bool REGGRAM = (KC == 16);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      ss_d4 q = ss_d4{0.0, 0.0, 0.0, 0.0};
      double a[NKS], va[4];
#pragma unroll
#if NK_SS_EXP == 4
      for (int ks = 0; ks < NKS; ++ks) a[ks] = vr[ks % KC];
      if constexpr (REGGRAM) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) va[rr] = w[rr];
      }
#else
      for (int ks = 0; ks < NKS; ++ks) a[ks] = pu[ks][g * 16];
      if constexpr (REGGRAM) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) va[rr] = pa[0][g * 16 + 4 * rr];   // V[row0 + q4 + 4 rr][li]
      }
#endif
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) q = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], tb[ks], q, 0, 0, 0);
      if constexpr (!REGGRAM || !NOSTORE) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) pq[g * 16 + 4 * rr] = q[rr];
      }
      if constexpr (REGGRAM) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(va[rr], q[rr], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(q[rr], q[rr], acc[1], 0, 0, 0);
        }
      }
    }
    // the updated columns leave through the row-per-thread layout (coalesced). Buffer stores: a lane without a row (ragged last
    // tile, the phantom tile of an odd count) carries an out-of-range offset and the hardware drops it — `if (ok) store` is a
    // branch around the stores, behind which the wait-count pass no longer knows how many operations follow a register set's
    // loads and waits for the stores it has just issued at the top of the next tile.
    const unsigned rbyte = (unsigned)r * 8u;
    if constexpr (!NOSTORE) {
#pragma unroll
    for (int c = 0; c < S; ++c) {
      const double qv = sX[(k + c) * SS_P + t];
      const unsigned off = ok ? (unsigned)c * ldvb + rbyte : 0xFFFFFFFFu;
      // (non-temporal hints measured in round 6 — `nt` on these stores −1.6 %, on the loads of the k read-only columns −0.9 %,
      //  both −2.3 % in Newton steps/s: profiles/r06_h_nt_hints_ab.txt — the basis is re-read from the Infinity Cache by the next launch)
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(ss_u2, qv), wrs, (int)off, 0, 0);
    }
    } else {
      (void)rbyte; (void)ldvb; (void)wrs;
      // a row without data (ragged last tile, phantom tile) must not reach the Gram block: the product wrote T-combinations of
      // zeros = zeros there already (the staged values of such a row are zero)
    }
    // Gram block [V Q]ᵀQ: 64 rows per wavefront, 4 per instruction
    if constexpr (!REGGRAM) {
#pragma unroll
    for (int kk = 0; kk < 16; kk += 4) {
      double bb[4], aa[NT][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        bb[u] = pb[(kk + u) * 4];
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) aa[mt][u] = pa[mt][(kk + u) * 4];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(aa[mt][u], bb[u], acc[mt], 0, 0, 0);
      }
    }
    }
  };
  auto pair = [&](int tile) __attribute__((always_inline)) {
    process(vr, w, tile, true, tile + 2 < tile1 ? tile + 2 : -1);
    process(vr2, w2, tile + 1, tile + 1 < tile1, tile + 3 < tile1 ? tile + 3 : -1);
  };
  // (first pair peeled: at the loop header the outstanding-load state of the entry edge then equals the back edge's, and the
  //  waits inside the loop are the steady state's vmcnt(46 … 61) instead of the prologue's vmcnt(31 …))
  if (tile0 < tile1) {
    pair(tile0);
    for (int tile = tile0 + 2; tile < tile1; tile += 2) pair(tile);
  }
  __syncthreads();   // (the scratch overlays rows the other wavefronts may still be reading)
#pragma unroll
  for (int mt = 0; mt < NT; ++mt) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) sX[((wv * NT + mt) * 4 + rr) * 64 + lane] = acc[mt][rr];
  }
  __syncthreads();
#pragma unroll
  for (int mt = 0; mt < NT; ++mt) {
    const int rr = t >> 6, ln = t & 63;
    const int mrow = mt * 16 + (ln >> 4) + 4 * rr, ncol = ln & 15;
    if (mrow < K && ncol < S) {
      const int e = (mt * 4 + rr) * 64 + ln;
      const double sum = (sX[e] + sX[NT * 256 + e]) + (sX[2 * NT * 256 + e] + sX[3 * NT * 256 + e]);
      partials[(size_t)(mrow * S + ncol) * gridDim.x + blockIdx.x] = sum;
    }
  }
}


// Sweep B of the cycle's LAST block behind 16 columns — the read-only form (round 6). k_ss_block_mm<15, 16, ·, true> stores nothing
// but still stages all 31 columns through the LDS tile (31 ds_write_b64 + 62 v_cndmask per wavefront and tile) and reads the
// update's operands back from it (32 of its 48 ds_read_b64): taken apart on the device (profiles/r06_s_sweep_decomposition.txt),
// its loop costs 29.7 µs for the matrix instructions alone, 38.5 with the operands' LDS reads, 53.6 with the staging — more than
// the 42.5 µs its loads take: the sweep was bound by LDS traffic and instruction issue, not by HBM. Here
//  * every lane loads its operands of the UPDATE product straight from global memory in the matrix instruction's own layout: lane
//    (li, q4) takes rows 2 li, 2 li + 1 of a 32-row block of column 4 ks + q4 — one 16-byte buffer load (a wavefront's load covers
//    4 columns × 256 contiguous bytes; 16 loads per register set instead of 31). The even rows of a block are one 16-row group of the
//    product, the odd rows another: the update is row-wise and the Gram block a sum over rows, so the order of rows inside a tile
//    is free as long as both Gram operands use the same one;
//  * only the 16 basis columns go to LDS (8 ds_write_b128 per wavefront and tile), for the Gram block's Vᵀ Q operand, which wants
//    them row-major: one ds_read_b128 delivers a row pair = the operand of the even AND the odd group (8 per tile instead of 48
//    ds_read_b64). Pitch 258 doubles: 16-byte aligned columns, an odd number of 16-byte units (conflict-free both ways);
//  * the updated rows go from the product's accumulator into the Gram instructions (as in k_ss_block_mm since REGGRAM);
//  * raw buffer loads: a column's offset rides in the scalar offset, the lane's in one VGPR for the whole kernel (no 64-bit
//    address arithmetic in the loop), a prefetch past the workgroup's last tile goes through a descriptor of zero records (returns
//    zeros, moves nothing), rows past n are masked in the ragged tile only (a wave-uniform branch around selects).
// Needs (k + S)·ldv·8 < 2³², ldv even and a 16-byte aligned basis; otherwise the launcher takes k_ss_block_mm.
constexpr int SS_P2 = SS_R + 2;
typedef double ss_d2 __attribute__((ext_vector_type(2)));
typedef unsigned int ss_u4 __attribute__((ext_vector_type(4)));
template <int S, int KC, bool HOST>
__global__ __launch_bounds__(SS_R) void k_ss_block_ro(int64_t n, const double *__restrict__ V, int64_t ldv,
                                                      const double *__restrict__ coef, double *__restrict__ partials,
                                                      const int *d_skip, int ntiles, int *mark, ss_tail_args hta, int hk, int hs) {
  static_assert(S == 15 && KC == 16, "the default cycle's second block");
  {
    const int dskip = (d_skip != nullptr) ? *d_skip : 0;
    if (mark != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *mark = dskip;
    if (dskip) return;
  }
  extern __shared__ ss_d2 sX2[];   // (16-byte aligned)
  double *sX = reinterpret_cast<double *>(sX2);
  constexpr int K = KC + S, NT = 2, NKS = 8;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, li = lane & 15, q4 = lane >> 4;
  if (HOST && blockIdx.x == 0) {
    for (int e = t; e < K * S; e += SS_R) partials[(size_t)e * gridDim.x] = 0.0;
    ss_hess_block(hk, hs, sX, hta, true);
    return;
  }
  const int nwk = (int)gridDim.x - (HOST ? 1 : 0), me = (int)blockIdx.x - (HOST ? 1 : 0);
  const int tpw = (ntiles + nwk - 1) / nwk;
  const int tq = ntiles / nwk, tr = ntiles - tq * nwk;
  const int tile0 = HOST ? me * tq + min(me, tr) : me * tpw;
  const int tile1 = HOST ? tile0 + tq + (me < tr ? 1 : 0) : min(tile0 + tpw, ntiles);
  // x[gp][ks]: rows (2 li, 2 li + 1) of the wavefront's 32-row block gp, column 4 ks + q4 (column 31 does not exist: past the
  // descriptor's records — zeros against T's zero row). Block gp of wavefront wv = rows 128 gp + 32 wv … of the tile: the four
  // wavefronts' loads of one (gp, ks) cover 1 KB of each column in one piece
  ss_d2 xa[2][NKS], xb[2][NKS];
  const unsigned nrec = (unsigned)((((int64_t)(K - 1)) * ldv + n) * 8);
  const unsigned voff = (unsigned)(((int64_t)q4 * ldv + 32 * wv + 2 * li) * 8);
  const unsigned colb = (unsigned)(ldv * 32);   // four columns, in bytes
  auto prefetch = [&](ss_d2 (&x)[2][NKS], int gp, int tile) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)V, 0, tile < 0 ? 0 : (int)nrec, 0x00020000);
    const unsigned tb0 = (unsigned)(tile < 0 ? 0 : tile) * (unsigned)(SS_R * 8) + (unsigned)gp * 1024u;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
      x[gp][ks] = __builtin_bit_cast(ss_d2, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, (int)(tb0 + (unsigned)ks * colb), 0));
  };
  constexpr int NCOEF = KC * S + S * S;
  const double cf0 = coef[t < NCOEF ? t : 0], cf1 = coef[t + SS_R < NCOEF ? t + SS_R : 0];   // (in front of the tiles: ss_mm_form_t)
  __builtin_amdgcn_sched_barrier(0);
  if (tile0 < tile1) {   // both register sets are in flight while T is formed
    prefetch(xa, 0, tile0);
    prefetch(xa, 1, tile0);
    prefetch(xb, 0, tile0 + 1 < tile1 ? tile0 + 1 : -1);
    prefetch(xb, 1, tile0 + 1 < tile1 ? tile0 + 1 : -1);
  }
  __builtin_amdgcn_sched_barrier(0);
  double tb[NKS];
  ss_mm_form_t<S, KC>(cf0, cf1, sX, tb);
  ss_d4 acc[NT];
#pragma unroll
  for (int mt = 0; mt < NT; ++mt) acc[mt] = ss_d4{0.0, 0.0, 0.0, 0.0};
  // staging: the row pair of basis column 4 ks + q4 (ks < 4); Gram operand: rows 2 q4 + 8 rr (+ 1) of column li
  ss_d2 *pst = reinterpret_cast<ss_d2 *>(sX + q4 * SS_P2 + 32 * wv + 2 * li);
  const ss_d2 *pva = reinterpret_cast<const ss_d2 *>(sX + li * SS_P2 + 32 * wv + 2 * q4);
  auto process = [&](ss_d2 (&x)[2][NKS], int tile, bool valid, int next) __attribute__((always_inline)) {
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
#pragma unroll
      for (int ks = 0; ks < KC / 4; ++ks) pst[(4 * ks * SS_P2 + 128 * gp) / 2] = x[gp][ks];
    }
    // rows of this tile that exist (a tile without data arrived as zeros: nothing to mask)
    const int64_t rv64 = n - (int64_t)tile * SS_R;
    const int rv = __builtin_amdgcn_readfirstlane(valid ? (rv64 > SS_R ? SS_R : (rv64 < 0 ? 0 : (int)rv64)) : SS_R);
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
      ss_d4 qe = ss_d4{0.0, 0.0, 0.0, 0.0}, qo = ss_d4{0.0, 0.0, 0.0, 0.0};   // the even rows' group, the odd rows' group
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) {
#if NK_SS_EXP == 1
        qe[ks & 3] += x[gp][ks].x; qo[ks & 3] += x[gp][ks].y;   // (the loads stay alive)
#else
        qe = __builtin_amdgcn_mfma_f64_16x16x4f64(x[gp][ks].x, tb[ks], qe, 0, 0, 0);
        qo = __builtin_amdgcn_mfma_f64_16x16x4f64(x[gp][ks].y, tb[ks], qo, 0, 0, 0);
#endif
      }
      ss_d2 va[4];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) va[rr] = pva[(128 * gp + 8 * rr) / 2];
      // the registers are consumed: request them again for the tile after next
      __builtin_amdgcn_sched_barrier(0);
#if NK_SS_EXP < 2
      prefetch(x, gp, next);
#endif
      __builtin_amdgcn_sched_barrier(0);
      if (rv < SS_R) {   // the ragged tile: product lane (li, q4), entry rr is row 128 gp + 32 wv + 2 (q4 + 4 rr) (+ 1) of the tile
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int r = 128 * gp + 32 * wv + 2 * q4 + 8 * rr;
          if (r >= rv) { qe[rr] = 0.0; va[rr].x = 0.0; }
          if (r + 1 >= rv) { qo[rr] = 0.0; va[rr].y = 0.0; }
        }
      }
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
#if NK_SS_EXP == 1
        acc[0][rr] += va[rr].x + qe[rr]; acc[1][rr] += va[rr].y + qo[rr];
#else
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(va[rr].x, qe[rr], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(qe[rr], qe[rr], acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(va[rr].y, qo[rr], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(qo[rr], qo[rr], acc[1], 0, 0, 0);
#endif
      }
    }
  };
  auto pair = [&](int tile) __attribute__((always_inline)) {
    process(xa, tile, true, tile + 2 < tile1 ? tile + 2 : -1);
    process(xb, tile + 1, tile + 1 < tile1, tile + 3 < tile1 ? tile + 3 : -1);
  };
  if (tile0 < tile1) {   // (first pair peeled: the loop's waits are then the steady state's)
    pair(tile0);
    for (int tile = tile0 + 2; tile < tile1; tile += 2) pair(tile);
  }
  __syncthreads();   // (the scratch overlays rows the other wavefronts may still be reading)
#pragma unroll
  for (int mt = 0; mt < NT; ++mt) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) sX[((wv * NT + mt) * 4 + rr) * 64 + lane] = acc[mt][rr];
  }
  __syncthreads();
#pragma unroll
  for (int mt = 0; mt < NT; ++mt) {
    const int rr = t >> 6, ln = t & 63;
    const int mrow = mt * 16 + (ln >> 4) + 4 * rr, ncol = ln & 15;
    if (mrow < K && ncol < S) {
      const int e = (mt * 4 + rr) * 64 + ln;
      const double sum = (sX[e] + sX[NT * 256 + e]) + (sX[2 * NT * 256 + e] + sX[3 * NT * 256 + e]);
      partials[(size_t)(mrow * S + ncol) * gridDim.x + blockIdx.x] = sum;
    }
  }
}

static size_t ss_tile_doubles(int k, int s, bool gram, int nt) {
  if (!gram) return 0;
  const size_t a = (size_t)(k + s) * SS_P, b = (size_t)4 * nt * 256;
  return a > b ? a : b;
}
static int ss_class(int k, int s) { return k + s <= 16 ? 1 : (k + s <= 32 ? 2 : (k + s <= 48 ? 3 : 0)); }
static size_t ss_lds_bytes(int k, int s, bool gram) {
  const int c = ss_class(k, s);
  return ss_tile_doubles(k, s, gram, c ? c : SS_MTMAX) * sizeof(double);
}
bool nk_ss_fusable(int k, int s) {
  static const bool off = getenv("NK_SS_FUSED") && atoi(getenv("NK_SS_FUSED")) == 0;
  return !off && ss_class(k, s) != 0;
}
// persistent workgroups per CU = what the runtime says a CU holds of the instance that will run (LDS tile and register
// footprint: 190–196 VGPRs for the Gram sweeps of a 15-column block behind 16 columns — two workgroups, not the three a table
// of size classes once said; the third of every CU ran as a second round on a third of the chip), queried once per shape
int nk_ss_sweep_occupancy(nk_ctx *ctx, int mode, int k, int s);
static int ss_per_cu(nk_ctx *ctx, int k, int s) {
  static const int forced = getenv("NK_SS_PER_CU") ? atoi(getenv("NK_SS_PER_CU")) : 0;   // A/B switch
  if (forced > 0) return forced > SS_MAX_WG_PER_CU ? SS_MAX_WG_PER_CU : forced;
  const int a = nk_ss_sweep_occupancy(ctx, 0, k, s), b = nk_ss_sweep_occupancy(ctx, 1, k, s);
  int per_cu = a < b ? a : b;
  return per_cu < 1 ? 1 : (per_cu > SS_MAX_WG_PER_CU ? SS_MAX_WG_PER_CU : per_cu);
}
// Workgroups of a sweep: as many per CU as fit — but not more than divide the tiles evenly. A workgroup walks ⌈tiles / grid⌉
// tiles, so a CU is busy for per_cu·⌈tiles / (CUs·per_cu)⌉ of them: with 4096 tiles (n = 2²⁰) on 256 CUs three workgroups per CU
// (what the 15-column block behind one column fits) make that 18 where two or four make it 16 — measured 25.6 → 24.2 and
// 47.4 → 45.7 µs for sweeps A and B of that shape with two. The largest count that reaches the minimum is taken.
int nk_ss_grid(nk_ctx *ctx, int64_t n, int k, int s) {
  const int ntiles = (int)((n + SS_R - 1) / SS_R);
  const int occ = ss_per_cu(ctx, k, s);
  int best = occ;
  int64_t best_cost = INT64_MAX;
  for (int p = occ; p >= 1; --p) {
    const int64_t g = (int64_t)ctx->num_cus * p;
    const int64_t cost = (int64_t)p * ((ntiles + g - 1) / g);
    if (cost < best_cost) { best_cost = cost; best = p; }
  }
  int g = ctx->num_cus * best;
  if (g > ntiles) g = ntiles;
  return g > 0 ? g : 1;
}
// Sweep A (read-only) of the default cycle's shapes runs ONE workgroup per CU — measured on the same box, stand-alone:
// 24.1 → 22.0 µs behind one column, 45.9 → 42.1 µs behind 16 (6.1–6.2 TB/s; the writing sweep B gains nothing from it). When it
// hosts the previous block's Hessenberg work the hosting workgroup is one of these (it streams nothing and has a CU to itself:
// 49 µs; as a 257th workgroup beside a streaming one 54 µs, inside a grid of 512 52 µs — the hosted scalar work, ≈ 45 µs under
// load against 29 µs as a launch of its own, is what that launch waits for, not its 255 streaming workgroups).
// Other shapes: the grid of sweep B.
int nk_ss_grid_a(nk_ctx *ctx, int64_t n, int k, int s, bool hosting) {
  static const bool off = getenv("NK_SS_GRID_A") && atoi(getenv("NK_SS_GRID_A")) == 0;   // A/B switch
  static const bool kc_off = getenv("NK_SS_KCONST") && atoi(getenv("NK_SS_KCONST")) == 0;
  const int ntiles = (int)((n + SS_R - 1) / SS_R);
  if (off || kc_off || s != 15 || (k != 1 && k != 16) || ntiles < 2 * ctx->num_cus) return nk_ss_grid(ctx, n, k, s);
  static const int host_extra = getenv("NK_SS_GRID_A_HOST") ? atoi(getenv("NK_SS_GRID_A_HOST")) : 0;   // A/B switch (1: a 257th workgroup)
  return ctx->num_cus + (hosting ? host_extra : 0);
}

// Sweep B that stores nothing takes the read-only kernel (k_ss_block_ro): behind 16 columns, 32-bit byte offsets over the whole
// basis, 16-byte row pairs
static bool ss_b_is_read_only(const double *V, int64_t ldv, int k, int s) {
  static const bool ro_on = !(getenv("NK_SS_RO") && atoi(getenv("NK_SS_RO")) == 0);   // A/B switch
  return ro_on && k == 16 && s == 15 && (int64_t)(k + s) * ldv * 8 < ((int64_t)1 << 32) - 8 && (ldv & 1) == 0 && ((uintptr_t)V & 15) == 0;
}
template <int S>
static int ss_launch_s(nk_ctx *ctx, int mode, int64_t n, int k, double *V, int64_t ldv, const double *coef, double *partials,
                       const int *d_skip, int grid, const ss_tail_args *tap, int *mark, int *occ_out = nullptr, int hk = 0,
                       int hs = 0, const ss_job *hjp = nullptr, int flags = 0) {
  const int ntiles = (int)((n + SS_R - 1) / SS_R);
  const int cls = ss_class(k, S);
  // "fused": a sweep whose workgroup 0 derives a block's Hessenberg columns while the others stream (its LDS: the scalar
  // workspace) — sweep C for its own block (hk = k, hs = S), sweep A for the PREVIOUS block when that was left at its first
  // pass (hk, hs: that block). The update coefficients always arrive through scalar loads from `coef`, where the reduction's
  // last workgroup (k_ss_reduce_factor) or the tail kernels left them.
  const bool fuse = tap != nullptr && (mode == 2 || mode == 0) && cls != 0;
  if (mode == 2) { hk = k; hs = S; }
  const size_t tile = ss_tile_doubles(k, S, mode != 2, cls ? cls : SS_MTMAX);
  // (the hosting workgroup streams no tiles — or, alone in the grid, is done with its tile when the scalar work starts: the
  //  workspace OVERLAYS the tile, so hosting costs the sweep no occupancy)
  ss_job hjv;
  std::memset(&hjv, 0, sizeof(hjv));
  if (hjp) hjv = *hjp;   // sweep A hosting a scalar launch in its last workgroups (the caller has checked ss_a_can_host_job)
  const size_t wsd = hjp ? ss_job_layout(hjv, hjv.mode).total : (fuse ? ss_ws_doubles(hk, hs, true) : 0);
  const size_t lds = (tile > wsd ? tile : wsd) * sizeof(double);
  ss_tail_args ta;
  std::memset(&ta, 0, sizeof(ta));
  if (tap) ta = *tap;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  const bool ev = !occ_out && ctx->prof.on && nk_prof_next(ctx, &e0, &e1);
#define SS_GO4(UPD, GRM, KM, FS, KCC)                                                                                     \
  do {                                                                                                                    \
    if (lds > 64 * 1024)                                                                                                  \
      NK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ss_block<S, UPD, GRM, KM, FS, KCC>),                   \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                                  \
    if (occ_out) {                                                                                                        \
      NK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(occ_out, k_ss_block<S, UPD, GRM, KM, FS, KCC>, SS_R, lds));     \
    } else if (ev) hipExtLaunchKernelGGL((k_ss_block<S, UPD, GRM, KM, FS, KCC>), dim3(g), dim3(SS_R), lds, ctx->stream, e0, e1, 0, n, \
                                  k, V, ldv, coef, partials, d_skip, ntiles, ta, mark, ws_off, hk, hs, hjv);              \
    else hipLaunchKernelGGL((k_ss_block<S, UPD, GRM, KM, FS, KCC>), dim3(g), dim3(SS_R), lds, ctx->stream, n, k, V, ldv, coef, \
                            partials, d_skip, ntiles, ta, mark, ws_off, hk, hs, hjv);                                     \
  } while (0)
  // the default cycle's shapes (blocks of 15 behind 1 and 16 columns) run the instances with a compile-time k
#define SS_GO3(UPD, GRM, KM, FS)                                                                                          \
  do {                                                                                                                    \
    if constexpr (S == 15 && KM == 1 && !UPD) {                                                                           \
      if (k == 1 && kc_on) SS_GO4(UPD, GRM, KM, FS, 1); else SS_GO4(UPD, GRM, KM, FS, 0);                                 \
    } else if constexpr (S == 15 && KM == 2 && !UPD) {                                                                    \
      if (k == 16 && kc_on) SS_GO4(UPD, GRM, KM, FS, 16); else SS_GO4(UPD, GRM, KM, FS, 0);                               \
    } else SS_GO4(UPD, GRM, KM, FS, 0);                                                                                   \
  } while (0)
#define SS_GO(UPD, GRM)                                                                                                   \
  do {                                                                                                                    \
    if (cls == 1) { if (fuse) SS_GO3(UPD, GRM, 1, (UPD != GRM)); else SS_GO3(UPD, GRM, 1, false); }                       \
    else if (cls == 2) { if (fuse) SS_GO3(UPD, GRM, 2, (UPD != GRM)); else SS_GO3(UPD, GRM, 2, false); }                  \
    else if (cls == 3) { if (fuse) SS_GO3(UPD, GRM, 3, (UPD != GRM)); else SS_GO3(UPD, GRM, 3, false); }                  \
    else SS_GO3(UPD, GRM, 0, false);                                                                                      \
  } while (0)
  static const int ws_off = (getenv("NK_SS_BARRIERS") && atoi(getenv("NK_SS_BARRIERS")) != 0) ? 1 : 0;
  int g = grid + (hjp ? hjv.host_wgs : 0);
  if constexpr (S == 15) {
    static const bool mm_on = !(getenv("NK_SS_MM") && atoi(getenv("NK_SS_MM")) == 0);   // A/B switch
    if (mode == 1 && mm_on && (k == 1 || k == 16) &&
        (occ_out || (int64_t)S * ldv * 8 < ((int64_t)1 << 32) - 8)) {
      const size_t tile_b = (size_t)(k + S + 1) * SS_P * sizeof(double);   // the tile + the spare column
      // tap: workgroup 0 hosts the previous block's Hessenberg work (hk, hs); its workspace overlays the tile it does not use
      const bool hostB = tap != nullptr && !occ_out;
      NK_REQUIRE(!hostB || g > 1, "internal: a hosting sweep B needs a second workgroup");
      const size_t ws_b = hostB ? ss_ws_doubles(hk, hs, true) * sizeof(double) : 0;
      const size_t lds = tile_b > ws_b ? tile_b : ws_b;
#define SS_MM(KCC, HST, NST)                                                                                              \
  do {                                                                                                                    \
    if (lds > 64 * 1024)                                                                                                  \
      NK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ss_block_mm<S, KCC, HST, NST>),                        \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                                  \
    if (occ_out) {                                                                                                        \
      NK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(occ_out, k_ss_block_mm<S, KCC, HST, NST>, SS_R, lds));          \
    } else if (ev) hipExtLaunchKernelGGL((k_ss_block_mm<S, KCC, HST, NST>), dim3(g), dim3(SS_R), lds, ctx->stream, e0, e1, 0, n, V, \
                                         ldv, coef, partials, d_skip, ntiles, mark, ta, hk, hs);                          \
    else hipLaunchKernelGGL((k_ss_block_mm<S, KCC, HST, NST>), dim3(g), dim3(SS_R), lds, ctx->stream, n, V, ldv, coef, partials, \
                            d_skip, ntiles, mark, ta, hk, hs);                                                            \
  } while (0)
      const bool nostore = (flags & 1) != 0;   // (the cycle's last block: nk_ss_cycle)
      // the read-only form behind 16 columns (k_ss_block_ro): 32-bit byte offsets over the whole basis, 16-byte row pairs
      const bool ro = nostore && (occ_out ? k == 16 : ss_b_is_read_only(V, ldv, k, S));
      if (ro) {
        const size_t tile_r = (size_t)16 * SS_P2 * sizeof(double), red_r = (size_t)4 * 2 * 256 * sizeof(double);
        size_t lds_r = tile_r > red_r ? tile_r : red_r;
        if (ws_b > lds_r) lds_r = ws_b;
#define SS_RO(HST)                                                                                                        \
  do {                                                                                                                    \
    if (lds_r > 64 * 1024)                                                                                                \
      NK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ss_block_ro<S, 16, HST>),                              \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_r));                                \
    if (occ_out) {                                                                                                        \
      NK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(occ_out, k_ss_block_ro<S, 16, HST>, SS_R, lds_r));              \
    } else if (ev) hipExtLaunchKernelGGL((k_ss_block_ro<S, 16, HST>), dim3(g), dim3(SS_R), lds_r, ctx->stream, e0, e1, 0, n, V, \
                                         ldv, coef, partials, d_skip, ntiles, mark, ta, hk, hs);                          \
    else hipLaunchKernelGGL((k_ss_block_ro<S, 16, HST>), dim3(g), dim3(SS_R), lds_r, ctx->stream, n, V, ldv, coef, partials, \
                            d_skip, ntiles, mark, ta, hk, hs);                                                            \
  } while (0)
        if (hostB) SS_RO(true); else SS_RO(false);
#undef SS_RO
      }
      else if (hostB && nostore) { NK_REQUIRE(k == 16, "internal: no hosting sweep B behind one column that stores nothing"); SS_MM(16, true, true); }
      else if (hostB) { if (k == 1) SS_MM(1, true, false); else SS_MM(16, true, false); }
      else if (nostore) { if (k == 1) SS_MM(1, false, true); else SS_MM(16, false, true); }
      else { if (k == 1) SS_MM(1, false, false); else SS_MM(16, false, false); }
#undef SS_MM
      NK_HIP(hipGetLastError());
      return NK_OK;
    }
  }
  static const bool kc_on = !(getenv("NK_SS_KCONST") && atoi(getenv("NK_SS_KCONST")) == 0);   // A/B switch
  NK_REQUIRE(!(mode == 1 && tap != nullptr), "internal: sweep B of this shape (k = %d, s = %d) cannot host a Hessenberg workgroup", k, S);
  NK_REQUIRE((flags & 1) == 0, "internal: sweep B of this shape (k = %d, s = %d) has no form that stores nothing", k, S);
  if (mode == 0) SS_GO(false, true);        // sweep A: Gram only
  else if (mode == 1) SS_GO(true, true);    // sweep B: update, then Gram of the result
  else {                                    // sweep C: update only — no LDS tile
    // as many workgroups as the register footprint lets a CU hold (≤ 6), + the Hessenberg workgroup
    if (!occ_out) {
      int per_cu = nk_ss_sweep_occupancy(ctx, 2, k, S);
      per_cu = per_cu < 1 ? 1 : (per_cu > 6 ? 6 : per_cu);
      g = ctx->num_cus * per_cu < ntiles ? ctx->num_cus * per_cu : ntiles;
      if (fuse) g += 1;
    }
    SS_GO(true, false);
  }
#undef SS_GO
#undef SS_GO3
#undef SS_GO4
  NK_HIP(hipGetLastError());
  return NK_OK;
}
// mode 0/1/2 = sweep A/B/C over V[:, 0..k) and the s columns behind them; tap != nullptr: the fused forms of B and C
static int ss_sweep_dispatch(nk_ctx *ctx, int mode, int64_t n, int k, int s, double *V, int64_t ldv, const double *coef, double *partials,
                             const int *d_skip, int grid, const ss_tail_args *tap, int *mark, int *occ_out, int hk = 0, int hs = 0,
                             const ss_job *hjp = nullptr, int flags = 0) {
  NK_REQUIRE(s >= 1 && s <= SS_SMAX && k >= 0 && k + s <= 16 * SS_MTMAX, "s-step sweep: s in 1..%d, k + s ≤ %d", SS_SMAX,
             16 * SS_MTMAX);
  NK_REQUIRE(ss_lds_bytes(k, s, true) <= 160 * 1024, "s-step sweep: %d columns do not fit the LDS tile", k + s);
  switch (s) {
    case 1: return ss_launch_s<1>(ctx, mode, n, k, V, ldv, coef, partials, d_skip, grid, tap, mark, occ_out, hk, hs, hjp, flags);
    case 2: return ss_launch_s<2>(ctx, mode, n, k, V, ldv, coef, partials, d_skip, grid, tap, mark, occ_out, hk, hs, hjp, flags);
    case 4: return ss_launch_s<4>(ctx, mode, n, k, V, ldv, coef, partials, d_skip, grid, tap, mark, occ_out, hk, hs, hjp, flags);
    case 6: return ss_launch_s<6>(ctx, mode, n, k, V, ldv, coef, partials, d_skip, grid, tap, mark, occ_out, hk, hs, hjp, flags);
    case 8: return ss_launch_s<8>(ctx, mode, n, k, V, ldv, coef, partials, d_skip, grid, tap, mark, occ_out, hk, hs, hjp, flags);
    case 15: return ss_launch_s<15>(ctx, mode, n, k, V, ldv, coef, partials, d_skip, grid, tap, mark, occ_out, hk, hs, hjp, flags);
    default: NK_FAIL(NK_E_INVALID, "internal: no s-step sweep for a block of %d columns", s);
  }
}
int nk_ss_sweep(nk_ctx *ctx, int mode, int64_t n, int k, int s, double *V, int64_t ldv, const double *coef, double *partials,
                const int *d_skip, int grid, const ss_tail_args *tap, int *mark, int hk, int hs, int flags) {
  return ss_sweep_dispatch(ctx, mode, n, k, s, V, ldv, coef, partials, d_skip, grid, tap, mark, nullptr, hk, hs, nullptr, flags);
}
// sweep A of a block whose last `hj.host_wgs` workgroups are the scalar launch that closes the previous block (ss_a_can_host_job)
static int ss_sweep_a_hosting_job(nk_ctx *ctx, int64_t n, int k, int s, double *V, int64_t ldv, double *partials, const int *d_skip,
                                  int grid, const ss_tail_args &pend_ta, int *mark, const ss_job &hj) {
  return ss_sweep_dispatch(ctx, 0, n, k, s, V, ldv, nullptr, partials, d_skip, grid, &pend_ta, mark, nullptr, 0, 0, &hj);
}
// workgroups of sweep `mode` for a block of s columns behind k that a CU holds at once (cached per shape)
int nk_ss_sweep_occupancy(nk_ctx *ctx, int mode, int k, int s) {
  static int cache[3][SS_SMAX + 1][16 * SS_MTMAX + 1];
  if (mode < 0 || mode > 2 || s < 1 || s > SS_SMAX || k < 0 || k + s > 16 * SS_MTMAX) return 1;
  int &c = cache[mode][s][k + s];
  if (c == 0) {
    int occ = 0;
    ss_tail_args ta;
    std::memset(&ta, 0, sizeof(ta));
    // (sweep C is asked about in its fused form when the shape has one: the Hessenberg workspace is part of its LDS)
    if (ss_sweep_dispatch(ctx, mode, 1, k, s, nullptr, 0, nullptr, nullptr, nullptr, 1, (mode == 2 && ss_class(k, s) != 0) ? &ta : nullptr,
                          nullptr, &occ) != NK_OK || occ < 1)
      occ = 1;
    c = occ;
  }
  return c;
}
// widths the sweeps are compiled for; any other block is cut into these (the last block of a cycle, odd block sizes)
int nk_ss_block_width(int want) {   // (round 6: 3, 5, 7, 10 and 12 left the list — 45 % of the sweep instantiations for ragged tails only)
  if (want >= 15) return 15;
  if (want >= 8) return 8;
  if (want >= 6) return 6;
  if (want >= 4) return 4;
  return want >= 2 ? 2 : 1;
}

// ----------------------------------------------------------------------------- the scalar work of the block scheme: one launch
// Everything between two sweeps that is NOT a sweep, in one kind of launch (round 5; rounds 3–4: k_ss_reduce_factor after every
// sweep + k_ss_hess + k_backsolve, 121 µs of a 532 µs Newton step in seven one-workgroup critical sections).
//   * stage-2 reduction: one wavefront per entry of up to TWO partial Gram blocks — set 0: this block's sweep A (its first
//     pass), set 1: the PREVIOUS block's sweep B (its second pass, deferred: nothing in front of the next block's sweep A needs
//     those factors, so that block's reduce-and-factor launch is gone and the reduction rides here) — fixed order, one ticket
//     per workgroup; several ranks on peer-mapped arenas: this is also the all-reduce (ONE message for both sets);
//   * the workgroup that draws the last ticket requests EVERYTHING its serial phases will read in one round trip (both reduced
//     blocks, the column scales, the earlier blocks' factors, the Hessenberg work's inputs, the rotated factor for the
//     back-substitution) and then runs, as the job's bits say:
//       SSJ_F2    the pending block's second factorisation (+ the first-pass departure test when it is left at that pass) → C₂, R₂
//       SSJ_PREP  its Wi = R₂⁻¹, D = C₂R₂⁻¹ (what the next blocks' reductions apply), kept in LDS for the factorisation below
//       SSJ_COEF2 … and the coefficients of an explicit third sweep (blocks that are not left at their first pass)
//       SSJ_F1    this block's first factorisation → the coefficients of sweep B, C₁, R₁
//       SSJ_HESS  the pending block's Hessenberg columns, rotations and stopping test (otherwise: hosted by a sweep)
//       SSJ_BACK  the cycle's back-substitution, the outcome published to the host (the cycle's LAST launch of this kind:
//                 reduce → factor → Hessenberg → y without leaving the workgroup's LDS — three launches and their global
//                 round trips in rounds 3–4).
// A failure (lost pivot, departure) or a cycle that was done before this launch skips everything but the back-substitution.
__host__ __device__ inline ss_job_lds ss_job_layout(const ss_job &j, int mode) {
  ss_job_lds L;
  const bool f1 = (mode & SSJ_F1) != 0, f2 = (mode & SSJ_F2) != 0, hs = (mode & SSJ_HESS) != 0, bk = (mode & SSJ_BACK) != 0;
  size_t o = (size_t)j.nslots0 + j.nslots1;
  L.sc = o; o += (size_t)((f1 && j.k0 > j.k1) ? j.k0 : (f2 ? j.k1 : j.k0)) + 1;
  L.fix = o; o += ss_fixc_doubles(j.cfix);
  L.w1 = o; o += f2 ? ss_ws_doubles(j.k1, j.sb1, hs) : 0;
  L.w0 = o; o += f1 ? ss_ws_doubles(j.k0, j.sb0, false) : 0;
  L.LK = j.m;   // (the LDS copy keeps the global pitch: it arrives through lane-linear LDS-direct loads)
  L.sR = o; o += bk ? (size_t)j.m * L.LK : 0;
  L.sg = o; o += bk ? (size_t)j.m + 2 : 0;
  L.verdict = o; o += 8;
  L.bfix = o; o += bk ? ss_fixc_doubles(j.bfx) : 0;
  L.rdv = o; o += bk ? (size_t)j.m + 16 * NK_SS_NFIX : 0;   // reciprocal diagonals of the back-substitution's factors
  L.total = o;
  return L;
}
// y = R⁻¹ g, then the blocks left at their first pass, last first: coefficients on [V_true Q] → on V_true and the block's stored
// columns (nk_gmres.hip's k_backsolve, on operands that are in LDS already). bc: (C₂, R₂) per block; rdv: k + 16 per block doubles.
// The serial part runs on one wavefront at ONE instruction per ≈ 5 cycles — what counts is the number of instructions per step
// (round 5's first form: 30 per step, 90 steps, 9 µs). So everything that is not the chain is done up front by the whole workgroup:
// the reciprocal diagonals, and every triangular factor scaled by them with its diagonal and lower part zeroed — a step of a solve
// is then {the next entry's LDS read (one step ahead), two v_readlane, one multiply-add}: no division, no select, no mask.
// (Not inlined: inlined into k_ss_job the compiler of ROCm 7.2 stops with "Illegal instruction detected: V_CMP_NE_U32_e32 0,
//  $src_shared_base" — a null test of a generic pointer it has itself proven to be an LDS one; the LDS block and the block list
//  arrive as address-space-3 pointers so that every access stays a DS instruction.) All 256 threads call it.
typedef __attribute__((address_space(3))) double *ss_lds_ptr;
typedef const __attribute__((address_space(3))) ss_fixc *ss_lds_fixc;
__device__ __noinline__ void ss_backsolve(int k, int failed, int oR, int LK, int og, int ordv, double *y, int m, ss_lds_fixc bcp,
                                          ss_lds_ptr lds) {
  const int t = threadIdx.x;
  k = __builtin_amdgcn_readfirstlane(k); failed = __builtin_amdgcn_readfirstlane(failed);
  oR = __builtin_amdgcn_readfirstlane(oR); LK = __builtin_amdgcn_readfirstlane(LK); og = __builtin_amdgcn_readfirstlane(og);
  ordv = __builtin_amdgcn_readfirstlane(ordv); m = __builtin_amdgcn_readfirstlane(m);
  const int nb = __builtin_amdgcn_readfirstlane(bcp->n);
  ss_lds_ptr sR = lds + oR, sg = lds + og, rdv = lds + ordv;
  if (failed) {
    if (t < m) y[t] = 0.0;
    return;
  }
  // reciprocal diagonals
  if (t < k) rdv[t] = 1.0 / sR[t * LK + t];
  for (int bq = 0; bq < nb; ++bq) {
    const int fsb = __builtin_amdgcn_readfirstlane(bcp->sb[bq]), oW = __builtin_amdgcn_readfirstlane(bcp->oWi[bq]);
    const int u = t - 64 - 16 * bq;
    if (u >= 0 && u < fsb) rdv[k + 16 * bq + u] = 1.0 / lds[oW + u * fsb + u];
  }
  __syncthreads();
  // scaled strictly upper parts, the rest zero
  for (int r = t >> 5; r < k; r += 8) {
    const double rd = rdv[r];
    for (int c = t & 31; c < k; c += 32) {
      const double v = sR[r * LK + c];
      sR[r * LK + c] = c > r ? v * rd : 0.0;
    }
  }
  for (int bq = 0; bq < nb; ++bq) {
    const int fsb = __builtin_amdgcn_readfirstlane(bcp->sb[bq]), oW = __builtin_amdgcn_readfirstlane(bcp->oWi[bq]);
    const int a = t >> 4, c = t & 15;
    if (a < fsb && c < fsb) {
      const double v = lds[oW + a * fsb + c];
      lds[oW + a * fsb + c] = c > a ? v * rdv[k + 16 * bq + a] : 0.0;
    }
  }
  __syncthreads();
  if (t >= 64) return;
  double gv = (t < k) ? sg[t] * rdv[t] : 0.0;
  if (k > 0) {
    ss_lds_ptr row = sR + (t < k ? t : k - 1) * LK;   // (lanes ≥ k: the last row — all zero now)
    // eight steps per round: their eight entries are requested together (one LDS latency per round, not per step); steps past
    // the end multiply a zero
#pragma unroll 1
    for (int i0 = k - 1; i0 >= 1; i0 -= 8) {
      double rv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) rv[u] = (i0 - u >= 1) ? row[i0 - u] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u) gv = __builtin_fma(-rv[u], ss_readlane(gv, i0 - u >= 1 ? i0 - u : 0), gv);
    }
  }
#pragma unroll 1
  for (int bq = nb - 1; bq >= 0; --bq) {
    const int fk0 = __builtin_amdgcn_readfirstlane(bcp->k0[bq]), fsb = __builtin_amdgcn_readfirstlane(bcp->sb[bq]);
    const int oW = __builtin_amdgcn_readfirstlane(bcp->oWi[bq]), oD = __builtin_amdgcn_readfirstlane(bcp->oD[bq]);
    if (k <= fk0) continue;   // (a cycle that ended before the block: nothing of it in y)
    const int cc = t - fk0;
    const bool inb = cc >= 0 && cc < fsb;
    ss_lds_ptr r2row = lds + oW + (inb ? cc : fsb - 1) * fsb;   // (other lanes: the last row — all zero now)
    ss_lds_ptr c2row = lds + oD + (t < fk0 ? t : 0) * fsb;
    if (inb) gv *= rdv[k + 16 * bq + cc];
#pragma unroll 1
    for (int c0 = fsb - 1; c0 >= 1; c0 -= 8) {       // b = R₂⁻¹ y_Q on lanes k0 … k0 + sb − 1 (y is zero from k on)
      double rv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) rv[u] = (c0 - u >= 1) ? r2row[c0 - u] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u) gv = __builtin_fma(-rv[u], ss_readlane(gv, fk0 + (c0 - u >= 1 ? c0 - u : 0)), gv);
    }
    double acc = 0.0;
#pragma unroll 1
    for (int c0 = 0; c0 < fsb; c0 += 8) {
      double cvv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) cvv[u] = (c0 + u < fsb) ? c2row[c0 + u] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = __builtin_fma(cvv[u], ss_readlane(gv, fk0 + (c0 + u < fsb ? c0 + u : 0)), acc);
    }
    if (t < fk0) gv -= acc;
  }
  if (t < m) y[t] = gv;
}
__device__ void ss_publish_outcome(nk_gmres_pub *pub, uint64_t seq, const uint64_t *peer_err, int k, int converged, int failed,
                                   double rnorm0, double rnorm) {
  if (pub == nullptr) return;
  pub->k = k;
  pub->converged = converged;
  pub->failed = failed;
  pub->rnorm0 = rnorm0;
  pub->rnorm = rnorm;
  pub->pad = peer_err ? (int)(__hip_atomic_load(peer_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0x7fffffff) : 0;
  __threadfence_system();
  __hip_atomic_store(&pub->end_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// requests of the back-substitution: the rotated factor, g, the earlier blocks' (C₂, R₂) (all but the last `skip_last`)
__device__ void ss_back_request(const ss_job &j, const ss_job_lds &L, double *lds, ss_fixc *bc, int skip_last) {
  const int t = threadIdx.x, m = j.m;
  double *sR = lds + L.sR, *sg = lds + L.sg;
  for (int e = t; e < m * m; e += SS_R) {
    const int i = e / m, c = e - i * m;
    sR[i * L.LK + c] = j.Rg[(size_t)i * m + c];
  }
  for (int e = t; e <= m; e += SS_R) sg[e] = j.g[e];
  if (t == 0) lds[L.verdict + 4] = j.ctl->rnorm0;
  ss_fixc_request(bc, j.bfx, lds, (int)L.bfix, skip_last, true);
}
// the back-substitution alone (the cycle was done before this launch): every operand from global memory
__device__ void ss_back_only(const ss_job &j, const ss_job_lds &L, double *lds, ss_fixc *bc, ss_lds_ptr lds3, ss_lds_fixc bc3) {
  ss_back_request(j, L, lds, bc, 0);
  double *vd = lds + L.verdict;
  if (threadIdx.x == 0) {
    vd[0] = (double)j.ctl->k; vd[1] = (double)j.ctl->converged; vd[2] = (double)j.ctl->failed; vd[3] = j.ctl->rnorm;
  }
  __syncthreads();
  if (threadIdx.x == 0) ss_publish_outcome(j.pub, j.seq, j.peer_err, (int)vd[0], (int)vd[1], (int)vd[2], vd[4], vd[3]);
  ss_backsolve((int)vd[0], (int)vd[2], (int)L.sR, L.LK, (int)L.sg, (int)L.rdv, j.y, j.m, bc3, lds3);
}
// ---- ONE memory round trip for everything the last workgroup's serial phases read from global memory.
// A loop `for (e = t; e < count; e += 256) lds[e] = global[e]` makes one round trip PER ITERATION and per list (the LDS store
// behind a load waits for it, the next load sits behind the store): 8.9 µs for the 15 lists of the cycle's last launch. Staging
// every list through registers instead (all requests, then all stores) needs the lists unrolled — and this code runs ONCE per
// launch on one workgroup, from a cold instruction cache: straight-line code is paid for per byte fetched (measured: the
// unrolled form was slower than the loops). So the lists go through the LDS-direct loads of gfx950 (global_load_lds_dword: a
// wavefront's 64 lanes deliver 64 consecutive dwords at a wave-uniform LDS base, every lane from its own global address):
// compact rolled loops that only ISSUE — nothing waits until the barrier behind all of them (which carries vmcnt(0)).
template <class F, int AUX = 0>
__device__ __forceinline__ void ss_gather_async(unsigned lbase /* LDS byte address */, int count, F src_of /* e → const double * */) {
  const int t = threadIdx.x, lane = t & 63;
  for (int d0 = t - lane; d0 < 2 * count; d0 += SS_R) {   // d0: the wavefront's first dword of this round
    const int d = d0 + lane;
    const unsigned la = __builtin_amdgcn_readfirstlane(lbase + 4u * (unsigned)d0);   // (M0: wave-uniform)
    if (d < 2 * count) {
      const double *p = src_of(d >> 1);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)((const char *)p + 4 * (d & 1)),
                                       (__attribute__((address_space(3))) void *)(size_t)la, 4, 0, AUX);
    }
  }
}
__device__ __forceinline__ void ss_fixc_carve(ss_fixc *fc, const nk_ss_fix &fix, int off) {
#pragma unroll
  for (int q = 0; q < NK_SS_NFIX; ++q) {
    if (q < fix.n) {
      const int oD = off; off += fix.k0[q] * fix.sb[q];
      const int oWi = off; off += fix.sb[q] * fix.sb[q];
      if (threadIdx.x == 0) { fc->k0[q] = fix.k0[q]; fc->sb[q] = fix.sb[q]; fc->oD[q] = oD; fc->oWi[q] = oWi; }
    }
  }
  if (threadIdx.x == 0) fc->n = fix.n;
}
// the cached factors of a list of blocks: its first `nload` slots from global memory ((D, Wi), or (C₂, R₂))
__device__ __forceinline__ void ss_fixc_gather(const nk_ss_fix &fix, unsigned b /* LDS byte address */, int nload, bool c2r2) {
#pragma unroll
  for (int q = 0; q < NK_SS_NFIX; ++q) {
    if (q < fix.n) {
      const unsigned sD = b; b += 8u * (unsigned)(fix.k0[q] * fix.sb[q]);
      const unsigned sWi = b; b += 8u * (unsigned)(fix.sb[q] * fix.sb[q]);
      if (q < nload) {
        const double *gD = c2r2 ? fix.C2[q] : fix.D[q], *gW = c2r2 ? fix.R2[q] : fix.Wi[q];
        ss_gather_async(sD, fix.k0[q] * fix.sb[q], [=](int e) { return gD + e; });
        ss_gather_async(sWi, fix.sb[q] * fix.sb[q], [=](int e) { return gW + e; });
      }
    }
  }
}
template <int MODE>
__device__ __forceinline__ void ss_job_request(const ss_job &j, const ss_tail_args &ta1, const ss_job_lds &L, double *lds, ss_fixc *fc,
                                               ss_fixc *bc, const ss_ws &w1, bool peer, unsigned lds0 /* LDS byte address of `lds` */) {
  constexpr bool f1 = (MODE & SSJ_F1) != 0, f2 = (MODE & SSJ_F2) != 0, hs = f2 && (MODE & SSJ_HESS) != 0, bk = (MODE & SSJ_BACK) != 0;
  const int t = threadIdx.x;
  const bool prep = f2 && (MODE & SSJ_PREP) != 0 && ta1.Wi != nullptr;
  const ss_ws_offs wo = ss_ws_off(f2 ? j.k1 : 0, f2 ? j.sb1 : 0);
  auto at = [=](int o) { return lds0 + 8u * (unsigned)(w1.o0 + o); };   // (an array of the pending block's workspace as an LDS address)
  ss_fixc_carve(fc, j.cfix, (int)L.fix);
  if (bk) ss_fixc_carve(bc, j.bfx, (int)L.bfix);
  if (!peer) {   // both reduced blocks (written by other workgroups: the acquire behind the ticket has invalidated the L1)
    const double *red = j.red;
    auto src = [=](int e) { return red + e; };
    ss_gather_async<decltype(src), 16>(lds0, j.nslots0 + j.nslots1, src);   // (sc1: written write-through by other workgroups of this launch)
  }
  {
    const int ksc = (f1 && j.k0 > j.k1) ? j.k0 : (f2 ? j.k1 : j.k0);
    const double *sc = j.sc;
    ss_gather_async(lds0 + 8u * (unsigned)L.sc, ksc, [=](int e) { return sc + e; });
  }
  ss_fixc_gather(j.cfix, lds0 + 8u * (unsigned)L.fix, j.cfix.n - ((f1 && prep) ? 1 : 0), false);   // (the pending block's slot is computed here)
  if (hs) {   // what ss_hess_load requests
    const int k = j.k1, sb = j.sb1, ko = k - 1, m = j.m;
    const double *pH = ta1.H, *pcs = ta1.cs, *psn = ta1.sn, *pg = j.g, *pR1 = ta1.R1, *pC1 = ta1.C1, *scal = ta1.scal;
    const double *ptol = &j.ctl->tol;
    ss_gather_async(at(wo.Hs), k * ko, [=](int e) { return pH + (size_t)(e / ko) * m + (e % ko); });
    ss_gather_async(at(wo.scs), ko, [=](int e) { return pcs + e; });
    ss_gather_async(at(wo.ssn), ko, [=](int e) { return psn + e; });
    ss_gather_async(at(wo.sg), ko + 1, [=](int e) { return pg + e; });
    ss_gather_async(at(wo.R1s), sb * sb, [=](int e) { return pR1 + e; });
    ss_gather_async(at(wo.F), k * sb, [=](int e) { return pC1 + e; });
    ss_gather_async(at(wo.Fx), SS_SMAX + 2, [=](int e) { return e < SS_SMAX ? scal + SS_TH + e : (e == SS_SMAX ? scal + 2 : ptol); });
    if (ta1.usb > 0) {
      const int usb = ta1.usb, uk0 = ta1.uk0;
      const double *puC = ta1.uC2, *puR = ta1.uR2;
      ss_gather_async(at(wo.uu), k, [=](int e) { return e < uk0 ? puC + e * usb + usb - 1 : puR + (e - uk0) * usb + usb - 1; });
    } else if (t < k) {
      w1.uu[t] = (t == k - 1) ? 1.0 : 0.0;
    }
  }
  if (bk) {   // the rotated factor (pitch m in LDS as well), g, ‖r₀‖, the earlier blocks' (C₂, R₂)
    const int m = j.m;
    const double *pRg = j.Rg, *pg = j.g, *prn0 = &j.ctl->rnorm0;
    // (the block's own columns are written by the Hessenberg work of this launch: rows ko … of the old columns are never read)
    ss_gather_async(lds0 + 8u * (unsigned)L.sR, (hs ? j.k1 - 1 : m) * m, [=](int e) { return pRg + e; });
    ss_gather_async(lds0 + 8u * (unsigned)L.sg, m + 1, [=](int e) { return pg + e; });
    ss_gather_async(lds0 + 8u * (unsigned)(L.verdict + 4), 1, [=](int e) { return prn0 + e; });
    ss_fixc_gather(j.bfx, lds0 + 8u * (unsigned)L.bfix, j.bfx.n - (f2 ? 1 : 0), true);
  }
}
// MODE: the job's bits at compile time — each combination the cycle uses is an instance of its own. This code runs once per
// launch from a cold instruction cache: what an instance does not do must not be in it.
// (slot, nwg: this workgroup's place among the job's workgroups — a launch of its own: (blockIdx.x, gridDim.x); hosted by a sweep A:
//  the sweep's last nwg workgroups, each of which then walks several rounds of entries)
template <bool PEER, int MODE>
__device__ __forceinline__ void ss_job_body(const ss_job &j, const ss_tail_args &ta0, const ss_tail_args &ta1, const nk_peer_ar_view &pv,
                                            int slot, int nwg, double *s_rf) {
  __shared__ unsigned int s_last;
  __shared__ ss_fixc s_fc, s_bc;
  const int skip = (j.d_skip != nullptr) ? *j.d_skip : 0;
  const int t = threadIdx.x;
  constexpr bool f1 = (MODE & SSJ_F1) != 0, f2 = (MODE & SSJ_F2) != 0, hs = f2 && (MODE & SSJ_HESS) != 0, bk = (MODE & SSJ_BACK) != 0;
  SS_STAMP(0);
  // one wavefront per entry (four entries per workgroup): fixed order — lane l adds partials l, l + 64, …, then the
  // butterfly —, and one ticket per workgroup (465 same-address atomics of a workgroup-per-entry launch took 6 µs)
  const int wv = t >> 6, lane = t & 63;
  const int nslots = j.nslots0 + j.nslots1;
  const ss_job_lds L = ss_job_layout(j, MODE);
  // (the LDS block and the block list as address-space-3 pointers, taken from the symbols themselves: a cast of a generic
  //  pointer further down trips the compiler — see ss_backsolve)
  const ss_lds_ptr lds3 = (ss_lds_ptr)s_rf;
  const ss_lds_fixc bc3 = (ss_lds_fixc)&s_bc;
  for (int entry = slot * 4 + wv; entry < nslots; entry += nwg * 4) {   // (a launch of its own: one round)
    double v = 0.0;
    const bool first = entry < j.nslots0;
    const int nblk = first ? j.nblk0 : j.nblk1;
    const double *p = first ? j.part0 + (size_t)entry * nblk : j.part1 + (size_t)(entry - j.nslots0) * nblk;
    for (int base = 0; base < nblk; base += 512) {   // eight loads in flight per lane (a rolled loop made eight round trips)
      double x[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int i = base + lane + 64 * q;
        x[q] = p[i < nblk ? i : nblk - 1];
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) v += (base + lane + 64 * q < nblk) ? x[q] : 0.0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) {
      if (PEER) {
        const int par = (int)(pv.seq & 1);
        for (int q = 0; q < pv.P; ++q) reinterpret_cast<nk_peer_hdr *>(pv.map[q])->ar_data[par][pv.me][entry] = v;
        __threadfence_system();
      } else {   // write-through: the hand-off below needs no L2 write-back
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(&j.red[entry]), (unsigned long long)__double_as_longlong(v),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
  }
  // (the flag was requested together with the partials: one round trip; a collective runs on every rank even when the cycle
  //  is done.) The cycle may have ended INSIDE a sweep in front of this launch (a hosted Hessenberg workgroup's stopping test):
  //  the sweeps and Hessenberg launches behind a skipped job look at pad1.
  if (skip && slot == 0 && t == 0) j.ctl->pad1 = 1;
  if (skip && !PEER) {
    if (bk && slot == 0) ss_back_only(j, L, s_rf, &s_bc, lds3, bc3);
    return;
  }
  // hand-off to the last workgroup (any XCD): write-through (sc1) stores, drained by their wavefronts → barrier → ticket; the
  // workgroup that takes the last ticket reads them with sc1 loads (MI355X_MICROARCH.md: "sc1 payload → drained → flag"; rounds
  // 3–4: plain stores + an agent-scope release fence = an L2 write-back behind a sweep that left the L2 full of dirty lines)
  __syncthreads();
  if (t == 0) s_last = (atomicAdd(j.ticket, 1u) == (unsigned)nwg - 1u) ? 1u : 0u;
  __syncthreads();
  if (!s_last) return;
  SS_STAMP(1);
  if (t == 0) *j.ticket = 0u;   // (for the next launch: the kernel boundary publishes it)
  if (PEER) {
    const int par = (int)(pv.seq & 1);
    nk_peer_hdr *mine = reinterpret_cast<nk_peer_hdr *>(pv.map[pv.me]);
    __threadfence_system();
    if (t < pv.P) {
      __hip_atomic_store(&reinterpret_cast<nk_peer_hdr *>(pv.map[t])->ar_flag[par][pv.me], pv.seq, __ATOMIC_RELEASE,
                         __HIP_MEMORY_SCOPE_SYSTEM);
      const unsigned long long t0 = wall_clock64();
      while (__hip_atomic_load(&mine->ar_flag[par][t], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < pv.seq) {
        if (__hip_atomic_load(&mine->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 4) break;
        if (wall_clock64() - t0 > nk_peer_timeout(&mine->err)) { atomicAdd((unsigned long long *)&mine->err, 1ull); break; }
        __builtin_amdgcn_s_sleep(2);
      }
    }
    __syncthreads();
    for (int e = t; e < nslots; e += SS_R) {   // combined in rank order, straight into LDS
      double acc = mine->ar_data[par][0][e];
      for (int q = 1; q < pv.P; ++q) acc += mine->ar_data[par][q][e];
      s_rf[e] = acc;
    }
    if (skip) {
      __syncthreads();
      if (bk) ss_back_only(j, L, s_rf, &s_bc, lds3, bc3);
      return;
    }
  }
  // ---- the last workgroup. Everything the serial phases read from global memory rides in the same round trip: both reduced
  // blocks (written by other workgroups: read past the L1), the column scales, the earlier blocks' factors (the pending
  // block's slot is filled below), the Hessenberg work's inputs, the back-substitution's
  const double *red0 = s_rf, *red1 = s_rf + j.nslots0;
  double *s_sc = s_rf + L.sc, *vd = s_rf + L.verdict;
  const bool prep = f2 && (MODE & SSJ_PREP) != 0 && ta1.Wi != nullptr;
  const ss_ws w1 = ss_ws_carve(s_rf + L.w1, f2 ? j.k1 : 0, f2 ? j.sb1 : 0, hs, (int)L.w1);
  const ss_ws w0 = ss_ws_carve(s_rf + L.w0, f1 ? j.k0 : 0, f1 ? j.sb0 : 0, false, (int)L.w0);
  ss_job_request<MODE>(j, ta1, L, s_rf, &s_fc, &s_bc, w1, PEER, (unsigned)(size_t)(__attribute__((address_space(3))) char *)s_rf);
  __syncthreads();
  SS_STAMP(2);
  bool alive = true;
  if (f2) {
    alive = ss_factor(j.k1, j.sb1, red1, s_sc, w1, s_fc, ta1.fix.n, ta1.ptol, s_rf);
    // (left at its first pass only if that pass was good)
    if (alive && ta1.Wi != nullptr && !ss_first_pass_departure_ok(j.k1, j.sb1, w1, 0.1)) alive = false;
    if (alive) {
      for (int e = t; e < j.k1 * j.sb1; e += SS_R) ta1.C2[e] = w1.Ct[e];
      if (t < j.sb1 * j.sb1) ta1.R2[t] = w1.Rm[t];
      if (MODE & SSJ_COEF2) {
        for (int e = t; e < j.k1 * j.sb1; e += SS_R) j.coef[e] = w1.U[e];
        if (t < j.sb1 * j.sb1) j.coef[(size_t)j.k1 * j.sb1 + t] = w1.Ri[t];
      }
      // the next block's matrix powers start from a column of unit scale
      if (t == 0) ta1.scal[0] = 1.0 / ta1.scal[2];
      if (prep) {
        const int q = ta1.fix.n;   // the pending block's slot in this block's list (= its position among the earlier blocks)
        ss_fix_prepare(j.k1, j.sb1, w1.Ct, w1.Rm, w1.Sm, ta1.Wi, ta1.D, f1, s_rf + (f1 ? s_fc.oWi[q] : 0), s_rf + (f1 ? s_fc.oD[q] : 0));
      }
    }
  }
  SS_STAMP(3);
  if (alive && f1) {
    if (f2) {   // the pending block's columns are normalised (its Hessenberg work, which says so in `sc`, may not have run yet)
      for (int e = j.k1 + t; e < j.k0; e += SS_R) s_sc[e] = 1.0;
      __syncthreads();
    }
    alive = ss_factor(j.k0, j.sb0, red0, s_sc, w0, s_fc, ta0.fix.n, ta0.ptol, s_rf);
    if (alive) {
      for (int e = t; e < j.k0 * j.sb0; e += SS_R) j.coef[e] = w0.U[e];
      if (t < j.sb0 * j.sb0) j.coef[(size_t)j.k0 * j.sb0 + t] = w0.Ri[t];
      ss_keep_pass1(j.k0, j.sb0, w0, ta0, red0);
      if (t == 0) ta0.scal[0] = 1.0 / ta0.scal[2];   // (this block's powers have run; the next block's start from a unit column)
    }
  }
  if (!alive) ss_fail(j.ctl, j.pub, j.seq);
  SS_STAMP(4);
  if (alive && hs) ss_hessenberg(j.k1, j.sb1, w1, ta1, true, s_rf + L.sR, bk ? L.LK : 0, f1, vd);
  if (bk) {
    __syncthreads();
    if (t == 0) {
      if (!(alive && hs)) {   // (a failure inside this launch: the columns closed before this block count, x stays as it is)
        vd[0] = (double)j.ctl->k; vd[1] = 0.0; vd[2] = alive ? (double)j.ctl->failed : 2.0; vd[3] = j.ctl->rnorm;
      }
    }
    if (alive && f2 && t == 0) {   // the pending block's factors: in LDS already
      const int q = j.bfx.n - 1;
      if (j.raw_last && hs) {
        // its sweep B stored nothing: the block's columns in memory are the matrix powers' X = V_true C + Q R, with C = C₁ + C₂R₁ and
        // R = R₂R₁ — the coordinates the Hessenberg recovery has just formed (F: C in rows 0 … k − 1, R in rows k …). The same
        // operation as for a block left at its first pass, Q y = X (R⁻¹ y) − V_true (C R⁻¹ y), with (C, R) in place of (C₂, R₂)
        const int oF = w1.o0 + ss_ws_off(j.k1, j.sb1).F;
        s_bc.oD[q] = oF;
        s_bc.oWi[q] = oF + j.k1 * j.sb1;
      } else {   // (C₂, R₂): the stored columns are the first pass's Q₁
        s_bc.oD[q] = w1.o0;
        s_bc.oWi[q] = w1.o0 + ss_ws_off(j.k1, j.sb1).Rm;
      }
    }
    __syncthreads();
    SS_STAMP(14);
    ss_backsolve((int)vd[0], (int)vd[2], (int)L.sR, L.LK, (alive && hs) ? w1.o0 + ss_ws_off(j.k1, j.sb1).sg : (int)L.sg, (int)L.rdv, j.y, j.m, bc3, lds3);
    // the outcome goes to the host from the LAST wavefront, which has left the back-substitution behind its parallel phase: the
    // system-scope fence (≈ 1.2 µs) runs beside the first wavefront's dependent chain instead of in front of it
    if (t == SS_R - 1) ss_publish_outcome(j.pub, j.seq, j.peer_err, (int)vd[0], (int)vd[1], (int)vd[2], vd[4], vd[3]);
    SS_STAMP(15);
  }
#ifdef NK_SS_STAMPS
  if (t == 0 && g_ss_stamp != nullptr) {
    const int bank = bk ? 2 : (((f1 && f2) || j.host_wgs > 0) ? 1 : 0);   // (1: both factorisations — or the job a sweep A hosts)
    for (int i = 0; i < 16; ++i) g_ss_stamp[16 * bank + i] = s_ss_stamps[i];
  }
#endif
}
template <bool PEER, int MODE>
__global__ __launch_bounds__(SS_R) void k_ss_job(const ss_job j, const ss_tail_args ta0, const ss_tail_args ta1, const nk_peer_ar_view pv) {
  extern __shared__ double s_rf[];
  ss_job_body<PEER, MODE>(j, ta0, ta1, pv, blockIdx.x, gridDim.x, s_rf);
}
extern "C" int nk_ss_debug_stamps(int enable, unsigned long long *out5) {
  static unsigned long long *d_st = nullptr;
  if (enable && !d_st) {
    if (hipMalloc(&d_st, 48 * sizeof(unsigned long long)) != hipSuccess) return NK_E_NOMEM;
    hipMemset(d_st, 0, 48 * sizeof(unsigned long long));   // (development hook: the caller synchronises the device)
    hipDeviceSynchronize();
    hipMemcpyToSymbol(HIP_SYMBOL(g_ss_stamp), &d_st, sizeof(d_st));
  }
  if (!enable && d_st) {
    unsigned long long *z = nullptr;
    hipMemcpyToSymbol(HIP_SYMBOL(g_ss_stamp), &z, sizeof(z));
  }
  if (out5 && d_st) {
    hipDeviceSynchronize();
    hipMemcpy(out5, d_st, 48 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  }
  return NK_OK;
}

// ----------------------------------------------------------------------------- development harness (not in the public header)
// Runs one sweep on caller data: V (n × (k+s), column-major, leading dimension n, HOST), coef = [U ; R] (k·s + s·s, HOST; the
// sweep computes X ← (X − V U) R⁻¹); returns the
// updated columns, the reduced Gram block ((k+s) × s, row-major) and the average kernel time over `iters` launches.
extern "C" int nk_ss_sweep_test(nk_ctx *ctx, int mode_in, int64_t n, int k, int s, double *V_host, const double *coef_host,
                                double *gram_out, int iters, double *avg_us) {
  NK_REQUIRE(ctx && V_host, "NULL argument");
  const int mode = mode_in == 3 ? 1 : mode_in, flags = mode_in == 3 ? 1 : 0;   // (3: the sweep B that stores nothing — the Gram block only)
  NK_HIP(hipSetDevice(ctx->device));
  double *dV = nullptr, *dc = nullptr, *dp = nullptr;
  const int grid = mode == 0 ? nk_ss_grid_a(ctx, n, k, s, false) : nk_ss_grid(ctx, n, k, s);
  const size_t nv = (size_t)n * (k + s), nslots = (size_t)(k + s) * s;
  NK_TRY(nk_dev_alloc(&dV, nv));
  NK_TRY(nk_dev_alloc(&dc, (size_t)k * s + s * s + 1));
  NK_TRY(nk_dev_alloc(&dp, nslots * grid + 1));
  NK_HIP(nk_memcpy(ctx, dV, V_host, nv * sizeof(double), hipMemcpyHostToDevice));
  if (coef_host) {  // [U ; R] as the caller wrote them → R with a reciprocal diagonal, as the sweeps take it
    std::vector<double> hc(coef_host, coef_host + (size_t)k * s + (size_t)s * s);
    for (int c = 0; c < s; ++c) hc[(size_t)k * s + (size_t)c * s + c] = 1.0 / hc[(size_t)k * s + (size_t)c * s + c];
    NK_HIP(nk_memcpy(ctx, dc, hc.data(), hc.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  NK_TRY(nk_ss_sweep(ctx, mode, n, k, s, dV, n, dc, dp, nullptr, grid, nullptr, nullptr, 0, 0, flags));
  NK_HIP(hipStreamSynchronize(ctx->stream));
  NK_HIP(nk_memcpy(ctx, V_host, dV, nv * sizeof(double), hipMemcpyDeviceToHost));
  if (mode != 2 && gram_out) {
    std::vector<double> hp(nslots * grid);
    NK_HIP(nk_memcpy(ctx, hp.data(), dp, hp.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (size_t e = 0; e < nslots; ++e) {
      double a = 0.0;
      for (int b = 0; b < grid; ++b) a += hp[e * grid + b];
      gram_out[e] = a;
    }
  }
  if (iters > 0 && avg_us) {
    hipEvent_t e0, e1;
    NK_HIP(hipEventCreate(&e0));
    NK_HIP(hipEventCreate(&e1));
    NK_HIP(hipEventRecord(e0, ctx->stream));
    for (int i = 0; i < iters; ++i) NK_TRY(nk_ss_sweep(ctx, mode, n, k, s, dV, n, dc, dp, nullptr, grid, nullptr, nullptr, 0, 0, flags));
    NK_HIP(hipEventRecord(e1, ctx->stream));
    NK_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    NK_HIP(hipEventElapsedTime(&ms, e0, e1));
    *avg_us = 1e3 * ms / iters;
    hipEventDestroy(e0);
    hipEventDestroy(e1);
  }
  hipFree(dV);
  hipFree(dc);
  hipFree(dp);
  return NK_OK;
}

// Start of a cycle, s-step form — ONE launch for what were up to four (k_reduce_sum of ‖b‖², k_max2_final of the Jacobian
// fill's Gershgorin partials, k_gmres_begin, the block basis' set-up):
//  * ss_part != nullptr (one rank, zero initial guess): ‖b‖² is still per-workgroup partial sums — summed here in
//    k_reduce_sum's order (bit-identical);
//  * bpart != nullptr (one rank): {max −lo, max hi} per row block → `ival` (the matrix's bounds cache), max is order-free;
//  * thread 0: the cycle's begin (nk_gmres_begin_body), then the shifts and the scale of the block basis and the scale of the
//    first operator application. newton: `ival` = {−lo, hi} bounds the spectrum; θ_j = c + h·t_j with t the Leja-ordered
//    Chebyshev points of [−1, 1] and σ = the interval's capacity h/2 rounded to a power of two (exact in binary floating point;
//    the basis polynomials then stay O(1) on the interval). Degenerate bounds fall back to the monomial basis: θ = 0, σ ≈ ‖A v₁‖.
typedef nk_ss_begin_args ss_begin_args;
__global__ __launch_bounds__(256) void k_ss_cycle_begin(const ss_begin_args a) {
  __shared__ double sm[16];
  nk_ss_begin_body(a, sm);
}

// development hook (not in the public header): the first block of restart cycle `cycle` of the next solves reports a
// Cholesky breakdown — exercises the column-by-column fall-back where no natural breakdown can be arranged (cycle ≥ 1, with a
// preconditioner); −1 switches it off
__global__ void k_ss_force_fail(nk_gmres_ctl *ctl, nk_gmres_pub *pub, uint64_t seq) {
  if (threadIdx.x != 0) return;
  ctl->failed = 2;
  ctl->done = 1;
  ctl->pad1 = 1;
  ss_pub_progress(pub, seq, ctl->k, 1);
}
extern "C" int nk_gmres_debug_force_breakdown(nk_gmres *G, int cycle) {
  if (G) G->ahead.valid = false;   // (a cycle begin run ahead of the next solve belongs to the old setting)
  NK_REQUIRE(G, "NULL argument");
  G->ss_force_break_cycle = cycle;
  return NK_OK;
}

// Chebyshev points of [−1, 1] in Leja order: t_0 = the point of largest modulus, t_j = the point that maximises
// Π_{i<j} |t − t_i| (ties: the first in the list cos((2i+1)π/2s), i = 0..s−1). A prefix of a Leja sequence is again well
// spread, so a shorter last block uses the first points of the same list.
extern "C" int nk_ss_leja_nodes(int s, double *out) {
  if (s < 1 || s > SS_SMAX || !out) return NK_E_INVALID;
  double pts[SS_SMAX];
  bool used[SS_SMAX];
  for (int i = 0; i < s; ++i) { pts[i] = cos((2.0 * i + 1.0) * M_PI / (2.0 * s)); used[i] = false; }
  for (int j = 0; j < s; ++j) {
    int best = -1;
    double bv = -1.0;
    for (int i = 0; i < s; ++i) {
      if (used[i]) continue;
      double v = (j == 0) ? fabs(pts[i]) : 1.0;
      for (int q = 0; q < j; ++q) v *= fabs(pts[i] - out[q]);
      if (v > bv) { bv = v; best = i; }
    }
    used[best] = true;
    out[j] = pts[best];
  }
  return NK_OK;
}

// ============================================================================= one restart cycle, s columns at a time
struct nk_sstep {
  int s = 0, grid = 0;
  double *part = nullptr, *part2 = nullptr, *red = nullptr, *coef = nullptr, *C1 = nullptr, *R1 = nullptr, *H = nullptr, *scal = nullptr;
  double *C2 = nullptr, *R2 = nullptr;       // pass 2's factors (for sweep C's Hessenberg workgroup), one slot per block
  double *Wi = nullptr, *D = nullptr;        // R₂⁻¹ and C₂R₂⁻¹ of the blocks left at their first pass (same slots)
  size_t c2_stride = 0;
  int nblk_slots = 0;
  unsigned int *ticket = nullptr;            // last-workgroup ticket of k_ss_job
  double *ival = nullptr, *nodes = nullptr;  // {−lo, hi} of the spectrum; Leja-ordered Chebyshev points for nodes_s columns
  const double *ival_use = nullptr;          // where this solve's bounds are: `ival`, or the matrix's cache (left by its fill kernel)
  const double *bpart = nullptr;             // the fill kernel's per-block bounds, to be reduced into ival_use by the next begin kernel
  int bnblk = 0;
  int nodes_s = 0;
  bool newton = false;                       // this solve builds Newton-basis blocks
};
void nk_ss_destroy(nk_sstep *W) {
  if (!W) return;
  hipFree(W->part); hipFree(W->part2); hipFree(W->red); hipFree(W->coef); hipFree(W->C1); hipFree(W->R1); hipFree(W->H); hipFree(W->scal);
  hipFree(W->ival); hipFree(W->nodes); hipFree(W->C2); hipFree(W->R2); hipFree(W->ticket); hipFree(W->Wi); hipFree(W->D);
  delete W;
}
static int ss_workspace(nk_gmres *G) {
  if (G->ss) return NK_OK;
  nk_sstep *W = new nk_sstep();
  auto guard = nk_make_guard(W, [](nk_sstep *w) { nk_ss_destroy(w); });
  const int m = G->m;
  W->grid = G->ctx->num_cus * SS_MAX_WG_PER_CU;
  const size_t nslots = (size_t)(m + 1 + SS_SMAX) * SS_SMAX;
  NK_TRY(nk_dev_alloc(&W->part, nslots * W->grid + 1));
  NK_TRY(nk_dev_alloc(&W->part2, nslots * W->grid + 1));   // sweep B's partial blocks while their reduction is deferred
  NK_TRY(nk_dev_alloc(&W->red, 2 * nslots + 1));   // (a launch may reduce two partial blocks)
  NK_TRY(nk_dev_alloc(&W->coef, nslots + 64));
  // the factors of both passes, one slot per block of a cycle: blocks left at their first pass need pass 2's until the
  // back-substitution, and a block's Hessenberg columns (which need pass 1's) may be derived while the next block is under way
  W->c2_stride = nslots + 1;
  W->nblk_slots = m + 2;
  NK_TRY(nk_dev_alloc(&W->C1, W->c2_stride * W->nblk_slots));
  NK_TRY(nk_dev_alloc(&W->R1, (size_t)SS_SS * W->nblk_slots));
  NK_TRY(nk_dev_alloc(&W->C2, W->c2_stride * W->nblk_slots));
  NK_TRY(nk_dev_alloc(&W->R2, (size_t)SS_SS * W->nblk_slots));
  NK_TRY(nk_dev_alloc(&W->Wi, (size_t)SS_SS * W->nblk_slots));
  NK_TRY(nk_dev_alloc(&W->D, W->c2_stride * W->nblk_slots));
  NK_TRY(nk_dev_alloc(&W->ticket, (size_t)2));
  NK_HIP(nk_memset(G->ctx, W->ticket, 0, 2 * sizeof(unsigned int)));
  NK_TRY(nk_dev_alloc(&W->H, (size_t)(m + 2 + SS_SMAX) * m));
  NK_TRY(nk_dev_alloc(&W->scal, (size_t)SS_TH + SS_SMAX));
  NK_TRY(nk_dev_alloc(&W->ival, (size_t)2));
  NK_TRY(nk_dev_alloc(&W->nodes, (size_t)SS_SMAX));
  NK_HIP(nk_memset(G->ctx, W->scal, 0, (SS_TH + SS_SMAX) * sizeof(double)));
  NK_HIP(nk_memset(G->ctx, W->H, 0, (size_t)(m + 2 + SS_SMAX) * m * sizeof(double)));
  G->ss = guard.release();
  return NK_OK;
}
bool nk_ss_eligible(const nk_gmres *G) { return G->m + 1 <= SS_KMAX - 1 && G->n > 0; }

// Once per linear solve: choose the block basis. Newton basis when real bounds of the operator's spectrum are at hand
// (nk_gmres_spectrum_interval_dev: they are computed on the device and stay there — the shifts are formed by k_ss_begin, no
// host round trip), monomial otherwise or on request.
int nk_ss_prepare(nk_gmres *G) {
  NK_TRY(ss_workspace(G));
  nk_sstep *W = G->ss;
  W->newton = false;
  if (G->ss_basis == NK_SS_BASIS_MONOMIAL) return NK_OK;
  bool have = false;
  W->bpart = nullptr;
  {  // a concrete Jacobian whose fill kernel left per-block Gershgorin bounds: this solve's first begin kernel reduces them
    double *dst = nullptr;
    if (!G->ss_ival_user && !G->prec_kind && !G->lprec_kind && !G->normal && G->shift == 0.0 && G->op_kind == 1 &&
        G->A->nblocks > 0 && nk_csr_take_pending_bounds(G->A, &W->bpart, &W->bnblk, &dst)) {
      W->ival_use = dst;
      W->newton = true;
      return NK_OK;
    }
  }
  NK_TRY(nk_gmres_spectrum_interval_dev(G, W->ival, &W->ival_use, &have));
  if (!have) {
    if (G->ss_basis == NK_SS_BASIS_NEWTON)
      NK_FAIL(NK_E_UNSUPPORTED, "s-step Newton basis: no bounds of this operator's spectrum are known "
                                "(nk_gmres_set_spectrum_interval supplies them)");
    return NK_OK;
  }
  W->newton = true;
  return NK_OK;
}
// the block size in effect: the caller's, else 15 with the Newton basis and 6 with the monomial one — narrowed to 8, then 4,
// once a block of this object has lost rank (a start vector concentrated in a corner of the spectrum, bounds much wider than
// the spectrum: κ of the block grows with its width)
int nk_ss_block_size(const nk_gmres *G) {
  if (G->ss_s > 0) return G->ss_s > SS_SMAX ? SS_SMAX : G->ss_s;
  const int s = (G->ss && G->ss->newton) ? 15 : 6;
  return (G->ss_s_cap > 0 && G->ss_s_cap < s) ? G->ss_s_cap : s;
}

extern "C" int nk_gmres_get_sstep_state(nk_gmres *G, int *block_size, int *newton_basis, int *breakdowns) {
  NK_REQUIRE(G, "NULL argument");
  if (block_size) *block_size = nk_ss_block_size(G);
  if (newton_basis) *newton_basis = (G->ss && G->ss->newton) ? 1 : 0;
  if (breakdowns) *breakdowns = G->ss_breakdowns;
  return NK_OK;
}

// Implicit second pass (A/B switch NK_SS_IMPLICIT=0): a block that is not the cycle's last is left at its first pass as well —
// no sweep C; the next blocks carry their Gram products through its (C₂, R₂) (ss_fix_to_true / _to_stored), its Hessenberg
// columns are a launch of their own, the back-substitution adapts y block by block. One sweep over k + 2s columns less per block.
static bool ss_implicit_on() {
  static const bool off = getenv("NK_SS_IMPLICIT") && atoi(getenv("NK_SS_IMPLICIT")) == 0;
  return !off;
}
static bool ss_skip_last_sweep() {
  static const bool off = getenv("NK_SS_LAST_SWEEP") && atoi(getenv("NK_SS_LAST_SWEEP")) != 0;   // A/B switch: run it anyway
  return !off;
}
// for the back-substitution of a cycle whose last block was left at its first pass: what turns y into coefficients on the
// stored columns (k_backsolve, nk_gmres.hip)
nk_ss_fix nk_ss_take_last_block(nk_gmres *G) {
  nk_ss_fix fx = G->ss_fix;
  G->ss_fix = nk_ss_fix{};
  return fx;
}

// The cycle begin's argument block (k_ss_cycle_begin / nk_ss_begin_body) for this object's workspace. ss_partials: ‖b‖² as
// per-workgroup partial sums (one rank), else G->d_ss holds it. nk_ss_prepare (or nk_ss_prepare_ahead) has chosen the basis.
int nk_ss_begin_args_for(nk_gmres *G, double atol, double rtol, int fixed, int first, uint64_t seq, const double *ss_partials,
                         int ss_grid, nk_ss_begin_args *out) {
  nk_ctx *ctx = G->ctx;
  NK_TRY(ss_workspace(G));
  nk_sstep *W = G->ss;
  const int s = nk_ss_block_size(G);
  if (W->newton && W->nodes_s != s) {
    double h_nodes[SS_SMAX] = {0};
    NK_TRY(nk_ss_leja_nodes(s, h_nodes));
    NK_HIP(hipMemcpyAsync(W->nodes, h_nodes, SS_SMAX * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    NK_HIP(hipStreamSynchronize(ctx->stream));  // (h_nodes is a stack array; once per block size)
    W->nodes_s = s;
  }
  ss_begin_args a;
  a.ctl = G->d_ctl; a.d_ss = G->d_ss; a.g = G->d_g; a.s = G->d_s; a.scal = W->scal;
  a.ival = const_cast<double *>(W->newton ? W->ival_use : (const double *)W->ival);
  a.ss_part = ss_partials; a.bpart = W->newton ? W->bpart : nullptr; a.nodes = W->nodes;
  a.pub = G->h_pub_dev; a.seq = seq; a.atol = atol; a.rtol = rtol;
  a.fixed = fixed; a.first = first; a.m = G->m; a.ss_grid = ss_grid; a.bnblk = W->bnblk; a.ns = s; a.newton = W->newton ? 1 : 0;
  *out = a;
  return NK_OK;
}
// The cycle's begin as a launch of its own.
int nk_ss_begin_cycle(nk_gmres *G, double atol, double rtol, int fixed, int first, uint64_t seq, const double *ss_partials,
                      int ss_grid) {
  nk_ctx *ctx = G->ctx;
  ss_begin_args a;
  NK_TRY(nk_ss_begin_args_for(G, atol, rtol, fixed, first, seq, ss_partials, ss_grid, &a));
  NK_LAUNCH(ctx, k_ss_cycle_begin, dim3(1), dim3(256), a);
  NK_HIP(hipGetLastError());
  if (a.bpart != nullptr && G->op_kind == 1 && G->A) nk_csr_commit_pending_bounds(G->A);
  G->ss->bpart = nullptr;   // (reduced by this launch; later cycles of the solve read the bounds where it left them)
  return NK_OK;
}
// nk_ss_prepare for a solve whose begin is handed to a kernel of the caller's (nk_gmres_begin_ahead): the Newton basis on the
// Gershgorin partials the caller names, reduced by that begin into `dst` (the matrix's bounds word)
int nk_ss_prepare_ahead(nk_gmres *G, const double *bpart, int bnblk, double *dst) {
  NK_TRY(ss_workspace(G));
  nk_sstep *W = G->ss;
  W->bpart = bpart;
  W->bnblk = bnblk;
  W->ival_use = dst;
  W->newton = true;
  return NK_OK;
}

static int ss_launch_hess(nk_ctx *ctx, int k, int sb, const ss_tail_args &ta) {
  const size_t lds = ss_ws_doubles(k, sb, true) * sizeof(double);
  if (lds > 64 * 1024)
    NK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ss_hess), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  nk_prof_scope prof_(ctx, NK_K_REDUCE_SMALL, 8.0 * (k + sb) * sb);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (ctx->prof.on && nk_prof_next(ctx, &e0, &e1))
    hipExtLaunchKernelGGL(k_ss_hess, dim3(1), dim3(256), lds, ctx->stream, e0, e1, 0, k, sb, ta);
  else
    hipLaunchKernelGGL(k_ss_hess, dim3(1), dim3(256), lds, ctx->stream, k, sb, ta);
  NK_HIP(hipGetLastError());
  return NK_OK;
}
// One launch of the block scheme's scalar work (k_ss_job): grid = one wavefront per entry of the partial blocks it reduces.
static int ss_launch_job(nk_ctx *ctx, const ss_job &j, const ss_tail_args &ta0, const ss_tail_args &ta1) {
  const int nslots = j.nslots0 + j.nslots1;
  NK_REQUIRE(nslots > 0, "internal: an s-step job without a block to reduce");
  nk_peer_ar_view pv{nullptr, 0, 0, 0, nullptr};
  if (!nk_ctx_is_single(ctx)) {
    pv = nk_peer_ar_next(ctx, nslots);
    NK_REQUIRE(pv.seq != 0, "internal: the fused s-step reduction needs the peer-mapped arenas (%d values)", nslots);
  }
  const ss_job_lds L = ss_job_layout(j, j.mode);
  const size_t lds = L.total * sizeof(double);
  NK_REQUIRE(lds <= 160 * 1024, "internal: the s-step scalar work needs %zu bytes of LDS", lds);
  const int grid = (nslots + 3) / 4;
  nk_prof_scope prof_(ctx, NK_K_REDUCE_SMALL, 8.0 * ((double)j.nslots0 * j.nblk0 + (double)j.nslots1 * j.nblk1));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  const bool ev = ctx->prof.on && nk_prof_next(ctx, &e0, &e1);
#define SS_JOB_GO(PR, MD)                                                                                                          \
  do {                                                                                                                             \
    if (lds > 64 * 1024)                                                                                                           \
      NK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ss_job<PR, MD>), hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                 (int)lds));                                                                                       \
    if (ev) hipExtLaunchKernelGGL((k_ss_job<PR, MD>), dim3(grid), dim3(SS_R), lds, ctx->stream, e0, e1, 0, j, ta0, ta1, pv);       \
    else hipLaunchKernelGGL((k_ss_job<PR, MD>), dim3(grid), dim3(SS_R), lds, ctx->stream, j, ta0, ta1, pv);                        \
  } while (0)
#define SS_JOB_MODE(MD)                                                                                                            \
  case MD:                                                                                                                         \
    if (pv.seq) SS_JOB_GO(true, MD); else SS_JOB_GO(false, MD);                                                                    \
    break
  switch (j.mode) {   // the combinations nk_ss_cycle uses
    SS_JOB_MODE(SSJ_F1);                                            // a block's first factorisation
    SS_JOB_MODE(SSJ_F1 | SSJ_F2 | SSJ_PREP);                        // … behind the pending block's second (its Hessenberg work: hosted by sweep B)
    SS_JOB_MODE(SSJ_F1 | SSJ_F2 | SSJ_PREP | SSJ_HESS);             // … and its Hessenberg columns here
    SS_JOB_MODE(SSJ_F2 | SSJ_PREP | SSJ_HESS);                      // the pending block closed in a launch of its own
    SS_JOB_MODE(SSJ_F2 | SSJ_PREP | SSJ_HESS | SSJ_BACK);           // the cycle's last launch
    SS_JOB_MODE(SSJ_F2 | SSJ_COEF2);                                // a block with an explicit third sweep
    default: NK_FAIL(NK_E_INVALID, "internal: no s-step scalar launch for mode %d", j.mode);
  }
#undef SS_JOB_MODE
#undef SS_JOB_GO
  NK_HIP(hipGetLastError());
  return NK_OK;
}
// Sweep B of this shape can host a Hessenberg workgroup (the matrix-core form of the default cycle's shapes)
static bool ss_b_can_host(int64_t ldv, int k, int s) {
  static const bool mm_on = !(getenv("NK_SS_MM") && atoi(getenv("NK_SS_MM")) == 0);
  static const bool host_on = !(getenv("NK_SS_HOST_B") && atoi(getenv("NK_SS_HOST_B")) == 0);   // A/B switch
  return host_on && mm_on && s == 15 && (k == 1 || k == 16) && (int64_t)s * ldv * 8 < ((int64_t)1 << 32) - 8;
}

// Sweep A of this shape can host the scalar launch that closes the previous block in extra workgroups (k_ss_block's JOBHOST
// instance: the compile-time-k form of the default cycle's second block). One rank only: a hosted job cannot wait for peers
// while the streaming workgroups of its own launch hold the chip.
// The job's workgroups take the place of streaming ones (the sweep's LDS tile admits two workgroups per CU and the grid fills
// them: workgroups added to a full grid start when the sweep is over).
static int ss_host_a_wgs() {
  static const int n = getenv("NK_SS_HOST_A_WGS") ? atoi(getenv("NK_SS_HOST_A_WGS")) : 20;   // 240 entries: three rounds of 4 × 20 wavefronts (8 left the job longer than the sweep, 30 cost the sweep)
  return n < 1 ? 1 : (n > 64 ? 64 : n);
}
static bool ss_a_can_host_job(nk_ctx *ctx, int k, int s) {
  static const bool host_on = !(getenv("NK_SS_HOST_A") && atoi(getenv("NK_SS_HOST_A")) == 0);   // A/B switch
  static const bool kc_on = !(getenv("NK_SS_KCONST") && atoi(getenv("NK_SS_KCONST")) == 0);
  return host_on && kc_on && nk_ctx_is_single(ctx) && s == 15 && k == 16 && ss_class(k, s) == 2;
}

// Enqueues the Arnoldi part of one cycle: `steps` columns in blocks of ≤ s (cut to the widths the sweeps are compiled for;
// the last block may be shorter). k_gmres_begin has run. `wait_progress(need)` (may be empty) blocks the host until `need`
// columns are closed or the cycle is done and returns false when no further block should be enqueued.
// *backsolved: the cycle's back-substitution (y, the outcome published to the host) rode in the cycle's last scalar launch —
// the caller must not launch k_backsolve.
//
// Round 5 — the DEFERRED second factorisation (NK_SS_DEFER, default on; with the implicit second pass on the fused path). A block
// left at its first pass needs its pass-2 factors (C₂, R₂) only where the NEXT block's reduction carries inner products through
// them — not in front of the next block's matrix powers (which start from the stored column) and not in front of its sweep A (a
// Gram block of stored columns). So sweep B's partial Gram block is not reduced by a launch of its own: the next block's
// reduce-and-factor launch sums BOTH partial blocks, factors the previous block's second pass, prepares Wi and D in LDS, then factors
// this block's first pass through them. The previous block's Hessenberg columns, rotations and stopping test follow in the same
// workgroup (a solve that stops on a tolerance: the verdict arrives where it did before) or ride in workgroup 0 of this block's
// sweep B (the fixed-work protocol: off the critical path). The cycle's last block is closed by one launch that reduces,
// factors, derives the Hessenberg columns and back-substitutes without leaving the workgroup. Per GMRES(30) cycle of two blocks:
// 3 scalar launches instead of 6 (4 × k_ss_reduce_factor, k_ss_hess, k_backsolve), 3 all-reduces instead of 4 on several ranks.
int nk_ss_cycle(nk_gmres *G, int steps, const std::function<bool(int)> &wait_progress, bool *backsolved) {
  nk_ctx *ctx = G->ctx;
  if (backsolved) *backsolved = false;
  NK_TRY(ss_workspace(G));
  nk_sstep *W = G->ss;
  const int64_t n = G->n, ldv = G->ldv;
  const int s = nk_ss_block_size(G);
  const int *done = &G->d_ctl->done, *skipC = &G->d_ctl->pad1;
  ss_tail_args ta;
  std::memset(&ta, 0, sizeof(ta));
  ta.ctl = G->d_ctl; ta.red = W->red; ta.sc = G->d_s; ta.C1 = W->C1; ta.R1 = W->R1; ta.C2 = W->C2; ta.R2 = W->R2; ta.H = W->H; ta.m = G->m;
  ta.Rg = G->d_R; ta.cs = G->d_cs; ta.sn = G->d_sn; ta.g = G->d_g; ta.scal = W->scal; ta.pub = G->h_pub_dev; ta.seq = G->cycle_seq;
  G->ss_fix = nk_ss_fix{};
  // Blocks are left at their first pass with the NEWTON basis only (monomial blocks of 6–8 columns live near the rank-loss bar
  // and keep the explicit second update), and only while the first pass leaves them NEARLY orthonormal: the Hessenberg recovery
  // of the next block starts from a stored column, a combination u of true basis vectors whose images carry this block's
  // recovery errors — harmless for u ≈ e_k, amplified column by column when pass 1 was far off (a block within a factor of ≈ 30
  // of losing rank: oracle, Arnoldi residual 5e-2 against 1e-7 for the explicit update at a departure of 0.75, equal up to 0.2).
  // ss_first_pass_departure measures max(|C₂|, |R₂ − I|) in the reduction's tail; above 0.1 the block counts as broken and
  // takes the fall-back (narrower blocks), exactly like a lost pivot.
  const bool implicit_mode = ss_implicit_on() && W->newton;
  static const bool defer_off = getenv("NK_SS_DEFER") && atoi(getenv("NK_SS_DEFER")) == 0;            // A/B switches
  static const bool tail_back_off = getenv("NK_SS_TAIL_BACK") && atoi(getenv("NK_SS_TAIL_BACK")) == 0;
  static const int hess_where = getenv("NK_SS_DEFER_HESS") ? atoi(getenv("NK_SS_DEFER_HESS")) : -1;   // 0: in the job, 1: hosted by sweep B
  // several ranks: the fused scalar launches are also the all-reduce — on peer-mapped arenas only
  // (one message may carry two partial blocks: ≤ 2·(steps + 1)·s values — 930 for GMRES(30) in blocks of 15)
  const bool peer_ok = nk_ctx_is_single(ctx) || nk_peer_ar_available(ctx, 2 * (steps + 1) * s);
  const bool deferred = !defer_off && implicit_mode && peer_ok && nk_ss_fusable(1, 1);
  const bool fixed_work = !G->ss_grow;
  ta.ptol = 1e-12;
  ss_job jb;   // what every scalar launch of this cycle shares
  std::memset(&jb, 0, sizeof(jb));
  jb.m = G->m; jb.red = W->red; jb.coef = W->coef; jb.d_skip = done; jb.ticket = W->ticket;
  jb.ctl = G->d_ctl; jb.sc = G->d_s; jb.pub = G->h_pub_dev; jb.seq = G->cycle_seq;
  jb.y = G->d_y; jb.Rg = G->d_R; jb.g = G->d_g;
  jb.peer_err = ctx->peer.on ? nk_peer_err_ptr(ctx) : nullptr;
  int blk = 0;            // index of the block within the cycle = its slot of pass-2 factors
  int prev_k0 = 0, prev_sb2 = 0;  // the previous block if it was left at its first pass (prev_sb2 = 0: it was not)
  ss_tail_args pend_ta;           // … and, while its Hessenberg columns wait for a sweep A to host them, its arguments
  std::memset(&pend_ta, 0, sizeof(pend_ta));
  int pend_k = 0, pend_sb = 0;
  // deferred form: the block whose sweep B has run and whose second factorisation has not (its partial Gram block: W->part2)
  struct { bool on; int k, sb, grid; ss_tail_args ta; } dp;
  std::memset(&dp, 0, sizeof(dp));
  // closes the pending block in a launch of its own: second factorisation, Wi / D, Hessenberg columns — and, at the cycle's end,
  // the back-substitution
  // the cycle's last block with a sweep B that stores nothing (k_ss_block_mm<…, NOSTORE>): its columns in memory stay as the matrix
  // powers left them, X; the back-substitution of the cycle's last launch takes the block's COMBINED factors (C, R) — which the
  // Hessenberg recovery forms anyway — where it takes (C₂, R₂) for a block whose first-pass columns were stored (ss_job.raw_last)
  bool raw_on = false;
  auto close_pending = [&](bool with_back) -> int {
    ss_job j = jb;
    j.part1 = W->part2; j.nblk1 = dp.grid; j.nslots1 = (dp.k + dp.sb) * dp.sb; j.k1 = dp.k; j.sb1 = dp.sb;
    j.mode = SSJ_F2 | SSJ_PREP | SSJ_HESS | (with_back ? SSJ_BACK : 0);
    j.cfix = dp.ta.fix;
    NK_REQUIRE(!raw_on || with_back, "internal: a block whose sweep B stored nothing needs the back-substitution of its cycle's last launch");
    if (with_back) {
      j.bfx = G->ss_fix;
      j.raw_last = raw_on ? 1 : 0;
    }
    NK_TRY(ss_launch_job(ctx, j, dp.ta, dp.ta));
    dp.on = false;
    if (with_back && backsolved) *backsolved = true;
    return NK_OK;
  };
  if (G->ss_force_break_cycle >= 0 && G->ss_force_break_cycle == G->ss_cycle_idx)
    NK_LAUNCH(ctx, k_ss_force_fail, dim3(1), dim3(64), G->d_ctl, G->h_pub_dev, G->cycle_seq);
  int k = 1;  // orthonormal columns so far (column 0 = r₀, un-normalised, scale s[0])
  int prev_sb = s;
  // A solve that stops on a tolerance may need 2 iterations or 200: a block's operator applications past the column that meets
  // the tolerance are wasted (a multigrid V-cycle each, under that preconditioner). Automatic block sizes therefore start small
  // in every cycle and double — 4, 8, 15, 15 … with the Newton basis, 2, 4, 6, 6 … with the monomial one —: a solve that needs
  // k iterations applies the operator < 2k times, and one that fills the cycle builds most of it in full-width blocks. The
  // fixed-work protocol and explicit block sizes take full blocks from the start.
  int grow = (G->ss_s == 0 && G->ss_grow) ? (W->newton ? 4 : 2) : s;
  while (k - 1 < steps) {
    int sb = (steps - (k - 1)) < s ? (steps - (k - 1)) : s;
    if (grow < sb) sb = grow;
    grow = grow * 2 > s ? s : grow * 2;
    sb = nk_ss_block_width(sb);
    if (k + sb > 48 && sb > 8) sb = 8;  // the streaming size class keeps its scalar workspace within the LDS
    if (wait_progress && k > 1 && !wait_progress(k - 1 - prev_sb)) break;
    prev_sb = sb;
    double *Wk = G->V + (size_t)k * ldv;
    // the block's basis vectors: (A − θ_j I) applied s times (right-preconditioned operator), scaled by 1/σ — in one launch
    // with the matrix held on the chip where that applies (nk_powers.hip), else one operator launch per column
    bool powers = false;
    NK_TRY(nk_gmres_op_powers(G, G->V + (size_t)(k - 1) * ldv, Wk, ldv, sb, done, W->scal, W->newton ? W->scal + SS_TH : nullptr,
                              &powers));
    for (int j = 0; j < sb && !powers; ++j)
      NK_TRY(nk_gmres_op_apply(G, G->V + (size_t)(k - 1 + j) * ldv, Wk + (size_t)j * ldv, done, W->scal + (j == 0 ? 0 : 1),
                               W->newton ? W->scal + SS_TH + j : nullptr));
    const int grid = nk_ss_grid(ctx, n, k, sb);
    const int nslots = (k + sb) * sb;
    if (ctx->audit.on)   // (development) the block's new columns as the operator left them
      for (int j = 0; j < sb; ++j) nk_audit(ctx, 100 + k - 1 + j, Wk + (size_t)j * ldv, (size_t)n);
    // fused: the block's scalar work rides in the stage-2 reduction (its last workgroup factors the reduced block and leaves
    // the update coefficients for the next sweep's scalar loads) and in sweep C (workgroup 0: the Hessenberg columns) — one
    // rank, or several on peer-mapped arenas (the reduction is then the all-reduce as well). Other transports and the
    // streaming size class (k + s > 48): reduction, all-reduce and the scalar work as launches of their own.
    const bool fused = nk_ss_fusable(k, sb) && peer_ok;
    if (pend_sb > 0 && !fused) {   // nobody to host it: the previous block's Hessenberg columns as a launch of their own
      NK_TRY(ss_launch_hess(ctx, pend_k, pend_sb, pend_ta));
      pend_sb = 0;
    }
    // this block's slots of factors; the blocks before it that were left at their first pass; the start vector's origin
    NK_REQUIRE(blk < W->nblk_slots, "internal: more s-step blocks in a cycle than factor slots");
    ta.C1 = W->C1 + (size_t)blk * W->c2_stride;
    ta.R1 = W->R1 + (size_t)blk * SS_SS;
    ta.C2 = W->C2 + (size_t)blk * W->c2_stride;
    ta.R2 = W->R2 + (size_t)blk * SS_SS;
    ta.fix = G->ss_fix;
    ta.usb = prev_sb2; ta.uk0 = prev_k0;
    ta.uC2 = prev_sb2 ? W->C2 + (size_t)(blk - 1) * W->c2_stride : nullptr;
    ta.uR2 = prev_sb2 ? W->R2 + (size_t)(blk - 1) * SS_SS : nullptr;
    // left at its first pass (no sweep C): the cycle's last block always; any other block while the list has room — whatever
    // the transport and the size class, so that every path runs the same arithmetic (results are compared bit for bit)
    const bool last_block = (k - 1 + sb >= steps) && ss_skip_last_sweep();
    const bool implicit = !last_block && implicit_mode && G->ss_fix.n < NK_SS_NFIX - 1;
    ta.Wi = implicit ? W->Wi + (size_t)blk * SS_SS : nullptr;
    ta.D = implicit ? W->D + (size_t)blk * W->c2_stride : nullptr;
    const bool defer_this = deferred && fused && (last_block || implicit);
    if (dp.on && !defer_this) NK_TRY(close_pending(false));   // (this block takes the older form: nobody to carry the pending one)
    if (defer_this) {
      // ---- sweep A, the job [second factorisation of the pending block ; first factorisation of this one], sweep B
      int grid_a = nk_ss_grid_a(ctx, n, k, sb, false);
      // the fixed-work protocol, second block of the default cycle: the pending block is closed (reduction, second factorisation,
      // Wi / D, Hessenberg columns) by extra workgroups of THIS sweep — nothing it writes is read by the streaming ones — and the
      // launch behind the sweep only factors this block's first pass
      const bool host_a = dp.on && grid_a > 4 * ss_host_a_wgs() && fixed_work && hess_where < 0 && ss_a_can_host_job(ctx, k, sb);
      if (host_a) grid_a -= ss_host_a_wgs();   // (streaming workgroups: the pitch of the partial blocks)
      {
        nk_prof_scope prof_(ctx, NK_K_MULTIDOT, 8.0 * (double)n * (k + sb));
        if (host_a) {
          ss_job hj = jb;
          hj.part1 = W->part2; hj.nblk1 = dp.grid; hj.nslots1 = (dp.k + dp.sb) * dp.sb; hj.k1 = dp.k; hj.sb1 = dp.sb;
          hj.mode = SSJ_F2 | SSJ_PREP | SSJ_HESS;
          hj.cfix = dp.ta.fix;
          hj.host_wgs = ss_host_a_wgs();
          NK_TRY(ss_sweep_a_hosting_job(ctx, n, k, sb, G->V, ldv, W->part, done, grid_a, dp.ta, &G->d_ctl->pad1, hj));
        } else
          NK_TRY(nk_ss_sweep(ctx, 0, n, k, sb, G->V, ldv, W->coef, W->part, done, grid_a, nullptr, &G->d_ctl->pad1, 0, 0));
      }
      // where the pending block's Hessenberg columns are derived: in workgroup 0 of this block's sweep B when nothing can stop
      // the cycle early (fixed work) and that sweep has the hosting form; else in the job itself (the verdict arrives before sweep B)
      const bool host_b = !host_a && dp.on && grid > 1 && ss_b_can_host(ldv, k, sb) &&
                          (hess_where < 0 ? fixed_work : hess_where == 1);
      // the last block's sweep B stores nothing where the matrix-core form runs it, the cycle's last scalar launch
      // back-substitutes, the list has room and nothing else wants the columns (development audit)
      static const bool nostore_off = getenv("NK_SS_NOSTORE") && atoi(getenv("NK_SS_NOSTORE")) == 0;   // A/B switch
      const bool raw_last = last_block && !nostore_off && !(host_b && k != 16) && ss_b_can_host(ldv, k, sb) && !tail_back_off &&
                            backsolved != nullptr && !ctx->audit.on;
      {
        ss_job j = jb;
        j.part0 = W->part; j.nblk0 = grid_a; j.nslots0 = nslots; j.k0 = k; j.sb0 = sb;
        j.mode = SSJ_F1;
        j.cfix = ta.fix;
        if (dp.on && !host_a) {
          j.part1 = W->part2; j.nblk1 = dp.grid; j.nslots1 = (dp.k + dp.sb) * dp.sb; j.k1 = dp.k; j.sb1 = dp.sb;
          j.mode |= SSJ_F2 | SSJ_PREP | (host_b ? 0 : SSJ_HESS);
        }
        NK_TRY(ss_launch_job(ctx, j, ta, (dp.on && !host_a) ? dp.ta : ta));
      }
      // the read-only sweep runs ONE workgroup per CU where the tiles allow (as the read-only sweeps A do: stand-alone 52 → 48 µs
      // at 1024² — half the prologues and partial sums, one wavefront per SIMD on the matrix pipe)
      static const bool ro_grid_off = getenv("NK_SS_RO_GRID") && atoi(getenv("NK_SS_RO_GRID")) == 0;   // A/B switch
      int grid_b = grid;
      if (raw_last && !host_b && !ro_grid_off && ss_b_is_read_only(G->V, ldv, k, sb) && (n + SS_R - 1) / SS_R >= 2 * (int64_t)ctx->num_cus &&
          grid > ctx->num_cus)
        grid_b = ctx->num_cus;
      {
        nk_prof_scope prof_(ctx, NK_K_MULTIDOT, 8.0 * (double)n * (k + (raw_last ? 1 : 2) * sb));
        ss_tail_args hta = dp.ta;
        hta.Wi = nullptr; hta.D = nullptr;   // (prepared by the job already)
        NK_TRY(nk_ss_sweep(ctx, 1, n, k, sb, G->V, ldv, W->coef, W->part2, done, grid_b, host_b ? &hta : nullptr, nullptr,
                           host_b ? dp.k : 0, host_b ? dp.sb : 0, raw_last ? 1 : 0));
      }
      if (raw_last) raw_on = true;
      dp.on = true; dp.k = k; dp.sb = sb; dp.grid = grid_b; dp.ta = ta;
      {
        nk_ss_fix &fx = G->ss_fix;
        NK_REQUIRE(fx.n < NK_SS_NFIX, "internal: too many s-step blocks left at their first pass");
        fx.k0[fx.n] = k; fx.sb[fx.n] = sb; fx.C2[fx.n] = ta.C2; fx.R2[fx.n] = ta.R2;
        fx.Wi[fx.n] = implicit ? ta.Wi : nullptr; fx.D[fx.n] = implicit ? ta.D : nullptr;
        fx.n++;
      }
      NK_HIP(hipGetLastError());
      prev_k0 = k;
      prev_sb2 = sb;
      k += sb;
      ++blk;
      continue;
    }
    const int grid_b = grid;
    for (int pass = 0; pass < 2; ++pass) {
      int grid = grid_b;   // (of THIS pass: the sweep, and the reduction behind it, which sums one partial per workgroup)
      {
        nk_prof_scope prof_(ctx, NK_K_MULTIDOT, 8.0 * (double)n * (k + sb + (pass ? sb : 0)));
        const bool host_prev = pass == 0 && pend_sb > 0 && fused;   // sweep A hosts the previous block's Hessenberg columns
        if (pass == 0) grid = nk_ss_grid_a(ctx, n, k, sb, host_prev);
        NK_TRY(nk_ss_sweep(ctx, pass, n, k, sb, G->V, ldv, W->coef, W->part, done, grid, host_prev ? &pend_ta : nullptr,
                           pass == 0 ? &G->d_ctl->pad1 : nullptr, pend_k, pend_sb));
        if (host_prev) pend_sb = 0;
      }
      if (fused) {
        ss_job j = jb;
        j.cfix = ta.fix;
        if (pass == 0) {
          j.part0 = W->part; j.nblk0 = grid; j.nslots0 = nslots; j.k0 = k; j.sb0 = sb;
          j.mode = SSJ_F1;
        } else {   // the block's own second factorisation: coefficients for sweep C, C₂ / R₂ for whoever derives its Hessenberg columns
          j.part1 = W->part; j.nblk1 = grid; j.nslots1 = nslots; j.k1 = k; j.sb1 = sb;
          j.mode = SSJ_F2 | SSJ_COEF2;
        }
        NK_TRY(ss_launch_job(ctx, j, ta, ta));
      } else {
        if (ctx->audit.on) {
          nk_audit(ctx, 1000 * (pass + 1) + 10 * blk + 1, W->part, (size_t)nslots * grid);            // the sweep's partial Gram blocks
          if (pass == 1) for (int j = 0; j < sb; ++j) nk_audit(ctx, 200 + k + j, Wk + (size_t)j * ldv, (size_t)n);   // … and updated columns
        }
        {
          nk_prof_scope prof_(ctx, NK_K_REDUCE_SMALL, 8.0 * nslots * grid);
          NK_TRY(nk_blas_reduce_slots_allreduce(ctx, W->part, grid, nslots, W->red, done));  // (k + s)·s values, one message
        }
        if (ctx->audit.on) nk_audit(ctx, 1000 * (pass + 1) + 10 * blk + 2, W->red, (size_t)nslots);   // the reduced (all-reduced) block
        const size_t lds = (ss_ws_doubles(k, sb, pass == 1) + ss_fixc_doubles(ta.fix)) * sizeof(double);
        if (pass == 0) {
          if (lds > 64 * 1024)
            NK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ss_tail1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
          hipLaunchKernelGGL(k_ss_tail1, dim3(1), dim3(256), lds, ctx->stream, k, sb, W->coef, ta);
        } else {
          if (lds > 64 * 1024)
            NK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ss_tail2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
          hipLaunchKernelGGL(k_ss_tail2, dim3(1), dim3(256), lds, ctx->stream, k, sb, W->coef, ta);
        }
        if (ctx->audit.on) {   // what the tail left: the next sweep's coefficients, the factors, the control block
          nk_audit(ctx, 1000 * (pass + 1) + 10 * blk + 3, W->coef, (size_t)k * sb + (size_t)sb * sb);
          nk_audit(ctx, 1000 * (pass + 1) + 10 * blk + 4, pass == 0 ? ta.C1 : ta.C2, (size_t)k * sb);
          nk_audit(ctx, 1000 * (pass + 1) + 10 * blk + 5, pass == 0 ? ta.R1 : ta.R2, (size_t)sb * sb);
          nk_audit(ctx, 1000 * (pass + 1) + 10 * blk + 6, G->d_ctl, sizeof(nk_gmres_ctl) / 8);
        }
      }
    }
    if (!last_block && !implicit) {
      nk_prof_scope prof_(ctx, NK_K_MULTIAXPY, 8.0 * (double)n * (k + 2 * sb));
      NK_TRY(nk_ss_sweep(ctx, 2, n, k, sb, G->V, ldv, W->coef, W->part, skipC, grid, fused ? &ta : nullptr, nullptr));
    } else {
      // left at its first pass: no third sweep (k_backsolve turns y into coefficients on the columns as they are; later blocks
      // carry their Gram products through this block's factors). Its Hessenberg columns — the work of sweep C's workgroup 0,
      // or already done by k_ss_tail2 on the unfused path —: the cycle's last block as a launch of its own, any other block
      // inside the NEXT block's sweep A (which also leaves Wi, D for the reductions behind it).
      if (implicit && fused) {
        pend_ta = ta; pend_k = k; pend_sb = sb;
      } else if (fused) {
        NK_TRY(ss_launch_hess(ctx, k, sb, ta));
      }   // (unfused: k_ss_tail2 has derived the Hessenberg columns — and Wi, D — already)
      {
        nk_ss_fix &fx = G->ss_fix;
        NK_REQUIRE(fx.n < NK_SS_NFIX, "internal: too many s-step blocks left at their first pass");
        fx.k0[fx.n] = k; fx.sb[fx.n] = sb; fx.C2[fx.n] = ta.C2; fx.R2[fx.n] = ta.R2;
        fx.Wi[fx.n] = implicit ? ta.Wi : nullptr; fx.D[fx.n] = implicit ? ta.D : nullptr;
        fx.n++;
      }
    }
    NK_HIP(hipGetLastError());
    prev_k0 = k;
    prev_sb2 = (last_block || implicit) ? sb : 0;
    k += sb;
    ++blk;
  }
  if (pend_sb > 0) NK_TRY(ss_launch_hess(ctx, pend_k, pend_sb, pend_ta));   // (the host stopped enqueueing blocks early)
  if (dp.on) NK_TRY(close_pending(!tail_back_off && backsolved != nullptr));
  return NK_OK;
}
