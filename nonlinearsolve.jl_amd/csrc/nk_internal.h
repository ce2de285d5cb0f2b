// Internal declarations of libmi355x_nk.so (gfx950 only). Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <chrono>
#include <functional>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <string>
#include <vector>

#include "mi355x_nk.h"

// ----------------------------------------------------------------------------- errors
void nk_set_error(const char *fmt, ...);
#define NK_FAIL(code, ...)        \
  do {                            \
    nk_set_error(__VA_ARGS__);    \
    return (code);                \
  } while (0)
#define NK_HIP(call)                                                                      \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess)                                                                 \
      NK_FAIL(NK_E_HIP, "%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
  } while (0)
#define NK_TRY(call)            \
  do {                          \
    int s_ = (call);            \
    if (s_ != NK_OK) return s_; \
  } while (0)
#define NK_REQUIRE(cond, ...) \
  do {                        \
    if (!(cond)) NK_FAIL(NK_E_INVALID, __VA_ARGS__); \
  } while (0)

// frees a half-built object when a create function leaves early (NK_TRY / NK_HIP / NK_REQUIRE return on error)
template <class T, class D>
struct nk_scope_guard {
  T *p;
  D d;
  ~nk_scope_guard() { if (p) d(p); }
  T *release() { T *q = p; p = nullptr; return q; }
};
template <class T, class D>
nk_scope_guard<T, D> nk_make_guard(T *p, D d) { return {p, d}; }

// ----------------------------------------------------------------------------- context
constexpr int NK_BLOCK = 256;          // 4 wavefronts of 64
constexpr int NK_MAX_RED_BLOCKS = 1024; // stage-1 reduction blocks (4 per CU)
constexpr int NK_MAX_ROW_TILES = 16384; // DCGS2-1R dot sweep: one block per 2048-row tile (≤ 33.5 M local rows)
constexpr int NK_LDV_PAD_DEFAULT = 544;   // extra elements between basis columns (tuned on hardware)
constexpr int NK_DOT_BLOCKS = 512;      // multi-dot kernels: fewer, fatter blocks (epilogue = NV block reductions)
constexpr int NK_MAX_NV = 64;          // max simultaneous dot products (restart m ≤ 63)

// per-kernel-family timing with HIP events on the launch stream (bench/roofline evidence; off by default)
enum nk_kernel_id {
  NK_K_SPMV = 0, NK_K_MULTIDOT, NK_K_MULTIAXPY, NK_K_JVP, NK_K_RESIDUAL, NK_K_SCALE, NK_K_REDUCE_SMALL,
  NK_K_JACFILL, NK_K_NEWTON_UPDATE, NK_K_OTHER, NK_K_POWERS, NK_K_COUNT
};
struct nk_prof {
  bool on = false;
  std::vector<hipEvent_t> ev;       // 2 per record
  std::vector<int> ids;
  std::vector<double> nbytes;
  size_t used = 0;
  double ms[NK_K_COUNT] = {0}, bytes[NK_K_COUNT] = {0};
  int cur_id = -1;          // kernel family of the active scope (-1: none)
  double cur_bytes = 0.0;   // algorithmic bytes not yet attributed to a launch of the active scope
  int64_t count[NK_K_COUNT] = {0};
};

// peer-mapped arenas (hipIpc over xGMI) for the small collectives of the Krylov loop (nk_ctx.hip)
constexpr int NK_PEER_MAX_RANKS = 16;
constexpr int NK_PEER_AR_MAX = 1024;                     // doubles per all-reduce message (two s-step blocks in one message: 2·(k + s)·s ≤ 2·31·15)
constexpr size_t NK_PEER_HDR_BYTES = 524288;             // flags + all-reduce slots + error word
// Layout of every rank's arena: a header (all-reduce flags and slots, error word) and a bump-allocated rest that holds
// the receive areas of the halo plans. All of it is uncached device memory, so that a kernel polling a flag sees the
// store a peer GPU made while the kernel was already running.
struct nk_peer_hdr {
  uint64_t ar_flag[2][NK_PEER_MAX_RANKS];
  uint64_t err;
  uint64_t timeout_ticks;   // bound of every device-side wait, in ticks of the 100 MHz wall clock (NK_PEER_TIMEOUT_MS; default 5 s)
  uint64_t pad[30];
  double ar_data[2][NK_PEER_MAX_RANKS][NK_PEER_AR_MAX];
};
static_assert(sizeof(nk_peer_hdr) <= NK_PEER_HDR_BYTES, "peer arena header too large");
struct nk_peer_seg {  // one neighbour of a halo plan, as the push / wait kernels see it
  int64_t send_off, send_cnt;     // into the plan's send index list
  double *dst[2];                 // the neighbour's receive area for my entries (both parities), in MY address space
  uint64_t *flag_remote;          // the neighbour's flag for me
  const uint64_t *flag_local;     // my flag for the neighbour
};
struct nk_peer {
  bool on = false;
  int P = 0, me = 0;
  char *arena = nullptr;
  size_t arena_bytes = 0, bump = 0;
  char *map[NK_PEER_MAX_RANKS] = {nullptr};   // every rank's arena in my address space (map[me] == arena)
  char **d_map = nullptr;                     // the same table on the device
  uint64_t ar_seq = 0;
  unsigned int *d_ticket = nullptr;           // last-workgroup ticket of the fused reduce + all-reduce kernel
};
// the all-reduce slots of the arena header, as nk_blas.hip's fused stage-2 reduction sees them (layout: nk_ctx.hip)
struct nk_peer_ar_view {
  char *const *map;   // device table of the arenas
  int P, me;
  uint64_t seq;       // 0: no peer path (plain stage-2 reduction)
  unsigned int *ticket;
};
nk_peer_ar_view nk_peer_ar_next(nk_ctx *ctx, int count);
bool nk_peer_ar_available(nk_ctx *ctx, int count);   // would nk_peer_ar_next take the fast path? (no side effect)
// Receive areas of the resident matrix-powers kernel on several ranks (nk_powers.hip): per rank two areas of 4 × 1024 doubles
// (from the rank above / below) and a flag word each, carved from the arena. COLLECTIVE: every rank passes whether its share of
// the matrix is eligible; *ok is the common verdict (all eligible and the areas fit everywhere).
struct nk_peer_powers {
  double *push_up = nullptr, *push_dn = nullptr;           // the neighbours' areas for my slices, in my address space
  uint64_t *flag_up = nullptr, *flag_dn = nullptr;         // the neighbours' flags for me
  double *recv_up = nullptr, *recv_dn = nullptr;           // my areas
  uint64_t *myflag_up = nullptr, *myflag_dn = nullptr;
};
int nk_peer_powers_setup(nk_ctx *ctx, bool eligible, nk_peer_powers *out, bool *ok);
// bound of a device-side wait: the word behind the arena's error counter (err[1]); 5 s if the arena predates it
__device__ __forceinline__ unsigned long long nk_peer_timeout(const uint64_t *err) {
  const unsigned long long t = err[1];
  return t ? t : 500000000ull;
}
uint64_t *nk_peer_err_ptr(nk_ctx *ctx);  // the arena's time-out counter (device pointer)  // claims the next sequence number if the fast path applies

// development: content hashes of device buffers at named points of a solve (nk_audit; tools/shared_device_probe.py --audit) —
// two runs of a deterministic path must produce the same sequence; the first entry that differs names the kernel whose output did
struct nk_audit_log {
  bool on = false;
  uint64_t *d_slots = nullptr;
  int cap = 0, count = 0;
  std::vector<int> tags;
};
struct nk_ctx {
  nk_audit_log audit;
  nk_prof prof;
  nk_peer peer;
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // halo exchange overlapped with the interior rows of the SpMV: the exchange runs on comm_stream, ordered by events
  int halo_overlap = 0;
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_halo_ready = nullptr, ev_halo_done = nullptr;
  int deterministic = 1;
  int num_cus = 256;
  // several ranks of this communicator run on ONE device (processes time-slicing a GPU: development boxes, CI): forms that need
  // every workgroup of a launch resident at once, or more than 64 KB of LDS per workgroup, are not used then (nk_ctx.hip)
  bool device_shared = false;
  // communicator
  int comm_kind = NK_COMM_NONE, nranks = 1, rank = 0;
  void *rccl_comm = nullptr;
  nk_comm_callbacks cb{};
  // scratch
  double *d_partials = nullptr;  // NK_MAX_NV * NK_MAX_RED_BLOCKS doubles
  double *d_partials_ss = nullptr;  // ‖·‖² partials of the axpy kernels (own buffer: survives later multidots)
  int last_red_grid = 0;
  int peer_err_reported = 0;     // peer-arena time-outs already surfaced as NK_E_COMM by some solver object of this context
  double *d_scal = nullptr;      // 4*NK_MAX_NV doubles of device scalars
  double *h_pinned = nullptr;    // 4*NK_MAX_NV doubles pinned host (coherent: kernels publish scalars into it)
  double *h_pinned_dev = nullptr;  // the same allocation as the device sees it
  uint64_t *h_seq = nullptr, *h_seq_dev = nullptr;  // sequence word of the last published batch (pinned, coherent)
  uint64_t seq = 0;
  nk_stats stats{};
};

// Host side of "a kernel publishes into coherent pinned memory, the host polls": spins on `ready`, checking that the stream
// is still alive (a faulted or drained stream must not leave the host spinning) — by the CLOCK, first after 2 ms and every
// 2 ms from then on: hipStreamQuery on a busy stream makes the runtime put a marker (a barrier packet with a completion
// signal) behind the last kernel enqueued; a poll count (round 5: every 16 384 polls ≈ 0.2 ms) put one into every Newton step
// whose linear solve outlasted it. (No measurable effect on the step time: profiles/r06_c_host_gaps_head_and_powers_handoff.md.)
template <class Pred>
static inline int nk_spin_wait(nk_ctx *ctx, Pred ready, const char *what) {
  auto next_check = std::chrono::steady_clock::time_point{};
  for (uint64_t it = 1;; ++it) {
    if (ready()) return NK_OK;
    if ((it & 0x3ff) == 0) {
      const auto now = std::chrono::steady_clock::now();
      if (next_check == std::chrono::steady_clock::time_point{}) next_check = now + std::chrono::milliseconds(2);
      else if (now >= next_check) {
        next_check = now + std::chrono::milliseconds(2);
        const hipError_t e = hipStreamQuery(ctx->stream);
        if (e == hipSuccess) {
          if (ready()) return NK_OK;
          NK_FAIL(NK_E_HIP, "the stream drained but %s never arrived", what);
        }
        if (e != hipErrorNotReady) NK_FAIL(NK_E_HIP, "stream error while waiting for %s: %s", what, hipGetErrorString(e));
      }
    }
    __builtin_ia32_pause();
  }
}

int nk_audit(nk_ctx *ctx, int tag, const void *p, size_t nwords64);   // (no-op unless the log is on)
// true when no collective has to be issued (1 rank and not in the NK_FORCE_COLLECTIVES test mode)
bool nk_ctx_is_single(const nk_ctx *ctx);
// Profiling: inside an nk_prof_scope every launch goes through hipExtLaunchKernelGGL with its own start/stop
// events, which carry the kernel's begin/end device timestamps (the same quantity rocprofv3's kernel trace
// reports) — not host-side event brackets, which also measure the record overhead and the launch gap.
void nk_prof_scope_begin(nk_ctx *ctx, int id, double bytes);
void nk_prof_scope_end(nk_ctx *ctx);
bool nk_prof_next(nk_ctx *ctx, hipEvent_t *start, hipEvent_t *stop);
void nk_prof_flush(nk_ctx *ctx);
struct nk_prof_scope {
  nk_ctx *c;
  nk_prof_scope(nk_ctx *ctx, int id, double bytes) : c(ctx->prof.on ? ctx : nullptr) { if (c) nk_prof_scope_begin(c, id, bytes); }
  ~nk_prof_scope() { if (c) nk_prof_scope_end(c); }
};
#define NK_LAUNCH(ctxp, kern, grid, block, ...)                                                              \
  do {                                                                                                      \
    hipEvent_t e0_, e1_;                                                                                    \
    if ((ctxp)->prof.on && nk_prof_next((ctxp), &e0_, &e1_))                                                \
      hipExtLaunchKernelGGL(kern, grid, block, 0, (ctxp)->stream, e0_, e1_, 0, __VA_ARGS__);                \
    else                                                                                                    \
      hipLaunchKernelGGL(kern, grid, block, 0, (ctxp)->stream, __VA_ARGS__);                                \
  } while (0)
int nk_comm_allreduce(nk_ctx *ctx, double *dbuf, int count, int op /*0 sum,1 max*/);
int nk_comm_allreduce_mixed(nk_ctx *ctx, double *dbuf, int count, int max_lo, int max_hi);  // [max_lo,max_hi): max, rest: +
int nk_comm_alltoallv(nk_ctx *ctx, const void *send, const int64_t *soff, const int64_t *sbytes,
                      void *recv, const int64_t *roff, const int64_t *rbytes, hipStream_t stream = nullptr /* ctx->stream */);
void nk_comm_destroy(nk_ctx *ctx);

// optional row epilogue of the SpMV / JVP kernels: mode 1 fuses one Chebyshev-iteration vector update, mode 2 (stencil JVP
// only) the residual b − J v, mode 3 a shift read from device memory: y = scale·(A x − θ x) — one step of the s-step Arnoldi
// process's Newton basis (nk_sstep.hip); x[row] is the diagonal gather the row has just made, so the shift moves no extra bytes
struct nk_spmv_epi {
  int mode = 0;
  double c1 = 0, c2 = 0;
  double *r = nullptr, *dnew = nullptr, *yacc = nullptr;
  const double *theta = nullptr;
  const double *dinv = nullptr;   // mode 1 (CSR kernel): the step runs on D⁻¹A. CSR modes 2 / 4: y = r − A x / r −= A x
};

// ----------------------------------------------------------------------------- halo plan
// recv_buf holds the off-rank entries a kernel needs ("halo"), in the order defined by the plan's owner.
struct nk_halo {
  int64_t n_send = 0, n_recv = 0;
  std::vector<int64_t> send_off, send_cnt, recv_off, recv_cnt;  // per peer (size nranks), in elements
  int32_t *d_send_idx = nullptr;  // local indices to gather (n_send)
  double *d_send = nullptr, *d_recv = nullptr;
  // every peer's send list is one contiguous local range (row-partitioned stencils: whole grid lines) → the exchange
  // sends straight from the vector, no gather launch
  std::vector<int64_t> contig_base;
  bool contig = false;
  // peer fast path: receive areas (two parities) live in this rank's arena; segs describes the neighbours
  bool peer = false;
  double *recv_buf[2] = {nullptr, nullptr};
  nk_peer_seg *d_segs = nullptr;
  int nsegs = 0;
  uint64_t seq = 0;
  bool active() const { return n_send > 0 || n_recv > 0; }
};
int nk_halo_setup(nk_ctx *ctx, nk_halo *H, const std::vector<std::vector<int32_t>> &send_idx_per_peer,
                  const std::vector<int64_t> &recv_cnt_per_peer);
// Plan for gathering arbitrary entries of a row-partitioned vector: this rank owns [my_begin, my_begin + my_count) of it
// and needs the global entries `needs` (sorted, unique, none of them owned). Collective. The receive buffer holds them in
// the order of `needs`.
int nk_halo_build_from_needs(nk_ctx *ctx, int64_t my_begin, int64_t my_count, const std::vector<int64_t> &needs, nk_halo *H);
int nk_halo_exchange(nk_ctx *ctx, nk_halo *H, const double *d_x_local);  // result in H->d_recv
// split form: begin = gather on the compute stream + exchange on ctx->comm_stream; end = compute stream waits for it
int nk_halo_exchange_begin(nk_ctx *ctx, nk_halo *H, const double *d_x_local);
int nk_halo_exchange_end(nk_ctx *ctx, nk_halo *H);
void nk_halo_free(nk_halo *H);

// ----------------------------------------------------------------------------- CSR
struct nk_csr {
  nk_ctx *ctx = nullptr;
  int64_t nrows = 0, n_global = 0, row_begin = 0, nnz = 0;
  int32_t *d_rowptr = nullptr, *d_col = nullptr;  // local columns: [0,nrows) owned, ≥ nrows → halo slot
  // SpMV-only copy of the column ids as 16-bit offsets from the row block's first row (banded matrices: 2 instead of
  // 4 bytes per non-zero on the stream); used for descriptors [0, n16) — the interior blocks, when all of them fit
  int16_t *d_col16 = nullptr;
  int n16 = 0;
  double *d_val = nullptr;
  int32_t *d_rowblocks = nullptr;
  int nblocks = 0;
  int nblocks_interior = 0;  // descriptors [0, nblocks_interior) touch no halo column (ordered first at creation)
  int tile = 2048, variant = 0;  // SpMV kernel configuration (tunable via NK_SPMV_TILE / NK_SPMV_VARIANT)
  nk_halo halo;
  std::vector<int64_t> halo_gcols;  // global column of each halo slot
  // host copies of the pattern (needed for transpose / banded LU / colouring)
  std::vector<int32_t> h_rowptr, h_col;
  // lazily built transpose of the local block: (nrows + n_halo) × nrows — rows ≥ nrows collect the contributions to
  // entries other ranks own, which the reverse halo exchange returns to their owners (nk_csr_spmv_t_dev)
  nk_csr *T = nullptr;
  int32_t *d_tperm = nullptr;
  double *d_ones = nullptr;   // a vector of ones (nk_csr_colsumsq_dev)
  int32_t *d_diagpos = nullptr;  // position of every row's diagonal entry in val (nk_csr_add_to_diagonal_dev)
  double *d_gersh = nullptr;     // per-row-block Gershgorin bounds (nk_csr_gershgorin_dev)
  double *d_bounds = nullptr;    // {−lo, hi} left by a fill kernel that computes the discs on the fly (valid while bounds_valid)
  int gersh_cap = 0;             // doubles allocated behind d_gersh
  const double *bounds_part = nullptr;   // a fill kernel's {max −lo, max hi} per block, not reduced into d_bounds yet
  int bounds_nblk = 0;
  bool bounds_pending = false;
  bool bounds_valid = false, raw_exposed = false;  // raw_exposed: nk_csr_values_device handed the value array out — never trust a cache
  int32_t *d_csc_src = nullptr;  // created from CSC arrays: index of every local entry in that call's nzval
  double *d_csc_stage = nullptr; // staging for host nzval (nk_csr_set_values_csc)
  int64_t csc_nnz = 0;
  bool t_values_stale = true;
  double *d_tz = nullptr, *d_trecv = nullptr;  // T·x (nrows + n_halo) and what the peers sent back (n_send)
  // column colouring of the pattern (structurally orthogonal columns), built on first use by coloured assembly
  int ncolors = 0;
  int32_t *d_color = nullptr, *d_nnzcolor = nullptr;
  double *d_seed = nullptr, *d_B = nullptr;
  double *d_xtmp = nullptr, *d_ytmp = nullptr;  // staging for host-memspace calls
  // problem-specific device tables attached by nk_problem_jac_csr (freed with the matrix)
  uint8_t *d_role = nullptr;   // Brusselator: role of every non-zero
  int32_t *d_node = nullptr;   // Brusselator: grid node of every non-zero's row
  // resident matrix-powers kernel (nk_powers.hip): built on first use; NULL = the matrix is not eligible
  struct nk_powers_plan *pw = nullptr;
  bool pw_tried = false;
  bool local_only = false;   // created without a global partition (helper matrices): nothing about it is collective
};
// s operator applications in one launch with the matrix held in registers (nk_powers.hip):
//   Y[:, p] = os_p·(A Y[:, p−1] − θ_p Y[:, p−1]),  Y[:, −1] = x0,  os_0 = *d_scal_first, os_p = *d_scal_rest (NULL = 1), θ NULL = 0
bool nk_csr_powers_ready(nk_csr *A);   // the matrix has a plan (built on first call) and no launch has timed out
int nk_csr_powers_dev(nk_csr *A, const double *d_x0, double *d_Y, int64_t ldy, int s, const double *d_scal_first,
                      const double *d_scal_rest, const double *d_theta, const int *d_skip);
int nk_csr_powers_check(nk_csr *A);    // NK_E_HIP once if a launch timed out (the plan is parked)
void nk_csr_powers_rearm(nk_csr *A);   // a parked plan comes back (one rank, fewer than three time-outs so far)
void nk_powers_plan_destroy(struct nk_powers_plan *P);
// d_out_scale (nullable): y = (*d_out_scale) · A x   (lagged normalisation of the Krylov basis)
int nk_csr_spmv_dev(nk_csr *A, const double *d_x, double *d_y, const int *d_skip, const double *d_out_scale = nullptr,
                    const nk_spmv_epi *epi = nullptr);
int nk_csr_spmv_t_dev(nk_csr *A, const double *d_x, double *d_y);
int nk_csr_colsumsq_dev(nk_csr *A, double *d_out);  // out_j = Σ_i A_ij² (diag AᵀA)
int nk_csr_add_to_diagonal_dev(nk_csr *A, double sigma, const double *d_m = nullptr);  // A += σ·diag(m) (m NULL: I)  // A ← A + σ I on the stored pattern (the diagonal must be stored)
// assembled normal matrix N = JᵀJ + λ·diag(d) on the pattern of JᵀJ (single rank; nk_csr.hip)
struct nk_normal_plan;
int nk_normal_plan_create(nk_csr *J, nk_normal_plan **out);
nk_csr *nk_normal_plan_matrix(nk_normal_plan *Pn);
int nk_normal_plan_values(nk_normal_plan *Pn, nk_csr *J, double lambda, const double *d_diag);
void nk_normal_plan_destroy(nk_normal_plan *Pn);
// local_only: every column is a local index already (rectangular helper matrices such as the transposed local block):
// no halo plan is built, i.e. the call is NOT collective
int nk_csr_create_local(nk_ctx *ctx, int64_t nrows, int64_t n_global, int64_t row_begin,
                        const std::vector<int32_t> &rowptr, const std::vector<int64_t> &gcol,
                        const double *vals_host, nk_csr **out, bool local_only = false);

// ----------------------------------------------------------------------------- problems
struct nk_problem {
  nk_ctx *ctx = nullptr;
  int kind = 0;
  int64_t n_local = 0, n_global = 0, row_begin = 0;
  double params[8] = {0};
  int nparams = 0;
  uint64_t params_version = 0;   // bumped by nk_problem_set_params (anything derived from the parameters and cached elsewhere checks it)
  bool replicated = false;  // every rank holds the WHOLE problem (coarsest multigrid level): no partition, no halo
  // grid problems
  int64_t ns = 0;           // side length
  int64_t j0 = 0, j1 = 0;   // owned grid lines [j0, j1)
  double c_lap = 0, c_exp = 0;  // Bratu coefficients
  nk_halo halo;             // grid-line halo (lower line(s) then upper line(s))
  double *d_diag = nullptr; // Bratu: c_exp*exp(u) at the linearisation point
  const double *d_u_lin = nullptr;
  // user problem
  nk_user_callbacks cb{};
  void *user = nullptr;
  nk_csr *user_pattern = nullptr;
  // forward-difference JVP for user problems without a jvp callback: f(u) at the linearisation point, u + εv, f(u + εv)
  double *d_fd_f0 = nullptr, *d_fd_up = nullptr, *d_fd_f1 = nullptr;
  // operator fallbacks through the Jacobian (prepare_jvp / prepare_vjp, SciMLJacobianOperators.jl:296-362,373-431): a private
  // copy of the jac_prototype pattern whose values are J(d_u_linJ) — filled by f.jac, or colour-compressed differences
  nk_csr *lin_J = nullptr;
  const double *d_u_linJ = nullptr;
  // staging buffers for host-memspace calls
  double *d_tmp[3] = {nullptr, nullptr, nullptr};
  // resident matrix-powers kernel for the matrix-free Bratu operator (nk_powers.hip); NULL = not eligible
  struct nk_powers_plan *pw = nullptr;
  bool pw_tried = false;
};
bool nk_problem_powers_ready(nk_problem *P);
int nk_problem_powers_dev(nk_problem *P, const double *d_u, const double *d_x0, double *d_Y, int64_t ldy, int s,
                          const double *d_scal_first, const double *d_scal_rest, const double *d_theta, const int *d_skip);
int nk_problem_powers_check(nk_problem *P);
void nk_problem_powers_rearm(nk_problem *P);
int nk_problem_create_bratu_replicated(nk_ctx *ctx, int64_t ns, double lambda, double scale, nk_problem **out);
int nk_problem_create_brus_replicated(nk_ctx *ctx, const double *params5, nk_problem **out);
int nk_problem_ghost_lines(nk_problem *P, const double *d_v, const double **lo, const double **hi);
int nk_problem_residual_dev(nk_problem *P, const double *d_u, double *d_f);
// gpart (nullable, 2·NK_MAX_RED_BLOCKS doubles, one rank): the Gershgorin partials {max −lo, max hi} per workgroup of J(d_u) —
// the numbers the fill kernel's own discs give (same expressions), here before that kernel has run
int nk_problem_residual_norms_dev(nk_problem *P, const double *d_u, double *d_f, double *partials, int *grid_out,
                                  double *f_copy = nullptr, double *ss_copy = nullptr, double *gpart = nullptr);
// forget what the problem was linearised at: the caller wrote new contents into a buffer it may have been keyed on
static inline void nk_problem_invalidate(nk_problem *P) { P->d_u_lin = nullptr; P->d_u_linJ = nullptr; }
int nk_csr_clone_pattern(nk_csr *A, nk_csr **out);  // same pattern and partition, own values (collective on several ranks)
int nk_problem_jvp_prepare(nk_problem *P, const double *d_u);  // linearise at u (u must stay alive)
int nk_problem_jvp_dev(nk_problem *P, const double *d_u, const double *d_v, double *d_jv, const int *d_skip,
                       const double *d_out_scale = nullptr, const struct nk_spmv_epi *epi = nullptr);
int nk_problem_vjp_dev(nk_problem *P, const double *d_u, const double *d_v, double *d_vj);
int nk_problem_spectrum_interval_dev(nk_problem *P, const double *d_u, double *d_out2);  // Bratu: {−lo, hi} of the stencil's discs
// the s-step cycle's begin (nk_sstep.hip: k_ss_cycle_begin — or, round 6, workgroup 0 of the Newton driver's speculative Jacobian
// fill, behind the norms' reduction: nk_gmres_begin_ahead): the stage-2 sum of ‖b‖² (k_reduce_sum's order), the max-reduction of a
// fill kernel's Gershgorin partials, the GMRES begin, the shifts and scales of the block basis
struct nk_gmres_ctl;
struct nk_gmres_pub;
constexpr int NK_SS_SMAX = 16;   // widest block
constexpr int NK_SS_TH = 8;      // scal[NK_SS_TH + j] = θ_j; scal[0..5): first-application scale, 1/σ, σ, carried σ estimate, Newton flag
struct nk_ss_begin_args {
  nk_gmres_ctl *ctl;               // nullptr: nothing to do (the folded form's "no begin")
  double *d_ss, *g, *s, *scal, *ival;
  const double *ss_part, *bpart, *nodes;
  nk_gmres_pub *pub;
  uint64_t seq;
  double atol, rtol;
  int fixed, first, m, ss_grid, bnblk, ns, newton;
};
// `fold` (optional): the stage-2 reduction of a residual kernel's norm partials — max |f|, Σ f², optionally a second sum — and
// their delivery to the host ride in workgroup 0 of the fill kernel instead of in a launch of their own (k_reduce_inf2): the
// Newton driver's speculative fill of the NEXT Jacobian sits directly behind the residual kernel, and its first workgroup has the
// norms out ≈ 3 µs into the launch while the others write the matrix. *folded: whether this problem's fill kernel took it (the
// Bratu fill on one rank does; otherwise the caller launches the reduction itself).
struct nk_fold_norms {
  const double *partials = nullptr;   // [nblk] maxima, then [nblk] sums (k_absmax_sumsq's layout)
  int nblk = 0;
  const double *extra = nullptr;      // [extra_n] further partial sums (nullable)
  int extra_n = 0;
  double *out = nullptr, *h_dst = nullptr;   // device scalars; coherent pinned host copy (nullable)
  uint64_t *h_seq = nullptr;
  uint64_t seq = 0;
};
// `begin` (optional, with `fold`): behind that reduction the same workgroup runs the NEXT linear solve's cycle begin
// (nk_ss_begin_body) — its inputs exist when the fill kernel starts (the residual kernel left Σ f² and, for the Bratu stencil,
// the Gershgorin partials of the very Jacobian this launch fills: nk_problem_residual_norms_dev's gpart).
int nk_problem_jac_values_dev(nk_problem *P, const double *d_u, nk_csr *J, const nk_fold_norms *fold = nullptr,
                              bool *folded = nullptr, const nk_ss_begin_args *begin = nullptr);
int nk_problem_jac_colored_dev(nk_problem *P, const double *d_u, nk_csr *J);  // ncolors JVPs + decompression

// ----------------------------------------------------------------------------- BLAS-1 launchers (device)
// reductions leave their (all-reduced) result in device memory at d_out
int nk_blas_dot(nk_ctx *ctx, int64_t n, const double *x, const double *y, double *d_out);
int nk_blas_sumsq(nk_ctx *ctx, int64_t n, const double *x, double *d_out);
int nk_blas_norm_inf(nk_ctx *ctx, int64_t n, const double *x, double *d_out);
// d_scales (nullable): per-column scale s_j of a lazily-normalised basis; h_j = s_j (ṽ_j·w), coefficient h_j s_j
int nk_blas_multidot(nk_ctx *ctx, int64_t n, int nv, const double *V, int64_t ldv, const double *w,
                     double *d_h, bool with_self, const int *d_skip, const double *d_scales);
// a Newton update riding in the pass that forms the linear solve's x: u_new = u_old + usign·w_new (out of place), the
// partial sums of (u_new − u_old)² per workgroup → partials[0 .. *grid_out) (k_newton_update's arithmetic)
struct nk_fused_update {
  const double *u_old = nullptr;
  double *u_new = nullptr;
  double usign = -1.0;
  double *partials = nullptr;
  int grid = 0;            // out: workgroups = partial sums written
  bool armed = false, done = false;
};
int nk_blas_multiaxpy(nk_ctx *ctx, int64_t n, int nv, const double *V, int64_t ldv, const double *d_h,
                      double sign, double *w, double *d_sumsq /*nullable*/, const int *d_skip,
                      const int *d_nv /*nullable: device count overrides nv*/, const double *d_scales,
                      bool overwrite = false /* w = sign·V h instead of w += … */, nk_fused_update *fu = nullptr);
constexpr int NK_MR_MAX = 6;  // inner products per nk_blas_multi_reduce launch
int nk_blas_multi_reduce(nk_ctx *ctx, int64_t n, int ndots, const double *const *xs, const double *const *ys,
                         const double *amax, const double *extra_partials, int extra_slots, int extra_n, double *d_out);
int nk_blas_reduce_one(nk_ctx *ctx, const double *partials, int nblk, double *d_out, const int *d_skip);
// y = x and *d_out = Σ x² (all-reduced) in one pass
int nk_blas_copy_sumsq(nk_ctx *ctx, int64_t n, const double *x, double *y, double *d_out);
// d_out[0] = max|x| (NaN-propagating), d_out[1] = Σ x²; optional third slot: Σ of `extra_partials[0..extra_n)` (per-block
// partial sums another kernel left behind) — one stage-2 launch and one fetch for the Newton driver's three norms
int nk_blas_norms_inf2(nk_ctx *ctx, int64_t n, const double *x, double *d_out, const double *extra_partials, int extra_n,
                       int have_partials = 0);
#define NK_SUMSQ_PARTIALS_ONLY ((double *)(uintptr_t)1)  // multiaxpy: leave ‖w‖² partials in ctx->d_partials_ss
// DCGS2 pass A: correct the pending column V[:,k] by −Σ a_j ṽ_j, turn V[:,k+1] (= s_k·A·pending) into the true next
// vector by −Σ b_j ṽ_j − b_k·corrected, and return d_h[0..k] = s_j·(ṽ_j·w) over the corrected basis
// DCGS2-1R sweeps: dots of the pending column p = V[:,k] (and, unless `flush`, of z = V[:,k+1] = A p) against the k final
// columns: d_red = [ṽ_j·p (k), p·p, ṽ_j·z (k), p·z], all-reduced, unscaled; axpy: p −= Σ a_j ṽ_j, z = b_{k+1} z − Σ b_j ṽ_j − b_k p
int nk_blas_dcgs2r_dots(nk_ctx *ctx, int64_t n, int k, bool flush, const double *V, int64_t ldv, double *d_red,
                        const int *d_skip);
int nk_blas_dcgs2r_axpy(nk_ctx *ctx, int64_t n, int k, double *V, int64_t ldv, const double *d_a, const double *d_b,
                        const int *d_skip, double *d_ss_out = nullptr);
int nk_blas_dcgs2_pass_a(nk_ctx *ctx, int64_t n, int k, double *V, int64_t ldv, const double *d_a, const double *d_b,
                         const double *d_scales, double *d_h, const int *d_skip);
int nk_blas_fused_axpy_dot(nk_ctx *ctx, int64_t n, int nv, const double *V, int64_t ldv, const double *d_h,
                           const double *d_scales, double *w, double *d_h2, const int *d_skip);
int nk_blas_axpby(nk_ctx *ctx, int64_t n, double a, const double *x, double b, double *y);  // y = a x + b y
int nk_blas_scale_to(nk_ctx *ctx, int64_t n, const double *d_scale, const double *x, double *y,
                     const int *d_skip);  // y = (*d_scale) * x
int nk_blas_copy(nk_ctx *ctx, int64_t n, const double *x, double *y);
int nk_blas_fill(nk_ctx *ctx, int64_t n, double a, double *y);
// z = a*x + b*y (three-operand)
int nk_blas_lincomb(nk_ctx *ctx, int64_t n, double a, const double *x, double b, const double *y, double *z);
int nk_scalars_to_host(nk_ctx *ctx, const double *d_src, int count, double *h_dst,
                       const std::function<int()> &before_wait = nullptr);  // synchronises
int nk_blas_minmax(nk_ctx *ctx, int64_t n, const double *x, double *d_out2 /*min,max*/);

// ----------------------------------------------------------------------------- multigrid preconditioner (nk_mg.hip)
struct nk_mg;
int nk_mg_create(nk_problem *P, int nu, int coarse_max, nk_mg **out);
int nk_mg_update(nk_mg *M, const double *d_u);
int nk_mg_apply(nk_mg *M, const double *src, double *dst, const int *d_skip);
int nk_mg_levels(const nk_mg *M);
void nk_mg_destroy(nk_mg *M);

// ----------------------------------------------------------------------------- GMRES
struct nk_gmres_ctl {  // lives in device memory, mirrored to pinned host memory
  int done, k, converged, failed, need_reorth, pad0, pad1, pad2;
  double tol, rnorm0, rnorm, inv_hn, hn, wnorm2_before, beta, r0;
};
// Written by the device into coherent pinned host memory, polled by the host (no copy, no stream synchronisation):
//   progress = seq << 16 | k << 1 | done   after k_gmres_begin (k = 0) and after every closed Hessenberg column
//   end_seq  = seq                          once the cycle's back-substitution has run; the fields below are then valid
struct nk_gmres_pub {
  uint64_t progress, end_seq;
  int k, converged, failed, pad;
  double rnorm0, rnorm;
};
// Blocks of the cycle whose stored columns were left at their FIRST pass (no sweep C): Q = (Q₁ − V_true C₂) R₂⁻¹ is never
// formed — everything that needs the true basis goes through (C₂, R₂): the back-substitution turns y into coefficients on the
// stored columns (block by block, last first), the next blocks' Gram products are carried into true coordinates
constexpr int NK_SS_NFIX = 3;   // (k_backsolve keeps their factors in 64 KB of static LDS next to the Hessenberg factor)
struct nk_ss_fix {
  int n;
  int k0[NK_SS_NFIX], sb[NK_SS_NFIX];
  const double *C2[NK_SS_NFIX], *R2[NK_SS_NFIX];
  // the same two factors in the form the NEXT blocks' reductions apply without dependent steps: Wi = R₂⁻¹ (sb × sb, upper) and
  // D = C₂ R₂⁻¹ (k0 × sb), left by the workgroup that derives the block's Hessenberg columns
  const double *Wi[NK_SS_NFIX], *D[NK_SS_NFIX];
};
struct nk_gmres {
  nk_ctx *ctx = nullptr;
  int64_t n = 0, ldv = 0;
  int m = 30, ortho = NK_ORTHO_CGS2;
  double *V = nullptr, *w = nullptr, *z = nullptr, *r = nullptr;
  double *x0_keep = nullptr;   // the warm start of a solve that runs on a resident matrix-powers plan (restored if a launch is torn)
  nk_fused_update fu;          // armed by the Newton driver for ONE solve (nk_gmres_arm_fused_update)
  struct { const double *b = nullptr, *ss = nullptr; int grid = 0; } pre;   // nk_gmres_preloaded_rhs (one solve)
  // nk_gmres_begin_ahead: the NEXT solve's cycle begin is in the queue already (inside a kernel of the caller's) — for exactly
  // these arguments and this set of operator values
  struct { bool valid = false; const double *b = nullptr, *val = nullptr; double atol = 0.0, rtol = 0.0;
           int maxiter = 0, fixed_iters = 0; uint64_t seq = 0; } ahead;
  double *d_Hraw = nullptr, *d_ca = nullptr, *d_cb = nullptr;  // DCGS2: un-rotated Hessenberg, pass-A coefficients
  double *d_tprev = nullptr, *d_red = nullptr;                 // DCGS2-1R: first-projection part of the open column, reduced dots
  double *d_h = nullptr, *d_h2 = nullptr, *d_R = nullptr, *d_cs = nullptr, *d_sn = nullptr,
         *d_g = nullptr, *d_y = nullptr, *d_ss = nullptr;
  double *d_s = nullptr;  // s_j: the basis is stored un-normalised, v_j = s_j ṽ_j (lagged normalisation)
  nk_gmres_ctl *d_ctl = nullptr, *h_ctl = nullptr;
  nk_gmres_pub *h_pub = nullptr, *h_pub_dev = nullptr;
  uint64_t cycle_seq = 0;
  int run_ahead = 0;  // > 0: at most that many Arnoldi steps are enqueued ahead of the device (0: whole cycle)
  // operator
  int op_kind = 0;  // 0 none, 1 csr, 2 problem jvp, 3 fn
  nk_csr *A = nullptr;
  nk_problem *P = nullptr;
  const double *d_u = nullptr;
  double *d_u_own = nullptr;
  nk_matvec_fn fn = nullptr;
  void *fn_user = nullptr;
  nk_matvec_fn prec = nullptr;
  void *prec_user = nullptr;
  bool normal = false;       // operator = AᵀA of the CSR / problem operator (normal form)
  double *nrm_tmp = nullptr; // A x between the two halves
  const double *nrm_diag = nullptr;  // damped normal form: AᵀA + nrm_lambda·diag(nrm_diag)
  double nrm_lambda = 0.0;
  double shift = 0.0;  // operator = A + shift·diag(m) (matrix-free pseudo-transient damping)
  const double *d_shift_w = nullptr;  // m (borrowed, local rows); NULL = identity
  bool fn_host = false, prec_host = false;  // the callbacks take HOST pointers: vectors are staged through h_stage
  double *h_stage = nullptr;                // pinned, 2 n doubles
  int prec_kind = 0;  // 0 none, 1 callback, 2 built-in Chebyshev polynomial, 3 built-in multigrid V-cycle, 4 nk_precond object
  struct nk_precond *rprec_obj = nullptr;
  // left preconditioner: Arnoldi runs on Pl⁻¹ A Pr⁻¹, residual norms are the preconditioned ones
  int lprec_kind = 0; // 0 none, 1 callback, 4 nk_precond object
  nk_matvec_fn lprec = nullptr;
  void *lprec_user = nullptr;
  bool lprec_host = false;
  struct nk_precond *lprec_obj = nullptr;
  double *lz = nullptr;  // A Pr⁻¹ x before Pl⁻¹
  struct nk_mg *mg = nullptr;
  int cheb_degree = 0;
  double cheb_lmin = 0, cheb_lmax = 0;
  double *cr = nullptr, *cd = nullptr, *ct = nullptr, *cd2 = nullptr;  // Chebyshev work vectors (cd/cd2 ping-pong)
  double *d_b = nullptr, *d_x = nullptr;  // staging for host-memspace calls
  // NK_GMRES_GRAPH=1 (A/B switch): the fixed-work cycle captured once into a HIP graph and replayed
  hipGraphExec_t gexec = nullptr;
  hipStream_t cap_stream = nullptr;
  bool graph_broken = false;
  const void *gkey[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  int gsteps = 0;
  // NK_ORTHO_SSTEP (nk_sstep.hip): s basis columns per block — matrix powers, then block CGS in Pythagorean form, twice
  struct nk_sstep *ss = nullptr;
  int ss_s = 0;           // block size s (1..16); 0 = automatic: 15 with the Newton basis, 6 with the monomial one
  int ss_basis = 0;       // NK_SS_BASIS_AUTO / _MONOMIAL / _NEWTON
  bool ss_ival_user = false;          // the caller supplied real bounds of the operator's spectrum
  double ss_ival[2] = {0.0, 0.0};
  int ss_breakdowns = 0;  // Cholesky breakdowns of a block (the rest of that solve ran with delayed CGS2)
  int ss_s_cap = 0;              // automatic block size only: narrowed (15 → 8 → 4) after a block lost rank; 0 = not narrowed
  int ss_cycle_idx = 0;          // restart cycle of the current solve (0-based)
  nk_ss_fix ss_fix{};            // blocks of the running cycle left at their first pass (no sweep C): k_backsolve adapts y
  bool ss_grow = false;          // this solve stops on a tolerance: automatic block sizes start small and double (4, 8, 15 …)
  int peer_err_seen = 0;         // the peer arena's cumulative time-out count as of the last cycle this object waited for
  int ss_force_break_cycle = -1; // development hook (nk_gmres_debug_force_breakdown): that cycle's first block "loses rank"
};
int nk_gmres_solve_dev(nk_gmres *G, const double *d_b, double *d_x, int use_x0, double atol, double rtol,
                       int maxiter, int fixed_iters, nk_gmres_info *info);

// aggregation algebraic multigrid built from a CSR matrix (nk_amg.hip); the object behind nk_precond_create_amg
struct nk_amg;
int nk_amg_create(nk_csr *A, const nk_amg_params *prm, nk_amg **out);
int nk_amg_update(nk_amg *M);                       // new values, same pattern
int nk_amg_apply_dev(nk_amg *M, const double *d_b, double *d_x, const int *d_skip);   // x = one V-cycle applied to b (b ≠ x)
void nk_amg_destroy(nk_amg *M);
int nk_amg_levels(const nk_amg *M);
int nk_amg_level_info(const nk_amg *M, int l, int64_t *n, int64_t *nnz, double *lmax);
int nk_amg_matching(const nk_amg *M);   // 1 = sequential pairwise pass (host set-up), 2 = handshaking (device set-up)
const int32_t *nk_amg_aggregates(const nk_amg *M, int l);   // host: row → coarse row of level l (NULL on the coarsest)
// in-place inverse of ONE dense n × n matrix, n ≤ 128, column-major with leading dimension ld, row pivoting (nk_bcr.hip)
int nk_dense_invert128_dev(nk_ctx *ctx, double *d_M, int ld, int n, int *d_fail);

// preconditioner objects (nk_precond.hip)
int nk_precond_apply_dev(struct nk_precond *P, const double *d_x, double *d_y, const int *d_skip);
int64_t nk_precond_size(struct nk_precond *P);

// s-step Arnoldi (nk_sstep.hip)
// y = scale·(A M⁻¹ x − θ x); d_scale / d_theta (device scalars) may be nullptr (= 1 / 0)
int nk_gmres_op_apply(nk_gmres *G, const double *d_x, double *d_y, const int *d_skip, const double *d_scale,
                      const double *d_theta = nullptr);
int nk_gmres_op_powers(nk_gmres *G, const double *d_x, double *d_Y, int64_t ldy, int s, const int *d_skip, const double *d_scale,
                       const double *d_theta, bool *done);
// real bounds [lo, hi] of the operator's spectrum, on the device as {−lo, hi} (all-reduced); false: none are known
int nk_gmres_spectrum_interval_dev(nk_gmres *G, double *d_out2, const double **where, bool *have);
// {−min_i(a_ii − r_i), max_i(a_ii + r_i)} of the local rows: *where = the matrix's cached bounds (left by a fused fill kernel)
// or d_out2 after a pass over the matrix
int nk_csr_gershgorin_dev(nk_csr *A, double *d_out2, const double **where);
int nk_csr_bounds_from_partials(nk_csr *A, const double *d_part, int nblk);   // fill kernels: {max −lo, max hi} per block → cache
bool nk_ss_eligible(const nk_gmres *G);
int nk_ss_prepare(nk_gmres *G);   // once per linear solve: Newton basis (spectrum bounds known) or monomial
int nk_ss_block_width(int want);
// start of a restart cycle (thread 0 of k_gmres_begin / of the s-step form's k_ss_cycle_begin): β = ‖r₀‖ from its square,
// tolerance of a new solve, flags, the first column's scale, the right-hand side of the small least-squares problem
#ifdef __HIPCC__
// stage 2 of the norms [max |·| (NaN-propagating), Σ, Σ]: ONE 256-thread workgroup, fixed order — k_reduce_inf2 (nk_blas.hip) as a
// launch of its own, workgroup 0 of a fill kernel when folded (nk_fold_norms): the same arithmetic either way
__device__ __forceinline__ double nk_nanmax(double a, double b) { return (a != a || b != b) ? __builtin_nan("") : (a > b ? a : b); }
__device__ __forceinline__ void nk_reduce_inf2_body(const nk_fold_norms &f, double *sm /* [12] shared */) {
  double m = -__builtin_inf(), s = 0.0, e = 0.0;
  for (int i = threadIdx.x; i < f.nblk; i += NK_BLOCK) { m = nk_nanmax(m, f.partials[i]); s += f.partials[f.nblk + i]; }
  for (int i = threadIdx.x; i < f.extra_n; i += NK_BLOCK) e += f.extra[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    m = nk_nanmax(m, __shfl_xor(m, o, 64));
    s += __shfl_xor(s, o, 64);
    e += __shfl_xor(e, o, 64);
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sm[w] = m; sm[4 + w] = s; sm[8 + w] = e; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double o0 = nk_nanmax(nk_nanmax(sm[0], sm[1]), nk_nanmax(sm[2], sm[3]));
    const double o1 = (sm[4] + sm[5]) + (sm[6] + sm[7]);
    const double o2 = (sm[8] + sm[9]) + (sm[10] + sm[11]);
    f.out[0] = o0;
    f.out[1] = o1;
    if (f.extra != nullptr) f.out[2] = o2;
    if (f.h_dst != nullptr) {
      f.h_dst[0] = o0;
      f.h_dst[1] = o1;
      if (f.extra != nullptr) f.h_dst[2] = o2;
      __threadfence_system();
      __hip_atomic_store(f.h_seq, f.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
__device__ __forceinline__ void nk_gmres_begin_body(nk_gmres_ctl *ctl, double ss, double atol, double rtol, int fixed, int first,
                                                    double *g, double *s, int m, nk_gmres_pub *pub, uint64_t seq) {
  const double beta = sqrt(ss);
  if (first) {  // 1: a new solve; 2: the cycle after an s-step breakdown — same solve, same tolerance, flags cleared
    if (first == 1) {
      ctl->rnorm0 = beta;
      ctl->tol = fixed ? -1.0 : atol + rtol * beta;
    }
    ctl->failed = 0;
    ctl->converged = 0;
  }
  ctl->beta = beta;
  ctl->rnorm = beta;
  ctl->k = 0;
  ctl->need_reorth = 0;
  ctl->pad0 = 0;
  const int bad = !(beta == beta) || isinf(beta);
  if (bad) ctl->failed = 1;
  if (!bad && (beta == 0.0 || (ctl->tol >= 0.0 && beta <= ctl->tol))) ctl->converged = 1;
  ctl->done = (ctl->failed || ctl->converged) ? 1 : 0;
  ctl->inv_hn = (beta > 0.0 && !bad) ? 1.0 / beta : 0.0;
  s[0] = ctl->inv_hn;
  g[0] = beta;
  for (int i = 1; i <= m; ++i) g[i] = 0.0;
  if (pub != nullptr)
    __hip_atomic_store(&pub->progress, (seq << 16) | (uint64_t)(ctl->done ? 1 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// (ONE 256-thread workgroup; sm: 16 doubles of LDS)
__device__ __forceinline__ void nk_ss_begin_body(const nk_ss_begin_args &a, double *sm) {
  double *sh_ch = sm + 12;   // centre, half width, Newton basis in effect
  const int t = threadIdx.x, wv = t >> 6;
  // (requested before the reductions: a lone thread's loads behind its own stores cost a round trip each — the 15 shifts alone
  //  were 7 of this launch's 12 µs)
  const double node_t = (t < NK_SS_SMAX) ? a.nodes[t] : 0.0;
  const double sigma_old = (t == 0) ? a.scal[3] : 0.0;
  double v = 0.0, blo = -INFINITY, bhi = -INFINITY;
  if (a.ss_part != nullptr)
    for (int i = t; i < a.ss_grid; i += 256) v += a.ss_part[i];
  if (a.bpart != nullptr) {
    for (int base = 0; base < a.bnblk; base += 2048) {   // sixteen loads in flight per lane (a rolled loop: one round trip each)
      double x[8], y[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int i = base + t + 256 * j, ic = i < a.bnblk ? i : a.bnblk - 1;   // (clamped: max is idempotent)
        x[j] = a.bpart[ic];
        y[j] = a.bpart[a.bnblk + ic];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) { blo = fmax(blo, x[j]); bhi = fmax(bhi, y[j]); }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    v += __shfl_xor(v, o, 64);
    blo = fmax(blo, __shfl_xor(blo, o, 64));
    bhi = fmax(bhi, __shfl_xor(bhi, o, 64));
  }
  if ((t & 63) == 0) { sm[wv] = v; sm[4 + wv] = blo; sm[8 + wv] = bhi; }
  __syncthreads();
  if (t == 0) {
  double ss;
  if (a.ss_part != nullptr) {
    ss = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    *a.d_ss = ss;
  } else {
    ss = *a.d_ss;
  }
  double lo_neg = 0.0, hi = 0.0;
  if (a.newton) {
    if (a.bpart != nullptr) {
      lo_neg = fmax(fmax(sm[4], sm[5]), fmax(sm[6], sm[7]));
      hi = fmax(fmax(sm[8], sm[9]), fmax(sm[10], sm[11]));
      a.ival[0] = lo_neg;
      a.ival[1] = hi;
    } else {
      lo_neg = a.ival[0];
      hi = a.ival[1];
    }
  }
  nk_gmres_begin_body(a.ctl, ss, a.atol, a.rtol, a.fixed, a.first, a.g, a.s, a.m, a.pub, a.seq);
  double *scal = a.scal;
  double sigma = sigma_old;
  double newton = 0.0, c = 0.0, h = 0.0;
  if (a.newton) {
    const double lo = -lo_neg;
    c = 0.5 * (lo + hi);
    h = 0.5 * (hi - lo);
    if (h > 0.0 && !isinf(h) && c == c && !isinf(c)) {
      sigma = exp2(rint(log2(0.5 * h)));
      newton = 1.0;
    }
  }
  sh_ch[0] = c; sh_ch[1] = h; sh_ch[2] = newton;
  if (!(sigma > 0.0) || isinf(sigma)) sigma = 1.0;
  scal[4] = newton;
  scal[3] = sigma;
  scal[2] = sigma;
  scal[1] = 1.0 / sigma;
  {  // s[0] as nk_gmres_begin_body has just stored it (recomputed: reading it back is a memory round trip)
    const double beta = sqrt(ss);
    const bool bad = !(beta == beta) || isinf(beta);
    scal[0] = ((beta > 0.0 && !bad) ? 1.0 / beta : 0.0) / sigma;
  }
  }
  __syncthreads();
  if (t < NK_SS_SMAX)   // the shifts, one lane each
    a.scal[NK_SS_TH + t] = (sh_ch[2] != 0.0 && t < a.ns) ? sh_ch[0] + sh_ch[1] * node_t : 0.0;
}
#endif
// s-step form: the cycle's begin kernel (gmres begin + the block basis' shifts and scales; on one rank also the stage-2
// reductions of ‖b‖² and of the Jacobian fill's Gershgorin partials, which are launches of their own otherwise)
int nk_ss_begin_cycle(nk_gmres *G, double atol, double rtol, int fixed, int first, uint64_t seq, const double *ss_partials,
                      int ss_grid);
// Everything of a CSR matrix that belongs to one SET OF VALUES on its fixed pattern: the value array, the fill kernel's
// Gershgorin partials and the flags of the caches derived from the values. Swapping it lets a second value set be filled
// behind the back of the live one (the Newton driver's speculative Jacobian fill, nk_solver.hip) and take its place later.
struct nk_csr_valstate {
  double *d_val = nullptr, *d_gersh = nullptr;
  int gersh_cap = 0;
  bool t_values_stale = true, bounds_valid = false, bounds_pending = false;
  const double *bounds_part = nullptr;
  int bounds_nblk = 0;
};
nk_csr_valstate nk_csr_get_valstate(const nk_csr *A);
void nk_csr_set_valstate(nk_csr *A, const nk_csr_valstate &v);
int nk_csr_alloc_values(nk_csr *A, double **out);   // a zero-padded value array of A's size (freed with hipFree)
// fill kernels' Gershgorin partials not reduced yet: hands them to a caller that reduces them into *dst in its own kernel
bool nk_csr_take_pending_bounds(nk_csr *A, const double **part, int *nblk, double **dst);
double *nk_csr_bounds_word(nk_csr *A);         // {−lo, hi} of the matrix's Gershgorin bounds (device; allocated on first use; NULL on failure)
void nk_csr_invalidate_bounds(nk_csr *A);     // the bounds word no longer belongs to the live values (recomputed on demand)
void nk_csr_commit_pending_bounds(nk_csr *A);   // the caller's reducing kernel is enqueued: the partials are no longer pending
int nk_blas_copy_sumsq_stage1(nk_ctx *ctx, int64_t n, const double *x, double *y, int *grid_out);
// (have_partials > 0: stage 1 has run inside the kernel that produced x — ctx->d_partials holds its have_partials workgroups' results)
// fold_into (optional, one rank, have_partials > 0): called INSTEAD of launching the stage-2 reduction, with what a kernel of the
// caller's needs to perform it in its first workgroup (nk_fold_norms); it returns whether such a kernel was enqueued — if not the
// reduction is launched as usual. Either way before_wait runs next, then the wait.
int nk_blas_norms_inf2_to_host(nk_ctx *ctx, int64_t n, const double *x, double *d_out, const double *extra_partials, int extra_n,
                               double *h_out, const std::function<int()> &before_wait = nullptr, int have_partials = 0,
                               const std::function<int(const nk_fold_norms &, bool *)> &fold_into = nullptr);
// the cycle's last s-step block as k_backsolve needs it (sb = 0: nothing to adapt); resets the record
nk_ss_fix nk_ss_take_last_block(nk_gmres *G);
int nk_ss_block_size(const nk_gmres *G);   // the block size in effect
int nk_ss_cycle(nk_gmres *G, int steps, const std::function<bool(int)> &wait_progress, bool *backsolved);
// The NEXT solve's last pass (x = V y) also forms u_new = u_old + usign·x and the partial sums of ‖u_new − u_old‖² — if that solve
// is a single cycle from a zero guess without a right preconditioner. nk_gmres_take_fused_update: whether it happened (disarms).
void nk_gmres_arm_fused_update(nk_gmres *G, const double *u_old, double *u_new, double usign, double *partials);
// Column 0 of the basis already holds b, and ss_partials[0 .. grid) sum to ‖b‖² (the kernel that produced b stored it twice):
// the NEXT solve — if its right-hand side is this b, from a zero guess, in the s-step form on one rank without a left
// preconditioner — skips the pass that copies b there. nk_gmres_rhs_column: where that kernel writes; NULL if this object's
// next solve could not use it anyway.
double *nk_gmres_rhs_column(nk_gmres *G);
void nk_gmres_preloaded_rhs(nk_gmres *G, const double *d_b, const double *ss_partials, int grid);
// The NEXT solve's cycle begin (k_ss_cycle_begin's work) handed to a kernel of the caller's instead of being launched: fills
// *out for nk_ss_begin_body when that solve will be the fixed-work, single-cycle, zero-guess s-step solve on one rank with a
// CSR operator and the Newton basis, its right-hand side d_b in column 0 already (nk_gmres_preloaded_rhs), ‖b‖² as the partial
// sums ss_partials[0 .. ss_grid), the operator's Gershgorin partials {max −lo, max hi} in bpart[0 .. 2·bnblk) — *done says
// whether. The caller must enqueue a kernel that runs nk_ss_begin_body(*out) behind the producers of those partials, on the
// operator values the solve will see. nk_gmres_solve_dev skips its begin when its arguments and the operator's value array are
// the ones recorded here, and otherwise starts from scratch; nk_gmres_drop_ahead forgets the record (every nk_gmres_set_* does).
int nk_gmres_begin_ahead(nk_gmres *G, const double *d_b, const double *ss_partials, int ss_grid, const double *bpart, int bnblk,
                         double atol, double rtol, int maxiter, int fixed_iters, nk_ss_begin_args *out, bool *done);
void nk_gmres_drop_ahead(nk_gmres *G);
void nk_gmres_ahead_values(nk_gmres *G, const double *d_val);   // the value array the begin's bounds belong to (set once its fill is enqueued)
// (nk_sstep.hip) the begin's argument block for this object's workspace; first: 1 a new solve
int nk_ss_prepare_ahead(nk_gmres *G, const double *bpart, int bnblk, double *dst);
int nk_ss_begin_args_for(nk_gmres *G, double atol, double rtol, int fixed, int first, uint64_t seq, const double *ss_partials,
                         int ss_grid, nk_ss_begin_args *out);
bool nk_gmres_take_fused_update(nk_gmres *G, int *grid);
void nk_ss_destroy(struct nk_sstep *W);
int nk_ss_grid(nk_ctx *ctx, int64_t n, int k, int s);
int nk_ss_grid_a(nk_ctx *ctx, int64_t n, int k, int s, bool hosting);   // sweep A's own grid (read-only: one workgroup per CU)
struct ss_tail_args;
int nk_ss_sweep(nk_ctx *ctx, int mode, int64_t n, int k, int s, double *V, int64_t ldv, const double *coef, double *partials,
                const int *d_skip, int grid, const ss_tail_args *tap, int *mark, int hk = 0, int hs = 0, int flags = 0 /* 1: sweep B stores nothing */);
int nk_blas_reduce_slots(nk_ctx *ctx, const double *partials, int nblk, int nslots, double *d_out, const int *d_skip);
int nk_blas_reduce_slots_allreduce(nk_ctx *ctx, const double *partials, int nblk, int nslots, double *d_out, const int *d_skip);

// ----------------------------------------------------------------------------- banded LU (direct linsolve, C2)
struct nk_bandlu {
  nk_ctx *ctx = nullptr;
  int64_t n = 0;
  int kl = 0, ku = 0, ldab = 0, nblk = 0;
  double *AB = nullptr;    // ldab × n band storage
  double *invL = nullptr;  // nblk × 32 × 32: L11⁻¹ of every diagonal block
  double *invU = nullptr;  // nblk × 32 × 32: U11⁻¹
  double *tmp = nullptr;
  int *d_fail = nullptr;
  struct nk_bcr *bcr = nullptr;  // block cyclic reduction engine (nk_bcr.hip); when set, factor/solve go through it
};
// engine: 0 = automatic (block cyclic reduction where it applies, else the band LU), 1 = band LU (NK_DIRECT=band forces it)
int nk_bandlu_create(nk_csr *A, nk_bandlu **out, int engine = 0);
struct nk_bcr;
int nk_bcr_create(nk_ctx *ctx, int64_t n, int b, nk_bcr **out);
int nk_bcr_factor(nk_bcr *S, nk_csr *A, int *ok);
int nk_bcr_solve(nk_bcr *S, const double *d_b, double *d_x);
void nk_bcr_destroy(nk_bcr *S);
int64_t nk_bcr_bytes(int64_t n, int b);
int nk_bcr_set_pivoting(nk_bcr *S, int on, int *changed);
void nk_bandlu_destroy(nk_bandlu *B);
int nk_bandlu_factor(nk_bandlu *B, nk_csr *A, int *ok);
int nk_bandlu_solve(nk_bandlu *B, const double *d_b, double *d_x);

// ----------------------------------------------------------------------------- misc helpers
// The library's "synchronous" memsets and copies, ORDERED ON THE CONTEXT'S STREAM. hipMemset / hipMemcpy run on the null stream:
// a caller's stream created with hipStreamNonBlocking (PyTorch's, AMDGPU.jl's) is not ordered against it, and hipMemset of device
// memory does not block the host — a buffer zeroed at allocation time was still being zeroed while the first kernel on the
// context's stream filled it (round 6: the spare Jacobian value set of the speculative fill, found by running a context on a
// non-default stream — tools/shared_device_probe.py --mode threads). nk_memset: asynchronous, in stream order. nk_memcpy: in
// stream order and complete on return (the host side may be a temporary).
static inline hipError_t nk_memset(const nk_ctx *ctx, void *p, int v, size_t bytes) {
  return hipMemsetAsync(p, v, bytes, ctx->stream);
}
static inline hipError_t nk_memcpy(const nk_ctx *ctx, void *dst, const void *src, size_t bytes, hipMemcpyKind kind) {
  const hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, ctx->stream);
  return e != hipSuccess ? e : hipStreamSynchronize(ctx->stream);
}
template <typename T>
static inline int nk_dev_alloc(T **p, size_t count) {
  *p = nullptr;
  if (count == 0) return NK_OK;
  hipError_t e = hipMalloc((void **)p, count * sizeof(T));
  if (e != hipSuccess) {
    nk_set_error("hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(e));
    return NK_E_NOMEM;
  }
  return NK_OK;
}
static inline int nk_grid_for(int64_t work_items, int per_block, int max_blocks) {
  int64_t b = (work_items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > max_blocks) b = max_blocks;
  return (int)b;
}
