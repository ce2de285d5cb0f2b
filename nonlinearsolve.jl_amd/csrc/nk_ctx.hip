// Context, error reporting, communicator (RCCL over xGMI via dlopen, or host callbacks) and halo exchange.
#include <dlfcn.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>

#include "nk_internal.h"
#include <cmath>

// ----------------------------------------------------------------------------- errors
static thread_local char g_err[1024] = "";
void nk_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char *nk_last_error(void) { return g_err; }
extern "C" const char *nk_version(void) { return "mi355x_nk 0.1 (gfx950)"; }

extern "C" int nk_device_count(int *count) {
  NK_REQUIRE(count, "count is NULL");
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) {
    *count = 0;
    NK_FAIL(NK_E_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  *count = c;
  return NK_OK;
}

// ----------------------------------------------------------------------------- context
extern "C" int nk_ctx_create(int device_id, void *stream, nk_ctx **out) {
  NK_REQUIRE(out, "out is NULL");
  *out = nullptr;
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess || c == 0)
    NK_FAIL(NK_E_HIP, "no HIP device available (%s); libmi355x_nk has no CPU fallback",
            e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  NK_REQUIRE(device_id >= 0 && device_id < c, "device_id %d out of range [0,%d)", device_id, c);
  NK_HIP(hipSetDevice(device_id));
  nk_ctx *ctx = new nk_ctx();
  auto guard = nk_make_guard(ctx, [](nk_ctx *c) { nk_ctx_destroy(c); });
  ctx->device = device_id;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) ctx->num_cus = prop.multiProcessorCount;
  ctx->stream = (hipStream_t)stream;  // NULL = the device's default (null) stream, as in every HIP API
  NK_TRY(nk_dev_alloc(&ctx->d_partials, (size_t)(2 * NK_MAX_NV + 2) * NK_MAX_ROW_TILES));  // DCGS2 dot sweep: 2k+2 slots
  NK_TRY(nk_dev_alloc(&ctx->d_partials_ss, (size_t)NK_MAX_RED_BLOCKS));
  NK_TRY(nk_dev_alloc(&ctx->d_scal, (size_t)4 * NK_MAX_NV));
  // coherent (fine-grained) pinned memory: kernels store scalars and a sequence word straight into it, the host polls
  NK_HIP(hipHostMalloc((void **)&ctx->h_pinned, sizeof(double) * 4 * NK_MAX_NV + 64, hipHostMallocCoherent | hipHostMallocMapped));
  NK_HIP(hipHostGetDevicePointer((void **)&ctx->h_pinned_dev, ctx->h_pinned, 0));
  ctx->h_seq = reinterpret_cast<uint64_t *>(ctx->h_pinned + 4 * NK_MAX_NV);
  ctx->h_seq_dev = reinterpret_cast<uint64_t *>(ctx->h_pinned_dev + 4 * NK_MAX_NV);
  *ctx->h_seq = 0;
  const char *ov = getenv("NK_HALO_OVERLAP");
  if (ov && atoi(ov) != 0) NK_TRY(nk_ctx_set_halo_overlap(ctx, 1));
  *out = guard.release();
  return NK_OK;
}

// Overlap of the SpMV's halo exchange with its interior rows: the exchange is enqueued on a second
// stream between two events, so the compute stream only waits for it before the row blocks that read halo columns.
extern "C" int nk_ctx_set_halo_overlap(nk_ctx *ctx, int on) {
  NK_REQUIRE(ctx, "ctx is NULL");
  NK_HIP(hipSetDevice(ctx->device));
  if (on && !ctx->comm_stream) {
    NK_HIP(hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
    NK_HIP(hipEventCreateWithFlags(&ctx->ev_halo_ready, hipEventDisableTiming));
    NK_HIP(hipEventCreateWithFlags(&ctx->ev_halo_done, hipEventDisableTiming));
  }
  ctx->halo_overlap = on ? 1 : 0;
  return NK_OK;
}

static void nk_peer_destroy(nk_ctx *ctx);
extern "C" int nk_ctx_destroy(nk_ctx *ctx) {
  if (!ctx) return NK_OK;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  nk_comm_destroy(ctx);
  nk_peer_destroy(ctx);
  hipFree(ctx->d_partials);
  hipFree(ctx->d_partials_ss);
  hipFree(ctx->audit.d_slots);
  hipFree(ctx->d_scal);
  hipHostFree(ctx->h_pinned);
  if (ctx->comm_stream) {
    hipStreamSynchronize(ctx->comm_stream);
    hipStreamDestroy(ctx->comm_stream);
    hipEventDestroy(ctx->ev_halo_ready);
    hipEventDestroy(ctx->ev_halo_done);
  }
  if (ctx->own_stream) hipStreamDestroy(ctx->stream);
  delete ctx;
  return NK_OK;
}
extern "C" int nk_ctx_set_stream(nk_ctx *ctx, void *stream) {
  NK_REQUIRE(ctx, "ctx is NULL");
  NK_HIP(hipStreamSynchronize(ctx->stream));
  ctx->stream = (hipStream_t)stream;
  return NK_OK;
}
extern "C" int nk_ctx_synchronize(nk_ctx *ctx) {
  NK_REQUIRE(ctx, "ctx is NULL");
  NK_HIP(hipStreamSynchronize(ctx->stream));
  return NK_OK;
}
extern "C" int nk_device_alloc(nk_ctx *ctx, int64_t bytes, void **out) {
  NK_REQUIRE(ctx && out && bytes >= 0, "bad argument");
  NK_HIP(hipSetDevice(ctx->device));
  *out = nullptr;
  if (bytes == 0) return NK_OK;
  hipError_t e = hipMalloc(out, (size_t)bytes);
  if (e != hipSuccess) NK_FAIL(NK_E_NOMEM, "hipMalloc(%lld bytes) failed: %s", (long long)bytes, hipGetErrorString(e));
  return NK_OK;
}
extern "C" int nk_device_free(nk_ctx *ctx, void *ptr) {
  NK_REQUIRE(ctx, "NULL argument");
  NK_HIP(hipSetDevice(ctx->device));
  if (ptr) {
    NK_HIP(hipStreamSynchronize(ctx->stream));
    NK_HIP(hipFree(ptr));
  }
  return NK_OK;
}
extern "C" int nk_device_copy(nk_ctx *ctx, void *dst, const void *src, int64_t bytes, int kind) {
  NK_REQUIRE(ctx && (bytes == 0 || (dst && src)) && bytes >= 0, "bad argument");
  NK_REQUIRE(kind >= 0 && kind <= 2, "kind must be 0 (host→device), 1 (device→host) or 2 (device→device)");
  NK_HIP(hipSetDevice(ctx->device));
  if (bytes == 0) return NK_OK;
  const hipMemcpyKind k = kind == 0 ? hipMemcpyHostToDevice : kind == 1 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
  NK_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, k, ctx->stream));
  NK_HIP(hipStreamSynchronize(ctx->stream));
  return NK_OK;
}
// ---- BLAS-1 on resident vectors (DEVICE pointers of local length n): what a host language needs to move a library-owned
// buffer through the reference's step! without a GPU array package of its own — `@bb axpy!(α, δu, u)`, `copyto!`
// (lib/NonlinearSolveFirstOrder/src/solve.jl:403,438,460); the reductions are nk_dot / nk_nrm2 / nk_norm_inf (nk_blas.hip).
extern "C" int nk_vec_axpby(nk_ctx *ctx, int64_t n, double a, const double *x, double b, double *y) {
  NK_REQUIRE(ctx && (n == 0 || (x && y)) && n >= 0, "bad argument");
  NK_HIP(hipSetDevice(ctx->device));
  return n ? nk_blas_axpby(ctx, n, a, x, b, y) : NK_OK;
}
extern "C" int nk_vec_fill(nk_ctx *ctx, int64_t n, double a, double *y) {
  NK_REQUIRE(ctx && (n == 0 || y) && n >= 0, "bad argument");
  NK_HIP(hipSetDevice(ctx->device));
  return n ? nk_blas_fill(ctx, n, a, y) : NK_OK;
}
// the communicator's all-reduce on a DEVICE buffer (in place), through whatever transport the context holds — for self-checks
// of a multi-GPU set-up (tools/multi_gpu_selfcheck.py) and for host-language code that needs a global reduction of its own
extern "C" int nk_ctx_comm_allreduce(nk_ctx *ctx, double *buf, int count, int op) {
  NK_REQUIRE(ctx && buf && count > 0 && (op == 0 || op == 1), "bad argument");
  NK_HIP(hipSetDevice(ctx->device));
  NK_TRY(nk_comm_allreduce(ctx, buf, count, op));
  NK_HIP(hipStreamSynchronize(ctx->stream));
  return NK_OK;
}
extern "C" int nk_ctx_set_deterministic(nk_ctx *ctx, int d) {
  NK_REQUIRE(ctx, "ctx is NULL");
  ctx->deterministic = d ? 1 : 0;
  return NK_OK;
}

// ----------------------------------------------------------------------------- kernel-family profiling
static const char *k_names[NK_K_COUNT] = {"spmv", "multidot", "multiaxpy", "jvp", "residual", "scale",
                                          "reduce_small", "jacfill", "newton_update", "other", "spmv_powers"};
void nk_prof_scope_begin(nk_ctx *ctx, int id, double bytes) {
  ctx->prof.cur_id = id;
  ctx->prof.cur_bytes = bytes;
}
void nk_prof_scope_end(nk_ctx *ctx) { ctx->prof.cur_id = -1; }
// hands out the event pair for the next launch of the active scope (false: no scope active → plain launch)
bool nk_prof_next(nk_ctx *ctx, hipEvent_t *start, hipEvent_t *stop) {
  nk_prof &p = ctx->prof;
  if (p.cur_id < 0) return false;
  if (p.used + 2 > p.ev.size()) {
    if (p.ev.size() >= 16384) nk_prof_flush(ctx);
    else {
      const size_t old = p.ev.size();
      p.ev.resize(old + 2048);
      for (size_t i = old; i < p.ev.size(); ++i) hipEventCreate(&p.ev[i]);
    }
  }
  p.ids.push_back(p.cur_id);
  p.nbytes.push_back(p.cur_bytes);  // a scope's bytes are attributed to its first launch
  p.cur_bytes = 0.0;
  *start = p.ev[p.used];
  *stop = p.ev[p.used + 1];
  p.used += 2;
  return true;
}
void nk_prof_flush(nk_ctx *ctx) {
  nk_prof &p = ctx->prof;
  if (p.used == 0) return;
  hipStreamSynchronize(ctx->stream);
  for (size_t r = 0; r < p.ids.size(); ++r) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.ev[2 * r], p.ev[2 * r + 1]) == hipSuccess) {
      p.ms[p.ids[r]] += ms;
      p.bytes[p.ids[r]] += p.nbytes[r];
      if (p.nbytes[r] > 0.0 || p.ids[r] == NK_K_REDUCE_SMALL || p.ids[r] == NK_K_OTHER) p.count[p.ids[r]]++;
    }
  }
  p.ids.clear();
  p.nbytes.clear();
  p.used = 0;
}
extern "C" int nk_ctx_profile_enable(nk_ctx *ctx, int on) {
  NK_REQUIRE(ctx, "ctx is NULL");
  nk_prof_flush(ctx);
  ctx->prof.on = on != 0;
  if (on) {
    for (int i = 0; i < NK_K_COUNT; ++i) { ctx->prof.ms[i] = 0; ctx->prof.bytes[i] = 0; ctx->prof.count[i] = 0; }
  }
  return NK_OK;
}
extern "C" int nk_ctx_profile_query(nk_ctx *ctx, int kernel_id, const char **name, int64_t *launches, double *ms,
                                    double *bytes) {
  NK_REQUIRE(ctx, "ctx is NULL");
  NK_REQUIRE(kernel_id >= 0 && kernel_id < NK_K_COUNT, "kernel_id out of range");
  nk_prof_flush(ctx);
  if (name) *name = k_names[kernel_id];
  if (launches) *launches = ctx->prof.count[kernel_id];
  if (ms) *ms = ctx->prof.ms[kernel_id];
  if (bytes) *bytes = ctx->prof.bytes[kernel_id];
  return NK_OK;
}
extern "C" int nk_ctx_profile_kernel_count(void) { return NK_K_COUNT; }

extern "C" int nk_partition_range(int64_t n_global, int64_t granule, int nranks, int rank,
                                  int64_t *row_begin, int64_t *row_end) {
  NK_REQUIRE(row_begin && row_end, "NULL output");
  NK_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank %d / %d", rank, nranks);
  if (granule < 1) granule = 1;
  NK_REQUIRE(n_global % granule == 0, "n_global %lld not a multiple of granule %lld",
             (long long)n_global, (long long)granule);
  int64_t units = n_global / granule;
  *row_begin = (units * rank / nranks) * granule;
  *row_end = (units * (rank + 1) / nranks) * granule;
  return NK_OK;
}


// ----------------------------------------------------------------------------- peer-mapped arenas (hipIpc over xGMI)

__device__ __forceinline__ bool peer_wait_ge(const uint64_t *flag, uint64_t seq, uint64_t *err) {
  const unsigned long long t0 = wall_clock64();
  while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
    // once a few waits have timed out the communicator is broken: fail fast instead of spending 5 s on every collective
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 4) return false;
    if (wall_clock64() - t0 > nk_peer_timeout(err)) {  // never hang the GPU: count the time-out and go on
      atomicAdd((unsigned long long *)err, 1ull);
      return false;
    }
    __builtin_amdgcn_s_sleep(2);
  }
  return true;
}

// One launch = one all-reduce of `count` ≤ NK_PEER_AR_MAX doubles: store my values into my slot of EVERY rank's arena, release my
// flag there, wait for every rank's flag in MY arena, combine the slots in rank order (elements [max_lo, max_hi) with a
// NaN-propagating max, the others with +). Slots and flags are double-buffered by the parity of the sequence number: a
// rank can only be one collective ahead of a peer, because it needs that peer's contribution to finish its own.
__global__ __launch_bounds__(NK_BLOCK) void k_peer_allreduce(char *const *map, int P, int me, double *buf, int count,
                                                             int max_lo, int max_hi, uint64_t seq) {
  const int t = threadIdx.x, par = (int)(seq & 1);
  for (int e = t; e < count; e += NK_BLOCK) {
    const double v = buf[e];
    for (int p = 0; p < P; ++p) reinterpret_cast<nk_peer_hdr *>(map[p])->ar_data[par][me][e] = v;
  }
  __threadfence_system();
  __syncthreads();
  if (t < P) {
    nk_peer_hdr *h = reinterpret_cast<nk_peer_hdr *>(map[t]);
    __hip_atomic_store(&h->ar_flag[par][me], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  nk_peer_hdr *mine = reinterpret_cast<nk_peer_hdr *>(map[me]);
  if (t < P) peer_wait_ge(&mine->ar_flag[par][t], seq, &mine->err);
  __syncthreads();
  for (int e = t; e < count; e += NK_BLOCK) {
    const bool mx = (e >= max_lo && e < max_hi);
    double acc = mine->ar_data[par][0][e];
    for (int p = 1; p < P; ++p) {
      const double w = mine->ar_data[par][p][e];
      if (mx) acc = (acc != acc || w != w) ? __builtin_nan("") : (w > acc ? w : acc);
      else acc += w;
    }
    buf[e] = acc;
  }
}

// One launch = one halo exchange: workgroup s serves neighbour s — it stores my entries for that neighbour into the
// neighbour's receive area (x gathered through the plan's index list, no staging buffer), releases my flag over there,
// and then waits for that neighbour's flag over here. Every workgroup pushes before it waits, so no cycle can form.
__global__ __launch_bounds__(NK_BLOCK) void k_peer_halo_xchg(const nk_peer_seg *segs, const int32_t *__restrict__ send_idx,
                                                             const double *__restrict__ x, uint64_t seq, uint64_t *err) {
  const nk_peer_seg *__restrict__ sg = segs + blockIdx.x;   // (through the pointer: a by-value copy indexed by the parity is private memory)
  double *__restrict__ dst = (seq & 1) ? sg->dst[1] : sg->dst[0];
  const int64_t send_cnt = sg->send_cnt;
  const int32_t *idx = send_idx + sg->send_off;
  for (int64_t i = threadIdx.x; i < send_cnt; i += NK_BLOCK) dst[i] = x[idx[i]];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_store(sg->flag_remote, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    peer_wait_ge(sg->flag_local, seq, err);
  }
}

static int comm_allreduce_base(nk_ctx *ctx, double *dbuf, int count, int op);
uint64_t *nk_peer_err_ptr(nk_ctx *ctx) { return reinterpret_cast<uint64_t *>(ctx->peer.arena + offsetof(nk_peer_hdr, err)); }
static bool peer_ar_unfused() {
  static const bool unfused = getenv("NK_PEER_UNFUSED") != nullptr;  // A/B switch: reduce and all-reduce as two launches
  return unfused;
}
// would nk_peer_ar_next(ctx, count) take the fast path? (no side effect)
bool nk_peer_ar_available(nk_ctx *ctx, int count) {
  return ctx->peer.on && !peer_ar_unfused() && count <= NK_PEER_AR_MAX && ctx->nranks > 1;
}
nk_peer_ar_view nk_peer_ar_next(nk_ctx *ctx, int count) {
  nk_peer_ar_view v{nullptr, 0, 0, 0, nullptr};
  if (!nk_peer_ar_available(ctx, count)) return v;
  v.map = ctx->peer.d_map;
  v.P = ctx->peer.P;
  v.me = ctx->peer.me;
  v.seq = ++ctx->peer.ar_seq;
  v.ticket = ctx->peer.d_ticket;
  ctx->stats.allreduces++;
  return v;
}

static void nk_peer_destroy(nk_ctx *ctx) {
  nk_peer &pr = ctx->peer;
  for (int p = 0; p < pr.P; ++p)
    if (p != pr.me && pr.map[p]) hipIpcCloseMemHandle(pr.map[p]);
  hipFree(pr.d_map);
  hipFree(pr.d_ticket);
  if (pr.arena) hipFree(pr.arena);
  pr = nk_peer{};
}

extern "C" int nk_ctx_comm_peer_handle(nk_ctx *ctx, int64_t arena_bytes, char handle_out[NK_IPC_HANDLE_BYTES]) {
  NK_REQUIRE(ctx && handle_out, "NULL argument");
  static_assert(sizeof(hipIpcMemHandle_t) == NK_IPC_HANDLE_BYTES, "IPC handle size");
  NK_HIP(hipSetDevice(ctx->device));
  nk_peer &pr = ctx->peer;
  NK_REQUIRE(!pr.arena, "the peer arena exists already");
  pr.arena_bytes = arena_bytes > 0 ? (size_t)arena_bytes : ((size_t)64 << 20);
  NK_REQUIRE(pr.arena_bytes >= 2 * NK_PEER_HDR_BYTES, "peer arena too small");
  void *a = nullptr;
  // uncached (fine-grained) device memory: stores from a peer GPU must be visible to a kernel that is already polling
  if (hipExtMallocWithFlags(&a, pr.arena_bytes, hipDeviceMallocUncached) != hipSuccess) {
    (void)hipGetLastError();
    if (hipExtMallocWithFlags(&a, pr.arena_bytes, hipDeviceMallocFinegrained) != hipSuccess) {
      (void)hipGetLastError();
      NK_FAIL(NK_E_NOMEM, "cannot allocate %zu bytes of uncached / fine-grained device memory for the peer arena", pr.arena_bytes);
    }
  }
  pr.arena = (char *)a;
  NK_HIP(nk_memset(ctx, pr.arena, 0, NK_PEER_HDR_BYTES));
  {  // the bound of every device-side wait (a rank stalled by a host callback, a JIT, a first kernel load): NK_PEER_TIMEOUT_MS
    const char *e = getenv("NK_PEER_TIMEOUT_MS");
    const double ms = e ? atof(e) : 5000.0;
    const uint64_t ticks = (uint64_t)((ms > 1.0 ? ms : 1.0) * 1e5);
    NK_HIP(nk_memcpy(ctx, pr.arena + offsetof(nk_peer_hdr, timeout_ticks), &ticks, sizeof(ticks), hipMemcpyHostToDevice));
  }
  NK_HIP(hipDeviceSynchronize());
  pr.bump = NK_PEER_HDR_BYTES;
  hipIpcMemHandle_t h;
  hipError_t e = hipIpcGetMemHandle(&h, pr.arena);
  if (e != hipSuccess) {
    hipFree(pr.arena);
    pr.arena = nullptr;
    NK_FAIL(NK_E_HIP, "hipIpcGetMemHandle: %s (is HSA_ENABLE_IPC_MODE_LEGACY=0 exported?)", hipGetErrorString(e));
  }
  memcpy(handle_out, &h, NK_IPC_HANDLE_BYTES);
  return NK_OK;
}

extern "C" int nk_ctx_comm_enable_peer(nk_ctx *ctx, const char *handles) {
  NK_REQUIRE(ctx && handles, "NULL argument");
  nk_peer &pr = ctx->peer;
  NK_REQUIRE(pr.arena, "call nk_ctx_comm_peer_handle first");
  NK_REQUIRE(ctx->comm_kind != NK_COMM_NONE && ctx->nranks > 1, "the peer path sits on top of an initialised communicator");
  NK_REQUIRE(ctx->nranks <= NK_PEER_MAX_RANKS, "at most %d ranks", NK_PEER_MAX_RANKS);
  NK_HIP(hipSetDevice(ctx->device));
  pr.P = ctx->nranks;
  pr.me = ctx->rank;
  for (int p = 0; p < pr.P; ++p) {
    if (p == pr.me) { pr.map[p] = pr.arena; continue; }
    hipIpcMemHandle_t h;
    memcpy(&h, handles + (size_t)p * NK_IPC_HANDLE_BYTES, NK_IPC_HANDLE_BYTES);
    void *q = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&q, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      pr.P = p;  // close what was opened
      NK_FAIL(NK_E_HIP, "hipIpcOpenMemHandle(rank %d): %s", p, hipGetErrorString(e));
    }
    pr.map[p] = (char *)q;
  }
  NK_TRY(nk_dev_alloc(&pr.d_map, (size_t)NK_PEER_MAX_RANKS));
  NK_HIP(nk_memcpy(ctx, pr.d_map, pr.map, sizeof(char *) * NK_PEER_MAX_RANKS, hipMemcpyHostToDevice));
  NK_TRY(nk_dev_alloc(&pr.d_ticket, (size_t)4));
  NK_HIP(nk_memset(ctx, pr.d_ticket, 0, 4 * sizeof(unsigned int)));
  pr.on = true;
  return NK_OK;
}

extern "C" int nk_ctx_comm_peer_disable(nk_ctx *ctx) {
  NK_REQUIRE(ctx, "NULL argument");
  ctx->peer.on = false;  // plans set up so far keep their transport; new ones use the base transport
  return NK_OK;
}
// A few all-reduces with known answers through the peer path, agreed on by all ranks through the base transport; on any
// mismatch or time-out the fast path is switched off everywhere (the base transport then serves every collective).
extern "C" int nk_ctx_comm_peer_selftest(nk_ctx *ctx, int *ok) {
  NK_REQUIRE(ctx && ok, "NULL argument");
  *ok = 0;
  if (!ctx->peer.on) return NK_OK;
  NK_HIP(hipSetDevice(ctx->device));
  const int P = ctx->nranks, me = ctx->rank, cnt = 64;
  std::vector<double> h(cnt);
  double *d = nullptr;
  NK_TRY(nk_dev_alloc(&d, (size_t)cnt + 2));
  int bad = 0;
  for (int round = 0; round < 4 && !bad; ++round) {
    for (int i = 0; i < cnt; ++i) h[i] = (double)(me + 1) * (round + 1) + 0.25 * i;
    NK_HIP(nk_memcpy(ctx, d, h.data(), cnt * sizeof(double), hipMemcpyHostToDevice));
    int st = nk_comm_allreduce_mixed(ctx, d, cnt, 8, 16);  // elements 8..15 with max, the others with +
    if (st == NK_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) st = NK_E_HIP;
    if (st != NK_OK) { bad = 1; break; }
    NK_HIP(nk_memcpy(ctx, h.data(), d, cnt * sizeof(double), hipMemcpyDeviceToHost));
    for (int i = 0; i < cnt; ++i) {
      const double sum = (double)(round + 1) * P * (P + 1) / 2.0 + 0.25 * i * P, mx = (double)P * (round + 1) + 0.25 * i;
      if (h[i] != ((i >= 8 && i < 16) ? mx : sum)) bad = 1;
    }
  }
  int64_t errs = 0;
  int en = 0;
  NK_TRY(nk_ctx_comm_peer_status(ctx, &en, &errs));
  if (errs != 0) bad = 1;
  double flag = bad ? 1.0 : 0.0;
  NK_HIP(nk_memcpy(ctx, d, &flag, sizeof(double), hipMemcpyHostToDevice));
  int st = comm_allreduce_base(ctx, d, 1, 1);
  if (st == NK_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) st = NK_E_HIP;
  if (st == NK_OK) NK_HIP(nk_memcpy(ctx, &flag, d, sizeof(double), hipMemcpyDeviceToHost));
  hipFree(d);
  NK_TRY(st);
  if (flag != 0.0) ctx->peer.on = false;
  *ok = ctx->peer.on ? 1 : 0;
  return NK_OK;
}

extern "C" int nk_ctx_comm_peer_status(nk_ctx *ctx, int *enabled, int64_t *errors) {
  NK_REQUIRE(ctx, "NULL argument");
  if (enabled) *enabled = ctx->peer.on ? 1 : 0;
  if (errors) {
    *errors = 0;
    if (ctx->peer.arena) {
      uint64_t e = 0;
      NK_HIP(hipStreamSynchronize(ctx->stream));
      NK_HIP(nk_memcpy(ctx, &e, ctx->peer.arena + offsetof(nk_peer_hdr, err), sizeof(e), hipMemcpyDeviceToHost));
      *errors = (int64_t)e;
    }
  }
  return NK_OK;
}

// ----------------------------------------------------------------------------- RCCL through dlopen
// Only the handful of entry points the Krylov loop needs; resolved at run time so that single-GPU use
// has no RCCL dependency and so that the process shares whichever librccl the host already loaded.
typedef struct { char internal[128]; } rccl_uid;
typedef int (*pfn_GetUniqueId)(rccl_uid *);
typedef int (*pfn_CommInitRank)(void **, int, rccl_uid, int);
typedef int (*pfn_CommDestroy)(void *);
typedef int (*pfn_AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef int (*pfn_Send)(const void *, size_t, int, int, void *, hipStream_t);
typedef int (*pfn_Recv)(void *, size_t, int, int, void *, hipStream_t);
typedef int (*pfn_Group)(void);
typedef const char *(*pfn_GetErrorString)(int);
static struct {
  void *h = nullptr;
  pfn_GetUniqueId GetUniqueId;
  pfn_CommInitRank CommInitRank;
  pfn_CommDestroy CommDestroy;
  pfn_AllReduce AllReduce;
  pfn_Send Send;
  pfn_Recv Recv;
  pfn_Group GroupStart, GroupEnd;
  pfn_GetErrorString GetErrorString;
} R;
enum { RCCL_INT8 = 0, RCCL_FLOAT64 = 8, RCCL_SUM = 0, RCCL_MAX = 2 };  // ncclDataType_t / ncclRedOp_t

static int rccl_load() {
  if (R.h) return NK_OK;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char *nm : names) {
    R.h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    if (R.h) break;
  }
  if (!R.h) NK_FAIL(NK_E_RCCL, "cannot dlopen librccl: %s", dlerror());
#define SYM(field, name)                                          \
  R.field = (decltype(R.field))dlsym(R.h, name);                  \
  if (!R.field) NK_FAIL(NK_E_RCCL, "librccl lacks symbol %s", name)
  SYM(GetUniqueId, "ncclGetUniqueId");
  SYM(CommInitRank, "ncclCommInitRank");
  SYM(CommDestroy, "ncclCommDestroy");
  SYM(AllReduce, "ncclAllReduce");
  SYM(Send, "ncclSend");
  SYM(Recv, "ncclRecv");
  SYM(GroupStart, "ncclGroupStart");
  SYM(GroupEnd, "ncclGroupEnd");
  SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  return NK_OK;
}
#define NK_RCCL(call)                                                                        \
  do {                                                                                       \
    int r_ = (call);                                                                         \
    if (r_ != 0) NK_FAIL(NK_E_RCCL, "%s:%d: %s -> %s", __FILE__, __LINE__, #call, R.GetErrorString(r_)); \
  } while (0)

extern "C" int nk_comm_unique_id(char id_out[128]) {
  NK_REQUIRE(id_out, "id_out is NULL");
  NK_TRY(rccl_load());
  rccl_uid id;
  NK_RCCL(R.GetUniqueId(&id));
  memcpy(id_out, id.internal, 128);
  return NK_OK;
}
// Do several ranks run on one device? Every rank contributes a hash of its device's PCI bus id (and host name) in its own slot
// of a vector that is summed over the ranks; equal entries = a shared device. Collective; the verdict is the same on every rank.
// Why it matters: two processes whose persistent kernels wait for each other (the resident matrix-powers kernel over the peer
// arenas) each need the whole device — on one device they only advance when the scheduler happens to run both, or not at all.
// That form is not used there. (Round 5 also kept sweep B's matrix-core form away from such ranks, blaming wrong results of
// two-process runs on preemption; round 6 found the cause in the product — a write-after-read race between the wavefronts of
// the Gram block's factorisation, ss_factor in nk_sstep.hip — and that restriction is gone: profiles/r06_a_shared_device_root_cause.md.)
static int comm_detect_shared_device(nk_ctx *ctx) {
  ctx->device_shared = false;
  if (ctx->nranks <= 1) return NK_OK;
  char bus[64] = {0}, host[256] = {0};
  const bool have_id = hipDeviceGetPCIBusId(bus, (int)sizeof(bus), ctx->device) == hipSuccess;
  if (!have_id) (void)hipGetLastError();
  gethostname(host, sizeof(host) - 1);
  uint64_t h = 1469598103934665603ull;
  for (const char *p = host; *p; ++p) h = (h ^ (unsigned char)*p) * 1099511628211ull;
  for (const char *p = bus; *p; ++p) h = (h ^ (unsigned char)*p) * 1099511628211ull;
  // 52 bits: exact in a double, and a sum with zeros keeps it exact. (A rank that cannot name its device still takes part in the
  //  collective, with a value no other rank can have.)
  const double mine = have_id ? (double)(h >> 12) : 0.5 + (double)ctx->rank;
  const int P = ctx->nranks;
  std::vector<double> tab((size_t)P, 0.0);
  tab[ctx->rank] = mine;
  double *d = nullptr;
  NK_TRY(nk_dev_alloc(&d, (size_t)P));
  NK_HIP(nk_memcpy(ctx, d, tab.data(), (size_t)P * sizeof(double), hipMemcpyHostToDevice));
  int st = comm_allreduce_base(ctx, d, P, 0);
  if (st == NK_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) st = NK_E_HIP;
  if (st == NK_OK && nk_memcpy(ctx, tab.data(), d, (size_t)P * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) st = NK_E_HIP;
  hipFree(d);
  NK_TRY(st);
  for (int a = 0; a < P; ++a)
    for (int b = a + 1; b < P; ++b)
      if (tab[a] == tab[b]) ctx->device_shared = true;
  // the override (both ways) is read BEHIND the collective: a rank that has it set still takes part, the others do not hang
  // (it must be set alike on every rank to mean anything — the forms it gates are collective decisions)
  if (const char *e = getenv("NK_DEVICE_SHARED")) ctx->device_shared = atoi(e) != 0;
  return NK_OK;
}
extern "C" int nk_ctx_comm_init_rccl(nk_ctx *ctx, int nranks, int rank, const char id[128]) {
  NK_REQUIRE(ctx && id, "NULL argument");
  NK_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank %d / %d", rank, nranks);
  NK_REQUIRE(ctx->comm_kind == NK_COMM_NONE, "communicator already initialised");
  NK_TRY(rccl_load());
  NK_HIP(hipSetDevice(ctx->device));
  rccl_uid uid;
  memcpy(uid.internal, id, 128);
  NK_RCCL(R.CommInitRank(&ctx->rccl_comm, nranks, uid, rank));
  ctx->comm_kind = NK_COMM_RCCL;
  ctx->nranks = nranks;
  ctx->rank = rank;
  return comm_detect_shared_device(ctx);
}
extern "C" int nk_ctx_comm_init_callbacks(nk_ctx *ctx, int nranks, int rank, const nk_comm_callbacks *cb) {
  NK_REQUIRE(ctx && cb && cb->allreduce && cb->alltoallv, "NULL argument / callback");
  NK_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank %d / %d", rank, nranks);
  NK_REQUIRE(ctx->comm_kind == NK_COMM_NONE, "communicator already initialised");
  ctx->cb = *cb;
  ctx->comm_kind = NK_COMM_CALLBACKS;
  ctx->nranks = nranks;
  ctx->rank = rank;
  return comm_detect_shared_device(ctx);
}
extern "C" int nk_ctx_comm_device_shared(nk_ctx *ctx, int *shared) {
  NK_REQUIRE(ctx && shared, "NULL argument");
  *shared = ctx->device_shared ? 1 : 0;
  return NK_OK;
}
extern "C" int nk_ctx_comm_info(nk_ctx *ctx, int *kind, int *nranks, int *rank) {
  NK_REQUIRE(ctx, "ctx is NULL");
  if (kind) *kind = ctx->comm_kind;
  if (nranks) *nranks = ctx->nranks;
  if (rank) *rank = ctx->rank;
  return NK_OK;
}
void nk_comm_destroy(nk_ctx *ctx) {
  if (ctx->comm_kind == NK_COMM_RCCL && ctx->rccl_comm && R.CommDestroy) R.CommDestroy(ctx->rccl_comm);
  ctx->rccl_comm = nullptr;
  ctx->comm_kind = NK_COMM_NONE;
  ctx->nranks = 1;
  ctx->rank = 0;
}

bool nk_ctx_is_single(const nk_ctx *ctx) {
  static const bool force = getenv("NK_FORCE_COLLECTIVES") != nullptr;
  return ctx->nranks <= 1 && !(force && ctx->comm_kind != NK_COMM_NONE);
}
static int comm_allreduce_base(nk_ctx *ctx, double *dbuf, int count, int op) {
  if (ctx->comm_kind == NK_COMM_RCCL) {
    NK_RCCL(R.AllReduce(dbuf, dbuf, (size_t)count, RCCL_FLOAT64, op == 1 ? RCCL_MAX : RCCL_SUM,
                        ctx->rccl_comm, ctx->stream));
    return NK_OK;
  }
  if (ctx->comm_kind == NK_COMM_CALLBACKS) {
    if (ctx->cb.allreduce(ctx->cb.user, dbuf, count, op, (void *)ctx->stream) != 0)
      NK_FAIL(NK_E_CALLBACK, "allreduce callback failed");
    return NK_OK;
  }
  NK_FAIL(NK_E_INVALID, "nranks>1 without a communicator");
}
// elements [max_lo, max_hi) are combined with max, all others with +: one message on the peer path, up to three
// collectives on the base transports
int nk_comm_allreduce_mixed(nk_ctx *ctx, double *dbuf, int count, int max_lo, int max_hi) {
  // NK_FORCE_COLLECTIVES=1: issue the collective even on a 1-rank communicator (exercises the RCCL entry points
  // on a single GPU; used by tests only)
  static const bool force = getenv("NK_FORCE_COLLECTIVES") != nullptr;
  if (count <= 0) return NK_OK;
  if (ctx->nranks <= 1 && !(force && ctx->comm_kind != NK_COMM_NONE)) return NK_OK;
  if (max_lo < 0) max_lo = 0;
  if (max_hi > count) max_hi = count;
  if (max_hi < max_lo) max_hi = max_lo;
  ctx->stats.allreduces++;
  if (ctx->peer.on && count <= NK_PEER_AR_MAX) {
    const uint64_t seq = ++ctx->peer.ar_seq;
    NK_LAUNCH(ctx, k_peer_allreduce, dim3(1), dim3(NK_BLOCK), (char *const *)ctx->peer.d_map, ctx->peer.P, ctx->peer.me, dbuf,
              count, max_lo, max_hi, seq);
    NK_HIP(hipGetLastError());
    return NK_OK;
  }
  if (max_lo > 0) NK_TRY(comm_allreduce_base(ctx, dbuf, max_lo, 0));
  if (max_hi > max_lo) NK_TRY(comm_allreduce_base(ctx, dbuf + max_lo, max_hi - max_lo, 1));
  if (count > max_hi) NK_TRY(comm_allreduce_base(ctx, dbuf + max_hi, count - max_hi, 0));
  if ((max_lo > 0) + (max_hi > max_lo) + (count > max_hi) > 1)
    ctx->stats.allreduces += (max_lo > 0) + (max_hi > max_lo) + (count > max_hi) - 1;
  return NK_OK;
}
int nk_comm_allreduce(nk_ctx *ctx, double *dbuf, int count, int op) {
  return op == 1 ? nk_comm_allreduce_mixed(ctx, dbuf, count, 0, count) : nk_comm_allreduce_mixed(ctx, dbuf, count, 0, 0);
}

int nk_comm_alltoallv(nk_ctx *ctx, const void *send, const int64_t *soff, const int64_t *sbytes,
                      void *recv, const int64_t *roff, const int64_t *rbytes, hipStream_t stream) {
  if (ctx->nranks <= 1) return NK_OK;
  if (!stream) stream = ctx->stream;
  if (ctx->comm_kind == NK_COMM_RCCL) {
    NK_RCCL(R.GroupStart());
    for (int p = 0; p < ctx->nranks; ++p) {
      if (p == ctx->rank) continue;
      if (sbytes[p] > 0)
        NK_RCCL(R.Send((const char *)send + soff[p], (size_t)sbytes[p], RCCL_INT8, p, ctx->rccl_comm, stream));
      if (rbytes[p] > 0)
        NK_RCCL(R.Recv((char *)recv + roff[p], (size_t)rbytes[p], RCCL_INT8, p, ctx->rccl_comm, stream));
    }
    NK_RCCL(R.GroupEnd());
    return NK_OK;
  }
  if (ctx->comm_kind == NK_COMM_CALLBACKS) {
    if (ctx->cb.alltoallv(ctx->cb.user, send, soff, sbytes, recv, roff, rbytes, (void *)stream) != 0)
      NK_FAIL(NK_E_CALLBACK, "alltoallv callback failed");
    return NK_OK;
  }
  NK_FAIL(NK_E_INVALID, "nranks>1 without a communicator");
}

// ----------------------------------------------------------------------------- halo exchange
__global__ __launch_bounds__(NK_BLOCK) void k_gather(int64_t n, const int32_t *__restrict__ idx,
                                                     const double *__restrict__ x, double *__restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * NK_BLOCK + threadIdx.x;
  if (i < n) out[i] = x[idx[i]];
}


// Receive areas of a halo plan inside this rank's arena ([flags | parity 0 | parity 1]) and the neighbour table: where my
// entries land over there, which flag I raise there, which flag of mine the neighbour raises. The arena offsets are
// exchanged once through the base transport (an all-reduce of a P × (P + 2) table, every rank filling its row).
static int halo_setup_peer(nk_ctx *ctx, nk_halo *H) {
  nk_peer &pr = ctx->peer;
  const int P = ctx->nranks, me = ctx->rank;
  const size_t stride = (((size_t)H->n_recv * 8) + 255) & ~(size_t)255;
  const size_t need = 256 + 2 * stride;
  const size_t off = (pr.bump + 255) & ~(size_t)255;
  // every rank must take the same decision: agree on "fits everywhere" first
  std::vector<double> tab((size_t)P * (P + 3), 0.0);
  double *row = tab.data() + (size_t)me * (P + 3);
  row[0] = (double)off;
  row[1] = (double)stride;
  row[2] = (off + need <= pr.arena_bytes) ? 0.0 : 1.0;
  for (int p = 0; p < P; ++p) row[3 + p] = (double)H->recv_off[p];
  double *d_tab = nullptr;
  NK_TRY(nk_dev_alloc(&d_tab, tab.size()));
  int st = NK_OK;   // (every exit below frees d_tab; a failed copy still takes part in the collective with what it has)
  if (nk_memcpy(ctx, d_tab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) st = NK_E_HIP;
  const int st_ar = comm_allreduce_base(ctx, d_tab, (int)tab.size(), 0);
  if (st == NK_OK) st = st_ar;
  if (st == NK_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) st = NK_E_HIP;
  if (st == NK_OK && nk_memcpy(ctx, tab.data(), d_tab, tab.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) st = NK_E_HIP;
  hipFree(d_tab);
  if (st == NK_E_HIP) nk_set_error("peer set-up: a copy of the negotiation table failed");
  NK_TRY(st);
  bool fits = true;
  for (int p = 0; p < P; ++p) fits = fits && tab[(size_t)p * (P + 3) + 2] == 0.0;
  if (!fits) {  // the arena is full somewhere: this plan uses the base transport
    NK_TRY(nk_dev_alloc(&H->d_recv, (size_t)H->n_recv));
    return NK_OK;
  }
  pr.bump = off + need;
  NK_HIP(hipMemsetAsync(pr.arena + off, 0, 256, ctx->stream));
  H->recv_buf[0] = reinterpret_cast<double *>(pr.arena + off + 256);
  H->recv_buf[1] = reinterpret_cast<double *>(pr.arena + off + 256 + stride);
  H->d_recv = H->recv_buf[0];
  std::vector<nk_peer_seg> segs;
  for (int p = 0; p < P; ++p) {
    if (H->send_cnt[p] == 0 && H->recv_cnt[p] == 0) continue;
    const double *prow = tab.data() + (size_t)p * (P + 3);
    const size_t poff = (size_t)prow[0], pstride = (size_t)prow[1];
    const int64_t there = (int64_t)prow[3 + me];  // where my entries start in p's receive area
    nk_peer_seg sg;
    sg.send_off = H->send_off[p];
    sg.send_cnt = H->send_cnt[p];
    sg.dst[0] = reinterpret_cast<double *>(pr.map[p] + poff + 256) + there;
    sg.dst[1] = reinterpret_cast<double *>(pr.map[p] + poff + 256 + pstride) + there;
    sg.flag_remote = reinterpret_cast<uint64_t *>(pr.map[p] + poff) + me;
    sg.flag_local = reinterpret_cast<const uint64_t *>(pr.arena + off) + p;
    segs.push_back(sg);
  }
  H->nsegs = (int)segs.size();
  if (H->nsegs) {
    NK_HIP(hipMalloc((void **)&H->d_segs, segs.size() * sizeof(nk_peer_seg)));
    NK_HIP(nk_memcpy(ctx, H->d_segs, segs.data(), segs.size() * sizeof(nk_peer_seg), hipMemcpyHostToDevice));
  }
  NK_HIP(hipStreamSynchronize(ctx->stream));
  // nobody may push into a receive area before its owner has zeroed the flags: one more collective as a barrier
  double *d_one = nullptr;
  NK_TRY(nk_dev_alloc(&d_one, (size_t)1));
  NK_HIP(nk_memset(ctx, d_one, 0, sizeof(double)));
  st = comm_allreduce_base(ctx, d_one, 1, 0);
  if (st == NK_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) st = NK_E_HIP;
  hipFree(d_one);
  NK_TRY(st);
  H->peer = true;
  return NK_OK;
}

// see nk_internal.h. Layout of a rank's block (offset `off` in its arena): flag from above, flag from below (128 bytes apart),
// then 4 × 1024 doubles from above, 4 × 1024 from below.
int nk_peer_powers_setup(nk_ctx *ctx, bool eligible, nk_peer_powers *out, bool *ok) {
  *ok = false;
  nk_peer &pr = ctx->peer;
  const int P = ctx->nranks, me = ctx->rank;
  constexpr size_t AREA = (size_t)4 * 1024 * sizeof(double), NEED = 256 + 2 * AREA;
  const size_t off = pr.on ? ((pr.bump + 255) & ~(size_t)255) : 0;
  std::vector<double> tab((size_t)P * 2, 0.0);
  tab[(size_t)me * 2 + 0] = (double)off;
  tab[(size_t)me * 2 + 1] = (eligible && pr.on && off + NEED <= pr.arena_bytes) ? 0.0 : 1.0;
  double *d_tab = nullptr;
  NK_TRY(nk_dev_alloc(&d_tab, tab.size()));
  int st = NK_OK;   // (every exit below frees d_tab; a failed copy still takes part in the collective with what it has)
  if (nk_memcpy(ctx, d_tab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) st = NK_E_HIP;
  const int st_ar = comm_allreduce_base(ctx, d_tab, (int)tab.size(), 0);
  if (st == NK_OK) st = st_ar;
  if (st == NK_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) st = NK_E_HIP;
  if (st == NK_OK && nk_memcpy(ctx, tab.data(), d_tab, tab.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) st = NK_E_HIP;
  hipFree(d_tab);
  if (st == NK_E_HIP) nk_set_error("peer set-up: a copy of the negotiation table failed");
  NK_TRY(st);
  for (int p = 0; p < P; ++p)
    if (tab[(size_t)p * 2 + 1] != 0.0) return NK_OK;   // somebody cannot: nobody does
  pr.bump = off + NEED;
  NK_HIP(hipMemsetAsync(pr.arena + off, 0, 256, ctx->stream));
  out->myflag_up = reinterpret_cast<uint64_t *>(pr.arena + off);
  out->myflag_dn = reinterpret_cast<uint64_t *>(pr.arena + off + 128);
  out->recv_up = reinterpret_cast<double *>(pr.arena + off + 256);
  out->recv_dn = reinterpret_cast<double *>(pr.arena + off + 256 + AREA);
  if (me > 0) {   // my first slice is the rank above's "from below"
    const size_t poff = (size_t)tab[(size_t)(me - 1) * 2];
    out->flag_up = reinterpret_cast<uint64_t *>(pr.map[me - 1] + poff + 128);
    out->push_up = reinterpret_cast<double *>(pr.map[me - 1] + poff + 256 + AREA);
  }
  if (me + 1 < P) {
    const size_t poff = (size_t)tab[(size_t)(me + 1) * 2];
    out->flag_dn = reinterpret_cast<uint64_t *>(pr.map[me + 1] + poff);
    out->push_dn = reinterpret_cast<double *>(pr.map[me + 1] + poff + 256);
  }
  NK_HIP(hipStreamSynchronize(ctx->stream));
  // nobody may push into an area before its owner has zeroed the flags: one more collective as a barrier
  double *d_one = nullptr;
  NK_TRY(nk_dev_alloc(&d_one, (size_t)1));
  NK_HIP(nk_memset(ctx, d_one, 0, sizeof(double)));
  st = comm_allreduce_base(ctx, d_one, 1, 0);
  if (st == NK_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) st = NK_E_HIP;
  hipFree(d_one);
  NK_TRY(st);
  *ok = true;
  return NK_OK;
}

int nk_halo_setup(nk_ctx *ctx, nk_halo *H, const std::vector<std::vector<int32_t>> &send_idx_per_peer,
                  const std::vector<int64_t> &recv_cnt_per_peer) {
  const int P = ctx->nranks;
  H->send_off.assign(P, 0);
  H->send_cnt.assign(P, 0);
  H->recv_off.assign(P, 0);
  H->recv_cnt.assign(P, 0);
  std::vector<int32_t> flat;
  int64_t so = 0, ro = 0;
  for (int p = 0; p < P; ++p) {
    H->send_off[p] = so;
    H->send_cnt[p] = (int64_t)send_idx_per_peer[p].size();
    flat.insert(flat.end(), send_idx_per_peer[p].begin(), send_idx_per_peer[p].end());
    so += H->send_cnt[p];
    H->recv_off[p] = ro;
    H->recv_cnt[p] = recv_cnt_per_peer[p];
    ro += H->recv_cnt[p];
  }
  H->n_send = so;
  H->n_recv = ro;
  H->contig_base.assign(P, 0);
  H->contig = getenv("NK_HALO_GATHER") == nullptr;  // NK_HALO_GATHER=1: always stage through the gather kernel (A/B)
  for (int p = 0; p < P && H->contig; ++p) {
    const std::vector<int32_t> &ix = send_idx_per_peer[p];
    if (ix.empty()) continue;
    H->contig_base[p] = ix[0];
    for (size_t q = 1; q < ix.size(); ++q)
      if (ix[q] != ix[0] + (int32_t)q) { H->contig = false; break; }
  }
  NK_TRY(nk_dev_alloc(&H->d_send_idx, (size_t)so));
  NK_TRY(nk_dev_alloc(&H->d_send, (size_t)so));
  if (so) NK_HIP(nk_memcpy(ctx, H->d_send_idx, flat.data(), so * sizeof(int32_t), hipMemcpyHostToDevice));
  if (ctx->peer.on && P > 1) return halo_setup_peer(ctx, H);  // (collective, like every halo set-up on several ranks)
  NK_TRY(nk_dev_alloc(&H->d_recv, (size_t)ro));
  return NK_OK;
}

int nk_halo_build_from_needs(nk_ctx *ctx, int64_t my_begin, int64_t my_count, const std::vector<int64_t> &needs, nk_halo *H) {
  const int P = ctx->nranks;
  NK_REQUIRE(P > 1, "a halo plan needs several ranks");
  // 1. everyone learns all row ranges: all-reduce a zero vector with our begin in slot `rank`
  std::vector<double> hb(P + 1, 0.0);
  hb[ctx->rank] = (double)my_begin;
  if (ctx->rank == P - 1) hb[P] = (double)(my_begin + my_count);
  double *d_tmp = nullptr;
  NK_TRY(nk_dev_alloc(&d_tmp, (size_t)P * P + P + 1));
  auto tmp_guard = nk_make_guard(d_tmp, [](double *q) { hipFree(q); });
  NK_HIP(nk_memcpy(ctx, d_tmp, hb.data(), (P + 1) * sizeof(double), hipMemcpyHostToDevice));
  NK_TRY(comm_allreduce_base(ctx, d_tmp, P + 1, 0));
  NK_HIP(hipStreamSynchronize(ctx->stream));
  NK_HIP(nk_memcpy(ctx, hb.data(), d_tmp, (P + 1) * sizeof(double), hipMemcpyDeviceToHost));
  std::vector<int64_t> begin(P + 1);
  for (int p = 0; p <= P; ++p) begin[p] = (int64_t)hb[p];
  // 2. needs per owner
  std::vector<std::vector<int64_t>> need(P);
  for (int64_t g : needs) {
    int p = (int)(std::upper_bound(begin.begin(), begin.end(), g) - begin.begin()) - 1;
    while (p > 0 && begin[p] == begin[p + 1]) --p;  // (empty ranges share a begin)
    NK_REQUIRE(p >= 0 && p < P && p != ctx->rank, "needed entry %lld has no owner", (long long)g);
    need[p].push_back(g);
  }
  // 3. counts: P×P matrix, row = requester, col = owner
  std::vector<double> cnt((size_t)P * P, 0.0);
  for (int p = 0; p < P; ++p) cnt[(size_t)ctx->rank * P + p] = (double)need[p].size();
  NK_HIP(nk_memcpy(ctx, d_tmp, cnt.data(), (size_t)P * P * sizeof(double), hipMemcpyHostToDevice));
  NK_TRY(comm_allreduce_base(ctx, d_tmp, P * P, 0));
  NK_HIP(hipStreamSynchronize(ctx->stream));
  NK_HIP(nk_memcpy(ctx, cnt.data(), d_tmp, (size_t)P * P * sizeof(double), hipMemcpyDeviceToHost));
  // 4. index lists (int64 global ids): what I need from p ↔ what p needs from me
  std::vector<int64_t> soff(P, 0), sbytes(P, 0), roff(P, 0), rbytes(P, 0), sendflat;
  int64_t rtotal = 0;
  for (int p = 0; p < P; ++p) {
    soff[p] = (int64_t)sendflat.size() * 8;
    sbytes[p] = (int64_t)need[p].size() * 8;
    sendflat.insert(sendflat.end(), need[p].begin(), need[p].end());
    roff[p] = rtotal * 8;
    const int64_t c = (int64_t)cnt[(size_t)p * P + ctx->rank];
    rbytes[p] = c * 8;
    rtotal += c;
  }
  int64_t *d_s = nullptr, *d_r = nullptr;
  NK_TRY(nk_dev_alloc(&d_s, sendflat.size() + 1));
  auto s_guard = nk_make_guard(d_s, [](int64_t *q) { hipFree(q); });
  NK_TRY(nk_dev_alloc(&d_r, (size_t)rtotal + 1));
  auto r_guard = nk_make_guard(d_r, [](int64_t *q) { hipFree(q); });
  if (!sendflat.empty()) NK_HIP(nk_memcpy(ctx, d_s, sendflat.data(), sendflat.size() * 8, hipMemcpyHostToDevice));
  NK_TRY(nk_comm_alltoallv(ctx, d_s, soff.data(), sbytes.data(), d_r, roff.data(), rbytes.data()));
  NK_HIP(hipStreamSynchronize(ctx->stream));
  std::vector<int64_t> wanted((size_t)rtotal);
  if (rtotal) NK_HIP(nk_memcpy(ctx, wanted.data(), d_r, (size_t)rtotal * 8, hipMemcpyDeviceToHost));
  std::vector<std::vector<int32_t>> send_idx(P);
  std::vector<int64_t> recv_cnt(P, 0);
  for (int p = 0; p < P; ++p) {
    const int64_t c = rbytes[p] / 8, o = roff[p] / 8;
    for (int64_t t = 0; t < c; ++t) {
      const int64_t g = wanted[o + t];
      NK_REQUIRE(g >= my_begin && g < my_begin + my_count, "peer %d asked for an entry this rank does not own", p);
      send_idx[p].push_back((int32_t)(g - my_begin));
    }
    recv_cnt[p] = (int64_t)need[p].size();
  }
  // needs are sorted by global id and owners are ordered by rank → the receive layout is the order of `needs`
  return nk_halo_setup(ctx, H, send_idx, recv_cnt);
}

// gather + (optionally on `xstream`) the exchange itself
static int halo_exchange_on(nk_ctx *ctx, nk_halo *H, const double *d_x_local, hipStream_t xstream) {
  const int P = ctx->nranks;
  if (H->peer) {  // one launch: push to every neighbour, wait for every neighbour (k_peer_halo_xchg)
    const uint64_t seq = ++H->seq;
    ctx->stats.halo_exchanges++;
    if (H->nsegs) {
      NK_LAUNCH(ctx, k_peer_halo_xchg, dim3(H->nsegs), dim3(NK_BLOCK), (const nk_peer_seg *)H->d_segs,
                (const int32_t *)H->d_send_idx, d_x_local, seq,
                reinterpret_cast<uint64_t *>(ctx->peer.arena + offsetof(nk_peer_hdr, err)));
      NK_HIP(hipGetLastError());
    }
    H->d_recv = H->recv_buf[seq & 1];
    return NK_OK;
  }
  const bool direct = H->contig && H->send_cnt[ctx->rank] == 0;  // send from the vector itself
  if (H->n_send && !direct) {
    int grid = (int)((H->n_send + NK_BLOCK - 1) / NK_BLOCK);
    NK_LAUNCH(ctx, k_gather, dim3(grid), dim3(NK_BLOCK), H->n_send, H->d_send_idx,
                       d_x_local, H->d_send);
  }
  // entries a rank "sends to itself" (periodic wrap on one rank) are a device copy
  if (H->send_cnt[ctx->rank] > 0) {
    NK_HIP(hipMemcpyAsync(H->d_recv + H->recv_off[ctx->rank], H->d_send + H->send_off[ctx->rank],
                          H->send_cnt[ctx->rank] * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  }
  if (P > 1) {
    std::vector<int64_t> so(P), sb(P), ro(P), rb(P);
    for (int p = 0; p < P; ++p) {
      so[p] = (direct ? H->contig_base[p] : H->send_off[p]) * 8;
      sb[p] = (p == ctx->rank) ? 0 : H->send_cnt[p] * 8;
      ro[p] = H->recv_off[p] * 8;
      rb[p] = (p == ctx->rank) ? 0 : H->recv_cnt[p] * 8;
    }
    ctx->stats.halo_exchanges++;
    if (xstream) {  // the exchange may start once the gather (and everything before it) is done
      NK_HIP(hipEventRecord(ctx->ev_halo_ready, ctx->stream));
      NK_HIP(hipStreamWaitEvent(xstream, ctx->ev_halo_ready, 0));
    }
    NK_TRY(nk_comm_alltoallv(ctx, direct ? (const void *)d_x_local : (const void *)H->d_send, so.data(), sb.data(),
                             H->d_recv, ro.data(), rb.data(), xstream));
    if (xstream) NK_HIP(hipEventRecord(ctx->ev_halo_done, xstream));
  }
  return NK_OK;
}

int nk_halo_exchange(nk_ctx *ctx, nk_halo *H, const double *d_x_local) {
  if (!H->active()) return NK_OK;
  return halo_exchange_on(ctx, H, d_x_local, nullptr);
}
int nk_halo_exchange_begin(nk_ctx *ctx, nk_halo *H, const double *d_x_local) {
  if (!H->active()) return NK_OK;
  return halo_exchange_on(ctx, H, d_x_local, H->peer ? nullptr : ctx->comm_stream);
}
int nk_halo_exchange_end(nk_ctx *ctx, nk_halo *H) {
  if (!H->active() || ctx->nranks <= 1 || H->peer) return NK_OK;
  NK_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_halo_done, 0));
  return NK_OK;
}

void nk_halo_free(nk_halo *H) {
  hipFree(H->d_send_idx);
  hipFree(H->d_send);
  if (!H->peer) hipFree(H->d_recv);  // (peer plans: the receive areas belong to the context's arena)
  hipFree(H->d_segs);
  H->d_segs = nullptr;
  H->peer = false;
  H->d_send_idx = nullptr;
  H->d_send = H->d_recv = nullptr;
  H->n_send = H->n_recv = 0;
}

// ----------------------------------------------------------------------------- development: the audit log (nk_internal.h)
__global__ __launch_bounds__(256) void k_audit_hash(const uint64_t *__restrict__ p, size_t n, uint64_t *__restrict__ out) {
  __shared__ uint64_t sa[256], sb[256];
  uint64_t a = 0, b = 0;
  for (size_t i = threadIdx.x; i < n; i += 256) {
    const uint64_t w = p[i];
    a += w * (2 * (uint64_t)i + 1);                       // position-dependent, order-independent
    b ^= (w << (i & 31)) | (w >> (64 - (i & 31) - 1) >> 1);
  }
  sa[threadIdx.x] = a; sb[threadIdx.x] = b;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int t = 1; t < 256; ++t) { a += sa[t]; b ^= sb[t]; }
    *out = a ^ (b * 0x9e3779b97f4a7c15ull);
  }
}
int nk_audit(nk_ctx *ctx, int tag, const void *p, size_t nwords64) {
  nk_audit_log &A = ctx->audit;
  if (!A.on || p == nullptr) return NK_OK;
  if (A.count >= A.cap) return NK_OK;
  hipLaunchKernelGGL(k_audit_hash, dim3(1), dim3(256), 0, ctx->stream, (const uint64_t *)p, nwords64, A.d_slots + A.count);
  A.tags.push_back(tag);
  A.count++;
  return NK_OK;
}
extern "C" int nk_debug_audit_enable(nk_ctx *ctx, int on) {
  NK_REQUIRE(ctx, "NULL argument");
  nk_audit_log &A = ctx->audit;
  if (on && !A.d_slots) {
    A.cap = 1 << 16;
    NK_TRY(nk_dev_alloc(&A.d_slots, (size_t)A.cap));
  }
  A.on = on != 0;
  A.count = 0;
  A.tags.clear();
  return NK_OK;
}
// copies the log out (tags and hashes) and clears it
extern "C" int nk_debug_audit_fetch(nk_ctx *ctx, int *tags, unsigned long long *hashes, int cap, int *count) {
  NK_REQUIRE(ctx && count, "NULL argument");
  nk_audit_log &A = ctx->audit;
  const int n = A.count < cap ? A.count : cap;
  if (n > 0) {
    NK_HIP(nk_memcpy(ctx, hashes, A.d_slots, (size_t)n * sizeof(uint64_t), hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) tags[i] = A.tags[i];
  }
  *count = n;
  A.count = 0;
  A.tags.clear();
  return NK_OK;
}
