// iqlhip.hip — C ABI (include/iqlhip.h) of the MI355X IQL step.  Host side:
// arena layout, scratch management, launches, hipGraph capture of K-step chunks.
#include "iqlhip_kernels.h"
#include "iqlhip_lb_kernels.h"

#include <dlfcn.h>

#include <time.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define GRAPH_STEPS IQLHIP_GRAPH_STEPS
// Captured chunk sizes: a call of n steps is composed of replays of these (every step runs inside a graph).  The even
// sizes start and end on staging buffer 0, so they chain in any order; the one-step chunk only ever ends a call.
static const int kChunkSizes[] = {GRAPH_STEPS, 16, 4, 2, 1};

static thread_local std::string g_err;

static int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIPCHK(expr)                                                                          \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) return fail(IQLHIP_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

static inline int64_t up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

// Diagnostic (IQLHIP_TRACE=1): host timestamps inside iqlhip_train_steps, printed to stderr at the end of the call.
static inline double now_us() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}
static const bool g_trace = getenv("IQLHIP_TRACE") != nullptr;

// Host-side index check of the entry points that see the indices: the reference's `self._states[indices]` raises
// IndexError for an index outside the tensor (iql.py:173-177); here that is IQLHIP_EINDEX before anything is launched.
static int check_host_indices(const int64_t* idx_host, int64_t n, int64_t n_rows) {
  for (int64_t i = 0; i < n; ++i)
    if (idx_host[i] < 0 || idx_host[i] >= n_rows)
      return fail(IQLHIP_EINDEX, "index %lld is out of bounds for dimension 0 with size %lld", (long long)idx_host[i], (long long)n_rows);
  return IQLHIP_OK;
}

// Make the context's GPU the current HIP device for the duration of an entry point and restore the caller's
// afterwards (a trainer on cuda:1 may be driven while cuda:0 is current; the library must not change that).
struct DevGuard {
  int prev = -1;
  bool switched = false;
  explicit DevGuard(int dev) {
    if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = (hipSetDevice(dev) == hipSuccess);
  }
  ~DevGuard() { if (switched) (void)hipSetDevice(prev); }
};

// What a captured chunk graph depends on through frozen kernel arguments.  The number of steps of a call is NOT part
// of it: a call is composed of replays of the fixed chunk graphs (64, 16, 4, 2 steps and one), nothing is ever
// captured per call length (kernel boundaries inside a graph are ~0.5 us shorter than between directly launched kernels).
struct GraphKey {
  const float* rows = nullptr;
  int64_t ld = 0;
  int32_t B = 0;
  int32_t K = 0;      // steps in the chunk: one of kChunkSizes, never the caller's step count
  float* params = nullptr;
  float drop_p = 0.f;
  float inv_batch = 0.f;
  int xch = 0;        // exchange mode the chunk was captured with
  int parity = 0;     // P2P exchange: which flat buffer step 0 of the chunk writes
  int head = 0;       // the chunk a call starts with: its graph begins with the call's set-up kernel (arguments set per replay)
  bool operator==(const GraphKey& o) const {
    return rows == o.rows && ld == o.ld && B == o.B && K == o.K && params == o.params && drop_p == o.drop_p &&
           inv_batch == o.inv_batch && xch == o.xch && parity == o.parity && head == o.head;
  }
};

// The few RCCL entry points the in-stream all-reduce needs, resolved at run time from the librccl.so.1 the process
// already has (PyTorch-ROCm brings one) or can load — the library has no link-time dependency on RCCL.
struct RcclId { char internal[IQLHIP_UNIQUE_ID_BYTES]; };
struct RcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, RcclId /* ncclUniqueId, by value */, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};

struct iqlhip_ctx {
  iqlhip_dims dims;
  iqlhip_hyper hyper;
  iqlhip_layout L;
  int device = 0;
  // bound (caller-owned)
  float *params = nullptr, *target = nullptr, *m = nullptr, *v = nullptr;
  // scratch (library-owned)
  DevScratch sc{};
  float* flat_tmp = nullptr;          // n_params + 4 (debug "grads")
  float* xb = nullptr;                // compact batch [max_batch][row_ld]: rows [s | a | s' | r | d | pad]
  // iqlhip_actor_forward's own staging (never aliases a training batch): packed states and policy head partials
  float* xb2 = nullptr;               // second staging batch: graph chunks alternate (step k reads one, gathers k+1 into the other)
  float* xb_act = nullptr;            // [act_cap][row_ld]
  float* heads_act = nullptr;         // [act_cap][A][NSPLIT]
  float* losses_host = nullptr;       // pinned landing pad of read_losses (a pageable D2H goes through a bounce copy)
  float* on_row_pin = nullptr;        // iqlhip_online_step: pinned, host-mapped staging of the new transition [row_ld]
  long long* on_idx_pin = nullptr;    // ... and of the sampled indices [max_batch]
  float* on_loss_pin = nullptr;       // ... and the landing words of the step's three losses [4]
  unsigned long long* done_pin = nullptr;   // host-mapped completion word of the synchronous entry points (the host spins on it)
  unsigned long long done_seq = 0;
  float* on_act_pin = nullptr;        // ... and of the follow-up act(): state in [IQLHIP_MAX_INPUT], action out [IQLHIP_MAX_ACTION]
  int act_cap = 0;
  unsigned long long act_calls = 0;   // Philox call counter of iqlhip_actor_sample
  int64_t row_ld = 0;
  // actor dropout
  unsigned* drop_bits = nullptr;      // [2 parities][2 layers][max_batch][8] keep-bits (a step reads one parity while the
                                      // forward's idle blocks draw the next step's into the other)
  float drop_p = 0.f;
  unsigned long long drop_seed = 0, drop_step = 0;
  bool drop_inject = false;           // tests: masks were written by iqlhip_debug_write_masks, do not regenerate
  int precision = 0;                  // 0: fp32 MFMA everywhere; 1: bf16 operands for the layer-0/1, dW1, dH0, dW0 products
  __bf16* wsh = nullptr;              // bf16 path: shadow of the parameter arena [n_params] (W1 is read from it) ...
  __bf16* tsh = nullptr;              // ... and of the target arena [n_target]; written by the update kernel, refreshed
                                      // from the fp32 masters at the start of every library call
  float* loss_ring = nullptr;         // [ring_cap][4], host-mapped pinned
  int ring_cap = 0;
  iqlhip_step_scalars* sched_cur = nullptr;   // [GRAPH_STEPS] device: per-step scalars of the chunk in flight
  iqlhip_step_scalars* sched_call = nullptr;  // [k_max] device copy of the scalar table of the call in flight
  iqlhip_step_scalars* sched_pin[4] = {nullptr, nullptr, nullptr, nullptr};  // pinned, host-mapped copies of a call's table [k_max]
  // a slot is free again once the set-up kernel that read it has acknowledged the call's number in sched_ack[slot]
  // (a pinned, host-mapped word the kernel writes; the host only reads memory — no event, no HIP call)
  unsigned long long* sched_ack = nullptr;      // pinned [4]
  unsigned long long sched_want[4] = {0, 0, 0, 0};
  unsigned long long call_seq = 0;
  unsigned* setup_arrivals = nullptr;           // device: block counter of iql_call_setup_kernel
  char* prep_save = nullptr;                    // device: prepare's copy of the four arenas, kept (a hipFree at the end of prepare
                                                //   idles the GPU right in front of the caller's first steps)
  hipStream_t sched_stream[4] = {nullptr, nullptr, nullptr, nullptr};   // (the stream a slot's reader was queued on)
  int sched_slot = 0;
  unsigned long long* hdr = nullptr;  // [HDR_WORDS] per-launch values of a chunk (ChunkHdr)
  unsigned long long* stamps = nullptr;  // diagnostic builds (-DIQL_STAMPS): [4096 blocks][16]
  int k_max = 0;
  int n_chunk_max = 0, n_rt_max = 0;
  size_t lds_fwd = 0, lds_fwd_solo = 0, lds_bwd = 0;
  int n_cus = 256;                    // compute units of the device (MI355X: 256)
  int w0_lds_k = 0;                   // widest layer-0 input whose weights the forward stages in LDS
  int bwd_donate_pct = -1;            // diagnostic (IQLHIP_BWD_DONATE_PCT): share of the policy's dW1 tiles run on the other XCDs
  int bwd_spb_force = -1;             // the same for the backward's (b) blocks (IQLHIP_BWD_SPB_L2)
  int fwd_spb_force = -1;             // diagnostic (IQLHIP_FWD_SPB_L2): fixed slices-per-block exponent of the forward
  // large-batch bf16 path (iqlhip_lb_kernels.h): bf16 precision and more than LB_MIN_ROWS rows per step
  float* pi_t = nullptr;              // [max_batch][32] T | [max_batch][32] G | [max_batch] L: the policy's loss terms without w
  __bf16* dh1g = nullptr;             // [4][max_batch][256] dH1 rows, [4][max_batch][256] dH0 rows, [max_batch][32] policy dY (allocated
  float* slab_x = nullptr;            // with the bf16 shadows); [64][n_params] the row blocks' partial sums
  __bf16* wimg = nullptr;             // bf16 path: operand images of W1 / W0, [6][IMG_STRIDE] (iqlhip_kernels.h)
  int lb_enabled = 1;                 // diagnostic (IQLHIP_LB=0): keep the small-batch kernels at every batch size
  int lb_nbb_force = -1, lb_cpb_force = -1, lb_nbi_force = -1;   // diagnostic (IQLHIP_LB_NBB / _CPB / _NBI)
  bool lb_pi_spread = true;           // IQLHIP_LB_PI_SPREAD=0: the policy's forward tiles on its own 32 blocks only (diagnostic)
  bool lb_csplit = true;              // IQLHIP_LB_CSPLIT=0: no column split of the row kernel at <= 1 024 rows (diagnostic)
  int lb_bwd_part = 0;                // iqlhip_debug_time_kernel only: 1 = launch the row kernel alone, 2 = the GEMM kernel alone
  size_t lds_bwd_lb = 0;
  // graph cache (a few (K,B,buffer) shapes: the steady chunk, the tail chunk, ...)
  hipStream_t cap_stream = nullptr;
  struct CachedGraph { GraphKey key; hipGraph_t graph; hipGraphExec_t exec; unsigned long long stamp; hipStream_t last;
                       IdleWork* work; /* [K] device records of the chunk's idle-block work */
                       hipGraphNode_t setup_node; /* head chunks: the set-up kernel's node */ };
  // Continuation of the index stream across calls: the last forward of a call stages the rows (and keep-bits) of the
  // step that would come next; a following call that IS that step (same rows / size / batch / seed, counter position
  // where the previous call stopped, staging untouched in between) starts without gathering anything.
  struct { bool valid = false; const float* rows = nullptr; int64_t ld = 0, size = 0; int32_t B = 0; uint64_t seed = 0;
           uint64_t next_offset = 0; float drop_p = 0.f; uint64_t drop_seed = 0, drop_step = 0; } cont;
  std::vector<CachedGraph> graphs;
  unsigned long long graph_clock = 0;
  // data-parallel gradient exchange
  int xch_mode = IQLHIP_XCH_NONE;
  int rank = 0, world = 1;
  float* xflat = nullptr;             // RCCL / split path: this rank's flat gradient [n_params + 4] (+ pad)
  void* nccl_comm = nullptr;
  // P2P: one exchange block per rank = [flags: IQLHIP_MAX_WORLD x 16 u64][flat 0][flat 1], exported through hipIpc
  char* xblk = nullptr;
  size_t xblk_bytes = 0, xflat_off[2] = {0, 0};   // per parity buffer: [flat n_params+4 | slab_b (8 row tiles) | loss_parts]
  size_t xslabb_off[2] = {0, 0}, xloss_off[2] = {0, 0};
  long long xslab_b_off[4] = {0, 0, 0, 0};        // per-net offsets inside an exchange block's slab_b region (8 row tiles)
  char* peer_blk[IQLHIP_MAX_WORLD] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  bool p2p_attached = false;
  unsigned long long* xstatus = nullptr;   // device: [0] first timed-out step, [1] spare
  unsigned long long* xstatus_host = nullptr;   // pinned landing pad of the status words
  unsigned long long xstep = 0;            // steps exchanged so far (the P2P flags count them)
  unsigned long long xtimeout_ticks = 500000000ull;   // 5 s of the 100 MHz wall clock
  // timing
  bool timing = false;
  std::vector<hipEvent_t> ev;         // 4 per recorded step
  int ev_used = 0;
  float t_acc[4] = {0, 0, 0, 0};
  int t_n = 0;
};

extern "C" int iqlhip_version(void) { return IQLHIP_VERSION; }
extern "C" const char* iqlhip_last_error(void) { return g_err.c_str(); }

static int check_dims(const iqlhip_dims* d) {
  if (!d) return fail(IQLHIP_EINVAL, "dims is NULL");
  if (d->state_dim < 1 || d->action_dim < 1) return fail(IQLHIP_EINVAL, "state_dim/action_dim must be >= 1");
  if (d->hidden_dim != IQLHIP_HIDDEN || d->n_hidden != 2)
    return fail(IQLHIP_EUNSUPPORTED, "kernels are built for hidden_dim=%d, n_hidden=2 (got %d, %d)", IQLHIP_HIDDEN,
                d->hidden_dim, d->n_hidden);
  if (d->state_dim + d->action_dim > IQLHIP_MAX_INPUT)
    return fail(IQLHIP_EUNSUPPORTED, "state_dim + action_dim > %d", IQLHIP_MAX_INPUT);
  if (d->action_dim > IQLHIP_MAX_ACTION) return fail(IQLHIP_EUNSUPPORTED, "action_dim > %d", IQLHIP_MAX_ACTION);
  if (d->policy != IQLHIP_POLICY_GAUSSIAN && d->policy != IQLHIP_POLICY_DETERMINISTIC)
    return fail(IQLHIP_EINVAL, "unknown policy kind %d", d->policy);
  if (d->max_batch < 1 || d->max_batch > 16384) return fail(IQLHIP_EINVAL, "max_batch must be in [1,16384]");
  return IQLHIP_OK;
}

extern "C" int iqlhip_arena_layout(const iqlhip_dims* d, iqlhip_layout* out) {
  int rc = check_dims(d);
  if (rc) return rc;
  if (!out) return fail(IQLHIP_EINVAL, "out is NULL");
  const int S = d->state_dim, A = d->action_dim, Hd = IQLHIP_HIDDEN;
  int64_t off = 0;
  for (int n = 0; n < 4; ++n) {
    iqlhip_net_layout& nl = out->net[n];
    nl.k_in = (n == IQLHIP_NET_Q1 || n == IQLHIP_NET_Q2) ? S + A : S;
    nl.d_out = (n == IQLHIP_NET_PI) ? A : 1;
    nl.seg_begin = off;
    nl.w1 = off; off += (int64_t)Hd * Hd;
    nl.w0 = off; off += (int64_t)Hd * nl.k_in;
    nl.b0 = off; off += Hd;
    nl.b1 = off; off += Hd;
    nl.w2 = off; off += (int64_t)nl.d_out * Hd;
    nl.b2 = off; off += up(nl.d_out, 4);
    if (n == IQLHIP_NET_PI && d->policy == IQLHIP_POLICY_GAUSSIAN) { nl.log_std = off; off += up(A, 4); }
    else nl.log_std = -1;
    off = up(off, 64);
    nl.seg_end = off;
  }
  out->n_params = off;
  out->target_src = out->net[IQLHIP_NET_Q1].seg_begin;
  out->n_target = out->net[IQLHIP_NET_Q2].seg_end - out->net[IQLHIP_NET_Q1].seg_begin;
  return IQLHIP_OK;
}

extern "C" int64_t iqlhip_row_stride(int32_t S, int32_t A) { return up(2 * (int64_t)S + A + 2, 4); }

static int xld_host(int k0) { int k0p = (k0 + 3) & ~3; return ((k0p + 29) / 32) * 32 + 2; }

extern "C" int iqlhip_destroy(iqlhip_ctx* c);
extern "C" int iqlhip_xch_shutdown(iqlhip_ctx* c);

static int create_impl(iqlhip_ctx* c, const iqlhip_dims* dims, const iqlhip_hyper* hyper, int device) {
  c->dims = *dims;
  c->hyper = *hyper;
  c->device = device;
  iqlhip_arena_layout(dims, &c->L);
  const int MB = dims->max_batch, A = dims->action_dim;
  c->n_chunk_max = (MB + CHUNK_ROWS - 1) / CHUNK_ROWS;
  c->n_rt_max = (MB + RT_ROWS - 1) / RT_ROWS;
  c->sc.max_batch = MB;
  auto dalloc = [&](float** p, size_t nfloat) -> hipError_t {
    hipError_t e = hipMalloc((void**)p, nfloat * sizeof(float));
    if (e != hipSuccess) return e;
    return hipMemset(*p, 0, nfloat * sizeof(float));
  };
  HIPCHK(dalloc(&c->sc.h0, (size_t)4 * MB * HID));
  HIPCHK(dalloc(&c->sc.h1, (size_t)4 * MB * HID));
  HIPCHK(dalloc(&c->sc.heads, (size_t)MB * HEAD_LD + (size_t)NSPLIT * MB * A));
  c->row_ld = iqlhip_row_stride(dims->state_dim, A);
  HIPCHK(dalloc(&c->xb, (size_t)MB * c->row_ld));
  HIPCHK(dalloc(&c->xb2, (size_t)MB * c->row_ld));
  HIPCHK(hipHostMalloc((void**)&c->losses_host, 4 * sizeof(float), hipHostMallocDefault));
  HIPCHK(hipHostMalloc((void**)&c->on_row_pin, (size_t)c->row_ld * sizeof(float), hipHostMallocDefault));
  HIPCHK(hipHostMalloc((void**)&c->on_idx_pin, (size_t)MB * sizeof(long long), hipHostMallocDefault));
  HIPCHK(hipHostMalloc((void**)&c->on_loss_pin, 4 * sizeof(float), hipHostMallocDefault));
  HIPCHK(hipHostMalloc((void**)&c->done_pin, 8 * sizeof(unsigned long long), hipHostMallocDefault));
  memset(c->done_pin, 0, 8 * sizeof(unsigned long long));
  HIPCHK(hipHostMalloc((void**)&c->on_act_pin, (size_t)(IQLHIP_MAX_INPUT + IQLHIP_MAX_ACTION) * sizeof(float), hipHostMallocDefault));
  c->act_cap = std::max(MB, IQLHIP_ACT_ROWS);
  HIPCHK(dalloc(&c->xb_act, (size_t)c->act_cap * c->row_ld));
  HIPCHK(dalloc(&c->heads_act, (size_t)c->act_cap * A * NSPLIT));
  HIPCHK(hipMalloc((void**)&c->drop_bits, (size_t)4 * MB * 8 * sizeof(unsigned)));
  HIPCHK(hipMemset(c->drop_bits, 0xFF, (size_t)4 * MB * 8 * sizeof(unsigned)));
  HIPCHK(dalloc(&c->sc.slab_a, (size_t)c->n_chunk_max * c->L.n_params));
  size_t sb = 0;
  for (int n = 0; n < 4; ++n) {
    c->sc.slab_b_off[n] = (long long)sb;
    sb += (size_t)c->n_rt_max * ((size_t)HID * c->L.net[n].k_in + HID);
  }
  HIPCHK(dalloc(&c->sc.slab_b, sb));
  HIPCHK(dalloc(&c->sc.loss_parts, 2 * 4 * 64));      // [4 losses][64 chunks / row blocks] (+ a second set: the large-batch path's totals)
  HIPCHK(dalloc(&c->pi_t, (size_t)MB * 65));
  HIPCHK(dalloc(&c->sc.losses, 4));
  HIPCHK(dalloc(&c->flat_tmp, (size_t)c->L.n_params + 4));
  c->k_max = 1024;
  c->ring_cap = c->k_max;
  // the loss ring lives in host-mapped pinned memory: each step's update kernel posts its 3 words there, and the
  // end of a train_steps call is ONE stream synchronisation queued right behind the work — no device-to-host copy
  HIPCHK(hipHostMalloc((void**)&c->loss_ring, (size_t)c->ring_cap * 4 * sizeof(float), hipHostMallocDefault));
  memset(c->loss_ring, 0, (size_t)c->ring_cap * 4 * sizeof(float));
  HIPCHK(hipMalloc((void**)&c->sched_cur, (size_t)GRAPH_STEPS * sizeof(iqlhip_step_scalars)));
  HIPCHK(hipMalloc((void**)&c->sched_call, (size_t)c->k_max * sizeof(iqlhip_step_scalars)));
  for (int i = 0; i < 4; ++i) {
    HIPCHK(hipHostMalloc((void**)&c->sched_pin[i], (size_t)c->k_max * sizeof(iqlhip_step_scalars), hipHostMallocDefault));
  }
  HIPCHK(hipHostMalloc((void**)&c->sched_ack, 8 * sizeof(unsigned long long), hipHostMallocDefault));   // [4] slots + a dummy word
  memset(c->sched_ack, 0, 8 * sizeof(unsigned long long));
  HIPCHK(hipMalloc((void**)&c->setup_arrivals, 64));
  HIPCHK(hipMemset(c->setup_arrivals, 0, 64));
  HIPCHK(hipMalloc((void**)&c->hdr, HDR_WORDS * sizeof(unsigned long long)));
  HIPCHK(hipMemset(c->hdr, 0, HDR_WORDS * sizeof(unsigned long long)));
  HIPCHK(hipMalloc((void**)&c->xstatus, 2 * sizeof(unsigned long long)));
  HIPCHK(hipMemset(c->xstatus, 0, 2 * sizeof(unsigned long long)));
  HIPCHK(hipHostMalloc((void**)&c->xstatus_host, 2 * sizeof(unsigned long long), hipHostMallocDefault));
  c->xstatus_host[0] = c->xstatus_host[1] = 0ull;
  HIPCHK(dalloc(&c->xflat, (size_t)up(c->L.n_params + 4, 64)));
  HIPCHK(hipStreamCreateWithFlags(&c->cap_stream, hipStreamNonBlocking));
#ifdef IQL_STAMPS
  HIPCHK(hipMalloc((void**)&c->stamps, 4096 * 16 * sizeof(unsigned long long)));
  HIPCHK(hipMemset(c->stamps, 0, 4096 * 16 * sizeof(unsigned long long)));
#endif
  // LDS sizes
  const int kq = dims->state_dim + dims->action_dim;
  const int ks_ = dims->state_dim;   // V / pi layer-0 width; Q nets use kq
  // layer-0 weights are staged in LDS for every instance whose input width fits: through registers up to
  // W0_LDS_MAX_K, by LDS-DMA (no registers) above that, as far as the CU's 160 KB allow (1 KiB-float4 slack for the
  // DMA's whole-wave granularity)
  const size_t fwd_fixed = (size_t)(RT_ROWS * H0_LD + RT_ROWS * T64_LD + RT_ROWS * (int)c->row_ld +
                                    (((A + 15) & ~15) * W2_LD + 32) + 512 + 16) * sizeof(float);
  auto fits = [&](int k) { return fwd_fixed + (size_t)HID * k * sizeof(float) + 4096 <= (size_t)160 * 1024 - 1024; };
  int w0_lds_k = 0;
  if (kq <= W0_DMA_MAX_K && fits(kq)) w0_lds_k = kq;
  else if (ks_ <= W0_DMA_MAX_K && fits(ks_)) w0_lds_k = ks_;
  if (const char* ov = getenv("IQLHIP_W0_LDS_K")) w0_lds_k = std::min(w0_lds_k, atoi(ov));   // diagnostic (tools/): force a narrower staging
  c->w0_lds_k = w0_lds_k;
  if (const char* ov = getenv("IQLHIP_BWD_DONATE_PCT")) c->bwd_donate_pct = std::max(0, std::min(100, atoi(ov)));   // diagnostic (tools/)
  if (const char* ov = getenv("IQLHIP_BWD_SPB_L2")) c->bwd_spb_force = std::max(0, std::min(2, atoi(ov)));   // diagnostic (tools/)
  if (const char* ov = getenv("IQLHIP_FWD_SPB_L2")) c->fwd_spb_force = std::max(0, std::min(2, atoi(ov)));   // diagnostic (tools/)
  if (const char* ov = getenv("IQLHIP_LB")) c->lb_enabled = atoi(ov) != 0;                                     // diagnostic (tools/)
  if (const char* ov = getenv("IQLHIP_LB_NBB")) c->lb_nbb_force = std::max(2, atoi(ov));
  if (const char* ov = getenv("IQLHIP_LB_CSPLIT")) c->lb_csplit = atoi(ov) != 0;
  if (const char* ov = getenv("IQLHIP_LB_PI_SPREAD")) c->lb_pi_spread = atoi(ov) != 0;
  if (const char* ov = getenv("IQLHIP_LB_CPB")) c->lb_cpb_force = std::max(1, atoi(ov));
  if (const char* ov = getenv("IQLHIP_LB_NBI")) c->lb_nbi_force = std::max(2, atoi(ov));
  c->lds_fwd = fwd_fixed + (size_t)HID * w0_lds_k * sizeof(float) + (w0_lds_k > W0_LDS_MAX_K ? 4096 : 0);
  // One block per CU while the grid fits the chip (co-resident blocks share a CU's L1 and fill rate and only slow
  // each other down); the exact size — two blocks per CU where it is <= 80 KB — once there are more blocks than CUs.
  c->lds_fwd_solo = std::max(c->lds_fwd, (size_t)(81 * 1024));
  {
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->device) == hipSuccess && ncu > 0) c->n_cus = ncu;
  }
  const int dyld = ((A + 15) & ~15) + 4;     // (a) blocks: [256][Dp + 4] rows for dY and dlog_std terms
  const size_t lds_a = (size_t)(4 * 32 * T64_LD + 2 * CHUNK_ROWS * dyld + 32 * 32 + 64 + CHUNK_ROWS) * sizeof(float);
  const size_t lds_b = (size_t)(RT_ROWS * H0_LD + 4 * 32 * T64_LD + RT_ROWS * T64_LD + RT_ROWS * 36 + 4 +
                                RT_ROWS * XR_LD_MAX) * sizeof(float);
  c->lds_bwd = std::max(lds_a, lds_b);
  {
    const void* fwd1[4] = {(const void*)iql_fwd_kernel<false, false, false, true>, (const void*)iql_fwd_kernel<false, true, false, true>,
                           (const void*)iql_fwd_kernel<true, false, false, true>,  (const void*)iql_fwd_kernel<true, true, false, true>};
    for (const void* f : fwd1) HIPCHK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_fwd_solo));
    const void* fwd[8] = {(const void*)iql_fwd_kernel<false, false, false>, (const void*)iql_fwd_kernel<false, true, false>,
                          (const void*)iql_fwd_kernel<true, false, false>,  (const void*)iql_fwd_kernel<true, true, false>,
                          (const void*)iql_fwd_kernel<false, false, true>,  (const void*)iql_fwd_kernel<false, true, true>,
                          (const void*)iql_fwd_kernel<true, false, true>,   (const void*)iql_fwd_kernel<true, true, true>};
    for (const void* f : fwd) HIPCHK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_fwd_solo));
  }
  {
    const void* bwd[8] = {(const void*)iql_bwd_kernel<false, false, false>, (const void*)iql_bwd_kernel<false, true, false>,
                          (const void*)iql_bwd_kernel<true, false, false>,  (const void*)iql_bwd_kernel<true, true, false>,
                          (const void*)iql_bwd_kernel<false, false, true>,  (const void*)iql_bwd_kernel<false, true, true>,
                          (const void*)iql_bwd_kernel<true, false, true>,   (const void*)iql_bwd_kernel<true, true, true>};
    for (const void* f : bwd) HIPCHK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_bwd));
  }
  {
    // large-batch backward, row blocks: [dH1 tile | H1 tile | dH0 tile | dY | dy] (iql_bwd_rows_kernel)
    c->lds_bwd_lb = (size_t)(3 * 32 * H0B_LD + 32 * LB_DYLD) * 2 + 32 * 4 + (size_t)(2 * 4 * 32 + 16) * 4;      // (+ the block sums' partials)
    HIPCHK(hipFuncSetAttribute((const void*)iql_bwd_rows_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_bwd_lb));
    HIPCHK(hipFuncSetAttribute((const void*)iql_bwd_rows_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_bwd_lb));
  }
  return IQLHIP_OK;
}

extern "C" int iqlhip_create(const iqlhip_dims* dims, const iqlhip_hyper* hyper, int device, iqlhip_ctx** out) {
  int rc = check_dims(dims);
  if (rc) return rc;
  if (!hyper || !out) return fail(IQLHIP_EINVAL, "hyper/out is NULL");
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(IQLHIP_EHIP, "device %d not available (%d visible)", device, ndev);
  DevGuard guard(device);               // the caller's current device is restored on return
  iqlhip_ctx* c = new iqlhip_ctx();
  rc = create_impl(c, dims, hyper, device);
  if (rc) {                             // free whatever was allocated before the failure (the message survives)
    const std::string msg = g_err;
    iqlhip_destroy(c);
    g_err = msg;
    return rc;
  }
  *out = c;
  return IQLHIP_OK;
}

static void drop_graph(iqlhip_ctx* c) {
  for (auto& g : c->graphs) {
    if (g.last) (void)hipStreamSynchronize(g.last);   // a replay may still be executing
    (void)hipGraphExecDestroy(g.exec);
    (void)hipGraphDestroy(g.graph);
    if (g.work) (void)hipFree(g.work);
  }
  c->graphs.clear();
  c->cont.valid = false;
}

extern "C" int iqlhip_destroy(iqlhip_ctx* c) {
  if (!c) return IQLHIP_OK;
  DevGuard guard(c->device);
  (void)hipDeviceSynchronize();
  drop_graph(c);
  (void)iqlhip_xch_shutdown(c);
  for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
  if (c->cap_stream) (void)hipStreamDestroy(c->cap_stream);
  void* bufs[] = {c->sc.h0, c->sc.h1, c->sc.heads, c->sc.slab_a, c->sc.slab_b, c->sc.loss_parts, c->sc.losses,
                  c->flat_tmp, c->sched_call, c->sched_cur, c->hdr, c->stamps, c->xb, c->xb2, c->xb_act,
                  c->heads_act, c->drop_bits, c->xstatus, c->xflat, c->wsh, c->tsh, c->pi_t, c->dh1g, c->slab_x, c->wimg};
  for (void* b : bufs) if (b) (void)hipFree(b);
  for (int i = 0; i < 4; ++i) {
    if (c->sched_pin[i]) (void)hipHostFree(c->sched_pin[i]);
  }
  if (c->sched_ack) (void)hipHostFree(c->sched_ack);
  if (c->loss_ring) (void)hipHostFree(c->loss_ring);
  if (c->xstatus_host) (void)hipHostFree(c->xstatus_host);
  if (c->setup_arrivals) (void)hipFree(c->setup_arrivals);
  if (c->prep_save) (void)hipFree(c->prep_save);
  if (c->losses_host) (void)hipHostFree(c->losses_host);
  if (c->on_row_pin) (void)hipHostFree(c->on_row_pin);
  if (c->on_idx_pin) (void)hipHostFree(c->on_idx_pin);
  if (c->on_loss_pin) (void)hipHostFree(c->on_loss_pin);
  if (c->done_pin) (void)hipHostFree(c->done_pin);
  if (c->on_act_pin) (void)hipHostFree(c->on_act_pin);
  delete c;
  return IQLHIP_OK;
}

extern "C" int iqlhip_set_hyper(iqlhip_ctx* c, const iqlhip_hyper* h) {
  if (!c || !h) return fail(IQLHIP_EINVAL, "NULL argument");
  c->hyper = *h;
  drop_graph(c);
  return IQLHIP_OK;
}

extern "C" int iqlhip_set_precision(iqlhip_ctx* c, int mode) {
  if (!c) return fail(IQLHIP_EINVAL, "NULL ctx");
  if (mode != 0 && mode != 1) return fail(IQLHIP_EINVAL, "precision mode must be 0 (f32) or 1 (bf16 operands)");
  if (mode != c->precision) drop_graph(c);
  if (mode == 1 && !c->wsh) {
    DevGuard guard(c->device);
    HIPCHK(hipMalloc((void**)&c->wsh, (size_t)up(c->L.n_params, 64) * sizeof(__bf16)));
    HIPCHK(hipMalloc((void**)&c->tsh, (size_t)up(c->L.n_target, 64) * sizeof(__bf16)));
    // scratch of the large-batch backward (iqlhip_lb_kernels.h)
    const size_t MB = (size_t)c->dims.max_batch;
    HIPCHK(hipMalloc((void**)&c->dh1g, (8 * MB * HID + MB * 32 + MB * LB_XLD + 4 * 65536 + 8192) * sizeof(__bf16)));
    HIPCHK(hipMemset(c->dh1g, 0, (8 * MB * HID + MB * 32 + MB * LB_XLD + 4 * 65536 + 8192) * sizeof(__bf16)));
    HIPCHK(hipMalloc((void**)&c->wimg, (size_t)6 * IMG_STRIDE * sizeof(__bf16)));
    HIPCHK(hipMemset(c->wimg, 0, (size_t)6 * IMG_STRIDE * sizeof(__bf16)));
    HIPCHK(hipMalloc((void**)&c->slab_x, (size_t)64 * c->L.n_params * sizeof(float)));
    HIPCHK(hipMemset(c->slab_x, 0, (size_t)64 * c->L.n_params * sizeof(float)));
  }
  c->precision = mode;
  return IQLHIP_OK;
}

// The context's two Philox stream positions: {dropout step, act() call}.  A caller that replaces a context (the shim
// re-creates it when a larger batch arrives) carries them over so that neither stream replays from its beginning.
extern "C" int iqlhip_get_counters(const iqlhip_ctx* c, uint64_t out[2]) {
  if (!c || !out) return fail(IQLHIP_EINVAL, "NULL argument");
  out[0] = c->drop_step;
  out[1] = c->act_calls;
  return IQLHIP_OK;
}
extern "C" int iqlhip_set_counters(iqlhip_ctx* c, const uint64_t in[2]) {
  if (!c || !in) return fail(IQLHIP_EINVAL, "NULL argument");
  c->drop_step = in[0];
  c->act_calls = in[1];
  return IQLHIP_OK;
}

extern "C" int iqlhip_set_dropout(iqlhip_ctx* c, float p, uint64_t seed) {
  if (!c) return fail(IQLHIP_EINVAL, "NULL ctx");
  if (!(p >= 0.f && p < 1.f)) return fail(IQLHIP_EINVAL, "dropout probability must be in [0,1)");
  c->drop_p = p;
  c->drop_seed = seed;
  c->drop_inject = false;
  return IQLHIP_OK;
}

extern "C" int iqlhip_debug_write_masks(iqlhip_ctx* c, const uint32_t* keep0, const uint32_t* keep1, int32_t rows,
                                        void* stream) {
  if (!c || !keep0 || !keep1) return fail(IQLHIP_EINVAL, "NULL argument");
  if (rows < 1 || rows > c->dims.max_batch) return fail(IQLHIP_EINVAL, "rows outside [1,max_batch]");
  hipStream_t st = (hipStream_t)stream;
  const size_t nb = (size_t)rows * 8 * sizeof(unsigned);
  HIPCHK(hipMemcpyAsync(c->drop_bits, keep0, nb, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(c->drop_bits + (size_t)c->dims.max_batch * 8, keep1, nb, hipMemcpyHostToDevice, st));
  HIPCHK(hipStreamSynchronize(st));
  c->drop_inject = true;
  c->cont.valid = false;
  return IQLHIP_OK;
}

extern "C" int iqlhip_bind(iqlhip_ctx* c, float* params, float* target, float* m, float* v) {
  if (!c || !params || !target || !m || !v) return fail(IQLHIP_EINVAL, "NULL argument");
  if (((uintptr_t)params | (uintptr_t)target | (uintptr_t)m | (uintptr_t)v) & 15)
    return fail(IQLHIP_EINVAL, "arenas must be 16-byte aligned");
  c->params = params; c->target = target; c->m = m; c->v = v;
  drop_graph(c);
  return IQLHIP_OK;
}

extern "C" int64_t iqlhip_grad_words(const iqlhip_ctx* c) { return c ? c->L.n_params + 4 : 0; }

// ---------------------------------------------------------------------------
static int check_batch(const iqlhip_ctx* c, const iqlhip_batch* b) {
  if (!c->params) return fail(IQLHIP_ENOTBOUND, "iqlhip_bind has not been called");
  if (!b) return fail(IQLHIP_EINVAL, "batch is NULL");
  if (b->rows < 1 || b->rows > c->dims.max_batch)
    return fail(IQLHIP_EINVAL, "batch rows %d outside [1, max_batch=%d]", b->rows, c->dims.max_batch);
  if (!b->s_dev || !b->a_dev || !b->r_dev || !b->ns_dev || !b->d_dev) return fail(IQLHIP_EINVAL, "NULL batch tensor");
  if (b->ld_s < c->dims.state_dim || b->ld_ns < c->dims.state_dim || b->ld_a < c->dims.action_dim || b->ld_r < 1 ||
      b->ld_d < 1)
    return fail(IQLHIP_EINVAL, "batch row strides smaller than the row widths");
  return IQLHIP_OK;
}

// True if the five arrays are views of ONE block of packed rows [s|a|s'|r|d|pad] with the library's row stride.
static bool is_packed(const iqlhip_ctx* c, const iqlhip_batch* b) {
  const int S = c->dims.state_dim, A = c->dims.action_dim;
  return b->ld_s == c->row_ld && b->ld_a == c->row_ld && b->ld_r == c->row_ld && b->ld_ns == c->row_ld &&
         b->ld_d == c->row_ld && b->a_dev == b->s_dev + S && b->ns_dev == b->s_dev + S + A &&
         b->r_dev == b->s_dev + 2 * S + A && b->d_dev == b->s_dev + 2 * S + A + 1;
}

// An indexed batch must be row-addressed storage in the packed layout (what ReplayBuffer holds).
static int check_indexed(const iqlhip_ctx* c, const iqlhip_batch* b) {
  const bool packed = is_packed(c, b);
  if (!packed) return fail(IQLHIP_EINVAL, "indexed batches must address packed rows [s|a|s'|r|d] with ld=%lld",
                           (long long)c->row_ld);
  if (((uintptr_t)b->s_dev) & 15) return fail(IQLHIP_EINVAL, "packed rows must be 16-byte aligned");
  return IQLHIP_OK;
}

static void launch_gather(const iqlhip_ctx* c, const float* rows, const long long* idx, int n, hipStream_t st) {
  const int total = n * (int)(c->row_ld / 4);
  hipLaunchKernelGGL(iql_gather_kernel, dim3((total + 255) / 256), dim3(256), 0, st, rows, (long long)c->row_ld, idx,
                     c->xb, n, 0ll);
}

// Bring the caller's batch into packed rows (the kernels read nothing else): gather by index, or pack five
// arrays into the staging buffer xb, or — when the caller's arrays already ARE one block of packed rows (what
// ReplayBuffer.sample returns) — consume them in place.  *xb_out = where the packed batch is.
static int stage_batch(const iqlhip_ctx* c, const iqlhip_batch* b, hipStream_t st, const float** xb_out) {
  *xb_out = c->xb;
  if (!b->idx_dev && is_packed(c, b) && !(((uintptr_t)b->s_dev) & 15)) {
    *xb_out = b->s_dev;
    return IQLHIP_OK;
  }
  if (b->idx_dev) {
    int rc = check_indexed(c, b);
    if (rc) return rc;
    launch_gather(c, b->s_dev, (const long long*)b->idx_dev, b->rows, st);
  } else {
    const int total = b->rows * (int)c->row_ld;
    hipLaunchKernelGGL(iql_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, st, c->xb, (int)c->row_ld,
                       c->dims.state_dim, c->dims.action_dim, b->rows, b->s_dev, (long long)b->ld_s, b->a_dev,
                       (long long)b->ld_a, b->r_dev, (long long)b->ld_r, b->ns_dev, (long long)b->ld_ns, b->d_dev,
                       (long long)b->ld_d);
  }
  return IQLHIP_OK;
}

static NetPtrs net_ptrs(const iqlhip_net_layout& nl, const float* base) {
  NetPtrs n;
  n.w0 = base + nl.w0; n.b0 = base + nl.b0; n.w1 = base + nl.w1; n.b1 = base + nl.b1;
  n.w2 = base + nl.w2; n.b2 = base + nl.b2; n.k0 = nl.k_in; n.d = nl.d_out;
  return n;
}

// ---------------------------------------------------------------------------
// Large-batch bf16 path (iqlhip_lb_kernels.h): which steps take it, and how their rows are spread over blocks.
#define LB_MIN_ROWS 512
static bool use_lb(const iqlhip_ctx* c, int rows) {
  if (c->precision != 1 || !c->lb_enabled || rows <= LB_MIN_ROWS) return false;
  if (c->fwd_spb_force >= 0 || c->bwd_spb_force >= 0) return false;      // (diagnostic layouts of the small-batch kernels)
  return c->dims.state_dim + c->dims.action_dim + 1 <= 16 * 5;           // [dW0 | db0] tiles a (b) block keeps in registers
}
struct LbGeom { int n_rt, n_chunk, nbi, nbb, cpb, n_cg, csplit; };
static LbGeom lb_geom(const iqlhip_ctx* c, int rows) {
  LbGeom g;
  g.n_rt = (rows + RT_ROWS - 1) / RT_ROWS;
  g.n_chunk = (rows + CHUNK_ROWS - 1) / CHUNK_ROWS;
  const int even_rt = (g.n_rt + 1) & ~1;
  // forward: 7 instances + the idle eighth share the chip: 32 blocks per instance, each walks ceil(n_rt / 32) row tiles
  g.nbi = std::min(even_rt, c->lb_nbi_force > 0 ? (c->lb_nbi_force + 1) & ~1 : 32);
  // backward: a net's blocks live on its two XCDs (64 CUs): (b) blocks of up to n_rt / nbb row tiles, (a) blocks of cpb chunks
  g.nbb = std::min(std::min(even_rt, 64), c->lb_nbb_force > 0 ? (c->lb_nbb_force + 1) & ~1 : 64);
  // up to 32 row tiles (1 024 rows): two blocks per tile, each half of the dH0 columns (iql_bwd_rows_kernel<true>) — a block
  // per tile would leave half the chip idle.  IQLHIP_LB_CSPLIT=0: the one-block-per-tile form at every size (diagnostic).
  g.csplit = (even_rt <= 32 && c->lb_nbb_force <= 0 && c->lb_csplit) ? 1 : 0;      // (nbb = even_rt then: one slab per tile)
  // GEMM blocks: 28 jobs per net and chunk group; about four chunk groups keep >= 400 blocks in flight
  g.cpb = c->lb_cpb_force > 0 ? std::min(c->lb_cpb_force, g.n_chunk) : std::max(1, std::min(8, g.n_chunk / 4));
  g.n_cg = (g.n_chunk + g.cpb - 1) / g.cpb;
  return g;
}
static LbArgs lb_args(const iqlhip_ctx* c, int rows) {
  const LbGeom g = lb_geom(c, rows);
  LbArgs a;
  a.pi_t = c->pi_t;
  a.pi_g = c->pi_t + (size_t)c->dims.max_batch * 32;
  a.pi_l = c->pi_t + (size_t)c->dims.max_batch * 64;
  a.n_rt = g.n_rt; a.n_chunk = g.n_chunk; a.nbi = g.nbi; a.nbb = g.nbb; a.cpb = g.cpb; a.n_cg = g.n_cg;
  const size_t MB = (size_t)c->dims.max_batch;
  a.dh1g = c->dh1g;
  a.dh0g = c->dh1g + 4 * MB * HID;
  a.dyg = c->dh1g + 8 * MB * HID;
  a.xbf = a.dyg + MB * 32;
  a.w1t = a.xbf + MB * LB_XLD;
  a.slab_x = c->slab_x;
  a.wimg = c->wimg;
  for (int n = 0; n < 4; ++n) { a.go_w0[n] = c->L.net[n].w0; a.go_b0[n] = c->L.net[n].b0; a.go_w1n[n] = c->L.net[n].seg_begin; }
  return a;
}

static StepParams make_step(const iqlhip_ctx* c, int rows, float inv_batch) {
  StepParams p;
  memset(&p, 0, sizeof p);
  p.stamps = c->stamps;
  const iqlhip_layout& L = c->L;
  const float* tb = c->target - L.target_src;
  const int S = c->dims.state_dim, A = c->dims.action_dim;
  // instances: V(s'), V(s), Qt1, Qt2, Q1, Q2, pi
  p.inst[0] = net_ptrs(L.net[IQLHIP_NET_V], c->params);  p.xoff[0] = S + A; p.slot[0] = -1;
  p.inst[1] = net_ptrs(L.net[IQLHIP_NET_V], c->params);  p.xoff[1] = 0;     p.slot[1] = 0;
  p.inst[2] = net_ptrs(L.net[IQLHIP_NET_Q1], tb);        p.xoff[2] = 0;     p.slot[2] = -1;
  p.inst[3] = net_ptrs(L.net[IQLHIP_NET_Q2], tb);        p.xoff[3] = 0;     p.slot[3] = -1;
  p.inst[4] = net_ptrs(L.net[IQLHIP_NET_Q1], c->params); p.xoff[4] = 0;     p.slot[4] = 1;
  p.inst[5] = net_ptrs(L.net[IQLHIP_NET_Q2], c->params); p.xoff[5] = 0;     p.slot[5] = 2;
  p.inst[6] = net_ptrs(L.net[IQLHIP_NET_PI], c->params); p.xoff[6] = 0;     p.slot[6] = 3;
  p.inst[7] = p.inst[6]; p.xoff[7] = 0; p.slot[7] = -1;
  if (c->precision == 1) {
    // bf16 path: the kernels read W1 from the bf16 shadows (same element offsets); the field carries that address
    const __bf16* wb = c->wsh;
    const __bf16* tbs = c->tsh - L.target_src;
    const int netof[7] = {IQLHIP_NET_V, IQLHIP_NET_V, IQLHIP_NET_Q1, IQLHIP_NET_Q2, IQLHIP_NET_Q1, IQLHIP_NET_Q2, IQLHIP_NET_PI};
    for (int i = 0; i < 7; ++i) p.inst[i].w1 = (const float*)(((i == 2 || i == 3) ? tbs : wb) + L.net[netof[i]].w1);
    p.inst[7].w1 = p.inst[6].w1;
  }
  for (int n = 0; n < 4; ++n) {
    p.net[n] = net_ptrs(L.net[n], c->params);
    if (c->precision == 1) p.net[n].w1 = (const float*)(c->wsh + L.net[n].w1);
    p.go[n].w1 = L.net[n].w1; p.go[n].b1 = L.net[n].b1; p.go[n].w2 = L.net[n].w2; p.go[n].b2 = L.net[n].b2;
    p.go[n].log_std = L.net[n].log_std;
  }
  p.log_std = (L.net[IQLHIP_NET_PI].log_std >= 0) ? c->params + L.net[IQLHIP_NET_PI].log_std : nullptr;
  p.hy = c->hyper;
  p.sc = c->sc;
  p.xb = c->xb;
  p.ld = (int)c->row_ld;
  p.rows = rows;
  p.S = S;
  p.A = A;
  p.policy = c->dims.policy;
  p.inv_batch = inv_batch;
  p.n_params = L.n_params;
  p.drop_bits = (c->drop_p > 0.f) ? c->drop_bits : nullptr;
  p.drop_scale = (c->drop_p > 0.f) ? 1.f / (1.f - c->drop_p) : 1.f;
  p.only_inst = -1;
  p.w0_lds_k = c->w0_lds_k;
  p.g_work = nullptr;
  return p;
}

static UpdParams make_upd(const iqlhip_ctx* c, const iqlhip_step_scalars* sc, int rows, const float* flat) {
  UpdParams u;
  u.L = c->L;
  u.sc = *sc;
  u.tau = c->hyper.tau;
  u.one_minus_tau = c->hyper.one_minus_tau;
  u.params = c->params; u.target = c->target; u.m = c->m; u.v = c->v;
  u.slab_a = c->sc.slab_a; u.slab_b = c->sc.slab_b;
  for (int n = 0; n < 4; ++n) u.slab_b_off[n] = c->sc.slab_b_off[n];
  u.flat_grads = flat;
  u.loss_parts = c->sc.loss_parts;
  u.losses = c->sc.losses;
  u.losses_mirror = nullptr;
  u.loss_ring = nullptr;
  u.ring_slot = 0;
  u.n_chunk = (rows + CHUNK_ROWS - 1) / CHUNK_ROWS;
  u.n_rt = (rows + RT_ROWS - 1) / RT_ROWS;
  u.n_loss = u.n_chunk;
  u.slab_x = nullptr;
  u.n_x = 0;
  if (use_lb(c, rows)) {      // the large-batch backward writes one slab per chunk group and one per row block
    const LbGeom gm = lb_geom(c, rows);
    u.n_chunk = gm.n_cg;
    u.slab_x = c->slab_x;
    u.n_x = gm.nbb;
    u.loss_parts = c->sc.loss_parts + 256;      // (entry 0 of each row: the total, summed by the GEMM launch's reduction job)
    u.n_loss = 1;
  }
  u.batch_rows = rows;
  u.sched = nullptr;
  u.sched_idx = 0;
  u.ring_hdr = nullptr;
  u.adv_hdr = nullptr; u.adv_k = 0; u.adv_rows = 0;
  u.done_flag = nullptr; u.done_val = 0;
  u.wsh = (c->precision == 1) ? c->wsh : nullptr;
  u.tsh = (c->precision == 1) ? c->tsh : nullptr;
  u.wimg = (c->precision == 1 && use_lb(c, rows)) ? c->wimg : nullptr;      // (read by the large-batch forward only)
  if (getenv("IQLHIP_LB_NOIMG")) u.wimg = nullptr;                          // timing experiment only: stale images
  u.n_peer = 0;
  u.peer_direct = 0;
  for (int r = 0; r < IQLHIP_MAX_WORLD; ++r) { u.peer_flat[r] = nullptr; u.peer_slab_b[r] = nullptr; u.peer_loss[r] = nullptr; }
  return u;
}

static size_t fwd_lds(const iqlhip_ctx* c, int n_blocks) { return (n_blocks <= c->n_cus) ? c->lds_fwd_solo : c->lds_fwd; }

static void launch_fwd_grid(const iqlhip_ctx* c, const StepParams& p, int nb, hipStream_t st) {
  const bool dma = c->w0_lds_k > W0_LDS_MAX_K;       // some instance stages wide layer-0 weights by LDS-DMA
  const size_t lds = fwd_lds(c, nb);
  const bool bf = c->precision == 1, multi = (p.spb_l2 & 3) > 0;
#define FWD_LAUNCH(BF, DMA, MU) hipLaunchKernelGGL((iql_fwd_kernel<BF, DMA, MU>), dim3(nb), dim3(256), lds, st, p)
#define FWD_LAUNCH_ONE(BF, DMA) hipLaunchKernelGGL((iql_fwd_kernel<BF, DMA, false, true>), dim3(nb), dim3(256), lds, st, p)
  if (p.only_inst >= 0) {        // policy inference (iqlhip_actor_forward): its own instantiation
    if (bf) { if (dma) FWD_LAUNCH_ONE(true, true); else FWD_LAUNCH_ONE(true, false); }
    else    { if (dma) FWD_LAUNCH_ONE(false, true); else FWD_LAUNCH_ONE(false, false); }
  } else if (multi) {
    if (bf) { if (dma) FWD_LAUNCH(true, true, true); else FWD_LAUNCH(true, false, true); }
    else    { if (dma) FWD_LAUNCH(false, true, true); else FWD_LAUNCH(false, false, true); }
  } else {
    if (bf) { if (dma) FWD_LAUNCH(true, true, false); else FWD_LAUNCH(true, false, false); }
    else    { if (dma) FWD_LAUNCH(false, true, false); else FWD_LAUNCH(false, false, false); }
  }
#undef FWD_LAUNCH
#undef FWD_LAUNCH_ONE
}

// Column slices per forward block (log2).  One block per (instance, row tile, slice) while that grid fits the chip in
// one round; beyond it every extra round costs a whole block time (prologue + layer 0 + one slice), so blocks take 2
// or 4 slices each — layer 0 and the prologue are then paid once per 2 / 4 slices (profiles/r02_slices_per_block.txt).
static int fwd_spb_l2(const iqlhip_ctx* c, int n_rt) {
  if (c->fwd_spb_force >= 0) return c->fwd_spb_force;
  // (two blocks sharing a CU each run ~1.8x slower — a second block per CU counts as no extra slot)
  if (8 * n_rt * NSPLIT <= c->n_cus) return 0;
  if (8 * n_rt * 2 <= c->n_cus) return 1;
  return 2;
}
static void launch_fwd(const iqlhip_ctx* c, const StepParams& p_in, hipStream_t st) {
  StepParams p = p_in;
  if (p.only_inst < 0 && use_lb(c, p.rows)) {
    const LbArgs a = lb_args(c, p.rows);
    const int kq = c->dims.state_dim + c->dims.action_dim;
    const dim3 grid(8 * a.nbi);
    // (the policy's tiles over its own and the idle blocks, iql_fwd_lb_kernel: from 3 tiles per block on the idle block takes 3 / 8 of them)
    const bool spread = a.n_rt > 2 * a.nbi && c->lb_pi_spread;
    if (spread) {
      if (kq <= 32) hipLaunchKernelGGL((iql_fwd_lb_kernel<1, true>), grid, dim3(256), 0, st, p, a);
      else if (kq <= 64) hipLaunchKernelGGL((iql_fwd_lb_kernel<2, true>), grid, dim3(256), 0, st, p, a);
      else hipLaunchKernelGGL((iql_fwd_lb_kernel<3, true>), grid, dim3(256), 0, st, p, a);
    } else {
      if (kq <= 32) hipLaunchKernelGGL((iql_fwd_lb_kernel<1, false>), grid, dim3(256), 0, st, p, a);
      else if (kq <= 64) hipLaunchKernelGGL((iql_fwd_lb_kernel<2, false>), grid, dim3(256), 0, st, p, a);
      else hipLaunchKernelGGL((iql_fwd_lb_kernel<3, false>), grid, dim3(256), 0, st, p, a);
    }
    return;
  }
  const int n_rt = (p.rows + RT_ROWS - 1) / RT_ROWS;
  const int l2 = fwd_spb_l2(c, n_rt);
  p.spb_l2 = l2;
  // (full-width blocks: each XCD of a net's pair takes the row tiles of one parity — iql_fwd_kernel's block map)
  const int nb = (l2 == 2) ? 8 * 2 * ((n_rt + 1) / 2) : 8 * n_rt * (NSPLIT >> l2);
  launch_fwd_grid(c, p, nb, st);
}
// Column slices per (b) block of the backward (log2): one while the whole grid — 4 nets x (32 dW1 tiles per 256-row
// chunk + 4 slices per row tile) — is at most two rounds of the chip (up to 512 rows: measured equal or better), else 4:
// the row tile's dY / dH1 tile is built once and the block's W1 fragments stream in under its MFMAs
// (profiles/r02_slices_per_block.txt).
static int bwd_spb_l2(const iqlhip_ctx* c, int n_chunk, int n_rt) {
  if (c->bwd_spb_force >= 0) return c->bwd_spb_force;
  return (4 * (32 * n_chunk + 4 * n_rt) <= 2 * c->n_cus) ? 0 : 2;
}
static void launch_bwd(const iqlhip_ctx* c, const StepParams& p_in, hipStream_t st) {
  StepParams p = p_in;
  if (use_lb(c, p.rows)) {
    const LbArgs a = lb_args(c, p.rows);
    const int kq = c->dims.state_dim + c->dims.action_dim;
    (void)kq;
    if (c->lb_bwd_part != 2) {
      if (lb_geom(c, p.rows).csplit) hipLaunchKernelGGL(iql_bwd_rows_kernel<true>, dim3(8 * a.nbb), dim3(256), c->lds_bwd_lb, st, p, a);
      else hipLaunchKernelGGL(iql_bwd_rows_kernel<false>, dim3(8 * (a.nbb / 2)), dim3(256), c->lds_bwd_lb, st, p, a);
    }
    if (c->lb_bwd_part != 1)
      hipLaunchKernelGGL(iql_bwd_gemm_kernel, dim3(8 * ((LB_NJOB * a.n_cg + 1 + 1) / 2)), dim3(256), 0, st, p, a);      // (+ 1: the reduction job)
    return;
  }
  const int n_rt = (p.rows + RT_ROWS - 1) / RT_ROWS;
  const int n_chunk = (p.rows + CHUNK_ROWS - 1) / CHUNK_ROWS;
  int l2 = bwd_spb_l2(c, n_chunk, n_rt);
  if (c->precision == 1 && l2 > 0) l2 = 2;      // bf16, large batches: a (b) block takes the whole row tile in one pass
  const int per_net = 32 * n_chunk + (4 >> l2) * n_rt;
  // multi-round launches: this many of the policy's dW1-tile blocks run at the ends of the scalar nets' XCD queues
  // (iql_bwd_kernel); share of its 32 n_chunk tiles tuned on obs 17 / act 6 and obs 39 / act 28 (profiles/r02_slices_per_block.txt)
  // Measured optimum of the share: 50 % at 28 action dims (1 024 rows 37.6 -> 30.8 us, bf16 32.7 -> 24.3; 2 048 rows
  // 76.0 -> 59.1), 30 % at 6-8 action dims (1 024 rows 28.7 -> 26.5); in between: linear in the action dims.
  int n_don = 0;
  if (l2 > 0) {
    const int pct = (c->bwd_donate_pct >= 0) ? c->bwd_donate_pct : std::max(30, std::min(50, 22 + c->dims.action_dim));
    n_don = std::min(32 * n_chunk, (32 * n_chunk * pct + 50) / 100);
  }
  p.spb_l2 = (l2 << 2) | (n_don << 8);
  const dim3 grid(8 * ((per_net + 1) / 2 + (n_don + 5) / 6));
  const bool full = (p.rows % CHUNK_ROWS) == 0;      // every tile of every block lies inside the batch: no clamps
  const bool bf = c->precision == 1, multi = l2 > 0;
  // the leading arguments are preloaded into SGPRs with each wave (build: -mllvm -amdgpu-kernarg-preload-count=14): what a
  // block needs to issue its first loads, see iql_bwd_kernel
  const float* q_heads = p.sc.heads; const float* q_xb = p.xb; const float* q_h1 = p.sc.h1; const float* q_h0 = p.sc.h0;
  const float* q_params = c->params;
  const unsigned q_dims = (unsigned)p.S | ((unsigned)p.A << 8) | ((unsigned)p.policy << 14);
  const unsigned q_ldB = (unsigned)p.ld | ((unsigned)p.rows << 10);
  const unsigned q_mbc = (unsigned)p.sc.max_batch | ((unsigned)n_chunk << 16);
  const unsigned q_rts = (unsigned)n_rt | ((unsigned)p.spb_l2 << 10);
#define BWD_LAUNCH(BF, FU, MU) hipLaunchKernelGGL((iql_bwd_kernel<BF, FU, MU>), grid, dim3(256), c->lds_bwd, st, q_heads, q_xb, q_h1, q_h0, q_params, q_dims, q_ldB, q_mbc, q_rts, p)
  if (multi) {
    if (bf) { if (full) BWD_LAUNCH(true, true, true); else BWD_LAUNCH(true, false, true); }
    else    { if (full) BWD_LAUNCH(false, true, true); else BWD_LAUNCH(false, false, true); }
  } else {
    if (bf) { if (full) BWD_LAUNCH(true, true, false); else BWD_LAUNCH(true, false, false); }
    else    { if (full) BWD_LAUNCH(false, true, false); else BWD_LAUNCH(false, false, false); }
  }
#undef BWD_LAUNCH
}
static unsigned drop_thresh(float p) {
  const double t = (double)p * 4294967296.0;
  return t >= 4294967295.0 ? 0xFFFFFFFFu : (unsigned)t;
}

static void launch_dropmask(const iqlhip_ctx* c, unsigned long long seed, unsigned long long step,
                            const unsigned long long* hdr, int k, hipStream_t st) {
  const int n_words = 2 * c->dims.max_batch * 8;
  hipLaunchKernelGGL(iql_dropmask_kernel, dim3((n_words + 255) / 256), dim3(256), 0, st, c->drop_bits, n_words,
                     drop_thresh(c->drop_p), seed, step, hdr, k);
}

static void launch_upd(const iqlhip_ctx* c, UpdParams u, hipStream_t st) {
  // grid: 8 x the 2 048-float windows of the longest net segment (iql_update_kernel's XCD-affine element map)
  long long seg_max = 0;
  for (int n = 0; n < 4; ++n) {
    const long long end = (n < 3) ? c->L.net[n + 1].seg_begin : c->L.n_params;
    seg_max = std::max(seg_max, end - c->L.net[n].seg_begin);
  }
  const int nb = 8 * (int)((seg_max + 2047) / 2048);
  const bool peer = u.n_peer > 0;
  // the leading arguments are preloaded into SGPRs with each wave (iql_update_kernel)
  const unsigned q_s0 = (unsigned)c->L.net[0].seg_begin, q_s1 = (unsigned)c->L.net[1].seg_begin, q_s2 = (unsigned)c->L.net[2].seg_begin,
                 q_s3 = (unsigned)c->L.net[3].seg_begin, q_end = (unsigned)c->L.net[3].seg_end;
  const unsigned q_flags = (!peer && !u.flat_grads && !u.slab_x && u.n_chunk == 1) ? UPD_EARLY_G : 0u;
#define UPD_LAUNCH(T, P) hipLaunchKernelGGL((iql_update_kernel<T, P>), dim3(nb), dim3(256), 0, st, u.params, u.m, u.v, u.slab_a, q_s0, q_s1, q_s2, q_s3, q_end, q_flags, u)
#define UPD_LAUNCH_LB(T, P) hipLaunchKernelGGL((iql_update_kernel<T, P, true>), dim3(nb), dim3(256), 0, st, u.params, u.m, u.v, u.slab_a, q_s0, q_s1, q_s2, q_s3, q_end, q_flags, u)
  if (u.slab_x) {      // large-batch bf16 step: the LB instantiations (gradient from the chunk-group slabs unless an exchange
                       // delivered it flat; the operand images of W0 / W1 written next to the bf16 shadows)
    if (u.sched) { if (peer) UPD_LAUNCH_LB(true, true); else UPD_LAUNCH_LB(true, false); }
    else         { if (peer) UPD_LAUNCH_LB(false, true); else UPD_LAUNCH_LB(false, false); }
  } else if (u.sched) { if (peer) UPD_LAUNCH(true, true); else UPD_LAUNCH(true, false); }
  else              { if (peer) UPD_LAUNCH(false, true); else UPD_LAUNCH(false, false); }
#undef UPD_LAUNCH
#undef UPD_LAUNCH_LB
}

// bf16 path: rebuild the bf16 shadows from the fp32 masters (the caller owns the masters and may have written them
// through its own tensors — load_state_dict, a broadcast — since the last update kernel kept the shadows current).
static void refresh_shadows(const iqlhip_ctx* c, hipStream_t st) {
  if (c->precision != 1) return;
  const long long n = std::max<long long>(c->L.n_params, c->L.n_target);
  hipLaunchKernelGGL(iql_shadow_refresh_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, c->params, c->target,
                     c->wsh, c->tsh, (long long)c->L.n_params, (long long)c->L.n_target, c->L, c->wimg);
}

static void launch_flatten(const iqlhip_ctx* c, const UpdParams& u, float* out, bool sys, hipStream_t st) {
  const int nb = (int)((c->L.n_params / 4 + 255) / 256);
  if (u.slab_x) {
    if (sys) hipLaunchKernelGGL((iql_grad_flatten_kernel<true, true>), dim3(nb), dim3(256), 0, st, u, out);
    else hipLaunchKernelGGL((iql_grad_flatten_kernel<false, true>), dim3(nb), dim3(256), 0, st, u, out);
  } else if (sys) hipLaunchKernelGGL(iql_grad_flatten_kernel<true>, dim3(nb), dim3(256), 0, st, u, out);
  else hipLaunchKernelGGL(iql_grad_flatten_kernel<false>, dim3(nb), dim3(256), 0, st, u, out);
}

// ---------------------------------------------------------------------------
// Gradient exchange between the ranks of a data-parallel group (include/iqlhip.h, "data-parallel gradient exchange").
static RcclApi g_rccl;

static int rccl_load() {
  if (g_rccl.lib) return IQLHIP_OK;
  void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);      // the copy the process already has (PyTorch-ROCm's)
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW);
  if (!h) return fail(IQLHIP_EHIP, "librccl.so.1 not found: %s", dlerror());
  RcclApi a;
  a.lib = h;
  a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  a.CommInitRank = (decltype(a.CommInitRank))dlsym(h, "ncclCommInitRank");
  a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
  a.AllReduce = (decltype(a.AllReduce))dlsym(h, "ncclAllReduce");
  a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
  if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce)
    return fail(IQLHIP_EHIP, "librccl.so.1 lacks an expected symbol");
  g_rccl = a;
  return IQLHIP_OK;
}
#define NCCLCHK(expr)                                                                              \
  do {                                                                                             \
    int r_ = (expr);                                                                               \
    if (r_ != 0)                                                                                   \
      return fail(IQLHIP_EHIP, "%s failed: %s", #expr, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "rccl error"); \
  } while (0)
enum { RCCL_FLOAT32 = 7, RCCL_SUM = 0 };     // ncclFloat32, ncclSum (rccl.h)

extern "C" int iqlhip_comm_unique_id(void* id_out) {
  if (!id_out) return fail(IQLHIP_EINVAL, "NULL argument");
  int rc = rccl_load();
  if (rc) return rc;
  NCCLCHK(g_rccl.GetUniqueId(id_out));
  return IQLHIP_OK;
}

extern "C" int iqlhip_allreduce_init(iqlhip_ctx* c, const void* unique_id, int rank, int world) {
  if (!c || !unique_id) return fail(IQLHIP_EINVAL, "NULL argument");
  if (world < 1 || rank < 0 || rank >= world) return fail(IQLHIP_EINVAL, "rank %d outside world %d", rank, world);
  if (c->p2p_attached && (world != c->world || rank != c->rank))
    return fail(IQLHIP_EINVAL, "rank/world differ from the attached P2P exchange");
  int rc = rccl_load();
  if (rc) return rc;
  DevGuard guard(c->device);
  if (c->nccl_comm) { (void)g_rccl.CommDestroy(c->nccl_comm); c->nccl_comm = nullptr; }
  RcclId id;
  memcpy(&id, unique_id, sizeof id);
  NCCLCHK(g_rccl.CommInitRank(&c->nccl_comm, world, id, rank));
  c->rank = rank;
  c->world = world;
  c->xch_mode = IQLHIP_XCH_RCCL;
  drop_graph(c);
  return IQLHIP_OK;
}

extern "C" int iqlhip_p2p_export(iqlhip_ctx* c, void* handle_out, int rank, int world) {
  if (!c || !handle_out) return fail(IQLHIP_EINVAL, "NULL argument");
  if (world < 1 || world > IQLHIP_MAX_WORLD || rank < 0 || rank >= world)
    return fail(IQLHIP_EINVAL, "rank %d / world %d outside [0, %d]", rank, world, IQLHIP_MAX_WORLD);
  if (c->nccl_comm && (world != c->world || rank != c->rank))
    return fail(IQLHIP_EINVAL, "rank/world differ from the RCCL communicator");
  if (getenv("IQLHIP_P2P_DISABLE"))     // diagnostic: lets the callers' fallback paths be exercised on any machine
    return fail(IQLHIP_EUNSUPPORTED, "the peer-to-peer exchange is disabled (IQLHIP_P2P_DISABLE)");
  DevGuard guard(c->device);
  if (!c->xblk) {
    const size_t flags_b = 4096;                                     // IQLHIP_MAX_WORLD x 128-B flag lines, padded
    const size_t flat_b = (size_t)up((c->L.n_params + 4) * (int64_t)sizeof(float), 4096);
    size_t sb = 0;                                                   // w0 / b0 partial slabs of <= 8 row tiles (<= 256 rows)
    for (int n = 0; n < 4; ++n) {
      c->xslab_b_off[n] = (long long)sb;
      sb += (size_t)8 * ((size_t)HID * c->L.net[n].k_in + HID);
    }
    const size_t slabb_b = (size_t)up((int64_t)(sb * sizeof(float)), 4096);
    const size_t loss_b = 4096;
    const size_t buf_b = flat_b + slabb_b + loss_b;
    for (int k = 0; k < 2; ++k) {
      c->xflat_off[k] = flags_b + (size_t)k * buf_b;
      c->xslabb_off[k] = c->xflat_off[k] + flat_b;
      c->xloss_off[k] = c->xslabb_off[k] + slabb_b;
    }
    c->xblk_bytes = flags_b + 2 * buf_b;
    // Coarse-grained device memory by default: the backward's plain stores into this block are complete AND written
    // back to the device's memory at the kernel boundary in front of the flag kernel (dirty L2 lines leave at a
    // boundary), and every peer reads them with system-scope loads that bypass its own caches — the data a peer can
    // see is in this device's HBM before the flag that announces it is stored.  IQLHIP_P2P_FINEGRAINED=1 allocates the
    // block fine-grained instead (no reliance on the boundary write-back; slower stores) where the runtime can export
    // such memory through hipIpc; the replicas-equal check of the callers is the gate either way.
    hipError_t ea = hipErrorUnknown;
    if (getenv("IQLHIP_P2P_FINEGRAINED")) ea = hipExtMallocWithFlags((void**)&c->xblk, c->xblk_bytes, hipDeviceMallocFinegrained);
    if (ea != hipSuccess) { (void)hipGetLastError(); HIPCHK(hipMalloc((void**)&c->xblk, c->xblk_bytes)); }
    HIPCHK(hipMemset(c->xblk, 0, c->xblk_bytes));
    HIPCHK(hipDeviceSynchronize());
  } else {
    // a second attach: every rank clears its OWN flag lines here, before the handles are exchanged (the callers'
    // all-gather is the barrier), so no stale flag of an earlier group satisfies the first waits of the new one
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemset(c->xblk, 0, 4096));
    HIPCHK(hipDeviceSynchronize());
  }
  hipIpcMemHandle_t h;
  static_assert(sizeof(hipIpcMemHandle_t) == IQLHIP_IPC_HANDLE_BYTES, "hipIpcMemHandle_t size");
  HIPCHK(hipIpcGetMemHandle(&h, c->xblk));
  memcpy(handle_out, &h, sizeof h);
  c->rank = rank;
  c->world = world;
  return IQLHIP_OK;
}

static void p2p_close(iqlhip_ctx* c) {
  for (int r = 0; r < IQLHIP_MAX_WORLD; ++r) {
    if (c->peer_blk[r] && c->peer_blk[r] != c->xblk) (void)hipIpcCloseMemHandle(c->peer_blk[r]);
    c->peer_blk[r] = nullptr;
  }
  c->p2p_attached = false;
}

extern "C" int iqlhip_p2p_attach(iqlhip_ctx* c, const void* handles, int timeout_ms) {
  if (!c || !handles) return fail(IQLHIP_EINVAL, "NULL argument");
  if (!c->xblk) return fail(IQLHIP_EINVAL, "iqlhip_p2p_export has not been called");
  DevGuard guard(c->device);
  p2p_close(c);
  for (int r = 0; r < c->world; ++r) {
    if (r == c->rank) { c->peer_blk[r] = c->xblk; continue; }
    hipIpcMemHandle_t h;
    memcpy(&h, (const char*)handles + (size_t)r * IQLHIP_IPC_HANDLE_BYTES, sizeof h);
    void* ptr = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      p2p_close(c);
      return fail(IQLHIP_EHIP, "hipIpcOpenMemHandle(rank %d): %s", r, hipGetErrorString(e));
    }
    c->peer_blk[r] = (char*)ptr;
  }
  c->p2p_attached = true;
  c->xtimeout_ticks = (unsigned long long)(timeout_ms > 0 ? timeout_ms : 5000) * 100000ull;   // 100 MHz wall clock
  c->xstep = 0;
  HIPCHK(hipMemset(c->xstatus, 0, 2 * sizeof(unsigned long long)));
  c->xch_mode = IQLHIP_XCH_P2P;
  drop_graph(c);
  return IQLHIP_OK;
}

extern "C" int iqlhip_xch_select(iqlhip_ctx* c, int mode) {
  if (!c) return fail(IQLHIP_EINVAL, "NULL ctx");
  if (mode == IQLHIP_XCH_RCCL && !c->nccl_comm) return fail(IQLHIP_EINVAL, "no RCCL communicator (iqlhip_allreduce_init)");
  if (mode == IQLHIP_XCH_P2P && !c->p2p_attached) return fail(IQLHIP_EINVAL, "no P2P exchange (iqlhip_p2p_attach)");
  if (mode != IQLHIP_XCH_NONE && mode != IQLHIP_XCH_RCCL && mode != IQLHIP_XCH_P2P) return fail(IQLHIP_EINVAL, "unknown exchange mode %d", mode);
  c->xch_mode = mode;
  return IQLHIP_OK;
}

extern "C" int iqlhip_xch_status(iqlhip_ctx* c, int64_t status[3], void* stream) {
  if (!c || !status) return fail(IQLHIP_EINVAL, "NULL argument");
  DevGuard guard(c->device);
  unsigned long long h[2] = {0, 0};
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  HIPCHK(hipMemcpy(h, c->xstatus, sizeof h, hipMemcpyDeviceToHost));
  status[0] = c->xch_mode;
  status[1] = (int64_t)h[0];
  status[2] = (int64_t)c->xstep;
  return IQLHIP_OK;
}

// Forget a recorded wait timeout (after the caller has re-synchronised the replicas and chosen another exchange).
extern "C" int iqlhip_xch_clear_status(iqlhip_ctx* c, void* stream) {
  if (!c) return fail(IQLHIP_EINVAL, "NULL ctx");
  DevGuard guard(c->device);
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  HIPCHK(hipMemset(c->xstatus, 0, 2 * sizeof(unsigned long long)));
  HIPCHK(hipDeviceSynchronize());
  drop_graph(c);
  return IQLHIP_OK;
}

extern "C" int iqlhip_xch_shutdown(iqlhip_ctx* c) {
  if (!c) return IQLHIP_OK;
  DevGuard guard(c->device);
  (void)hipDeviceSynchronize();
  drop_graph(c);
  if (c->nccl_comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->nccl_comm);
  c->nccl_comm = nullptr;
  p2p_close(c);
  if (c->xblk) (void)hipFree(c->xblk);
  c->xblk = nullptr;
  c->xch_mode = IQLHIP_XCH_NONE;
  c->world = 1;
  c->rank = 0;
  return IQLHIP_OK;
}

// A P2P wait that timed out is sticky and silent on the device (later waits return at once and the update kernels sum
// whatever the peers' buffers hold): the entry points that synchronise anyway read the status word along with the
// losses and turn it into an error, so a run cannot keep training on unsynchronised gradients unnoticed.
static int xch_poisoned(iqlhip_ctx* c, const unsigned long long* status_host) {
  if (status_host[0] == 0ull) return IQLHIP_OK;
  return fail(IQLHIP_EEXCHANGE, "the peer-to-peer gradient exchange timed out at exchange step %llu: a peer rank did not arrive; "
              "the replicas are no longer synchronised", status_host[0]);
}

static XchParams make_xch(const iqlhip_ctx* c, bool from_hdr) {
  XchParams x;
  memset(&x, 0, sizeof x);
  for (int r = 0; r < IQLHIP_MAX_WORLD; ++r)
    x.peer_flags[r] = (unsigned long long*)c->peer_blk[std::min(r, c->world - 1)];
  x.status = c->xstatus;
  x.hdr = from_hdr ? c->hdr : nullptr;
  x.xstep = c->xstep;
  x.timeout_ticks = c->xtimeout_ticks;
  x.rank = c->rank;
  x.world = c->world;
  return x;
}

// One step's launches after the batch has been staged: forward, backward and — by exchange mode — the update, or
// flatten + all-reduce + update, or flatten + flag handshake + the update that reads every rank's buffer.
// `k` = position inside the chunk (selects the P2P buffer together with `parity`, and the flag value hdr[XSTEP]+k+1).
static int enqueue_step(iqlhip_ctx* c, const StepParams& p_in, UpdParams u, int mode, int parity, int k, bool from_hdr,
                        hipStream_t st, hipEvent_t* ev) {
  StepParams p = p_in;
  const int buf = (parity + k) & 1;
  // P2P with batches of <= 256 rows: no flatten kernel.  The backward writes its chunk slab (= the w1 / b1 / w2 / b2 /
  // log_std gradients themselves), its <= 8 row-tile slabs of w0 / b0 partials and its loss sums straight into this
  // rank's exchange block; they are complete and written back at the kernel boundary in front of the flag kernel, and
  // every rank's update kernel reads all ranks' blocks.
  const bool direct = (mode == IQLHIP_XCH_P2P) && (p.rows <= CHUNK_ROWS);
  if (direct) {
    p.sc.slab_a = (float*)(c->xblk + c->xflat_off[buf]);
    p.sc.slab_b = (float*)(c->xblk + c->xslabb_off[buf]);
    for (int n = 0; n < 4; ++n) p.sc.slab_b_off[n] = c->xslab_b_off[n];
    p.sc.loss_parts = (float*)(c->xblk + c->xloss_off[buf]);
  }
  launch_fwd(c, p, st);
  if (ev) HIPCHK(hipEventRecord(ev[1], st));
  launch_bwd(c, p, st);
  if (ev) HIPCHK(hipEventRecord(ev[2], st));
  if (mode == IQLHIP_XCH_RCCL) {
    launch_flatten(c, u, c->xflat, false, st);
    NCCLCHK(g_rccl.AllReduce(c->xflat, c->xflat, (size_t)c->L.n_params + 4, RCCL_FLOAT32, RCCL_SUM, c->nccl_comm, st));
    u.flat_grads = c->xflat;
  } else if (mode == IQLHIP_XCH_P2P) {
    if (!direct) launch_flatten(c, u, (float*)(c->xblk + c->xflat_off[buf]), true, st);
    if (c->world > 1) hipLaunchKernelGGL(iql_xch_signal_wait_kernel, dim3(1), dim3(64), 0, st, make_xch(c, from_hdr), k);
    for (int r = 0; r < IQLHIP_MAX_WORLD; ++r) {
      const char* blk = c->peer_blk[std::min(r, c->world - 1)];
      u.peer_flat[r] = (const float*)(blk + c->xflat_off[buf]);
      u.peer_slab_b[r] = (const float*)(blk + c->xslabb_off[buf]);
      u.peer_loss[r] = (const float*)(blk + c->xloss_off[buf]);
    }
    u.n_peer = c->world;
    u.peer_direct = direct ? 1 : 0;
    if (direct) for (int n = 0; n < 4; ++n) u.slab_b_off[n] = c->xslab_b_off[n];
  }
  launch_upd(c, u, st);
  if (ev) HIPCHK(hipEventRecord(ev[3], st));
  return IQLHIP_OK;
}

static int ensure_events(iqlhip_ctx* c, int n) {
  while ((int)c->ev.size() < n) {
    hipEvent_t e;
    HIPCHK(hipEventCreate(&e));
    c->ev.push_back(e);
  }
  return IQLHIP_OK;
}

static int harvest_timing(iqlhip_ctx* c) {
  for (int i = 0; i + 3 < c->ev_used; i += 4) {
    HIPCHK(hipEventSynchronize(c->ev[i + 3]));
    float a = 0, b = 0, d = 0;
    HIPCHK(hipEventElapsedTime(&a, c->ev[i], c->ev[i + 1]));
    HIPCHK(hipEventElapsedTime(&b, c->ev[i + 1], c->ev[i + 2]));
    HIPCHK(hipEventElapsedTime(&d, c->ev[i + 2], c->ev[i + 3]));
    c->t_acc[0] += a * 1e3f; c->t_acc[1] += b * 1e3f; c->t_acc[2] += d * 1e3f; c->t_acc[3] += (a + b + d) * 1e3f;
    c->t_n += 1;
  }
  c->ev_used = 0;
  return IQLHIP_OK;
}

extern "C" int iqlhip_set_timing(iqlhip_ctx* c, int enabled) {
  if (!c) return fail(IQLHIP_EINVAL, "NULL ctx");
  c->timing = enabled != 0;
  c->ev_used = 0;
  c->t_n = 0;
  for (float& t : c->t_acc) t = 0.f;
  return IQLHIP_OK;
}

extern "C" int iqlhip_get_timing(iqlhip_ctx* c, float out_us[4]) {
  if (!c || !out_us) return fail(IQLHIP_EINVAL, "NULL argument");
  int rc = harvest_timing(c);
  if (rc) return rc;
  for (int k = 0; k < 4; ++k) out_us[k] = c->t_n ? c->t_acc[k] / c->t_n : 0.f;
  return IQLHIP_OK;
}

// Wait for the completion word of a synchronous entry point: spin on host-mapped memory (no HIP call: a stream
// synchronise was measured at 12-17 us of host time AFTER the GPU had finished, profiles/r03_sync_cost.txt); bounded —
// after 2 s the stream is synchronised the ordinary way, so a lost store cannot hang the caller.
static int wait_done(iqlhip_ctx* c, unsigned long long val, hipStream_t st) {
  // (acquire loads: the callers read the pinned loss words right behind this — those loads must not move in front of
  //  the flag's)
  const unsigned long long* f = c->done_pin;
  if (__atomic_load_n(f, __ATOMIC_ACQUIRE) == val) return IQLHIP_OK;
  const double t0 = now_us();
  for (;;) {
    for (int i = 0; i < 256; ++i) if (__atomic_load_n(f, __ATOMIC_ACQUIRE) == val) return IQLHIP_OK;
    if (now_us() - t0 > 2e6) break;
  }
  HIPCHK(hipStreamSynchronize(st));
  if (__atomic_load_n(f, __ATOMIC_ACQUIRE) != val) return fail(IQLHIP_EHIP, "the step's completion word was not written");
  return IQLHIP_OK;
}

static int step_impl(iqlhip_ctx* c, const iqlhip_batch* b, const iqlhip_step_scalars* sc, float* out_sync, void* stream,
                     bool defer_wait = false);

extern "C" int iqlhip_step(iqlhip_ctx* c, const iqlhip_batch* b, const iqlhip_step_scalars* sc, void* stream) {
  return step_impl(c, b, sc, nullptr, stream);
}

// ImplicitQLearning.train(batch) WITH its host synchronisation (the three .item() calls of iql.py:491,509,535 as one):
// iqlhip_step + the losses, which land in host-mapped pinned words followed by a completion word the host spins on.
// Returns once the losses are there; everything else the step does stays ordered by the stream as usual.
extern "C" int iqlhip_step_sync(iqlhip_ctx* c, const iqlhip_batch* b, const iqlhip_step_scalars* sc, float out[3], void* stream) {
  if (!out) return fail(IQLHIP_EINVAL, "NULL argument");
  return step_impl(c, b, sc, out, stream);
}

// The two halves of iqlhip_step_sync for a host that has something to do while the GPU runs the step (the Python shim
// computes the NEXT step's float64 Adam / cosine scalars there): begin launches, wait returns the losses.
extern "C" int iqlhip_step_begin(iqlhip_ctx* c, const iqlhip_batch* b, const iqlhip_step_scalars* sc, void* stream) {
  float dummy[3];
  return step_impl(c, b, sc, dummy, stream, /*defer_wait=*/true);
}
extern "C" int iqlhip_step_wait(iqlhip_ctx* c, float out[3], void* stream) {
  if (!c || !out) return fail(IQLHIP_EINVAL, "NULL argument");
  hipStream_t st = (hipStream_t)stream;
  if (c->xch_mode == IQLHIP_XCH_P2P && c->world > 1) {
    HIPCHK(hipMemcpyAsync(c->xstatus_host, c->xstatus, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    int rc = xch_poisoned(c, c->xstatus_host);
    if (rc) return rc;
  } else {
    int rc = wait_done(c, c->done_seq, st);
    if (rc) return rc;
  }
  out[0] = c->on_loss_pin[0]; out[1] = c->on_loss_pin[1]; out[2] = c->on_loss_pin[2];
  return IQLHIP_OK;
}

static int step_impl(iqlhip_ctx* c, const iqlhip_batch* b, const iqlhip_step_scalars* sc, float* out_sync, void* stream,
                     bool defer_wait) {
  if (!c || !sc) return fail(IQLHIP_EINVAL, "NULL argument");
  int rc = check_batch(c, b);
  if (rc) return rc;
  DevGuard guard(c->device);
  hipStream_t st = (hipStream_t)stream;
  c->cont.valid = false;                 // (the staging buffers / keep-bits a following train_steps call might continue from)
  const float* xb_cur = nullptr;
  rc = stage_batch(c, b, st, &xb_cur);
  if (rc) return rc;
  refresh_shadows(c, st);
  if (c->drop_p > 0.f && !c->drop_inject) launch_dropmask(c, c->drop_seed, c->drop_step++, nullptr, 0, st);
  StepParams p = make_step(c, b->rows, sc->inv_batch);
  p.xb = xb_cur;
  UpdParams u = make_upd(c, sc, b->rows, nullptr);
  const bool p2p_x = c->xch_mode == IQLHIP_XCH_P2P && c->world > 1;
  unsigned long long done_val = 0;
  if (out_sync) {
    u.losses_mirror = c->on_loss_pin;
    u.done_flag = c->done_pin;
    u.done_val = done_val = ++c->done_seq;
  }
  hipEvent_t* ev = nullptr;
  if (c->timing) {
    if (c->ev_used + 4 > 4096) { rc = harvest_timing(c); if (rc) return rc; }
    rc = ensure_events(c, c->ev_used + 4);
    if (rc) return rc;
    ev = &c->ev[c->ev_used];
    c->ev_used += 4;
    HIPCHK(hipEventRecord(ev[0], st));
  }
  rc = enqueue_step(c, p, u, c->xch_mode, (int)(c->xstep & 1ull), 0, /*from_hdr=*/false, st, ev);
  if (rc) return rc;
  if (c->xch_mode != IQLHIP_XCH_NONE) c->xstep += 1;
  HIPCHK(hipGetLastError());
  if (out_sync && !defer_wait) {
    if (p2p_x) {       // (the exchange's status word has to come back too: the ordinary synchronise)
      HIPCHK(hipMemcpyAsync(c->xstatus_host, c->xstatus, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      rc = xch_poisoned(c, c->xstatus_host);
      if (rc) return rc;
    } else {
      rc = wait_done(c, done_val, st);
      if (rc) return rc;
    }
    out_sync[0] = c->on_loss_pin[0]; out_sync[1] = c->on_loss_pin[1]; out_sync[2] = c->on_loss_pin[2];
  }
  return IQLHIP_OK;
}

static int actor_forward_impl(iqlhip_ctx* c, const float* states_dev, int64_t ld_s, int32_t rows, const float* noise_dev,
                              int64_t ld_noise, uint64_t rng_seed, uint64_t rng_call, float max_action,
                              float* actions_dev, int64_t ld_a, void* stream, unsigned long long* done_flag = nullptr,
                              unsigned long long done_val = 0);

// One iteration of the online loop's device work (algorithms/finetune/iql.py:741-773: add_transition -> sample ->
// train) in ONE call and four launches: ring write + gather straight from pinned host words, forward, backward, update
// with the losses landing in pinned host words; then one stream synchronisation.
extern "C" int iqlhip_online_step(iqlhip_ctx* c, float* rows_dev, int64_t ld, int64_t capacity, int64_t pointer,
                                  const float* row_host, const int64_t* idx_host, int32_t n,
                                  const iqlhip_step_scalars* sc, float out[3], const float* act_state_host,
                                  float max_action, uint64_t act_seed, float* act_out_host, void* stream) {
  if (!c || !rows_dev || !row_host || !idx_host || !sc || !out) return fail(IQLHIP_EINVAL, "NULL argument");
  if (act_state_host && !act_out_host) return fail(IQLHIP_EINVAL, "act_state_host without act_out_host");
  if (!c->params) return fail(IQLHIP_ENOTBOUND, "iqlhip_bind has not been called");
  if (ld != c->row_ld) return fail(IQLHIP_EINVAL, "row stride must be iqlhip_row_stride(S,A)=%lld", (long long)c->row_ld);
  if (((uintptr_t)rows_dev) & 15) return fail(IQLHIP_EINVAL, "packed rows must be 16-byte aligned");
  if (n < 1 || n > c->dims.max_batch) return fail(IQLHIP_EINVAL, "batch rows %d outside [1, max_batch=%d]", n, c->dims.max_batch);
  if (capacity < 1 || pointer < 0 || pointer >= capacity) return fail(IQLHIP_EINVAL, "ring pointer outside the buffer");
  {                                       // (the reference's torch indexing raises on such an index; a gather would fault)
    int rc_i = check_host_indices(idx_host, n, capacity);
    if (rc_i) return rc_i;
  }
  DevGuard guard(c->device);
  hipStream_t st = (hipStream_t)stream;
  c->cont.valid = false;
  memcpy(c->on_row_pin, row_host, (size_t)ld * sizeof(float));
  memcpy(c->on_idx_pin, idx_host, (size_t)n * sizeof(long long));
  const int total = n * (int)(ld / 4);
  hipLaunchKernelGGL(iql_online_gather_kernel, dim3((total + 255) / 256), dim3(256), 0, st, rows_dev, (long long)ld,
                     (long long)pointer, (const float*)c->on_row_pin, (const long long*)c->on_idx_pin, c->xb, n);
  refresh_shadows(c, st);
  if (c->drop_p > 0.f && !c->drop_inject) launch_dropmask(c, c->drop_seed, c->drop_step++, nullptr, 0, st);
  StepParams p = make_step(c, n, sc->inv_batch);
  UpdParams u = make_upd(c, sc, n, nullptr);
  u.losses_mirror = c->on_loss_pin;
  const unsigned long long done_val = ++c->done_seq;
  if (!act_state_host) { u.done_flag = c->done_pin; u.done_val = done_val; }      // (else the follow-up act() signals)
  int rc = enqueue_step(c, p, u, c->xch_mode, (int)(c->xstep & 1ull), 0, /*from_hdr=*/false, st, nullptr);
  if (rc) return rc;
  if (c->xch_mode != IQLHIP_XCH_NONE) c->xstep += 1;
  const int S = c->dims.state_dim, A = c->dims.action_dim;
  if (act_state_host) {
    // the NEXT iteration's actor.act(next_state) with the just-updated policy, in the same stream and under the
    // same synchronisation: state and action travel through host-mapped pinned words
    memcpy(c->on_act_pin, act_state_host, (size_t)S * sizeof(float));
    float* a_out = c->on_act_pin + IQLHIP_MAX_INPUT;
    rc = (act_seed != 0 && c->dims.policy == IQLHIP_POLICY_GAUSSIAN)
             ? actor_forward_impl(c, c->on_act_pin, S, 1, nullptr, 0, act_seed, c->act_calls++, max_action, a_out, A, stream, c->done_pin, done_val)
             : actor_forward_impl(c, c->on_act_pin, S, 1, nullptr, 0, 0, 0, max_action, a_out, A, stream, c->done_pin, done_val);
    if (rc) return rc;
  }
  // (a synchronous call: the pinned staging words are free again on return)
  const bool p2p_x = c->xch_mode == IQLHIP_XCH_P2P && c->world > 1;
  if (p2p_x) {
    HIPCHK(hipMemcpyAsync(c->xstatus_host, c->xstatus, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    int rx = xch_poisoned(c, c->xstatus_host);
    if (rx) return rx;
  } else {
    // the completion word comes from the last kernel of the call; the whole call's kernels precede it in the stream
    // EXCEPT that the update's flag is stored by its first block — the ring write and the gather (first launch) are
    // long done by then, and the pinned words read below were written before the flag (release)
    rc = wait_done(c, done_val, st);
    if (rc) return rc;
  }
  out[0] = c->on_loss_pin[0]; out[1] = c->on_loss_pin[1]; out[2] = c->on_loss_pin[2];
  if (act_state_host) memcpy(act_out_host, c->on_act_pin + IQLHIP_MAX_INPUT, (size_t)A * sizeof(float));
  HIPCHK(hipGetLastError());
  return IQLHIP_OK;
}

extern "C" int iqlhip_forward_backward(iqlhip_ctx* c, const iqlhip_batch* b, const iqlhip_step_scalars* sc,
                                       float* grads_dev, void* stream) {
  if (!c || !sc || !grads_dev) return fail(IQLHIP_EINVAL, "NULL argument");
  int rc = check_batch(c, b);
  if (rc) return rc;
  DevGuard guard(c->device);
  hipStream_t st = (hipStream_t)stream;
  c->cont.valid = false;                 // (the staging buffers / keep-bits a following train_steps call might continue from)
  const float* xb_cur = nullptr;
  rc = stage_batch(c, b, st, &xb_cur);
  if (rc) return rc;
  refresh_shadows(c, st);
  if (c->drop_p > 0.f && !c->drop_inject) launch_dropmask(c, c->drop_seed, c->drop_step++, nullptr, 0, st);
  StepParams p = make_step(c, b->rows, sc->inv_batch);
  p.xb = xb_cur;
  UpdParams u = make_upd(c, sc, b->rows, nullptr);
  launch_fwd(c, p, st);
  launch_bwd(c, p, st);
  launch_flatten(c, u, grads_dev, false, st);
  HIPCHK(hipGetLastError());
  return IQLHIP_OK;
}

extern "C" int iqlhip_apply_update(iqlhip_ctx* c, const float* grads_dev, const iqlhip_step_scalars* sc, void* stream) {
  if (!c || !sc || !grads_dev) return fail(IQLHIP_EINVAL, "NULL argument");
  if (!c->params) return fail(IQLHIP_ENOTBOUND, "iqlhip_bind has not been called");
  DevGuard guard(c->device);
  UpdParams u = make_upd(c, sc, 1, grads_dev);
  launch_upd(c, u, (hipStream_t)stream);
  HIPCHK(hipGetLastError());
  return IQLHIP_OK;
}

extern "C" int iqlhip_read_losses(iqlhip_ctx* c, float out[3], void* stream) {
  if (!c || !out) return fail(IQLHIP_EINVAL, "NULL argument");
  float* h = c->losses_host;
  const bool p2p = c->xch_mode == IQLHIP_XCH_P2P && c->world > 1;
  HIPCHK(hipMemcpyAsync(h, c->sc.losses, 4 * sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream));
  if (p2p) HIPCHK(hipMemcpyAsync(c->xstatus_host, c->xstatus, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  out[0] = h[0]; out[1] = h[1]; out[2] = h[2];
  return p2p ? xch_poisoned(c, c->xstatus_host) : IQLHIP_OK;
}

extern "C" int iqlhip_read_loss_ring(iqlhip_ctx* c, float* out, int32_t n_steps, void* stream) {
  if (!c || !out) return fail(IQLHIP_EINVAL, "NULL argument");
  if (n_steps < 1 || n_steps > c->ring_cap) return fail(IQLHIP_EINVAL, "n_steps outside [1,%d]", c->ring_cap);
  const bool p2p = c->xch_mode == IQLHIP_XCH_P2P && c->world > 1;
  const float* h = c->loss_ring;
  if (p2p) HIPCHK(hipMemcpyAsync(c->xstatus_host, c->xstatus, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  for (int k = 0; k < n_steps; ++k)
    for (int j = 0; j < 3; ++j) out[3 * k + j] = h[4 * (size_t)k + j];
  return p2p ? xch_poisoned(c, c->xstatus_host) : IQLHIP_OK;
}

// ---------------------------------------------------------------------------
extern "C" int iqlhip_draw_indices(int64_t* idx_dev, int64_t n, int64_t size, uint64_t seed, uint64_t offset, void* stream) {
  if (!idx_dev || n < 1 || size < 1) return fail(IQLHIP_EINVAL, "bad argument");
  const int nb = (int)std::min<int64_t>((n / 2 + 255) / 256 + 1, 1024);
  hipLaunchKernelGGL(iql_draw_indices_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, (long long*)idx_dev,
                     (long long)n, (long long)size, (unsigned long long)seed, (unsigned long long)offset,
                     (const unsigned long long*)nullptr);
  HIPCHK(hipGetLastError());
  return IQLHIP_OK;
}

// ---------------------------------------------------------------------------
// The multi-step driver.  A call of n steps = ONE directly launched set-up kernel (header words, the call's scalar
// table, step 0's rows unless the previous call left them staged) + its first 2 / 4 steps launched directly + replays
// of the fixed chunk graphs.  Everything a
// replay needs to differ in lives in device words that the set-up kernel writes once and each chunk's last update
// kernel advances, so the chunks of a call follow each other with no host-side launch in between; the rows, scalars and
// keep-bits of step k + 1 are staged by the idle eighth of step k's forward grid (IdleWork).
static void fill_idle_work(const iqlhip_ctx* c, IdleWork* w, const float* rows_dev, int B, int K) {
  const int MB = c->dims.max_batch;
  for (int k = 0; k < K; ++k) {
    IdleWork& i = w[k];
    memset(&i, 0, sizeof i);
    i.rows = rows_dev;
    i.ld = c->row_ld;
    i.xb_dst = (k & 1) ? c->xb : c->xb2;              // step k reads buffer k & 1, step k + 1 the other one
    i.hdr = c->hdr;
    i.sched_call = c->sched_call;
    i.sched_dst = c->sched_cur + k;
    i.drop_dst = (c->drop_p > 0.f) ? c->drop_bits + (size_t)((k + 1) & 1) * 2 * MB * 8 : nullptr;
    i.drop_words = 2 * MB * 8;
    i.drop_thresh = drop_thresh(c->drop_p);
    i.n = B;
    i.k = k;
  }
}

static int enqueue_chunk(iqlhip_ctx* c, hipStream_t st, int B, int K, float inv_batch, int mode, int parity,
                         const IdleWork* work_dev) {
  const int MB = c->dims.max_batch;
  iqlhip_step_scalars sc0;
  memset(&sc0, 0, sizeof sc0);
  sc0.inv_batch = inv_batch;
  for (int k = 0; k < K; ++k) {
    StepParams p = make_step(c, B, inv_batch);
    p.xb = (k & 1) ? c->xb2 : c->xb;
    if (p.drop_bits) p.drop_bits = c->drop_bits + (size_t)(k & 1) * 2 * MB * 8;
    p.g_work = work_dev + k;
    UpdParams u = make_upd(c, &sc0, B, nullptr);
    u.sched = c->sched_cur;
    u.sched_idx = k;
    u.loss_ring = c->loss_ring;
    u.ring_slot = k;
    u.ring_hdr = c->hdr;
    if (k + 1 == K) { u.adv_hdr = c->hdr; u.adv_k = K; u.adv_rows = B; }
    int rc = enqueue_step(c, p, u, mode, parity, k, /*from_hdr=*/true, st, nullptr);
    if (rc) return rc;
  }
  return IQLHIP_OK;
}

// A pinned table slot that no queued set-up kernel still has to read.
static int acquire_sched_slot(iqlhip_ctx* c, int* slot_out) {
  const int slot = c->sched_slot;
  c->sched_slot = (c->sched_slot + 1) & 3;
  const volatile unsigned long long* ack = c->sched_ack + slot;
  if (*ack < c->sched_want[slot]) {
    // (four calls deep in flight: wait for that set-up kernel — it runs at the head of its call — then re-check)
    for (int spin = 0; spin < 20000 && *ack < c->sched_want[slot]; ++spin) { /* ~ a few tens of us */ }
    if (*ack < c->sched_want[slot] && c->sched_stream[slot]) HIPCHK(hipStreamSynchronize(c->sched_stream[slot]));
  }
  *slot_out = slot;
  return IQLHIP_OK;
}

// The set-up launch of a call (iql_call_setup_kernel): header, scalar table, and (gather) step 0's rows + keep-bits.
// Its arguments as values + the pointer array both launch forms take: a direct launch, or — head chunk graphs, whose
// first node is this kernel — hipGraphExecKernelNodeSetParams in front of the replay.
struct SetupArgs {
  unsigned long long* hdr; ChunkHdr h; iqlhip_step_scalars* sched_call; const iqlhip_step_scalars* sched_src; int n_steps;
  const float* rows; long long ld; float* xb; int B; unsigned* drop_dst; int drop_words; unsigned drop_thresh;
  unsigned* arrivals; unsigned long long* ack; unsigned long long ack_val;
  void* ptrs[15];
  int nb;
};
static void fill_setup_args(iqlhip_ctx* c, SetupArgs& a, const ChunkHdr& h, int slot, int n_steps, const float* rows_dev,
                            int B, bool gather, hipStream_t st) {
  const int MB = c->dims.max_batch;
  const bool drop = gather && c->drop_p > 0.f;
  if (slot >= 0) {
    c->sched_want[slot] = ++c->call_seq;
    c->sched_stream[slot] = st;
  }
  a.hdr = c->hdr; a.h = h; a.sched_call = c->sched_call; a.sched_src = slot >= 0 ? c->sched_pin[slot] : c->sched_pin[0];
  a.n_steps = slot >= 0 ? n_steps : 0;
  a.rows = rows_dev; a.ld = (long long)c->row_ld; a.xb = c->xb; a.B = gather ? B : 0;
  a.drop_dst = drop ? c->drop_bits : nullptr; a.drop_words = 2 * MB * 8; a.drop_thresh = drop_thresh(c->drop_p);
  a.arrivals = c->setup_arrivals;
  a.ack = slot >= 0 ? c->sched_ack + slot : c->sched_ack + 4;        // (slot < 0: a capture-time placeholder, word 4 is a dummy)
  a.ack_val = slot >= 0 ? c->sched_want[slot] : 0ull;
  void* p[15] = {&a.hdr, &a.h, &a.sched_call, &a.sched_src, &a.n_steps, &a.rows, &a.ld, &a.xb, &a.B, &a.drop_dst,
                 &a.drop_words, &a.drop_thresh, &a.arrivals, &a.ack, &a.ack_val};
  memcpy(a.ptrs, p, sizeof p);
  long long want = ((long long)n_steps * 3 + 255) / 256;
  if (gather) want = std::max(want, ((long long)B * (c->row_ld / 4) + 255) / 256);
  if (drop) want += (2 * MB * 8 + 255) / 256;
  a.nb = (int)std::max<long long>(1, std::min<long long>(want, 256));
}
static int launch_call_setup(iqlhip_ctx* c, hipStream_t st, const ChunkHdr& h, int slot, int n_steps,
                             const float* rows_dev, int B, bool gather) {
  SetupArgs a;
  fill_setup_args(c, a, h, slot, n_steps, rows_dev, B, gather, st);
  HIPCHK(hipLaunchKernel((const void*)iql_call_setup_kernel, dim3(a.nb), dim3(256), a.ptrs, 0, st));
  return IQLHIP_OK;
}

static int chunk_graph(iqlhip_ctx* c, const GraphKey& key, hipGraphExec_t* out, iqlhip_ctx::CachedGraph** slot) {
  for (auto& g : c->graphs)
    if (g.key == key) { g.stamp = ++c->graph_clock; *out = g.exec; if (slot) *slot = &g; return IQLHIP_OK; }
  if (c->graphs.size() >= 12) {   // evict the least recently used — after its last replay has finished
    size_t lru = 0;
    for (size_t i = 1; i < c->graphs.size(); ++i) if (c->graphs[i].stamp < c->graphs[lru].stamp) lru = i;
    if (c->graphs[lru].last) HIPCHK(hipStreamSynchronize(c->graphs[lru].last));
    (void)hipGraphExecDestroy(c->graphs[lru].exec);
    (void)hipGraphDestroy(c->graphs[lru].graph);
    if (c->graphs[lru].work) (void)hipFree(c->graphs[lru].work);
    c->graphs.erase(c->graphs.begin() + lru);
  }
  // the chunk's idle-work records: device memory written once, here (their content is part of what the key freezes)
  IdleWork* work = nullptr;
  {
    std::vector<IdleWork> hw((size_t)key.K);
    fill_idle_work(c, hw.data(), key.rows, key.B, key.K);
    HIPCHK(hipMalloc((void**)&work, hw.size() * sizeof(IdleWork)));
    hipError_t e = hipMemcpy(work, hw.data(), hw.size() * sizeof(IdleWork), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(work); return fail(IQLHIP_EHIP, "chunk_graph: hipMemcpy: %s", hipGetErrorString(e)); }
  }
  hipStream_t cs = c->cap_stream;
  // (relaxed: a collective library may make calls during capture that the stricter modes forbid)
  hipError_t e = hipStreamBeginCapture(cs, key.xch == IQLHIP_XCH_RCCL ? hipStreamCaptureModeRelaxed : hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) { (void)hipFree(work); return fail(IQLHIP_EHIP, "hipStreamBeginCapture: %s", hipGetErrorString(e)); }
  int rc = IQLHIP_OK;
  if (key.head) {        // placeholder arguments: every replay sets the real ones (grid included)
    ChunkHdr h0;
    memset(&h0, 0, sizeof h0);
    h0.w[HDR_SIZE] = 1ull;
    SetupArgs a;
    fill_setup_args(c, a, h0, -1, 0, key.rows, key.B, true, cs);
    if (hipLaunchKernel((const void*)iql_call_setup_kernel, dim3(a.nb), dim3(256), a.ptrs, 0, cs) != hipSuccess)
      rc = fail(IQLHIP_EHIP, "capture of the set-up kernel failed");
  }
  if (!rc) rc = enqueue_chunk(c, cs, key.B, key.K, key.inv_batch, key.xch, key.parity, work);
  hipGraph_t graph = nullptr;
  e = hipStreamEndCapture(cs, &graph);
  if (rc) { if (graph) (void)hipGraphDestroy(graph); (void)hipFree(work); return rc; }
  if (e != hipSuccess) { (void)hipFree(work); return fail(IQLHIP_EHIP, "hipStreamEndCapture: %s", hipGetErrorString(e)); }
  hipGraphExec_t gexec = nullptr;
  e = hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0);
  if (e != hipSuccess) { (void)hipGraphDestroy(graph); (void)hipFree(work); return fail(IQLHIP_EHIP, "hipGraphInstantiate: %s", hipGetErrorString(e)); }
  hipGraphNode_t setup_node = nullptr;
  if (key.head) {
    size_t n_root = 1;
    e = hipGraphGetRootNodes(graph, &setup_node, &n_root);
    hipGraphNodeType ty = hipGraphNodeTypeEmpty;
    if (e == hipSuccess && n_root == 1) e = hipGraphNodeGetType(setup_node, &ty);
    if (e != hipSuccess || n_root != 1 || ty != hipGraphNodeTypeKernel) {
      (void)hipGraphExecDestroy(gexec); (void)hipGraphDestroy(graph); (void)hipFree(work);
      return fail(IQLHIP_EHIP, "head chunk graph: no single kernel root node");
    }
  }
  c->graphs.push_back({key, graph, gexec, ++c->graph_clock, nullptr, work, setup_node});
  *out = gexec;
  if (slot) *slot = &c->graphs.back();
  return IQLHIP_OK;
}

static int check_train_args(const iqlhip_ctx* c, const float* rows_dev, int64_t ld, int32_t B) {
  if (!c || !rows_dev) return fail(IQLHIP_EINVAL, "NULL argument");
  if (!c->params) return fail(IQLHIP_ENOTBOUND, "iqlhip_bind has not been called");
  if (B < 1 || B > c->dims.max_batch) return fail(IQLHIP_EINVAL, "batch_rows outside [1,max_batch]");
  if (ld != c->row_ld) return fail(IQLHIP_EINVAL, "row stride must be iqlhip_row_stride(S,A)=%lld", (long long)c->row_ld);
  if (((uintptr_t)rows_dev) & 15) return fail(IQLHIP_EINVAL, "packed rows must be 16-byte aligned");
  return IQLHIP_OK;
}

static GraphKey make_key(const iqlhip_ctx* c, const float* rows_dev, int64_t ld, int32_t B, int32_t K, float inv_batch,
                         int parity, int head = 0) {
  GraphKey key;
  key.head = head;
  key.rows = rows_dev; key.ld = ld; key.B = B; key.K = K; key.params = c->params; key.drop_p = c->drop_p;
  key.inv_batch = inv_batch; key.xch = c->xch_mode;
  key.parity = (c->xch_mode == IQLHIP_XCH_P2P) ? parity : 0;
  return key;
}

// Replay one chunk graph on `st`.  head: the call's first chunk — its set-up node gets this call's arguments first.
// How a call's first steps reach the GPU (IQLHIP_HEAD): "direct" (default) — the set-up kernel and the first 2 or 4
// steps are launched kernel by kernel, so the GPU has work ~3 us after the call instead of after a graph launch
// (~12 us), and the chunk launches that follow are hidden behind their execution; "graph" — round 3's first form, a head
// chunk graph whose first node is the set-up kernel; "plain" — set-up kernel + plain chunks only (diagnostic).
// profiles/r03_ab_experiments.txt #7: the driver's 20-step command 41.7 k -> 42.45 k steps/s with "direct".
static const int g_head_mode = [] {
  const char* e = getenv("IQLHIP_HEAD");
  if (e && !strcmp(e, "graph")) return 1;
  if (e && !strcmp(e, "plain")) return 2;
  return 0;
}();
static const bool g_direct_head = g_head_mode == 0;
static const bool g_no_head_graph = g_head_mode == 2;
static const bool g_direct_all = getenv("IQLHIP_DIRECT_ALL") != nullptr;           // diagnostic: no graph replays at all
static int n_chunks_for(int rem) {
  int n = rem / GRAPH_STEPS;
  rem %= GRAPH_STEPS;
  for (int cs_ : {16, 4, 2, 1}) { n += rem / cs_; rem %= cs_; }
  return n;
}
static int replay_chunk(iqlhip_ctx* c, hipStream_t st, const float* rows_dev, int64_t ld, int B, int n, float inv_batch,
                        const SetupArgs* head_args) {
  const int parity = (int)(c->xstep & 1ull);
  hipGraphExec_t gexec = nullptr;
  iqlhip_ctx::CachedGraph* cg = nullptr;
  int r = chunk_graph(c, make_key(c, rows_dev, ld, B, n, inv_batch, parity, head_args ? 1 : 0), &gexec, &cg);
  if (r) return r;
  if (g_direct_all && !head_args) {      // experiment: the chunk's kernels launched one by one instead of the graph replay
    r = enqueue_chunk(c, st, B, n, inv_batch, c->xch_mode, parity, cg->work);
    if (r) return r;
    cg->last = st;
    c->drop_step += (unsigned long long)n;
    if (c->xch_mode != IQLHIP_XCH_NONE) c->xstep += (unsigned long long)n;
    return IQLHIP_OK;
  }
  if (head_args) {
    hipKernelNodeParams np;
    memset(&np, 0, sizeof np);
    np.func = (void*)iql_call_setup_kernel;
    np.gridDim = dim3(head_args->nb);
    np.blockDim = dim3(256);
    np.kernelParams = const_cast<void**>(head_args->ptrs);
    HIPCHK(hipGraphExecKernelNodeSetParams(gexec, cg->setup_node, &np));
  }
  HIPCHK(hipGraphLaunch(gexec, st));
  cg->last = st;
  c->drop_step += (unsigned long long)n;
  if (c->xch_mode != IQLHIP_XCH_NONE) c->xstep += (unsigned long long)n;
  return IQLHIP_OK;
}

extern "C" int iqlhip_train_steps_prepare(iqlhip_ctx* c, const float* rows_dev, int64_t ld, int32_t B, float inv_batch,
                                          void* stream) {
  int rc = check_train_args(c, rows_dev, ld, B);
  if (rc) return rc;
  DevGuard guard(c->device);
  HIPCHK(hipDeviceSynchronize());       // a one-off set-up call: ordered after everything queued on any stream
  c->cont.valid = false;
  // the rehearsal replays run on the CALLER's stream — the first replay of a graph on a stream it has not run on yet was
  // measured ~20 us slower than the following ones, rehearsed on another stream or not (capture itself happens on the
  // library's own stream: the legacy default stream cannot be captured)
  hipStream_t cs = (hipStream_t)stream;
  // Each chunk graph is captured, instantiated, uploaded AND replayed once, so that its first replay inside a caller's
  // timed region costs what every later one does (a first replay is ~50-100 us slower, and the first launch of a
  // kernel loads its code).  The rehearsal must not train: the parameter, moment and target arenas are saved before
  // and restored after it; it reads row 0 only (size = 1); scratch, loss words and ring are transient anyway.  Under
  // data parallelism the rehearsal runs the exchange too — every rank must call prepare (the same number of times).
  const size_t np_b = (size_t)c->L.n_params * sizeof(float), nt_b = (size_t)c->L.n_target * sizeof(float);
  if (!c->prep_save) HIPCHK(hipMalloc((void**)&c->prep_save, 3 * np_b + nt_b));     // (arena sizes are fixed per context)
  char* const save = c->prep_save;
  auto copy_all = [&](bool restore) -> hipError_t {
    float* arenas[4] = {c->params, c->m, c->v, c->target};
    size_t off = 0;
    for (int i = 0; i < 4; ++i) {
      const size_t nb = (i == 3) ? nt_b : np_b;
      hipError_t e = restore ? hipMemcpyAsync(arenas[i], save + off, nb, hipMemcpyDeviceToDevice, cs)
                             : hipMemcpyAsync(save + off, arenas[i], nb, hipMemcpyDeviceToDevice, cs);
      if (e != hipSuccess) return e;
      off += nb;
    }
    return hipSuccess;
  };
  hipError_t e = copy_all(false);
  if (e != hipSuccess) return fail(IQLHIP_EHIP, "prepare: save arenas: %s", hipGetErrorString(e));
  refresh_shadows(c, cs);
  int slot = 0;
  rc = acquire_sched_slot(c, &slot);
  if (rc) return rc;
  const unsigned long long drop_step0 = c->drop_step;
  do {
    iqlhip_step_scalars benign;
    memset(&benign, 0, sizeof benign);
    benign.bc2_sqrt[0] = benign.bc2_sqrt[1] = benign.bc2_sqrt[2] = 1.f;
    benign.beta2 = 1.f; benign.eps = 1e-8f; benign.grad_scale = 1.f; benign.inv_batch = inv_batch;
    for (int k = 0; k < c->k_max; ++k) c->sched_pin[slot][k] = benign;
    // every chunk graph once: the head chunks (2 steps, and the one-step call's) with their set-up node, the plain ones
    // behind a directly launched set-up kernel.  P2P: a one-step chunk flips the buffer parity, so a second pass over
    // the same list reaches the other captured variant of each.
    const int passes = (c->xch_mode == IQLHIP_XCH_P2P) ? 2 : 1;
    struct Item { int K; int head; };
    const Item items[] = {{2, 1}, {GRAPH_STEPS, 0}, {16, 0}, {4, 0}, {2, 0}, {1, 0}};
    auto rehearse = [&](const Item& it) -> int {
      ChunkHdr h;
      memset(&h, 0, sizeof h);
      h.w[HDR_SIZE] = 1ull;
      h.w[HDR_DROP_STEP] = c->drop_step;
      h.w[HDR_DROP_SEED] = c->drop_seed;
      h.w[HDR_XSTEP] = c->xstep;
      if (it.head) {
        SetupArgs a;
        fill_setup_args(c, a, h, slot, it.K, rows_dev, B, /*gather=*/true, cs);
        return replay_chunk(c, cs, rows_dev, ld, B, it.K, inv_batch, &a);
      }
      int r = launch_call_setup(c, cs, h, slot, it.K, rows_dev, B, /*gather=*/true);
      if (r) return r;
      return replay_chunk(c, cs, rows_dev, ld, B, it.K, inv_batch, nullptr);
    };
    for (int pass = 0; pass < passes && !rc; ++pass)
      for (const Item& it : items) { if (it.head && g_head_mode != 1) continue; rc = rehearse(it); if (rc) break; }
    for (int pass = 0; pass < passes && !rc && g_head_mode == 1; ++pass) rc = rehearse(Item{1, 1});
    // Sustained replays of the 64-step chunk (benign scalars, row 0, arenas restored below like the rest of the
    // rehearsal): the chip's clocks ramp up over the first milliseconds of a workload — a timed region that starts right
    // behind a capture-heavy (GPU-idle) prepare call measured its first 20 steps ~5 % slower than the following ones
    // (profiles/r03_first_call.txt).  IQLHIP_PREPARE_WARM_CHUNKS: how many (default 16 = 1 024 steps, ~22 ms; 0 = none).
    static const int n_warm = getenv("IQLHIP_PREPARE_WARM_CHUNKS") ? std::max(0, atoi(getenv("IQLHIP_PREPARE_WARM_CHUNKS"))) : 16;
    if (!rc && n_warm > 0 && c->xch_mode == IQLHIP_XCH_NONE) {
      ChunkHdr h;
      memset(&h, 0, sizeof h);
      h.w[HDR_SIZE] = 1ull;
      h.w[HDR_DROP_STEP] = c->drop_step;
      h.w[HDR_DROP_SEED] = c->drop_seed;
      for (int i = 0; i < n_warm && !rc; ++i) {
        if ((i % (c->k_max / GRAPH_STEPS)) == 0) rc = launch_call_setup(c, cs, h, slot, c->k_max, rows_dev, B, /*gather=*/true);
        if (!rc) rc = replay_chunk(c, cs, rows_dev, ld, B, GRAPH_STEPS, inv_batch, nullptr);
      }
    }
  } while (0);
  c->drop_step = drop_step0;            // (the rehearsal drew keep-bits from the stream's current position; it is not advanced)
  e = copy_all(true);
  refresh_shadows(c, cs);
  hipError_t e2 = hipStreamSynchronize(cs);
  if (rc) return rc;
  if (e != hipSuccess || e2 != hipSuccess) return fail(IQLHIP_EHIP, "prepare: restore arenas: %s", hipGetErrorString(e != hipSuccess ? e : e2));
  return IQLHIP_OK;
}

extern "C" int iqlhip_train_steps(iqlhip_ctx* c, const float* rows_dev, int64_t ld, int64_t size, int32_t B,
                                  const iqlhip_step_scalars* sc, int32_t K, uint64_t seed, uint64_t stream_offset,
                                  int32_t flags, void* stream) {
  if (!sc) return fail(IQLHIP_EINVAL, "NULL argument");
  int rc = check_train_args(c, rows_dev, ld, B);
  if (rc) return rc;
  if (K < 1 || K > c->k_max) return fail(IQLHIP_EINVAL, "n_steps outside [1,%d]", c->k_max);
  if (size < 1) return fail(IQLHIP_EINVAL, "empty buffer");
  double tr_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (g_trace) tr_t[0] = now_us();
  DevGuard guard(c->device);
  hipStream_t st = (hipStream_t)stream;
  // the call's scalar table goes into a pinned, host-mapped slot that the set-up kernel reads in place; a slot is
  // reused only after the set-up kernel that read it has run (event), so the caller's array is free on return
  int slot = 0;
  rc = acquire_sched_slot(c, &slot);
  if (rc) return rc;
  memcpy(c->sched_pin[slot], sc, (size_t)K * sizeof(iqlhip_step_scalars));
  const float inv_batch = sc[0].inv_batch;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (c->timing) {
    rc = ensure_events(c, c->ev_used + 2);
    if (rc) return rc;
    ev0 = c->ev[c->ev_used]; ev1 = c->ev[c->ev_used + 1];
    HIPCHK(hipEventRecord(ev0, st));
  }
  // does this call continue the previous one's index stream with its step 0 already staged?
  const bool cont = (flags & IQLHIP_TS_CONTINUE) && c->cont.valid && c->cont.rows == rows_dev && c->cont.ld == ld &&
                    c->cont.size == size && c->cont.B == B && c->cont.seed == seed && c->cont.next_offset == stream_offset &&
                    c->cont.drop_p == c->drop_p && c->cont.drop_seed == c->drop_seed && c->cont.drop_step == c->drop_step;
  c->cont.valid = false;
  ChunkHdr h;
  memset(&h, 0, sizeof h);
  h.w[HDR_SIZE] = (unsigned long long)size;
  h.w[HDR_SEED] = (unsigned long long)seed;
  h.w[HDR_OFFSET] = (unsigned long long)stream_offset;     // index j of the call = counter offset + j / 2, pair word j & 1
  h.w[HDR_DROP_STEP] = c->drop_step;
  h.w[HDR_DROP_SEED] = c->drop_seed;
  h.w[HDR_XSTEP] = c->xstep;
  refresh_shadows(c, st);
  if (g_trace) tr_t[1] = now_us();
  // The call's HEAD — the set-up kernel and the first 2 or 4 steps (1 for a one-step call) — is launched kernel by
  // kernel (g_head_mode; a graph launch costs ~10 us + 0.4 us per node on the host and the GPU starts when it returns,
  // a kernel launch ~2.5 us).  The rest follows as chunk graphs, the even sizes ascending (each launch is hidden behind
  // the execution of what was launched before), the 64-step chunk as often as it fits, a one-step chunk — odd calls
  // only — last.
  int rem;
  if (g_direct_head) {
    // 2 or 4 steps (even: a chunk's step 0 reads staging buffer 0), whichever leaves fewer chunk launches behind it —
    // every launch boundary between chunks costs the GPU ~5 us (profiles/r03_train_steps_call_length.txt)
    const int head_k = (K >= 4 && n_chunks_for(K - 4) < n_chunks_for(K - 2)) ? 4 : ((K >= 2) ? 2 : 1);
    const int parity = (int)(c->xstep & 1ull);
    hipGraphExec_t gexec = nullptr;
    iqlhip_ctx::CachedGraph* cg = nullptr;
    rc = chunk_graph(c, make_key(c, rows_dev, ld, B, head_k, inv_batch, parity, 0), &gexec, &cg);     // (for its idle-work records)
    if (rc) return rc;
    rc = launch_call_setup(c, st, h, slot, K, rows_dev, B, /*gather=*/!cont);
    if (rc) return rc;
    rc = enqueue_chunk(c, st, B, head_k, inv_batch, c->xch_mode, parity, cg->work);
    if (rc) return rc;
    cg->last = st;
    c->drop_step += (unsigned long long)head_k;
    if (c->xch_mode != IQLHIP_XCH_NONE) c->xstep += (unsigned long long)head_k;
    rem = K - head_k;
  } else if (g_no_head_graph) {
    rc = launch_call_setup(c, st, h, slot, K, rows_dev, B, /*gather=*/!cont);
    if (rc) return rc;
    rem = K;
  } else {
    const int head_k = (K >= 2) ? 2 : 1;
    SetupArgs a;
    fill_setup_args(c, a, h, slot, K, rows_dev, B, /*gather=*/!cont, st);
    rc = replay_chunk(c, st, rows_dev, ld, B, head_k, inv_batch, &a);
    if (rc) return rc;
    rem = K - head_k;
  }
  if (g_trace) tr_t[2] = tr_t[3] = now_us();
  const int n64 = rem / GRAPH_STEPS;
  rem %= GRAPH_STEPS;
  int small[8], ns = 0;
  for (int cs_ : {16, 4, 2, 1}) while (rem >= cs_) { small[ns++] = cs_; rem -= cs_; }     // (<= 3 + 3 + 1 + 1 entries)
  for (int i = ns - 1; i >= 0; --i) if (small[i] != 1) { rc = replay_chunk(c, st, rows_dev, ld, B, small[i], inv_batch, nullptr); if (rc) return rc; }
  for (int i = 0; i < n64; ++i) { rc = replay_chunk(c, st, rows_dev, ld, B, GRAPH_STEPS, inv_batch, nullptr); if (rc) return rc; }
  if (ns > 0 && small[ns - 1] == 1) { rc = replay_chunk(c, st, rows_dev, ld, B, 1, inv_batch, nullptr); if (rc) return rc; }
  // what a following call must look like to start on the rows this call's last forward has staged: an even number of
  // steps ends on staging buffer 0, where a chunk's step 0 reads; the next counter follows from the rows drawn
  if ((K & 1) == 0 && (((unsigned long long)K * (unsigned long long)B) & 1ull) == 0) {
    c->cont.valid = true;
    c->cont.rows = rows_dev; c->cont.ld = ld; c->cont.size = size; c->cont.B = B; c->cont.seed = seed;
    c->cont.next_offset = stream_offset + ((unsigned long long)K * (unsigned long long)B) / 2ull;
    c->cont.drop_p = c->drop_p; c->cont.drop_seed = c->drop_seed; c->cont.drop_step = c->drop_step;
  }
  if (ev0) {
    HIPCHK(hipEventRecord(ev1, st));
    HIPCHK(hipEventSynchronize(ev1));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, ev0, ev1));
    c->t_acc[3] += ms * 1e3f;   // total per call; per-step = /K done by the caller
    c->t_n += 1;
  }
  HIPCHK(hipGetLastError());
  if (g_trace)
    fprintf(stderr, "[iqlhip trace] train_steps K=%d cont=%d: entry->launch %.1f us, head chunk (set-up + launch) %.1f, rest %.1f\n",
            K, (int)cont, tr_t[1] - tr_t[0], tr_t[2] - tr_t[1], now_us() - tr_t[3]);
  return IQLHIP_OK;
}

// ---------------------------------------------------------------------------
// Dataset ingest on the device.
extern "C" int iqlhip_cols_mean_std(const float* x_dev, int64_t ld, int32_t ncols, int64_t n, float eps, float* mean_dev,
                                    float* std_dev, void* stream) {
  if (!x_dev || !mean_dev || !std_dev) return fail(IQLHIP_EINVAL, "NULL argument");
  if (ncols < 1 || ncols > 4096 || n < 1 || ld < ncols) return fail(IQLHIP_EINVAL, "bad cols_mean_std geometry");
  hipStream_t st = (hipStream_t)stream;
  const int nb = (int)std::min<int64_t>((n + 7) / 8, MS_BLOCKS);
  double* scratch = nullptr;                         // [nb][ncols] partials + [ncols] float64 means
  HIPCHK(hipMallocAsync((void**)&scratch, ((size_t)nb * ncols + ncols) * sizeof(double), st));
  double* mean64 = scratch + (size_t)nb * ncols;
  hipLaunchKernelGGL(iql_cols_moment_kernel<0>, dim3(nb), dim3(256), 0, st, x_dev, (long long)ld, ncols, (long long)n,
                     (const double*)nullptr, scratch);
  hipLaunchKernelGGL(iql_cols_finish_kernel<0>, dim3((ncols + 63) / 64), dim3(64), 0, st, (const double*)scratch, nb, ncols,
                     (long long)n, eps, mean64, mean_dev);
  hipLaunchKernelGGL(iql_cols_moment_kernel<1>, dim3(nb), dim3(256), 0, st, x_dev, (long long)ld, ncols, (long long)n,
                     (const double*)mean64, scratch);
  hipLaunchKernelGGL(iql_cols_finish_kernel<1>, dim3((ncols + 63) / 64), dim3(64), 0, st, (const double*)scratch, nb, ncols,
                     (long long)n, eps, mean64, std_dev);
  HIPCHK(hipFreeAsync(scratch, st));
  HIPCHK(hipGetLastError());
  return IQLHIP_OK;
}

extern "C" int iqlhip_rows_normalize(float* rows_dev, int64_t ld, int32_t S, int32_t A, int64_t row0, int64_t n,
                                     const float* mean_dev, const float* std_dev, void* stream) {
  if (!rows_dev || !mean_dev || !std_dev) return fail(IQLHIP_EINVAL, "NULL argument");
  if (n < 0 || row0 < 0 || S < 1 || A < 1 || ld < 2 * (int64_t)S + A + 2) return fail(IQLHIP_EINVAL, "bad rows_normalize geometry");
  if (n == 0) return IQLHIP_OK;
  const long long total = (long long)n * 2 * S;
  const int nb = (int)std::min<long long>((total + 255) / 256, 8192);
  hipLaunchKernelGGL(iql_rows_normalize_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, rows_dev, (long long)ld, S, A,
                     (long long)row0, (long long)n, mean_dev, std_dev);
  HIPCHK(hipGetLastError());
  return IQLHIP_OK;
}

// ---------------------------------------------------------------------------
extern "C" int iqlhip_rows_fill_synth(float* rows_dev, int64_t ld, int32_t S, int32_t A, int64_t row0, int64_t n,
                                      uint64_t seed, float p_done, int32_t antmaze_rewards, void* stream) {
  if (!rows_dev) return fail(IQLHIP_EINVAL, "NULL argument");
  if (n < 0 || row0 < 0 || S < 1 || A < 1 || ld < 2 * (int64_t)S + A + 2) return fail(IQLHIP_EINVAL, "bad rows_fill_synth geometry");
  if (!(p_done >= 0.f && p_done <= 1.f)) return fail(IQLHIP_EINVAL, "p_done outside [0,1]");
  if (n == 0) return IQLHIP_OK;
  const long long total = (long long)n * (2 * S + A + 2);
  const int nb = (int)std::min<long long>((total + 255) / 256, 16384);
  hipLaunchKernelGGL(iql_rows_fill_synth_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, rows_dev, (long long)ld, S, A,
                     (long long)row0, (long long)n, (unsigned long long)seed, p_done, (int)antmaze_rewards);
  HIPCHK(hipGetLastError());
  return IQLHIP_OK;
}

// ---------------------------------------------------------------------------
extern "C" int iqlhip_rows_write(float* rows_dev, int64_t ld, int32_t S, int32_t A, int64_t row0, int64_t n,
                                 const float* s, const float* a, const float* r, const float* ns, const float* d,
                                 void* stream) {
  if (!rows_dev || !s || !a || !r || !ns || !d) return fail(IQLHIP_EINVAL, "NULL argument");
  if (n < 0 || row0 < 0 || ld < 2 * (int64_t)S + A + 2) return fail(IQLHIP_EINVAL, "bad rows_write geometry");
  if (n == 0) return IQLHIP_OK;
  const long long total = (long long)n * (2 * S + A + 2);
  const int nb = (int)std::min<long long>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(iql_rows_write_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, rows_dev, (long long)ld, S, A,
                     (long long)row0, (long long)n, s, a, r, ns, d);
  HIPCHK(hipGetLastError());
  return IQLHIP_OK;
}

extern "C" int iqlhip_rows_gather(const float* rows_dev, int64_t ld, int64_t n_rows, int32_t S, int32_t A,
                                  const int64_t* idx_dev, int64_t n, float* s, float* a, float* r, float* ns, float* d,
                                  void* stream) {
  if (!rows_dev || !idx_dev || !s || !a || !r || !ns || !d) return fail(IQLHIP_EINVAL, "NULL argument");
  if (n < 0 || n_rows < 1 || ld < 2 * (int64_t)S + A + 2) return fail(IQLHIP_EINVAL, "bad rows_gather geometry");
  if (n == 0) return IQLHIP_OK;
  const long long total = (long long)n * (2 * S + A + 2);
  const int nb = (int)std::min<long long>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(iql_rows_gather_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, rows_dev, (long long)ld,
                     (long long)n_rows, S, A, (const long long*)idx_dev, (long long)n, s, a, r, ns, d);
  HIPCHK(hipGetLastError());
  return IQLHIP_OK;
}

// ReplayBuffer.sample as ONE coalesced row gather: out[i] = rows[idx[i]] (whole packed rows).
extern "C" int iqlhip_rows_gather_packed(const float* rows_dev, int64_t ld, int64_t n_rows, const int64_t* idx_dev,
                                         int64_t n, float* out_rows_dev, void* stream) {
  if (!rows_dev || !idx_dev || !out_rows_dev) return fail(IQLHIP_EINVAL, "NULL argument");
  if (n < 0 || n_rows < 1 || ld < 4 || (ld & 3)) return fail(IQLHIP_EINVAL, "bad rows_gather_packed geometry");
  if ((((uintptr_t)rows_dev) | ((uintptr_t)out_rows_dev)) & 15) return fail(IQLHIP_EINVAL, "rows must be 16-byte aligned");
  if (n == 0) return IQLHIP_OK;
  const long long total = (long long)n * (ld / 4);
  if (total > 0x7fffffffLL) return fail(IQLHIP_EINVAL, "gather too large for one call");
  hipLaunchKernelGGL(iql_gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rows_dev,
                     (long long)ld, (const long long*)idx_dev, out_rows_dev, (int)n, (long long)n_rows);
  HIPCHK(hipGetLastError());
  return IQLHIP_OK;
}

extern "C" int iqlhip_stream_synchronize(void* stream) {
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  return IQLHIP_OK;
}

// The same with the indices still on the host (pinned): one H2D copy into idx_scratch_dev, then the gather —
// ReplayBuffer.sample's whole device side in one call.
extern "C" int iqlhip_rows_gather_packed_h(const float* rows_dev, int64_t ld, int64_t n_rows, const int64_t* idx_host,
                                           int64_t* idx_scratch_dev, int64_t n, float* out_rows_dev, void* stream) {
  if (!idx_host || !idx_scratch_dev) return fail(IQLHIP_EINVAL, "NULL argument");
  if (n < 0 || n_rows < 1) return fail(IQLHIP_EINVAL, "bad rows_gather_packed_h geometry");
  if (n == 0) return IQLHIP_OK;
  int rc = check_host_indices(idx_host, n, n_rows);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(idx_scratch_dev, idx_host, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, (hipStream_t)stream));
  return iqlhip_rows_gather_packed(rows_dev, ld, n_rows, idx_scratch_dev, n, out_rows_dev, stream);
}

// ReplayBuffer.sample in one call from ORDINARY host memory (the array np.random.randint returned): the indices are
// copied into a pinned ring slot owned by the library (guarded by an event, so a slot is never rewritten while its
// H2D copy may still be queued), sent to a device scratch array on `stream`, and the rows gathered.  Per device
// state, created on first use; like the rest of the ABI not thread-safe.
namespace {
struct SampleStage {
  int device = -1;
  int64_t cap = 0;
  enum { SLOTS = 8 };
  int64_t* host[SLOTS] = {};             // pinned, host-mapped index slots the gather kernel reads in place
  unsigned long long want[SLOTS] = {};   // the call number whose kernel must have acknowledged the slot before reuse
  hipStream_t stream[SLOTS] = {};
  unsigned long long* ack = nullptr;     // pinned [SLOTS]: written by the gather kernels
  unsigned* arrivals = nullptr;          // device block counter
  unsigned long long seq = 0;
  int slot = 0;
};
SampleStage g_stage[16];
}  // namespace

// ReplayBuffer.sample in one call from ORDINARY host memory (the array np.random.randint returned): the indices are
// copied into a pinned, host-mapped slot owned by the library and the gather kernel reads them THERE (2 KB over PCIe) —
// no H2D copy, no event; a slot is reused only after its kernel has acknowledged it in a host-mapped word.  Per-device
// state, created on first use; like the rest of the ABI not thread-safe.
extern "C" int iqlhip_rows_sample_packed(const float* rows_dev, int64_t ld, int64_t n_rows, const int64_t* idx_host,
                                         int64_t n, float* out_rows_dev, void* stream) {
  if (!rows_dev || !idx_host || !out_rows_dev) return fail(IQLHIP_EINVAL, "NULL argument");
  if (n < 0 || n_rows < 1 || ld < 4 || (ld & 3)) return fail(IQLHIP_EINVAL, "bad rows_sample_packed geometry");
  if ((((uintptr_t)rows_dev) | ((uintptr_t)out_rows_dev)) & 15) return fail(IQLHIP_EINVAL, "rows must be 16-byte aligned");
  if (n == 0) return IQLHIP_OK;
  {
    int rc = check_host_indices(idx_host, n, n_rows);
    if (rc) return rc;
  }
  const long long total = (long long)n * (ld / 4);
  if (total > 0x7fffffffLL) return fail(IQLHIP_EINVAL, "gather too large for one call");
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 16) return fail(IQLHIP_EUNSUPPORTED, "device index %d", dev);
  SampleStage& sg = g_stage[dev];
  if (sg.cap < n) {
    HIPCHK(hipDeviceSynchronize());          // nothing may still read the old staging
    for (int i = 0; i < SampleStage::SLOTS; ++i) if (sg.host[i]) { (void)hipHostFree(sg.host[i]); sg.host[i] = nullptr; }
    sg.cap = std::max<int64_t>(n, 1024);
    for (int i = 0; i < SampleStage::SLOTS; ++i) {
      HIPCHK(hipHostMalloc((void**)&sg.host[i], (size_t)sg.cap * sizeof(int64_t), hipHostMallocDefault));
      sg.want[i] = 0;
    }
    if (!sg.ack) {
      HIPCHK(hipHostMalloc((void**)&sg.ack, SampleStage::SLOTS * sizeof(unsigned long long), hipHostMallocDefault));
      memset(sg.ack, 0, SampleStage::SLOTS * sizeof(unsigned long long));
      HIPCHK(hipMalloc((void**)&sg.arrivals, 64));
      HIPCHK(hipMemset(sg.arrivals, 0, 64));
      HIPCHK(hipDeviceSynchronize());
    }
    sg.device = dev;
  }
  const int k = sg.slot;
  sg.slot = (sg.slot + 1) % SampleStage::SLOTS;
  {
    const volatile unsigned long long* ack = sg.ack + k;
    if (*ack < sg.want[k]) {                 // eight samples deep in flight: wait for that slot's kernel
      for (int spin = 0; spin < 20000 && *ack < sg.want[k]; ++spin) {}
      if (*ack < sg.want[k] && sg.stream[k]) HIPCHK(hipStreamSynchronize(sg.stream[k]));
    }
  }
  memcpy(sg.host[k], idx_host, (size_t)n * sizeof(int64_t));
  sg.want[k] = ++sg.seq;
  sg.stream[k] = (hipStream_t)stream;
  hipLaunchKernelGGL(iql_gather_hostidx_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rows_dev,
                     (long long)ld, (const long long*)sg.host[k], out_rows_dev, (int)n, (long long)n_rows, sg.arrivals, sg.ack + k, sg.want[k]);
  HIPCHK(hipGetLastError());
  return IQLHIP_OK;
}

// ---------------------------------------------------------------------------
// Policy inference: pack states -> forward of the policy instance only -> tanh / noise / scale / clamp.

extern "C" int iqlhip_actor_forward(iqlhip_ctx* c, const float* states_dev, int64_t ld_s, int32_t rows,
                                    const float* noise_dev, int64_t ld_noise, float max_action, float* actions_dev,
                                    int64_t ld_a, void* stream) {
  return actor_forward_impl(c, states_dev, ld_s, rows, noise_dev, ld_noise, 0, 0, max_action, actions_dev, ld_a, stream);
}

// dist.sample() with the noise drawn on the device: seed != 0 selects the stream, the library counts the calls.
extern "C" int iqlhip_actor_sample(iqlhip_ctx* c, const float* states_dev, int64_t ld_s, int32_t rows, uint64_t seed,
                                   float max_action, float* actions_dev, int64_t ld_a, void* stream) {
  if (!c) return fail(IQLHIP_EINVAL, "NULL argument");
  if (seed == 0) return fail(IQLHIP_EINVAL, "seed must be non-zero");
  return actor_forward_impl(c, states_dev, ld_s, rows, nullptr, 0, seed, c->act_calls++, max_action, actions_dev, ld_a,
                            stream);
}

static int actor_forward_impl(iqlhip_ctx* c, const float* states_dev, int64_t ld_s, int32_t rows, const float* noise_dev,
                              int64_t ld_noise, uint64_t rng_seed, uint64_t rng_call, float max_action,
                              float* actions_dev, int64_t ld_a, void* stream, unsigned long long* done_flag,
                              unsigned long long done_val) {
  if (!c || !states_dev || !actions_dev) return fail(IQLHIP_EINVAL, "NULL argument");
  if (!c->params) return fail(IQLHIP_EINVAL, "iqlhip_bind has not been called");
  const int S = c->dims.state_dim, A = c->dims.action_dim;
  if (rows < 0 || rows > c->act_cap) return fail(IQLHIP_EINVAL, "rows %d outside [0, %d]", rows, c->act_cap);
  if (ld_s < S || ld_a < A || (noise_dev && ld_noise < A)) return fail(IQLHIP_EINVAL, "row stride smaller than the row");
  if (rows == 0) return IQLHIP_OK;
  hipStream_t st = (hipStream_t)stream;
  const int total = rows * (int)c->row_ld;
  hipLaunchKernelGGL(iql_pack_states_kernel, dim3((total + 255) / 256), dim3(256), 0, st, c->xb_act, (int)c->row_ld, S,
                     rows, states_dev, (long long)ld_s);
  refresh_shadows(c, st);
  StepParams p = make_step(c, rows, 1.f / (float)rows);
  p.xb = c->xb_act;
  p.only_inst = 6;
  p.slot[6] = -1;            // inference keeps no activations
  p.drop_bits = nullptr;     // eval-mode forward; a training-mode policy with dropout stays on the caller's side
  p.sc.heads = c->heads_act; // the policy partials of row r land at heads[max_batch * HEAD_LD + r * A * NSPLIT ...]:
  p.sc.max_batch = 0;        // with max_batch = 0 that is heads_act[r * A * NSPLIT ...]
  const int n_rt = (rows + RT_ROWS - 1) / RT_ROWS;
  launch_fwd_grid(c, p, n_rt * NSPLIT, st);
  hipLaunchKernelGGL(iql_actor_finish_kernel, dim3((rows * A + 255) / 256), dim3(256), 0, st, c->heads_act, rows, A,
                     max_action, p.log_std, c->hyper.log_std_min, c->hyper.log_std_max, noise_dev, (long long)ld_noise,
                     (unsigned long long)rng_seed, (unsigned long long)rng_call, actions_dev, (long long)ld_a,
                     (rows * A <= 256) ? done_flag : (unsigned long long*)nullptr, done_val);
  if (done_flag && rows * A > 256) return fail(IQLHIP_EINVAL, "a completion flag needs a one-block finish launch");
  HIPCHK(hipGetLastError());
  return IQLHIP_OK;
}

// Diagnostic: queue a kernel that writes a fresh number into a host-mapped word, then spin on that word from the host
// (no HIP call): *spin_us = time until the GPU has drained `stream`, as the host sees it without any synchronise call.
extern "C" int iqlhip_debug_drain_spin(iqlhip_ctx* c, void* stream, double* spin_us) {
  if (!c || !spin_us) return fail(IQLHIP_EINVAL, "NULL argument");
  const double t0 = now_us();
  const unsigned long long v = ++c->call_seq;
  hipLaunchKernelGGL(iql_debug_flag_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, c->sched_ack + 5, v);
  const volatile unsigned long long* f = c->sched_ack + 5;
  while (*f != v) { if (now_us() - t0 > 5e6) return fail(IQLHIP_EHIP, "drain_spin: timeout"); }
  *spin_us = now_us() - t0;
  return IQLHIP_OK;
}

// ---------------------------------------------------------------------------
// Micro-benchmark hook: launch ONE kernel of the step `repeat` times back to back and return the
// average time per launch (hipEvents on `stream`).  which: 0 fwd, 1 bwd, 2 update (no-op scalars:
// step_size 0, so parameters do not move), 3 fwd+bwd+update; large-batch bf16 path: 4 = the backward's row kernel alone,
// 5 = its GEMM kernel alone.  Synchronous.
extern "C" int iqlhip_debug_time_kernel(iqlhip_ctx* c, const iqlhip_batch* b, int which, int repeat, float* avg_us,
                                        void* stream) {
  if (!c || !avg_us || repeat < 1) return fail(IQLHIP_EINVAL, "bad argument");
  int rc = check_batch(c, b);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  c->cont.valid = false;
  const float* xb_cur = nullptr;
  rc = stage_batch(c, b, st, &xb_cur);
  if (rc) return rc;
  iqlhip_step_scalars sc;
  memset(&sc, 0, sizeof sc);
  sc.bc2_sqrt[0] = sc.bc2_sqrt[1] = sc.bc2_sqrt[2] = 1.f;
  sc.beta2 = 1.f; sc.eps = 1e-8f; sc.grad_scale = 1.f; sc.inv_batch = 1.f / b->rows;
  StepParams p = make_step(c, b->rows, sc.inv_batch);
  p.xb = xb_cur;
  UpdParams u = make_upd(c, &sc, b->rows, nullptr);
  u.tau = 0.f; u.one_minus_tau = 1.f;
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  refresh_shadows(c, st);
  for (int w = 0; w < 3; ++w) { launch_fwd(c, p, st); launch_bwd(c, p, st); }
  HIPCHK(hipEventRecord(e0, st));
  c->lb_bwd_part = (which == 4) ? 1 : ((which == 5) ? 2 : 0);
  for (int i = 0; i < repeat; ++i) {
    if (which == 0 || which == 3) launch_fwd(c, p, st);
    if (which == 1 || which == 3 || which == 4 || which == 5) launch_bwd(c, p, st);
    if (which == 2 || which == 3) launch_upd(c, u, st);
  }
  c->lb_bwd_part = 0;
  HIPCHK(hipEventRecord(e1, st));
  HIPCHK(hipEventSynchronize(e1));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  *avg_us = ms * 1e3f / repeat;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return IQLHIP_OK;
}

// ---------------------------------------------------------------------------
extern "C" int iqlhip_debug_read(iqlhip_ctx* c, const char* name, float* host_out, int64_t max_floats, int64_t* n_out,
                                 void* stream) {
  if (!c || !name || !host_out) return fail(IQLHIP_EINVAL, "NULL argument");
  hipStream_t st = (hipStream_t)stream;
  const float* src = nullptr;
  int64_t n = 0;
  const int MB = c->dims.max_batch;
  if (!strcmp(name, "h0")) { src = c->sc.h0; n = (int64_t)4 * MB * HID; }
  else if (!strcmp(name, "h1")) { src = c->sc.h1; n = (int64_t)4 * MB * HID; }
  else if (!strcmp(name, "heads")) { src = c->sc.heads; n = (int64_t)MB * HEAD_LD + (int64_t)NSPLIT * MB * c->dims.action_dim; }
  else if (!strcmp(name, "loss_parts")) { src = c->sc.loss_parts; n = 4 * 64; }
  else if (!strcmp(name, "drop_bits")) { src = (const float*)c->drop_bits; n = (int64_t)2 * MB * 8; }
  else if (!strcmp(name, "stamps")) {   // 64-bit stamps returned as pairs of 32-bit words
    if (!c->stamps) return fail(IQLHIP_EINVAL, "library built without -DIQL_STAMPS");
    src = (const float*)c->stamps; n = 4096 * 16 * 2;
  }
  else if (!strcmp(name, "grads")) {
    // flatten the slabs of the LAST forward_backward/step with the batch size implied by max_batch slabs in use
    return fail(IQLHIP_EINVAL, "use iqlhip_forward_backward to obtain the flat gradient");
  } else return fail(IQLHIP_EINVAL, "unknown scratch array '%s'", name);
  if (n > max_floats) n = max_floats;
  HIPCHK(hipMemcpyAsync(host_out, src, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  if (n_out) *n_out = n;
  return IQLHIP_OK;
}
