// iqlhip_kernels.h — device code of the IQL step for gfx950 (MI355X, CDNA4).
//
// One step = three launches (cut at every all-to-all seam):
//   iql_fwd_kernel     7 MLP instances x row-tiles(32 rows) x 4 column slices
//   iql_bwd_kernel     (a) dW1 tiles over a 256-row chunk, (b) dH0/dW0 per row-tile
//   iql_update_kernel  slab-sum of gradients + Adam (3 lr groups) + Polyak + losses
//                      (+ extra blocks that gather the NEXT step's rows into the compact batch)
//
// The step is latency-bound at B=256 (0.54 GFLOP against ~20 dependent memory
// round trips), so every kernel is written as: issue ALL global loads of the
// block first -> one wait -> LDS staging -> MFMA phases fed from LDS/registers.
//
// All GEMMs use v_mfma_f32_16x16x4_f32 (exact fp32 fma chain).  Lane maps
// (wave64, l = lane, l15 = l&15, g = l>>4):
//   A operand: A[m=l15][k=g]      B operand: B[k=g][n=l15]
//   C/D:       D[m=4g+reg][n=l15] (reg = 0..3)
// "float4-along-k": a lane loads 4 consecutive k of its row once and feeds element t to
// the t-th of 4 MFMAs — A and B use the same (g,t)->k map, so 4 MFMAs cover 16 k once.
// "float4-along-n": a lane loads 4 consecutive output columns; MFMA t produces columns {4n+t}.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "iqlhip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte access at 4-byte alignment

#define HID 256
#define RT_ROWS 32          // rows per forward / (b) block
#define CHUNK_ROWS 256      // rows per (a) block chunk
#define H0_LD 260           // LDS row stride of a [rows][256] tile (16-B aligned, bank-shifted)
#define H0B_LD 264          // ... of the same tile kept in bf16 (bf16 path): 528-byte rows, 16 rows cover all 64 banks
#define T64_LD 68           // LDS row stride of a [rows][64] tile
#define NSPLIT 4            // column slices of the hidden layer per row tile
#define HEAD_LD 24          // scalar head partials per row: [inst 0..5][ns 0..3]
#define W0_LDS_MAX_K 64     // layer-0 weights are staged in LDS through registers when k_in <= this (64 KiB)
#define W0_DMA_MAX_K 96     // ... and by LDS-DMA up to this width, LDS permitting (host: iqlhip_create)
#define W2_LD 68            // row stride of the forward's head-weight tile in LDS (64 units + 4: bank spread)

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// bf16-operand variant of the large products (layer 0 / layer 1 forward, dW1, dH0, dW0): fp32 accumulate, fp32
// master weights / activations in memory; operands are rounded to bf16 in registers (v_cvt_pk_bf16_f32) and
// eight fp32 MFMAs (8 x 4 k) collapse into one v_mfma_f32_16x16x32_bf16 (32 k).  The operand lane map of the
// bf16 instruction is "8 k per lane"; since both operands are built from the same (lane, element) -> k
// assignment, the loads are exactly those of the fp32 path.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
__device__ __forceinline__ bf16x8 pack8(const f32x4 lo, const f32x4 hi) {
  bf16x8 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) { r[i] = (__bf16)lo[i]; r[4 + i] = (__bf16)hi[i]; }
  return r;
}
__device__ __forceinline__ bf16x8 pack8s(const float (&v)[8]) {
  bf16x8 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = (__bf16)v[i];
  return r;
}
// bf16 STORAGE on the bf16 path (round 3): W1 is read from a bf16 shadow of the fp32 master weights (written by the
// update kernel next to the master, refreshed from it at the start of every library call), and the activations H0 / H1
// the backward re-reads live in memory as bf16.  Every (lane, element) -> index map is that of the fp32 kernels: a
// lane's load shrinks from 16 to 8 bytes (4 elements) or from 8 to 4 (2 elements) — half the bytes through the CU's
// vector-memory pipe, which is what these kernels wait for — and the operands need no conversion any more.
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
template <bool BF16> struct Frag4 { typedef f32x4 type; };
template <> struct Frag4<true> { typedef bf16x4 type; };
template <bool BF16> struct Frag2 { typedef f32x2 type; };
template <> struct Frag2<true> { typedef bf16x2 type; };
// 4 / 2 consecutive elements at element offset `off` of an array whose element type is float (fp32 path) or __bf16
// (bf16 path; `base` then is the bf16 array's address carried in a float pointer)
template <bool BF16> __device__ __forceinline__ typename Frag4<BF16>::type ld4(const float* base, unsigned off) {
  if constexpr (BF16) return *(const bf16x4*)((const __bf16*)base + off);
  else return *(const f32x4*)(base + off);
}
template <bool BF16> __device__ __forceinline__ typename Frag2<BF16>::type ld2(const float* base, unsigned off) {
  if constexpr (BF16) return *(const bf16x2*)((const __bf16*)base + off);
  else return *(const f32x2*)(base + off);
}
template <bool BF16> __device__ __forceinline__ void st4(float* base, unsigned off, const f32x4 v) {
  if constexpr (BF16) {
    bf16x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = (__bf16)v[i];
    *(bf16x4*)((__bf16*)base + off) = r;
  } else {
    *(f32x4*)(base + off) = v;
  }
}
__device__ __forceinline__ bf16x8 cat8(const bf16x4 lo, const bf16x4 hi) {
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

struct DevScratch {
  float* h0;        // [4][max_batch][256]  post-ReLU layer-0 activations of V(s),Q1,Q2,pi
  float* h1;        // [4][max_batch][256]
  float* heads;     // [max_batch][HEAD_LD] scalar partials (bias folded into slice 0), then pi: [max_batch][A][NSPLIT]
  float* slab_a;    // [n_chunk_max][n_params]  chunk slabs: w1,b1,w2,b2,log_std grads
  float* slab_b;    // per net [n_rt_max][256*k_in+256]  row-tile slabs: w0,b0 grads
  float* loss_parts;// [4][64]: value, q1, q2 (err^2 sums), actor — per chunk
  float* losses;    // [4]
  long long slab_b_off[4];  // float offset of net's region in slab_b
  int max_batch;
};

// Diagnostic build only (-DIQL_STAMPS): lane 0 of every block writes s_memtime at phase
// boundaries to a buffer of its own; no product path reads it.
#ifdef IQL_STAMPS
__device__ __forceinline__ unsigned long long iql_memtime() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define STAMP(p, i)                                                                        \
  do {                                                                                     \
    const unsigned long long t_ = iql_memtime();                                           \
    if (stamps_ && threadIdx.x == 0 && blockIdx.x < stamps_blocks_) stamps_[(long long)blockIdx.x * 16 + (i)] = t_; \
  } while (0)
// (stamps_ is a LOCAL copy of p.stamps: modifying the by-value kernel-argument struct itself makes the compiler
//  copy all of it to scratch at entry — 1.1 KB per thread, ~3 us — which is what an earlier stamps build measured)
// (the buffer holds 4 096 blocks x 16 stamps: a launch with more blocks than fit behind its base stamps only the first ones)
#define STAMP_BASE(p, off) unsigned long long* const stamps_ = (p).stamps ? (p).stamps + (off) : nullptr; \
  const unsigned stamps_blocks_ = (unsigned)((4096 * 16 - (off)) / 16)
// the constant 100 MHz clock shared by the whole chip (s_memtime counters are local and not comparable across blocks)
__device__ __forceinline__ unsigned long long iql_realtime() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define RT_ENTRY() const unsigned long long rt_entry_ = iql_realtime()
#define RT_STAMP(p, i, t)                                                                  \
  do {                                                                                     \
    if (stamps_ && threadIdx.x == 0 && blockIdx.x < stamps_blocks_) stamps_[(long long)blockIdx.x * 16 + (i)] = (t); \
  } while (0)
#else
#define STAMP(p, i) do {} while (0)
#define STAMP_BASE(p, off) do {} while (0)
#define RT_ENTRY() do {} while (0)
#define RT_STAMP(p, i, t) do {} while (0)
#define iql_realtime() 0ull
#define rt_entry_ 0ull
#endif

// Direct pointers of one MLP (host-built, so a block needs ONE scalar-load round to find its weights).
struct NetPtrs {
  const float *w0, *b0, *w1, *b1, *w2, *b2;
  int k0, d;
};
// Offsets of one net's gradient tensors inside a chunk slab (= arena layout).
struct NetGrad {
  long long w1, b1, w2, b2, log_std;
};

// The batch is ALWAYS the packed staging buffer xb: rows [s(S) | a(A) | s'(S) | r | d | pad], stride ld.
struct StepParams {
  unsigned long long* stamps;
  NetPtrs inst[8];     // forward instances 0..6: V(s'), V(s), Qt1, Qt2, Q1, Q2, pi
  int xoff[8];         // first column of the instance's input inside a packed row (0 or S+A)
  int slot[8];         // activation slot (0..3) of trainable instances, -1 otherwise
  NetPtrs net[4];      // trainable nets V, Q1, Q2, pi (backward)
  NetGrad go[4];
  const float* log_std;   // Gaussian policy log_std (A floats) or nullptr
  iqlhip_hyper hy;
  DevScratch sc;
  const float* xb;
  int ld, rows;
  int S, A, policy;
  float inv_batch;
  long long n_params;
  // actor dropout (nn.Dropout(p) after each hidden ReLU of the policy MLP, iql.py:331-333): keep-bits
  // [2 layers][max_batch][8 words] (bit j of word w = unit 32w + j), scale = 1/(1-p); null when off
  const unsigned* drop_bits;
  float drop_scale;
  // >= 0: forward ONE instance only (grid = n_rt * NSPLIT, blockIdx = row tile * NSPLIT + column slice) — the
  // policy-inference entry point iqlhip_actor_forward; -1: the training forward over all 7 instances
  int only_inst;
  int w0_lds_k;             // instances with k_in <= this stage their layer-0 weights in LDS
  // Bits 0..1: log2 of the column slices a forward block walks (0, 1, 2): grid = 8 x row tiles x (NSPLIT >> that).  Batches of
  // more row tiles than the chip has CUs take 2 or 4 slices per block: layer 0 and the block prologue are then paid
  // once per 2 / 4 slices instead of once per slice (host: launch_fwd).
  // Bits 2..3: the same for the backward's (b) blocks (dH0 / dW0 of a row tile): the dY / dH1 tile of the 32 rows is
  // built once and 2 or 4 of the 64-column slices are walked with it (host: launch_bwd).  (One word for both: a
  // second one grew the kernel-argument block past 0x478 bytes and every backward launch took 0.17 us longer.)
  int spb_l2;
  // hipGraph chunks: what the forward's idle blocks (blockIdx & 7 == 7, one per row tile and column slice) do while the 7
  // instances run — stage the NEXT step's rows (indices drawn on the spot) into the other staging buffer, copy this
  // step's optimiser scalars into place, draw the next step's dropout keep-bits.  One pointer to a device-resident
  // record (frozen per captured step) instead of the fields themselves: the kernel-argument block stays below the
  // size at which every backward launch was measured 0.17 us slower (0x480 bytes).  Null: nothing to do.
  const struct IdleWork* g_work;
};

// Force kernel-argument fields into SGPRs NOW.  hipcc sinks each s_load next to its first use, which
// turns one scalar-cache miss (~1k cycles at kernel start) into 3-5 dependent ones; an empty asm that
// "uses" the values makes the compiler issue all the loads in one batch behind a single wait.
#define PIN_S(x) asm volatile("" ::"s"(x))
#define PIN_P(x) asm volatile("" ::"s"((unsigned long long)(uintptr_t)(x)))

// ---------------------------------------------------------------------------
// A 32-row tile of packed rows is staged whole (flat float4 copy).  The packed stride 2S+A+2 (padded to 4)
// reaches 260 floats at the dimension limits, i.e. up to 9 float4 per thread; the common dims need 2-4, so the
// copy is instantiated for 3 / 5 / 9 (straight-line loads each: a load under a run-time trip count would be
// waited for individually).
#define XR_MAX_F4 9
#define XR_LD_MAX 260
template <int Q0, int Q1>
__device__ __forceinline__ void xr_issue(f32x4 (&xr)[XR_MAX_F4], const float* xb, int first_f4, int n_x, int x_last) {
#pragma unroll
  for (int q = Q0; q < Q1; ++q) {
    const int f = min(first_f4 + min((int)threadIdx.x + 256 * q, n_x - 1), x_last);
    xr[q] = *(const f32x4*)(xb + 4u * (unsigned)f);
  }
}
__device__ __forceinline__ void xr_load(f32x4 (&xr)[XR_MAX_F4], const float* xb, int first_f4, int n_x, int x_last) {
#pragma unroll
  for (int q = 0; q < XR_MAX_F4; ++q) xr[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
  xr_issue<0, 3>(xr, xb, first_f4, n_x, x_last);
  if (n_x > 3 * 256) {
    xr_issue<3, 5>(xr, xb, first_f4, n_x, x_last);
    if (n_x > 5 * 256) xr_issue<5, 9>(xr, xb, first_f4, n_x, x_last);
  }
}
__device__ __forceinline__ void xr_store(const f32x4 (&xr)[XR_MAX_F4], float* Xr, int n_x) {
#pragma unroll
  for (int q = 0; q < XR_MAX_F4; ++q) {
    const int f = (int)threadIdx.x + 256 * q;
    if (f < n_x) *(f32x4*)(Xr + 4 * f) = xr[q];
  }
}

__device__ __forceinline__ void lds_dma16(const float* gsrc, float* lds_wave_base) {
  // 64 lanes x 16 B: global (per-lane address) -> LDS (wave-uniform base + lane*16), no VGPR staging
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// ---------------------------------------------------------------------------
// rows[idx[r]] -> xb[r], r < n: one float4 per thread-iteration (rows are 16-B aligned, ld % 4 == 0)
// n_rows > 0: an index outside [0, n_rows) is never dereferenced — its output row is filled with NaN instead (the
// reference's torch indexing raises IndexError, iql.py:173-177; the Python shim raises it on the host before the
// launch — this guard only makes sure that a caller of the C ABI cannot fault the GPU with a bad index).
__device__ __forceinline__ void gather_rows_flat(const float* rows, long long ld, const long long* idx, float* xb,
                                                 int n, int first, int stride, long long n_rows = 0) {
  const int q = (int)(ld >> 2);
  const int total = n * q;
  for (int e = first; e < total; e += stride) {
    const int r = e / q, c4 = e - r * q;
    const long long i = idx[r];
    f32x4 v;
    if (n_rows > 0 && (i < 0 || i >= n_rows)) {
      const float nanv = __builtin_nanf("");
      v = (f32x4){nanv, nanv, nanv, nanv};
    } else {
      v = *(const f32x4*)(rows + i * ld + 4 * c4);
    }
    *(f32x4*)(xb + (long long)r * ld + 4 * c4) = v;
  }
}

// ---------------------------------------------------------------------------
// Device words a run of steps (hipGraph chunks) reads its per-launch values from: kernel arguments of a captured graph
// are frozen, these words are not.  iql_call_setup_kernel writes them ONCE per iqlhip_train_steps call; every chunk
// then advances them itself (thread 0 of its last update kernel: POS, DROP_STEP, BASE, XSTEP), so the chunks of a
// call chain on the device with no host-side launch between them.
//   SIZE      rows the index draw covers           SEED / OFFSET  Philox key / the call's first counter
//   POS       indices drawn before this chunk (index j of the call = counter OFFSET + j / 2, word pair j & 1)
//   BASE      steps of the call before this chunk (row of the call's scalar table, slot of the loss ring)
//   DROP_*    dropout stream                        XSTEP          steps exchanged before this chunk (P2P flags)
enum { HDR_SIZE = 0, HDR_SEED = 1, HDR_OFFSET = 2, HDR_DROP_STEP = 3, HDR_DROP_SEED = 4, HDR_BASE = 5, HDR_XSTEP = 6,
       HDR_POS = 7, HDR_WORDS = 8 };
struct ChunkHdr { unsigned long long w[HDR_WORDS]; };

// Philox4x32-10 (Salmon et al. 2011).
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

// Dropout keep-bits for one step: word w (of n_words = 2 * max_batch * 8) gets 32 independent Bernoulli(1-p)
// bits: keep iff u32 >= thresh (thresh = p * 2^32).  Stream: key = seed, counter = (word, call, step).
__device__ __forceinline__ void dropmask_words(unsigned* bits, int n_words, unsigned thresh, unsigned long long seed,
                                               unsigned long long step, int first, int stride) {
  for (int w = first; w < n_words; w += stride) {
    unsigned word = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      uint32_t c[4] = {(uint32_t)w, (uint32_t)j | 0x44524F50u /* "DROP" */, (uint32_t)step, (uint32_t)(step >> 32)};
      philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
      for (int q = 0; q < 4; ++q) word |= (c[q] >= thresh ? 1u : 0u) << (4 * j + q);
    }
    bits[w] = word;
  }
}

// Index j of a call's draw: Philox4x32-10, counter = ctr0 + j / 2, key = seed; the counter's four words give two
// indices, uniform over [0, size) by multiply-high of 64 random bits (bias <= size / 2^64) — the stream
// iql_draw_indices_kernel writes out (np.random.randint's distribution, iql.py:172).
__device__ __forceinline__ long long draw_index(unsigned long long seed, unsigned long long ctr0, unsigned long long j,
                                                unsigned long long size) {
  const unsigned long long ctr = ctr0 + (j >> 1);
  uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0x49514C48u /* "IQLH" */, 0u};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  const unsigned long long r = (j & 1ull) ? (((unsigned long long)c[3] << 32) | c[2]) : (((unsigned long long)c[1] << 32) | c[0]);
  return (long long)__umul64hi(r, size);
}

// rows[draw(j0 + r)] -> xb[r], r < n: the gather of a step whose indices nobody stores — every thread of a row's
// float4s draws that row's index itself (~150 ALU instructions instead of a dependent idx -> row round trip to HBM).
__device__ __forceinline__ void gather_rows_drawn(const float* rows, long long ld, float* xb, int n, unsigned long long seed,
                                                  unsigned long long ctr0, unsigned long long j0, unsigned long long size,
                                                  int first, int stride) {
  const int q = (int)(ld >> 2);
  const int total = n * q;
  for (int e = first; e < total; e += stride) {
    const int r = e / q, c4 = e - r * q;
    const long long i = draw_index(seed, ctr0, j0 + (unsigned long long)r, size);
    *(f32x4*)(xb + (long long)r * ld + 4 * c4) = *(const f32x4*)(rows + i * ld + 4 * c4);
  }
}

// What the idle eighth of a captured step's forward grid does (StepParams::g_work; one record per step of a chunk,
// written once when the chunk is captured):
struct IdleWork {
  const float* rows;                      // replay rows the NEXT step's batch is drawn from
  long long ld;
  float* xb_dst;                          // the staging buffer the next step reads (the other one of the two)
  const unsigned long long* hdr;
  const iqlhip_step_scalars* sched_call;  // the call's scalar table (device copy)
  iqlhip_step_scalars* sched_dst;         // where THIS step's update kernel reads its scalars (frozen address)
  unsigned* drop_dst;                     // keep-bits of the next step (the other parity's buffer); null: no dropout
  int drop_words;
  unsigned drop_thresh;
  int n;                                  // rows per step
  int k;                                  // this step's number inside the chunk
};
__device__ __forceinline__ void idle_block_work(const IdleWork* wk, int blk, int nblk) {
  const IdleWork w = *wk;
  const unsigned long long size = w.hdr[HDR_SIZE], seed = w.hdr[HDR_SEED], ctr0 = w.hdr[HDR_OFFSET], pos = w.hdr[HDR_POS];
  const unsigned long long base = w.hdr[HDR_BASE], dseed = w.hdr[HDR_DROP_SEED], dstep = w.hdr[HDR_DROP_STEP];
  // the next step's rows: index j = POS + (k + 1) n + r of the call (for the chunk's last step that is step 0 of
  // whatever runs next: the following chunk, or the next call when it continues this one's stream)
  gather_rows_drawn(w.rows, w.ld, w.xb_dst, w.n, seed, ctr0, pos + (unsigned long long)(w.k + 1) * (unsigned long long)w.n,
                    size, blk * 256 + (int)threadIdx.x, nblk * 256);
  // this step's optimiser scalars: row BASE + k of the call's table -> the slot this step's update kernel reads
  if (blk == nblk - 1 && threadIdx.x < sizeof(iqlhip_step_scalars) / sizeof(float))
    ((float*)w.sched_dst)[threadIdx.x] = ((const float*)(w.sched_call + base + (unsigned long long)w.k))[threadIdx.x];
  // the next step's dropout keep-bits (the last blocks first: the gather occupies the first ones)
  if (w.drop_dst)
    dropmask_words(w.drop_dst, w.drop_words, w.drop_thresh, dseed, dstep + (unsigned long long)(w.k + 1),
                   (nblk - 1 - blk) * 256 + (int)threadIdx.x, nblk * 256);
}

// Forward: block = (instance, row tile of 32 rows, column slice ns of 64 hidden-1 units).
// grid = 8 * n_rt * NSPLIT; blockIdx & 7 = instance (7 = idle) so that the
// blocks of one instance share an XCD and hence one L2 copy of its weights.
// W0DMA: the variant that may stage wide layer-0 weights by LDS-DMA.  A separate instantiation because the mere
// presence of an LDS-DMA makes the compiler wait vmcnt(0) before LDS reads on every path it may reach (measured:
// +1 us on the narrow-input configs, whose W1 prefetch then no longer streams under layer 0).
// Layer 0 with bf16 operands: the 8 k-steps of a chunk (lane group g holds k = 4 ks + g, ks = 0..7, of both operands)
// are one v_mfma_f32_16x16x32_bf16 per (row tile, unit tile): 8 MFMAs of 16 cycles instead of 64 of 32.
__device__ __forceinline__ void l0_chunk_bf16(f32x4 (&acc)[2][4], const float (&bq)[8][4], const float (&aq)[8][2]) {
  bf16x8 B[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    float t8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) t8[e] = aq[e][r];
    B[r] = pack8s(t8);
  }
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    float t8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) t8[e] = bq[e][ct];
    const bf16x8 A = pack8s(t8);
    acc[0][ct] = MFMA_BF16(A, B[0], acc[0][ct]);
    acc[1][ct] = MFMA_BF16(A, B[1], acc[1][ct]);
  }
}

// (two waves per SIMD — 256 registers, accumulators included — so that two blocks can share a CU when LDS allows)
// ONE: the policy-inference launch (one instance, p.only_inst).  A template flag rather than a run-time test of
// p.only_inst: the test was a scalar load + wait + branch in FRONT of the argument batch below — two dependent
// scalar-cache misses at the start of every training forward instead of one.
template <bool BF16, bool W0DMA, bool MULTI, bool ONE = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void iql_fwd_kernel(StepParams p) {
  RT_ENTRY();
  const int bid = blockIdx.x;
  // XCD-affine block map (consecutive workgroups go round the 8 XCDs: XCD x = blockIdx & 7).  Across a kernel boundary an
  // XCD reads back what it wrote ITSELF much faster than what another XCD wrote (profiles/r01_l2_retention_microbench.txt:
  // 11.7 vs 20.4 us for the same reads; the L2's FETCH_SIZE counters are the same either way — r03_pmc_summary.json — so
  // the difference is on the memory side of the L2).  So the three kernels agree on who touches what: XCDs n and n + 4 belong to net
  // n (V, Q1, Q2, pi) — the backward's blocks of net n run there, the update kernel's blocks there own the net's arena
  // segment in 64-float stripes (even stripes on XCD n, odd ones on n + 4; W1 [unit][k] leads the segment with 4 stripes
  // per row, so the k-slice i of W1 — a dW1 tile's columns, a (b) block's slice — is the stripes of parity i & 1), and
  // HERE the two forward instances that read net n's
  // weights (or their target copy) share those two XCDs by column slice: slices of parity h on XCD n + 4 h.
  //   XCD pair   0 / 4          1 / 5       2 / 6       3 / 7
  //   instances  V(s), V(s')    Q1, Qt1     Q2, Qt2     pi, idle            (which = bit 0 of the block's index on its XCD)
  // One-slice grids: the H0 columns a block saves are the ones the backward's dW1 tiles and (b) slices of the same parity
  // read on this XCD (its own W1 rows span all k: half of their stripes were written here, half on the partner XCD —
  // for every block alike, whatever the map).  Blocks that walk 2 slices take the
  // pair {2 h, 2 h + 1}; blocks that walk all 4 take the row tiles of parity h.
  const int fx = bid & 7, fh = fx >> 2, fr = bid >> 3;
  constexpr unsigned FWD_PAIR_A = 0x6541u, FWD_PAIR_B = 0x7320u;      // nibble (x & 3): V(s) Q1 Q2 pi | V(s') Qt1 Qt2 idle
  const int inst = ONE ? p.only_inst : (int)((((fr & 1) ? FWD_PAIR_B : FWD_PAIR_A) >> (4 * (fx & 3))) & 7u);
  if (inst >= 7) {     // the idle eighth of the grid: the chunk's bookkeeping for the NEXT step (graph chunks), else exits
    if (p.g_work) idle_block_work(p.g_work, (fr >> 1) * 2 + fh, (int)(gridDim.x >> 3));
    return;
  }
  const int spb_l2 = MULTI ? (p.spb_l2 & 3) : 0;      // (MULTI = false: exactly the one-slice code, no loop)
  const int spb = 1 << spb_l2;
  int ns, rt;
  if (ONE) {           // blockIdx = row tile * NSPLIT + column slice
    ns = bid & (NSPLIT - 1);
    rt = bid >> 2;
  } else if (spb_l2 == 0) {
    ns = 2 * ((fr >> 1) & 1) + fh;
    rt = fr >> 2;
  } else if (spb_l2 == 1) {
    ns = 2 * fh;
    rt = fr >> 1;
  } else {
    ns = 0;
    rt = 2 * (fr >> 1) + fh;
    if (rt * RT_ROWS >= p.rows) return;      // (odd row-tile counts: the grid is rounded up to pairs of row tiles)
  }
  const int row0 = rt * RT_ROWS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;

  const NetPtrs np = p.inst[inst];
  const int xoff = p.xoff[inst];
  const int slot = p.slot[inst];
  const int k0 = np.k0;
  const int k0p = (k0 + 3) & ~3;
  const int D = np.d;
  const int ld = p.ld;
  const int B = p.rows;
  const int w0k = p.w0_lds_k;
  const float* xb = p.xb;
  float* h0g = p.sc.h0;
  float* h1g = p.sc.h1;
  float* headsg = p.sc.heads;
  const int MB = p.sc.max_batch;
  const int Aact = p.A;
  PIN_P(np.w0); PIN_P(np.b0); PIN_P(np.w1); PIN_P(np.b1); PIN_P(np.w2); PIN_P(np.b2);
  PIN_S(k0); PIN_S(D); PIN_S(xoff); PIN_S(slot); PIN_S(ld); PIN_S(B); PIN_S(MB); PIN_S(Aact);
  PIN_P(xb); PIN_P(h0g); PIN_P(h1g); PIN_P(headsg); PIN_S(w0k);
  const bool w0_lds = W0DMA ? (k0 <= w0k) : (k0 <= min(w0k, W0_LDS_MAX_K));
  const bool w0_dma = W0DMA && w0_lds && (k0 > W0_LDS_MAX_K);      // wide inputs: copied by LDS-DMA, no staging registers

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* H0s = smem;                         // [32][H0_LD]
  float* H1s = H0s + RT_ROWS * H0_LD;        // [32][T64_LD]
  float* Xr = H1s + RT_ROWS * T64_LD;        // [32][ld]  packed rows of this tile
  // (regions sized by the ACTUAL dims, not the limits: at S=17/A=6 the block needs 75 KB instead of 109 KB, so two
  //  blocks fit a CU's 160 KB when a large batch brings more than one block per CU; host: fwd_lds_floats())
  // [Dp][W2_LD] head weights of this column slice (rows beyond D zero: MFMA operand), then b2[D]; D <= A
  float* W2s = Xr + RT_ROWS * ld;
  const int w2s_words = ((Aact + 15) & ~15) * W2_LD + 32;
  unsigned* Mk = (unsigned*)(W2s + w2s_words);       // [2][32][8] dropout keep-bits of the tile
  float* W0s = W2s + w2s_words + 512;        // [256*k0] flat copy of layer-0 weights (when w0_lds); 16-B aligned
  // (no integer casts on LDS pointers: they would demote every access to a flat load, and a flat load
  //  waits vmcnt(0) — it would drain the W1 stream that is meant to stay in flight under layer 0)

  STAMP_BASE(p, 0);
  STAMP(p, 0);
  // ======== issue every global load of the block.  vmcnt retires in issue order: the small operands of
  // layer 0 go first, the 64 KiB W1 slice last — it keeps streaming while layer 0 runs (no LDS-DMA
  // here: a DMA in flight would make __syncthreads() wait vmcnt(0), i.e. for W1 as well).
  // (a) the 32 packed input rows, contiguous in xb: n_x float4, clamped at the end of the batch
  const int n_x = RT_ROWS * ld / 4;
  const int x_last = B * ld / 4 - 1;
  f32x4 xr[XR_MAX_F4];
  xr_load(xr, xb, row0 * ld / 4, n_x, x_last);
  // (b) head weights of this slice + b2, biases
  f32x4 w2pre[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int e = min(tid + 256 * q, D * 16 - 1);
    w2pre[q] = *(const f32x4*)(np.w2 + (unsigned)((e >> 4) * HID + ns * 64 + 4 * (e & 15)));
  }
  const float b2v = np.b2[min(tid, D - 1)];
  f32x4 bias0[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) bias0[ct] = *(const f32x4*)(np.b0 + (unsigned)(wave * 64 + ct * 16 + 4 * g));
  f32x4 bias1 = *(const f32x4*)(np.b1 + (unsigned)(ns * 64 + wave * 16 + 4 * g));
  // (b2) dropout keep-bits of this row tile (policy instance only): thread -> (row tid >> 3, word tid & 7)
  const bool drop = (inst == 6) && (p.drop_bits != nullptr);
  unsigned mk0 = 0xFFFFFFFFu, mk1 = 0xFFFFFFFFu;
  if (drop) {
    const int mrow = min(row0 + (tid >> 3), B - 1);
    mk0 = p.drop_bits[mrow * 8 + (tid & 7)];
    mk1 = p.drop_bits[(MB + mrow) * 8 + (tid & 7)];
  }
  // (c) layer-0 weights: flat float4 copy of 64*k0 float4 (thread handles tid + 256 j); 8 loads cover k0 <= 32
  const int n_w0v = 64 * k0;
  f32x4 w0v[16];
  if (w0_dma) {
    // whole waves of 64 x 16 B: global (per-lane address, clamped) -> LDS (wave base + lane * 16); the tail wave
    // writes into the region's 4 KiB slack.  The barrier below then waits for every outstanding load (the DMA is
    // tracked by vmcnt), so on this path the W1 fragments are requested after it and stream in under layer 0.
    const int nj = (n_w0v + 255) >> 8;
    for (int j = 0; j < nj; ++j)
      lds_dma16(np.w0 + 4 * min(tid + 256 * j, n_w0v - 1), W0s + 4 * (256 * j + 64 * wave));
  } else if (w0_lds) {
#pragma unroll
    for (int j = 0; j < 8; ++j) w0v[j] = *(const f32x4*)(np.w0 + 4u * (unsigned)min(tid + 256 * j, n_w0v - 1));
    if (k0 > 32) {
#pragma unroll
      for (int j = 8; j < 16; ++j) w0v[j] = *(const f32x4*)(np.w0 + 4u * (unsigned)min(tid + 256 * j, n_w0v - 1));
    }
  }
  // (d) this wave's W1 rows (16 output units x 256 k) as MFMA fragments: 64 KiB per block
  int n1 = ns * 64 + wave * 16 + l15;  // hidden-1 unit of this lane
  // fp32: 16 fragments of 4 k (k = 16 ks + 4 g + t).  bf16 (np.w1 addresses the bf16 shadow of W1): 8 fragments of 8
  // CONTIGUOUS k (k = 32 j + 8 g + e) — the bf16 MFMA's native operand, one 16-byte load each; the H0 tile in LDS is
  // bf16 too and is read with the same map, one ds_read_b128 per operand, no conversion anywhere in layer 1.
  f32x4 bw[BF16 ? 1 : 16];
  bf16x8 bwb[BF16 ? 8 : 1];
  __bf16* H0b = (__bf16*)H0s;          // [32][H0B_LD] (bf16 path: the H0 tile lives here instead of H0s)

  xr_store(xr, Xr, n_x);
  const int Dp = (D + 15) & ~15;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int e = tid + 256 * q;
    if (e < Dp * 16) *(f32x4*)(W2s + (e >> 4) * W2_LD + 4 * (e & 15)) = (e < D * 16) ? w2pre[q] : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  if (tid < D) W2s[Dp * W2_LD + tid] = b2v;
  Mk[tid] = mk0;
  Mk[256 + tid] = mk1;
  if (w0_lds && !w0_dma) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int f = tid + 256 * j;
      if (f < n_w0v) *(f32x4*)(W0s + 4 * f) = w0v[j];
    }
    if (k0 > 32) {
#pragma unroll
      for (int j = 8; j < 16; ++j) {
        const int f = tid + 256 * j;
        if (f < n_w0v) *(f32x4*)(W0s + 4 * f) = w0v[j];
      }
    }
  }
  __syncthreads();
  // the W1 fragments are requested only now: 16 x 1 KB per wave of row-fragment loads take ~1.5 k cycles of the CU's
  // one vector-memory pipe (64 B/clk) just to ISSUE — in front of the barrier they delayed layer 0 by that much.  The
  // fp32 paths with LDS-staged weights go one step further and request them BETWEEN the groups of layer-0 MFMAs (an MFMA
  // holds the SIMD's issue for 8 of its 32 cycles: four loads per 8 MFMAs trickle out at 42 B/clk over the four waves),
  // so that not even the issue time stands in front of layer 0.
  const bool bw_in_l0 = !BF16 && w0_lds;
#define BW_LOAD(ks_) bw[ks_] = *(const f32x4*)(np.w1 + (unsigned)(n1 * HID + 16 * (ks_) + 4 * g))
#define BWB_AT(unit_, j_) (*(const bf16x8*)((const __bf16*)np.w1 + (unsigned)((unit_) * HID + 32 * (j_) + 8 * g)))
  if constexpr (BF16) {
#pragma unroll
    for (int j = 0; j < 8; ++j) bwb[j] = BWB_AT(n1, j);
  } else if (!bw_in_l0) {
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) BW_LOAD(ks);
  }
  STAMP(p, 1);

  // ---- layer 0: this wave computes H0[32][64*wave .. +64).  Operand roles: A = W0 (m = hidden unit),
  // B = X (n = row), so a lane's 4 accumulator registers are 4 consecutive hidden units of ONE row:
  // one ds_write_b128 into the row-major H0 tile.
  {
    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nks = k0p >> 2;
    const float* x0 = Xr + l15 * ld + xoff;
    const float* x1 = Xr + (16 + l15) * ld + xoff;
    float bcur[4], bnxt[4], acur[2], anxt[2];
    if (w0_lds && nks <= 8) {
      // all operands of the (<= 8) k-steps are read up front, then the MFMAs run back to back
      const float* wl[4];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) wl[ct] = W0s + (wave * 64 + ct * 16 + l15) * k0;
      // columns kk >= k0 of a packed row hold other fields: they are zeroed on the X side (2 selects per k-step, made
      // here, in the read phase); the weight operand is read with a clamped column and used as it is (finite x 0 = 0).
      // With the selects on the four weight operands the compiler sank each v_cndmask in front of its MFMA pair
      // (VALU write -> s_nop -> MFMA, 24 times): the 48 MFMAs of this phase took 2.25 k cycles instead of 1.5 k.
      float bq[8][4], aq[8][2];
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const int kk = 4 * ks + g;
        const int kc = min(kk, k0 - 1);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) bq[ks][ct] = wl[ct][kc];
        const float xa = x0[kc], xb_ = x1[kc];
        aq[ks][0] = (kk < k0) ? xa : 0.f;
        aq[ks][1] = (kk < k0) ? xb_ : 0.f;
      }
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {      // (pin the selected values: no re-evaluation next to the MFMAs)
        asm volatile("" : "+v"(aq[ks][0]), "+v"(aq[ks][1]));
      }
      STAMP(p, 5);
      if constexpr (BF16) {       // (k-steps beyond nks: X side selected to zero above, weight side a clamped finite value)
        l0_chunk_bf16(acc, bq, aq);
      } else {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          if (ks < nks) {
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
              acc[0][ct] = MFMA16(bq[ks][ct], aq[ks][0], acc[0][ct]);
              acc[1][ct] = MFMA16(bq[ks][ct], aq[ks][1], acc[1][ct]);
            }
          }
          if (ks < 4) {       // W1 fragments 4 ks .. 4 ks + 3 behind this group of MFMAs
#pragma unroll
            for (int k2 = 4 * ks; k2 < 4 * ks + 4; ++k2) BW_LOAD(k2);
          }
        }
      }
      STAMP(p, 6);
    } else if (w0_lds) {
      // wide inputs (9..24 k-steps): chunks of 8 k-steps in straight-line code, the operands of a chunk read in one
      // batch like above and the next chunk's batch issued before this chunk's MFMAs.  (As a run-time loop with a
      // one-step look-ahead the compiler waited for each step's six reads in front of its eight MFMAs: 550 cycles
      // per k-step instead of 256.)  Only the last k-step can reach beyond k0; the selects are made per batch.
      const float* wl[4];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) wl[ct] = W0s + (wave * 64 + ct * 16 + l15) * k0;
      float bqA[8][4], aqA[8][2], bqB[8][4], aqB[8][2];
      auto rd = [&](float (&bq)[8][4], float (&aq)[8][2], const int base) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (base + 4 * h < nks) {
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
              const int ks = 4 * h + k4;
              const int kk = 4 * (base + ks) + g;
              const int kc = min(kk, k0 - 1);
#pragma unroll
              for (int ct = 0; ct < 4; ++ct) bq[ks][ct] = wl[ct][kc];
              const float xa = x0[kc], xb_ = x1[kc];
              aq[ks][0] = (kk < k0) ? xa : 0.f;
              aq[ks][1] = (kk < k0) ? xb_ : 0.f;
            }
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) asm volatile("" : "+v"(aq[4 * h + k4][0]), "+v"(aq[4 * h + k4][1]));
          } else if (BF16) {      // the bf16 MFMA takes all 8 k-steps of a chunk: the unread half contributes zeros
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
              aq[4 * h + k4][0] = 0.f;
              aq[4 * h + k4][1] = 0.f;
#pragma unroll
              for (int ct = 0; ct < 4; ++ct) bq[4 * h + k4][ct] = 0.f;
            }
          }
        }
      };
      auto mm = [&](const float (&bq)[8][4], const float (&aq)[8][2], const int base) {
        if constexpr (BF16) {
          l0_chunk_bf16(acc, bq, aq);
        } else {
#pragma unroll
          for (int ks = 0; ks < 8; ++ks) {
            if (base + ks < nks) {
#pragma unroll
              for (int ct = 0; ct < 4; ++ct) {
                acc[0][ct] = MFMA16(bq[ks][ct], aq[ks][0], acc[0][ct]);
                acc[1][ct] = MFMA16(bq[ks][ct], aq[ks][1], acc[1][ct]);
              }
            }
            if (base == 0) {    // the W1 fragments, two behind each MFMA group of the first chunk
              BW_LOAD(2 * ks);
              BW_LOAD(2 * ks + 1);
            }
          }
        }
      };
      rd(bqA, aqA, 0);
      rd(bqB, aqB, 8);
      STAMP(p, 5);
      mm(bqA, aqA, 0);
      if (nks > 16) rd(bqA, aqA, 16);
      mm(bqB, aqB, 8);
      if (nks > 16) mm(bqA, aqA, 16);
      STAMP(p, 6);
    } else {
      const float* wrow[4];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) wrow[ct] = np.w0 + (wave * 64 + ct * 16 + l15) * k0;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) { const float v = wrow[ct][min(g, k0 - 1)]; bcur[ct] = (g < k0) ? v : 0.f; }
      for (int ks = 0; ks < nks; ++ks) {
        const int kn = 4 * (ks + 1) + g;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) { const float v = wrow[ct][min(kn, k0 - 1)]; bnxt[ct] = (kn < k0) ? v : 0.f; }
        const int kc = min(4 * ks + g, k0 - 1);
        const float a0 = x0[kc];
        const float a1 = x1[kc];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          acc[0][ct] = MFMA16(bcur[ct], a0, acc[0][ct]);
          acc[1][ct] = MFMA16(bcur[ct], a1, acc[1][ct]);
        }
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) bcur[ct] = bnxt[ct];
      }
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
#pragma unroll
      for (int rtile = 0; rtile < 2; ++rtile) {
        f32x4 h;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) h[reg] = fmaxf(acc[rtile][ct][reg] + bias0[ct][reg], 0.f);
        if (drop) {   // units wave*64 + ct*16 + 4g .. +3 of row rtile*16 + l15
          const unsigned bits = Mk[(rtile * 16 + l15) * 8 + wave * 2 + (ct >> 1)] >> ((ct & 1) * 16 + 4 * g);
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) h[reg] = ((bits >> reg) & 1u) ? h[reg] * p.drop_scale : 0.f;
        }
        if constexpr (BF16) {
          bf16x4 hb_;
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) hb_[reg] = (__bf16)h[reg];
          *(bf16x4*)(H0b + (rtile * 16 + l15) * H0B_LD + wave * 64 + ct * 16 + 4 * g) = hb_;
        } else {
          *(f32x4*)(H0s + (rtile * 16 + l15) * H0_LD + wave * 64 + ct * 16 + 4 * g) = h;
        }
      }
    }
  }
  STAMP(p, 7);
  __syncthreads();
  STAMP(p, 2);

  // ======== per column slice: layer 1 over the block's H0 tile, head partials.  One pass when the grid holds a block
  // per slice; 2 or 4 passes for large batches — the next slice's W1 fragments, head weights and bias are requested
  // right after this slice's layer-1 MFMAs and arrive under its head phase.
  for (int it = 0;; ++it) {
  const bool more = MULTI && (it + 1 < spb);
  f32x4 bias1n = bias1;
  if (more) {      // the next slice's head weights and bias: requested a whole layer 1 ahead of their LDS store at the
                   // end of this pass (requested after the layer they waited ~1 k cycles in front of that store)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int e = min(tid + 256 * q, D * 16 - 1);
      w2pre[q] = *(const f32x4*)(np.w2 + (unsigned)((e >> 4) * HID + (ns + 1) * 64 + 4 * (e & 15)));
    }
    bias1n = *(const f32x4*)(np.b1 + (unsigned)((ns + 1) * 64 + wave * 16 + 4 * g));
  }
  // save H0 columns [64*ns, +64) of the trainable instances for the backward pass
  if (slot >= 0) {
    const int rl = tid >> 3;
    const int row = row0 + rl;
    if (row < B) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = ns * 64 + 4 * ((tid & 7) + 8 * j);
        if constexpr (BF16)
          *(bf16x4*)((__bf16*)h0g + (unsigned)((slot * MB + row) * HID + col)) = *(const bf16x4*)(H0b + rl * H0B_LD + col);
        else
          *(f32x4*)(h0g + (unsigned)((slot * MB + row) * HID + col)) = *(const f32x4*)(H0s + rl * H0_LD + col);
      }
    }
  }

  if (it == 0) STAMP(p, 8);
  // ---- layer 1: this wave computes H1[32][16 units]; A = W1 fragments (m = unit), B = H0 (n = row)
  {
    f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    if constexpr (BF16) {
      // all 16 operand reads of the tile first (one wait), then the 16 MFMAs back to back: written as read -> convert ->
      // MFMA per k-block the loop ran at one LDS latency + 8 conversions per pair of MFMAs (3.7 k cycles per slice at
      // 1 024 rows against 256 cycles of matrix work, profiles/r03_stamps_config5_1024_bf16.txt)
      bf16x8 b0[8], b1[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        b0[j] = *(const bf16x8*)(H0b + l15 * H0B_LD + 32 * j + 8 * g);
        b1[j] = *(const bf16x8*)(H0b + (16 + l15) * H0B_LD + 32 * j + 8 * g);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {     // one bf16 MFMA per 32 k: lane (g) supplies k = 32 j + 8 g .. + 7 of both operands
        acc0 = MFMA_BF16(bwb[j], b0[j], acc0);
        acc1 = MFMA_BF16(bwb[j], b1[j], acc1);
        if ((j & 1) && more) {      // the next slice's fragments replace the two just used
          bwb[j - 1] = BWB_AT(n1 + 64, j - 1);
          bwb[j] = BWB_AT(n1 + 64, j);
        }
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const f32x4 a0 = *(const f32x4*)(H0s + l15 * H0_LD + 16 * ks + 4 * g);
        const f32x4 a1 = *(const f32x4*)(H0s + (16 + l15) * H0_LD + 16 * ks + 4 * g);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          acc0 = MFMA16(bw[ks][t], a0[t], acc0);
          acc1 = MFMA16(bw[ks][t], a1[t], acc1);
        }
        // more slices to come: the next slice's W1 fragments are requested into the registers whose MFMAs have just
        // been issued, four k-steps at a time — they arrive under the rest of this layer and the head phase
        if ((ks & 3) == 3 && more) {
#pragma unroll
          for (int k2 = ks - 3; k2 <= ks; ++k2)
            bw[k2] = *(const f32x4*)(np.w1 + (unsigned)((n1 + 64) * HID + 16 * k2 + 4 * g));
        }
      }
    }
    if (it == 0) STAMP(p, 9);
    f32x4 h0, h1;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      h0[reg] = fmaxf(acc0[reg] + bias1[reg], 0.f);
      h1[reg] = fmaxf(acc1[reg] + bias1[reg], 0.f);
    }
    if (drop) {   // hidden-1 units ns*64 + wave*16 + 4g .. +3 of rows l15 and 16 + l15
      const int word = ns * 2 + (wave >> 1), sh = (wave & 1) * 16 + 4 * g;
      const unsigned ba = Mk[256 + l15 * 8 + word] >> sh, bb_ = Mk[256 + (16 + l15) * 8 + word] >> sh;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        h0[reg] = ((ba >> reg) & 1u) ? h0[reg] * p.drop_scale : 0.f;
        h1[reg] = ((bb_ >> reg) & 1u) ? h1[reg] * p.drop_scale : 0.f;
      }
    }
    *(f32x4*)(H1s + l15 * T64_LD + wave * 16 + 4 * g) = h0;
    *(f32x4*)(H1s + (16 + l15) * T64_LD + wave * 16 + 4 * g) = h1;
  }
  if (it == 0) STAMP(p, 10);
  __syncthreads();
  if (it == 0) STAMP(p, 11);
  STAMP(p, 3);
  if (more) n1 += 64;

  {
    const int rl = tid >> 3;
    const int row = row0 + rl;
    const int sub = tid & 7;
    if (slot >= 0 && row < B) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int cl = 4 * (sub + 8 * j);
        st4<BF16>(h1g, (unsigned)((slot * MB + row) * HID + ns * 64 + cl), *(const f32x4*)(H1s + rl * T64_LD + cl));
      }
    }
    // ---- head partial sums over this block's 64 hidden-1 units (slice 0 also adds the bias)
    // thread (row rl, sub): units 4 sub..4 sub+3 and 32+4 sub..; its H1 values are read once, not once per dim
    const f32x4 ha = *(const f32x4*)(H1s + rl * T64_LD + 4 * sub);
    const f32x4 hb = *(const f32x4*)(H1s + rl * T64_LD + 32 + 4 * sub);
    if (D == 1) {
      const f32x4 wa = *(const f32x4*)(W2s + 4 * sub);
      const f32x4 wb = *(const f32x4*)(W2s + 32 + 4 * sub);
      float acc = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = fmaf(ha[e], wa[e], acc);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = fmaf(hb[e], wb[e], acc);
      acc += __shfl_xor(acc, 1);
      acc += __shfl_xor(acc, 2);
      acc += __shfl_xor(acc, 4);
      if (ns == 0) acc += W2s[Dp * W2_LD];
      if (sub == 0 && row < B) {
        if (inst < 6) headsg[row * HEAD_LD + inst * NSPLIT + ns] = acc;
        else headsg[MB * HEAD_LD + row * NSPLIT + ns] = acc;           // a policy with one action dim
      }
    } else {
      // policy head on the matrix cores: partial[32 rows][Dp] = H1s[32][64] x W2s^T — wave w takes row tile w & 1
      // and the 16 action dims of tile w >> 1 (waves beyond Dp / 16 tiles idle), 16 dependent MFMAs over the block's
      // 64 units.  A = H1 (m = row, k = unit), B = W2 (k = unit, n = dim, zero rows beyond D).  (As scalar code the
      // policy instance was the forward's long pole: ~500 cycles per action dim.)
      if (16 * (wave >> 1) < Dp) {
        const int i = wave & 1, nt = wave >> 1;
        float a[16], b[16];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
          a[ks] = H1s[(16 * i + l15) * T64_LD + 4 * ks + g];
          b[ks] = W2s[(16 * nt + l15) * W2_LD + 4 * ks + g];
        }
        f32x4 hacc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) hacc = MFMA16(a[ks], b[ks], hacc);
        const int dd = 16 * nt + l15;
        const float bias = (ns == 0) ? W2s[Dp * W2_LD + min(dd, D - 1)] : 0.f;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int prow_ = row0 + 16 * i + 4 * g + reg;
          if (dd < D && prow_ < B) headsg[MB * HEAD_LD + (prow_ * Aact + dd) * NSPLIT + ns] = hacc[reg] + bias;
        }
      }
    }
  }
  if (it == 0) STAMP(p, 12);
  if (!more) break;
  __syncthreads();      // every thread is done with this slice's H1s / W2s
  if (it == 0) STAMP(p, 13);
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int e = tid + 256 * q;
    if (e < D * 16) *(f32x4*)(W2s + (e >> 4) * W2_LD + 4 * (e & 15)) = w2pre[q];
  }
  bias1 = bias1n;
  ++ns;
  }   // (the next slice's H1s / W2s writes are ordered before their readers by the barrier after its layer 1)
  STAMP(p, 4);
  RT_STAMP(p, 14, rt_entry_);
  RT_STAMP(p, 15, iql_realtime());
}

// ---------------------------------------------------------------------------
// Loss gradients at the heads.  The per-row inputs are loaded first (RowIn, issue
// only) so that the caller can overlap them with its other loads; row_finish does
// the arithmetic.  Head = sum of the 4 column-slice partials in fixed order
// (the bias is already inside slice 0).
struct RowIn {
  f32x4 h[6];       // scalar head partials: inst 0..5 x ns 0..3
  float r, d;
};

// The scalar per-row loss inputs of one row (issue only).  The policy's per-(row, dim) inputs are loaded by the
// callers as coalesced (row, dim) work items.
// (offsets are formed as 32-bit UNSIGNED values on uniform base pointers: the load then takes the SGPR-base +
//  VGPR-offset form and needs no 64-bit address arithmetic on the vector ALU — with signed indices every load of
//  the backward's issue phase cost a sign extension and a 64-bit multiply-add, ~300 extra instructions per block)
__device__ __forceinline__ void row_issue(const StepParams& p, int row, RowIn& in) {
  const float* hb = p.sc.heads;
  const unsigned o = (unsigned)row * (unsigned)HEAD_LD;
#pragma unroll
  for (int i = 0; i < 6; ++i) in.h[i] = *(const f32x4*)(hb + (o + 4u * i));
  const float* xb = p.xb;
  const unsigned ox = (unsigned)row * (unsigned)p.ld + (unsigned)(2 * p.S + p.A);
  in.r = xb[ox];
  in.d = xb[ox + 1u];
}

// (the same with every input an explicit argument: the backward passes values that arrive preloaded in SGPRs)
__device__ __forceinline__ void row_issue_hot(const float* heads, const float* xb, int ld, int S, int A, int row, RowIn& in) {
  const unsigned o = (unsigned)row * (unsigned)HEAD_LD;
#pragma unroll
  for (int i = 0; i < 6; ++i) in.h[i] = *(const f32x4*)(heads + (o + 4u * i));
  const unsigned ox = (unsigned)row * (unsigned)ld + (unsigned)(2 * S + A);
  in.r = xb[ox];
  in.d = xb[ox + 1u];
}

__device__ __forceinline__ float sum4(const f32x4 v) { return ((v[0] + v[1]) + v[2]) + v[3]; }

// tanh(x) = 1 - 2 / (exp(2x) + 1) on the hardware transcendentals: exp2 (v_exp_f32, 1 ulp) of 2x*log2(e) and
// v_rcp_f32 (1 ulp).  Absolute error < 1e-7 (the argument rounding |2x log2e| * 6e-8 is damped by
// d tanh / d e = 2 / (e + 1)^2); saturates correctly: e -> inf gives 1, e -> 0 gives -1.  The policy blocks of the
// backward evaluate it 256 x A times EACH (every one needs every row's dY), so libm's expf + IEEE division
// (~35 instructions) were a measurable part of those blocks — the kernel's long pole.
__device__ __forceinline__ float tanh_via_exp(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f);     // exp(2x)
  return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
}

// row_finish: dL/d(head pre-activation) of a SCALAR net (V, Q1, Q2) for one row -> dyrow[0]; loss terms:
//   V: lossA = w*u^2     Q1/Q2: lossA = e1^2, lossB = e2^2
// (the policy's per-(row, dim) terms are formed in the callers as coalesced work items.)
// PiConst: per-action-dim constants of the Gaussian policy (clamped log_std, 1/var): lane dd holds dim dd's; callers
// fetch them with a lane shuffle, so all lanes of the wave must be active at that point.
struct PiConst { float ls, ivar; };
// The raw log_std word is loaded by pi_ls_issue() BEFORE the block's big prefetches: vmcnt retires in issue
// order, so a load issued after them would make the loss arithmetic wait for all of them.
__device__ __forceinline__ float pi_ls_issue(const StepParams& p) {
  const float* src = (p.policy == IQLHIP_POLICY_GAUSSIAN) ? p.log_std : p.xb;      // any valid address when unused
  return src[min((int)(threadIdx.x & 63), p.A - 1)];
}
__device__ __forceinline__ float pi_ls_issue_hot(const float* log_std, const float* xb, int policy, int A) {
  const float* src = (policy == IQLHIP_POLICY_GAUSSIAN) ? log_std : xb;      // any valid address when unused
  return src[min((int)(threadIdx.x & 63), A - 1)];
}
__device__ __forceinline__ PiConst pi_consts_hot(const StepParams& p, int policy, int net, float lsr) {
  PiConst c;
  c.ls = 0.f;
  c.ivar = 1.f;
  if (net == IQLHIP_NET_PI && policy == IQLHIP_POLICY_GAUSSIAN) {
    c.ls = fminf(fmaxf(lsr, p.hy.log_std_min), p.hy.log_std_max);
    const float sig = expf(c.ls);
    c.ivar = 1.f / (sig * sig);
  }
  return c;
}
__device__ __forceinline__ PiConst pi_consts(const StepParams& p, int net, float lsr) {
  PiConst c;
  c.ls = 0.f;
  c.ivar = 1.f;
  if (net == IQLHIP_NET_PI && p.policy == IQLHIP_POLICY_GAUSSIAN) {
    c.ls = fminf(fmaxf(lsr, p.hy.log_std_min), p.hy.log_std_max);
    const float sig = expf(c.ls);
    c.ivar = 1.f / (sig * sig);
  }
  return c;
}

__device__ __forceinline__ void row_finish(const StepParams& p, int net, const RowIn& in, float* dyrow,
                                           float& lossA, float& lossB) {
  const float invB = p.inv_batch;
  lossA = 0.f;
  lossB = 0.f;
  if (net == IQLHIP_NET_V) {
    const float tq = fminf(sum4(in.h[2]), sum4(in.h[3]));
    const float u = tq - sum4(in.h[1]);
    const float wgt = fabsf(p.hy.iql_tau - ((u < 0.f) ? 1.f : 0.f));
    lossA = wgt * u * u;
    dyrow[0] = (-2.f * wgt * u) * invB;
    return;
  }
  const float nv = sum4(in.h[0]);
  const float y = in.r + ((1.f - in.d) * p.hy.discount) * nv;
  const float e1 = sum4(in.h[4]) - y;
  const float e2 = sum4(in.h[5]) - y;
  lossA = e1 * e1;
  lossB = e2 * e2;
  dyrow[0] = ((net == IQLHIP_NET_Q1) ? e1 : e2) * invB;
}

__device__ __forceinline__ float block_sum_256(float v, float* red /*>=4 floats*/) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// ---------------------------------------------------------------------------
// Backward.  blockIdx & 7 = x: net = x & 3, half = x >> 2 (two XCD groups per net).
// Within a net, local id < 32*n_chunk  -> (a) block: chunk c, j-tile jt (32 rows of W1), i-tile it (64 cols)
//               otherwise              -> (b) block: row tile rt (32 rows), i-slice is (64 cols)
// dH1pre tiles of one wave's 64 rows: pre[t][ta] += dY[16t.., 4 k1 + g] x W2s[4 k1 + g][2 l15 + ta], NK k-steps
// (straight-line: all LDS operands of a k-step in one batch).
template <int NK>
__device__ __forceinline__ void dh1_mfma(f32x4 (&pre)[4][2], const float* dYs, const float* W2s, int DYA, int wave,
                                         int g, int l15) {
#pragma unroll
  for (int k1 = 0; k1 < NK; ++k1) {
    float a1[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) a1[t] = dYs[(64 * wave + 16 * t + l15) * DYA + 4 * k1 + g];
    const f32x2 b1 = *(const f32x2*)(W2s + (4 * k1 + g) * 32 + 2 * l15);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      pre[t][0] = MFMA16(a1[t], b1[0], pre[t][0]);
      pre[t][1] = MFMA16(a1[t], b1[1], pre[t][1]);
    }
  }
}

// Policy loss terms of 8 (row, dim) items — one action dim, eight rows — as straight-line code: both policy kinds are
// evaluated and selected on the (uniform) kind, dims beyond A on `live`; no branch, no memory access.
//   gaussian: -log N(a; mu, sigma) = q/2 + log_sigma + log(2 pi)/2 with q = (a - mu)^2 / sigma^2   (iql.py:527)
//   deterministic: (mu - a)^2                                                                     (iql.py:531)
__device__ __forceinline__ void pi_items8(const f32x4 (&hv)[8], const float (&acv)[8], const float (&wv)[8], bool live,
                                          bool gauss, float ivar, float ls, float invB, float (&dyv)[8], float (&dlv)[8],
                                          float& lossA) {
#pragma unroll
  for (int cc = 0; cc < 8; ++cc) {
    const float w = wv[cc];
    const float mu = tanh_via_exp(sum4(hv[cc]));
    const float diff = acv[cc] - mu;
    const float q = diff * diff * ivar;
    const float l_g = w * (0.5f * q + ls + 0.918938533204672742f);
    const float l_d = w * (diff * diff);
    const float dmu_g = (-(w * diff) * ivar) * invB;
    const float dmu_d = (-2.f * w * diff) * invB;
    const float dmu = gauss ? dmu_g : dmu_d;
    lossA += live ? (gauss ? l_g : l_d) : 0.f;
    dyv[cc] = live ? dmu * (1.f - mu * mu) : 0.f;
    dlv[cc] = (live && gauss) ? w * (1.f - q) : 0.f;
  }
}

// FULL: every row of every block's tile is a row of the batch (B % 256 == 0; the host selects the instantiation):
// no row index is clamped, so consecutive loads differ by compile-time constants.
#define BROW(r) (FULL ? (r) : min((r), B - 1))
// Kernel-argument PRELOAD (library built with -mllvm -amdgpu-kernarg-preload-count=14): the first 14 dwords of the
// explicit arguments arrive in SGPRs with the wave instead of through s_load from the freshly written argument block
// (measured 450-700 cycles until a field of the by-value StepParams is usable, 40 with preloading:
// profiles/r03_kernarg_latency_microbench.txt).  They are exactly what the block needs to ISSUE its first loads — the
// four scratch / batch pointers, the parameter arena (W2 and log_std addresses follow from the dims: arena_off below
// restates iqlhip_arena_layout) and the dims; everything else still comes from `p`, whose fetch now overlaps them.
//   q_dims = S | A << 8 | policy << 14      q_ldB = ld | rows << 10      q_mbc = max_batch | n_chunk << 16
//   q_rts  = n_rt | spb_l2 word << 10
struct ArenaOff { unsigned w0, b0, b1, w2, b2, log_std; int k0, d; };
__device__ __forceinline__ ArenaOff arena_off(int net, int S, int A, bool gauss) {
  auto seg = [&](int k, int d, bool ls) -> unsigned {
    const unsigned n = 65536u + 256u * (unsigned)k + 512u + 256u * (unsigned)d + (((unsigned)d + 3u) & ~3u) + (ls ? (((unsigned)A + 3u) & ~3u) : 0u);
    return (n + 63u) & ~63u;
  };
  const unsigned sV = seg(S, 1, false), sQ = seg(S + A, 1, false);
  const unsigned begin = (net == IQLHIP_NET_V) ? 0u : ((net == IQLHIP_NET_Q1) ? sV : ((net == IQLHIP_NET_Q2) ? sV + sQ : sV + 2u * sQ));
  ArenaOff o;
  o.k0 = (net == IQLHIP_NET_Q1 || net == IQLHIP_NET_Q2) ? S + A : S;
  o.d = (net == IQLHIP_NET_PI) ? A : 1;
  o.w0 = begin + 65536u;
  o.b0 = o.w0 + 256u * (unsigned)o.k0;
  o.b1 = o.b0 + 256u;
  o.w2 = o.b1 + 256u;
  o.b2 = o.w2 + 256u * (unsigned)o.d;
  o.log_std = o.b2 + (((unsigned)o.d + 3u) & ~3u);
  return o;
}
template <bool BF16, bool FULL, bool MULTI>
__global__ __launch_bounds__(256) void iql_bwd_kernel(const float* q_heads, const float* q_xb, const float* q_h1, const float* q_h0,
                                                      const float* q_params, unsigned q_dims, unsigned q_ldB, unsigned q_mbc,
                                                      unsigned q_rts, StepParams p) {
  RT_ENTRY();
  const int bid = blockIdx.x;
  const int x = bid & 7;
  const int h_S = (int)(q_dims & 255u), h_A = (int)((q_dims >> 8) & 63u), h_pol = (int)((q_dims >> 14) & 1u);
  const int h_ld = (int)(q_ldB & 1023u), h_rows = (int)(q_ldB >> 10);
  const int h_MB = (int)(q_mbc & 0xFFFFu), n_chunk = (int)(q_mbc >> 16);
  const int n_rt = (int)(q_rts & 1023u), h_spb = (int)(q_rts >> 10);
  // Touch every 64-byte line of `p` this block will read, NOW and without waiting (one-slice instantiations only): the
  // fetch in PIN_REST() below then finds the lines on their way — hipcc splits it into three to four dependent groups,
  // each a scalar-cache miss of its own otherwise (interleaved A/B with the deferred fetch: backward 8.53 -> 8.29 us).
  // hipcc does not see that an asm's scalar loads complete late, so the destination registers stay allocated — as
  // operands of the waiting asm in PIN_REST() — until that wait, and the instantiations that do this must not spill
  // SGPRs (a spilled destination's register is handed to a live value at once: the late write then corrupts it — a memory
  // fault at 600 rows when the MULTI instantiations still did it); __graft_entry__.build() fails the build otherwise.
  unsigned kpf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if constexpr (!MULTI) {
    const unsigned long long ka = (unsigned long long)(uintptr_t)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr unsigned KA_P = 56;       // `p` follows five pointers and four words in the argument block
    const unsigned o_net = KA_P + (unsigned)offsetof(StepParams, net) + (unsigned)sizeof(NetPtrs) * (unsigned)(x & 3);
    const unsigned o_go = KA_P + (unsigned)offsetof(StepParams, go) + (unsigned)sizeof(NetGrad) * (unsigned)(x & 3);
    const unsigned o_t0 = (KA_P + (unsigned)offsetof(StepParams, log_std)) & ~63u;
    static_assert(KA_P + sizeof(StepParams) - ((KA_P + offsetof(StepParams, log_std)) & ~(size_t)63) <= 256, "kernel-argument tail: more than 4 lines");
    unsigned d0, d1, d2, d3, d4, d5, d6, d7;
    asm volatile(
        "s_load_dword %0, %8, %9\n\ts_load_dword %1, %8, %10\n\ts_load_dword %2, %8, %11\n\t"
        "s_load_dword %3, %8, %12\n\ts_load_dword %4, %8, %13\n\ts_load_dword %5, %8, %14\n\t"
        "s_load_dword %6, %8, %15\n\ts_load_dword %7, %8, %16"
        : "=&s"(d0), "=&s"(d1), "=&s"(d2), "=&s"(d3), "=&s"(d4), "=&s"(d5), "=&s"(d6), "=&s"(d7)
        : "s"(ka), "s"(o_net), "s"(o_net + (unsigned)sizeof(NetPtrs) - 4u), "s"(o_go), "s"(o_go + (unsigned)sizeof(NetGrad) - 4u),
          "s"(o_t0), "s"(o_t0 + 64u), "s"(o_t0 + 128u), "s"(o_t0 + 192u));
    kpf[0] = d0; kpf[1] = d1; kpf[2] = d2; kpf[3] = d3; kpf[4] = d4; kpf[5] = d5; kpf[6] = d6; kpf[7] = d7;
  }
  // A net's blocks stay on two XCDs (net = x & 3: its weights and activations live in those two L2s; rotating the nets
  // over all XCDs made multi-round launches 5-8 % SLOWER).  But the policy's blocks are 1.5-2.5x as long as the scalar
  // nets', and in a multi-round launch XCDs 3 and 7 finished at 33 us while the other six idled from 17 us on (obs 39 /
  // act 28, 1 024 rows, bf16).  MULTI: the policy's LAST n_don dW1-tile blocks are therefore moved to the FRONT of the
  // other six XCDs' queues (the first ceil(n_don / 6) grid rows; host: launch_bwd) — long blocks first: at the ends of
  // those queues they started at 22 us and finished at 36; their old slots return at once.
  const int n_a = 32 * n_chunk;
  const int bsl2 = MULTI ? ((h_spb >> 2) & 3) : 0;     // (b) blocks: log2 of the column slices per block
  const int n_b = (4 >> bsl2) * n_rt;
  int net = x & 3;
  int local_ = (bid >> 3) * 2 + (x >> 2);
  if (MULTI) {
    const int n_don = h_spb >> 8;
    const int n_e = (n_don + 5) / 6;
    const int q = bid >> 3;
    if (q < n_e) {
      const int j = q * 6 + (x - (x >> 2));       // x in {0,1,2,4,5,6} -> 0..5
      if (net == IQLHIP_NET_PI || j >= n_don) return;
      net = IQLHIP_NET_PI;
      local_ = n_a + n_b - n_don + j;
    } else {
      local_ -= 2 * n_e;
      if (net == IQLHIP_NET_PI && local_ >= n_a + n_b - n_don) return;
    }
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int B = h_rows;
  const int MB = h_MB;
  const ArenaOff ao = arena_off(net, h_S, h_A, h_pol == IQLHIP_POLICY_GAUSSIAN);

  const NetPtrs np = p.net[net];
  const NetGrad go = p.go[net];
  const int D = ao.d;
  const int Dp = (D + 15) & ~15;      // 16 or 32
  const int DYLD = Dp + 4;            // dY row stride in LDS: 16-B aligned rows (float4 reads of a row's dims)
  const float* w2 = q_params + ao.w2;
  // (bf16 path: H0 / H1 are stored as bf16 — the same element offsets, half the bytes; the pointers below then carry
  //  the bf16 arrays' addresses and are only ever dereferenced through ld4 / ld2)
  const float* H1g = BF16 ? (const float*)((const __bf16*)q_h1 + net * MB * HID) : q_h1 + net * MB * HID;
  const float* H0g = BF16 ? (const float*)((const __bf16*)q_h0 + net * MB * HID) : q_h0 + net * MB * HID;
  // Everything the block still needs from the by-value StepParams is fetched by PIN_REST(), which each branch invokes
  // right BEHIND the issue of its first global loads: those depend on preloaded arguments only, so the ~500-cycle fetch
  // of the argument block now runs under their latency instead of in front of them (one batch of scalar loads behind
  // one wait; in the one-slice instantiations the lines were touched at the top and the prefetch registers' live range
  // ends at the wait below, which is free by then).
#define PIN_REST()                                                                                                          \
  do {                                                                                                                      \
    float* sa_ = p.sc.slab_a; float* sb_ = p.sc.slab_b;                                                                     \
    const long long sbo_ = p.sc.slab_b_off[net], npar_ = p.n_params;                                                        \
    PIN_P(np.w1); PIN_P(sa_); PIN_P(sb_);                                                                                   \
    PIN_S(sbo_); PIN_S(npar_); PIN_S(go.w1); PIN_S(go.b1); PIN_S(go.w2); PIN_S(go.b2); PIN_S(go.log_std);                   \
    PIN_S(p.inv_batch); PIN_S(p.hy.iql_tau); PIN_S(p.hy.beta); PIN_S(p.hy.discount); PIN_S(p.hy.exp_adv_max);               \
    if constexpr (!MULTI)                                                                                                   \
      asm volatile("s_waitcnt lgkmcnt(0)" ::"s"(kpf[0]), "s"(kpf[1]), "s"(kpf[2]), "s"(kpf[3]), "s"(kpf[4]), "s"(kpf[5]),   \
                   "s"(kpf[6]), "s"(kpf[7]));                                                                               \
  } while (0)
  if (local_ >= n_a + n_b) return;
  // MULTI: the (b) blocks walk 2 / 4 slices and run 2-3x as long as a dW1 tile — they take the FIRST block indices so
  // that the launch ends on short blocks (longest first); one-slice grids keep the dW1 tiles first
  const int local = MULTI ? ((local_ < n_b) ? n_a + local_ : local_ - n_b) : local_;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  STAMP_BASE(p, 2048 * 16);   // second half of the stamp buffer: the forward kernel owns the first
  STAMP(p, 0);

  if (local < n_a) {
    // ===================== (a): dW1[j-tile][i-tile] over one 256-row chunk =====================
    const int c = local >> 5;
    const int jt = (local >> 2) & 7;
    const int it = local & 3;
    const int j0 = jt * 32, i0 = it * 64;
    const int cbase = c * CHUNK_ROWS;
    float* red = smem;                               // [4][32][T64_LD]
    const int DYA = Dp + 4;                          // row stride of dYs / dLs here: 16-B aligned rows (float4 reads)
    float* dYs = red + 4 * 32 * T64_LD;              // [256][DYA]
    float* dLs = dYs + CHUNK_ROWS * DYA;            // [256][DYA]  (gaussian pi designated block only)
    float* W2s = dLs + CHUNK_ROWS * DYA;            // [D][32]
    float* rsm = W2s + 32 * 32;                      // [64] small reductions
    float* wS = rsm + 64;                            // [256] policy: per-row advantage weight
    const bool designated = (jt == 0 && it == 2);    // db2 and dlog_std: a block without other extras
    const bool loss_block = (jt == 1 && it == 2);    // the loss sums: another one
    // the column-independent extras of this j tile: db1 by the it == 0 block; dW2 by the it == 0 block when D == 1
    // (two fmas per row) but, when D > 1 (policy: MFMAs and an LDS round trip), one half of the j columns each by
    // the it == 1 and it == 3 blocks — all on one block made that block the last to finish in the whole kernel
    const bool do_db1 = (it == 0);
    const bool do_dw2 = (D == 1) ? (it == 0) : (it == 1 || it == 3);
    const int tb_own = (it == 3) ? 1 : 0;            // D > 1: which of a lane's two j columns this block's dW2 covers
    const bool extras = do_db1 || do_dw2;

    // ---- loads, in the order they are needed (vmcnt retires in issue order): the per-row loss
    // inputs first, then the 96 KiB of activation tiles, which stream in under the dY arithmetic.
    const int prow = cbase + tid;
    RowIn in;
    const float lsr = pi_ls_issue_hot(q_params + ao.log_std, q_xb, h_pol, h_A);
    const bool is_pi = (net == IQLHIP_NET_PI);
    row_issue_hot(q_heads, q_xb, h_ld, h_S, h_A, BROW(prow), in);           // scalar partials, r, d (the policy needs h[1..3] for w)
    // Policy: its per-(row, dim) inputs are loaded as (row, dim) work items — thread (r8 = tid >> 3, sub = tid & 7)
    // takes rows r8 + 32c, c = 0..7, and action dim sub (+ 8e) — so that one load instruction touches 6-8 cache
    // lines.  With thread = row every such load touched 48-64 lines; the 16 of them held the load queue for 8.5 k
    // cycles and made the policy's (a) blocks (10-13 us) the long pole of the whole kernel (others: 6-9 us).
    const int r8 = tid >> 3, sub = tid & 7;
    const f32x4* hpb = (const f32x4*)(q_heads + MB * HEAD_LD);
    const float* hpf = q_heads + MB * HEAD_LD;
    const float* xbp = q_xb;
    f32x4 php[8];
    float pac[8];
    if (is_pi) {
      const unsigned dd0 = (unsigned)min(sub, h_A - 1);
      const unsigned uA = (unsigned)h_A, uld = (unsigned)h_ld, uS = (unsigned)h_S;
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) {
        const unsigned rowc = (unsigned)BROW(cbase + r8 + 32 * cc);
        php[cc] = *(const f32x4*)(hpf + 4u * (rowc * uA + dd0));
        pac[cc] = xbp[rowc * uld + uS + dd0];
      }
    }
    float w2pre[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ec = min(tid + 256 * q, D * 32 - 1);      // clamped: unconditional load
      w2pre[q] = w2[(unsigned)((ec >> 5) * HID + j0 + (ec & 31))];
    }
    // wave w reduces its 64 rows, 4 per MFMA: instruction ks of lane group g takes row AROW(ks) = 64w + 16(ks>>2) +
    // 4g + (ks&3) — the row a lane's accumulator register (ks&3) of the 16-row tile (ks>>2) holds when dH1 itself
    // comes out of an MFMA (wide heads, below), so that result feeds the dW1 MFMA without a shuffle.  Rows >= B are
    // clamped to a valid row: their dY is 0, so they contribute nothing.
#define AROW(ks) (64 * wave + 16 * ((ks) >> 2) + 4 * g + ((ks) & 3))
    typename Frag2<BF16>::type hh[16];
    typename Frag4<BF16>::type bb[16];
    // The 96 KB of activation tiles are requested in four groups with the chunk's loss arithmetic BETWEEN them: a wave spends
    // ~4 k cycles just issuing its ~60 loads (the four waves share the CU's one vector-memory pipe, ~64 cycles per 1 KB
    // instruction) with the vector ALU idle, and the per-row inputs requested first are back after the first third of that —
    // the policy's 2.7 k cycles of advantage weights and (row, dim) terms, which made its blocks the kernel's last, now run
    // inside the issue phase instead of behind it.  (Scheduling barriers: hipcc otherwise gathers all loads in front again.)
#define HB_LOAD(k0_, k1_)                                                                               \
    _Pragma("unroll") for (int ks = (k0_); ks < (k1_); ++ks) {                                          \
      const unsigned row = (unsigned)BROW(cbase + AROW(ks));                                            \
      hh[ks] = ld2<BF16>(H1g, row * (unsigned)HID + (unsigned)(j0 + 2 * l15));                          \
      bb[ks] = ld4<BF16>(H0g, row * (unsigned)HID + (unsigned)(i0 + 4 * l15));                          \
    }
    HB_LOAD(0, 4);
    PIN_REST();
    float* slab = p.sc.slab_a + (long long)c * p.n_params;
    // dropout: the saved activations are post-dropout, so (h > 0) already encodes relu AND keep; the chain
    // rule only adds the 1/(1-p) multiplier
    const float dscale = (net == IQLHIP_NET_PI && p.drop_bits != nullptr) ? p.drop_scale : 1.f;
    STAMP(p, 10);
    // ---- dY for the 256 rows of the chunk (thread = row)
    {
      const int row = prow;
#pragma unroll
      for (int q = 0; q < 4; ++q) {           // [Dp][32], zero rows beyond D (operand of the dH1 MFMA)
        const int e = tid + 256 * q;
        if (e < Dp * 32) W2s[e] = (e < D * 32) ? w2pre[q] : 0.f;
      }
      float lossA = 0.f, lossB = 0.f;
      const PiConst pc = pi_consts_hot(p, h_pol, net, lsr);
      if (!is_pi) {
        float* dyrow = dYs + tid * DYA;
        for (int dd = 0; dd < Dp; ++dd) dyrow[dd] = 0.f;
        if (row < B) row_finish(p, net, in, dyrow, lossA, lossB);
      } else {
        // phase 1 (thread = row): the advantage weight (iql.py:519); rows >= B get w = 0, hence dY = 0
        float wrow = 0.f;
        if (row < B) {
          const float tq = fminf(sum4(in.h[2]), sum4(in.h[3]));
          const float u = tq - sum4(in.h[1]);
          wrow = fminf(expf(p.hy.beta * u), p.hy.exp_adv_max);
        }
        STAMP(p, 5);
        wS[tid] = wrow;
      }
      __builtin_amdgcn_sched_barrier(0);
      HB_LOAD(4, 8);
      __builtin_amdgcn_sched_barrier(0);
      const bool gauss = (h_pol == IQLHIP_POLICY_GAUSSIAN);
      const bool want_dls = designated && gauss;
      float wv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (is_pi) {
        __syncthreads();                       // (net is block-uniform)
        STAMP(p, 6);
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) wv[cc] = wS[r8 + 32 * cc];
      }
      __builtin_amdgcn_sched_barrier(0);
      HB_LOAD(8, 12);
      __builtin_amdgcn_sched_barrier(0);
      if (is_pi) {
        // phase 2 (thread = (row, dim)): mean, log-prob term, dL/dpre and dL/dlog_std of every (row, dim) — eight
        // rows of one dim per thread and group of 8 dims, as straight-line select code (pi_items8): written with
        // per-item branches this phase was ~110 LDS / branch round trips (3.3 k cycles, the policy blocks' long pole)
        const int A = h_A;
        const float invB = p.inv_batch;
        const int DpZ = (D <= 8) ? 8 : Dp;      // dims the dH1 / dW2 products read as operands (zero-filled beyond A)
        for (int e = 0; 8 * e < DpZ; ++e) {
          const int dd = sub + 8 * e;
          float dyv[8], dlv[8];
#pragma unroll
          for (int cc = 0; cc < 8; ++cc) { dyv[cc] = 0.f; dlv[cc] = 0.f; }
          if (8 * e < A) {                     // (block-uniform; groups beyond A are padding up to Dp: zeros, no loads)
            const int ddc = min(dd, A - 1);
            const float ivar = __shfl(pc.ivar, ddc);     // lane ddc holds dim ddc's constants; the whole wave is here
            const float ls = __shfl(pc.ls, ddc);
            if (e == 0) {
              pi_items8(php, pac, wv, dd < A, gauss, ivar, ls, invB, dyv, dlv, lossA);
            } else {                           // action dims >= 8 (wide action spaces): loaded here, 8 at a time
              f32x4 hv[8];
              float acv[8];
#pragma unroll
              for (int cc = 0; cc < 8; ++cc) {
                const unsigned rowc = (unsigned)BROW(cbase + r8 + 32 * cc);
                hv[cc] = *(const f32x4*)(hpf + 4u * (rowc * (unsigned)A + (unsigned)ddc));
                acv[cc] = xbp[rowc * (unsigned)h_ld + (unsigned)(h_S + ddc)];
              }
              pi_items8(hv, acv, wv, dd < A, gauss, ivar, ls, invB, dyv, dlv, lossA);
            }
          }
#pragma unroll
          for (int cc = 0; cc < 8; ++cc) dYs[(r8 + 32 * cc) * DYA + dd] = dyv[cc];
          if (want_dls) {
#pragma unroll
            for (int cc = 0; cc < 8; ++cc) dLs[(r8 + 32 * cc) * DYA + dd] = dlv[cc];
          }
        }
        STAMP(p, 7);
      }
      __builtin_amdgcn_sched_barrier(0);
      HB_LOAD(12, 16);
      __builtin_amdgcn_sched_barrier(0);
#undef HB_LOAD
      STAMP(p, 11);
      if (loss_block) {
        const float sA = block_sum_256(lossA, rsm);
        if (net == IQLHIP_NET_V && tid == 0) p.sc.loss_parts[0 * 64 + c] = sA;
        if (net == IQLHIP_NET_PI && tid == 0) p.sc.loss_parts[3 * 64 + c] = sA;
        if (net == IQLHIP_NET_Q1) {
          const float sB = block_sum_256(lossB, rsm + 8);
          if (tid == 0) { p.sc.loss_parts[1 * 64 + c] = sA; p.sc.loss_parts[2 * 64 + c] = sB; }
        }
      }
    }
    __syncthreads();
    STAMP(p, 1);
    if (designated && D > 8) {
      // db2[dd] = sum_r dY[r][dd];  dlog_std[dd] = sum_r w (1 - diff^2/var) * inv_batch (inside clamp range only).
      // Wide heads: thread (dim tid & 31, row group tid >> 5) sums 32 rows, the 8 partial sums meet in LDS (the
      // tile-reduction buffer is idle until after the MFMA phase; only wave 0's part of it is touched here).  A
      // dim per wave and iteration, each with its own load and store, took ~1.4 k cycles per dim: 9.7 k at D = 28
      // (for D <= 8 that loop, at most two dims per wave, is the cheaper one and stays).
      const bool gls = (net == IQLHIP_NET_PI && h_pol == IQLHIP_POLICY_GAUSSIAN);
      const int dd = tid & 31, rg = tid >> 5;
      float s = 0.f, sl = 0.f;
      if (dd < D) {
#pragma unroll 8
        for (int r = 0; r < 32; ++r) {
          s += dYs[(rg * 32 + r) * DYA + dd];
          if (gls) sl += dLs[(rg * 32 + r) * DYA + dd];
        }
      }
      red[rg * 64 + dd] = s;
      red[rg * 64 + 32 + dd] = sl;
      __syncthreads();                 // (block-uniform condition)
      if (tid < D) {
        float ts = 0.f, tl = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) { ts += red[k * 64 + tid]; tl += red[k * 64 + 32 + tid]; }
        slab[go.b2 + tid] = ts;
        if (gls) {     // (lsr: this lane's raw log_std, loaded at the top of the block — tid < D <= 32 is lane tid of wave 0;
                       //  a load issued HERE would queue behind the whole activation-tile stream)
          const bool inside = (lsr >= p.hy.log_std_min) && (lsr <= p.hy.log_std_max);
          slab[go.log_std + tid] = inside ? tl * p.inv_batch : 0.f;
        }
      }
    } else if (designated && net != IQLHIP_NET_PI) {
      // scalar heads (V, Q1, Q2; D = 1): one 256-term sum by wave 0 — 4 rows per lane, then a shuffle tree.  (Kept as it is:
      // these blocks are not the kernel's last ones, and db2 of a Q net, sum_r (q - y) / B, cancels so heavily that
      // ANY other summation order moves it by ~2e-5 of itself against the reference's equally arbitrary order.)
      if (wave == 0) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) s += dYs[(lane + 64 * q) * DYA];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) slab[go.b2] = s;
      }
    } else if (designated) {
      // the policy with D <= 8: thread (dim tid & 7, row group tid >> 3) sums 8 rows, the 32 partial sums per dim meet in LDS and are
      // added in row-group order by one thread per dim (a dim per wave and pass, with a 6-level shuffle tree per dim,
      // took 2.9 k cycles on the one block that does this — the last block of the whole kernel)
      const bool gls = (net == IQLHIP_NET_PI && h_pol == IQLHIP_POLICY_GAUSSIAN);
      const int dd = tid & 7, rg = tid >> 3;
      // (both stages are balanced trees: these sums cancel heavily — db2 of a Q net is sum_r (q - y) / B — and a
      //  sequential 256-term sum lost a digit against the reference: 2.1e-5 instead of 2.6e-6 on one fixture)
      float s = 0.f, sl = 0.f;
      if (dd < D) {
        float a[8], b[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          a[r] = dYs[(rg * 8 + r) * DYA + dd];
          b[r] = gls ? dLs[(rg * 8 + r) * DYA + dd] : 0.f;
        }
        s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
        sl = ((b[0] + b[1]) + (b[2] + b[3])) + ((b[4] + b[5]) + (b[6] + b[7]));
      }
      red[rg * 16 + dd] = s;
      red[rg * 16 + 8 + dd] = sl;
      __syncthreads();                 // (block-uniform condition)
      if (tid < D) {
        float u[32], w[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) { u[k] = red[k * 16 + tid]; w[k] = red[k * 16 + 8 + tid]; }
#pragma unroll
        for (int st = 16; st > 0; st >>= 1) {
#pragma unroll
          for (int k = 0; k < 16; ++k) if (k < st) { u[k] += u[k + st]; w[k] += w[k + st]; }
        }
        const float ts = u[0], tl = w[0];
        slab[go.b2 + tid] = ts;
        if (gls) {
          const bool inside = (lsr >= p.hy.log_std_min) && (lsr <= p.hy.log_std_max);
          slab[go.log_std + tid] = inside ? tl * p.inv_batch : 0.f;
        }
      }
    }
    STAMP(p, 2);

    // ---- operand phase: A values av[ks][ta] = dH1[row][j0 + 2*l15 + ta] from registers + LDS
    float av[16][2];
    float db1a[2] = {0.f, 0.f};
    float dw2a[2] = {0.f, 0.f};    // D == 1
    if (D == 1) {
      const float w2a = W2s[2 * l15], w2b = W2s[2 * l15 + 1];
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const float dy = dYs[AROW(ks) * DYA];
        av[ks][0] = ((float)hh[ks][0] > 0.f) ? dy * w2a * dscale : 0.f;
        av[ks][1] = ((float)hh[ks][1] > 0.f) ? dy * w2b * dscale : 0.f;
        if (do_dw2) {
          dw2a[0] = fmaf(dy, (float)hh[ks][0], dw2a[0]);
          dw2a[1] = fmaf(dy, (float)hh[ks][1], dw2a[1]);
        }
      }
    } else {
      // wide heads (policy): dH1pre[row][j] = sum_dd dY[row][dd] W2[dd][j] on the matrix cores — per wave 4 row
      // tiles x 2 j tiles x Dp/4 k-steps (32 or 64 MFMAs) instead of 2 D fmas per (row, j) on the vector ALU (1 024
      // per thread at D = 28).  A = dY (m = row 16t + l15, k = dd), B = W2 (k = dd, n = j = 2 l15 + ta, zero rows
      // beyond D); lane (g, l15) gets rows 16t + 4g + reg = AROW(4t + reg): its own operand rows of the dW1 MFMA.
      // fp32 MFMA is an exact fma chain over k, i.e. the same sum in the same dim order as the scalar code.
      f32x4 pre[4][2];
#pragma unroll
      for (int t = 0; t < 4; ++t) { pre[t][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; pre[t][1] = pre[t][0]; }
      // k-steps of 4 action dims, rounded up to 2 / 4 / 8 (dims 4 NK .. are zero operands: not multiplied at all)
      if (D <= 8) dh1_mfma<2>(pre, dYs, W2s, DYA, wave, g, l15);
      else if (D <= 16) dh1_mfma<4>(pre, dYs, W2s, DYA, wave, g, l15);
      else dh1_mfma<8>(pre, dYs, W2s, DYA, wave, g, l15);
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        av[ks][0] = ((float)hh[ks][0] > 0.f) ? pre[ks >> 2][0][ks & 3] * dscale : 0.f;
        av[ks][1] = ((float)hh[ks][1] > 0.f) ? pre[ks >> 2][1][ks & 3] * dscale : 0.f;
      }
    }
    if (do_db1) {
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) { db1a[0] += av[ks][0]; db1a[1] += av[ks][1]; }
    }

    // ---- MFMA phase
    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (BF16) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {       // rows (k) 8q..8q+7 of this lane's 16
        bf16x8 A[2], Bv[4];
#pragma unroll
        for (int ta = 0; ta < 2; ++ta) {
          float t8[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) t8[e] = av[8 * q + e][ta];
          A[ta] = pack8s(t8);
        }
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) {      // (H0 arrives as bf16: the operand is assembled, not converted)
#pragma unroll
          for (int e = 0; e < 8; ++e) Bv[tb][e] = bb[8 * q + e][tb];
        }
#pragma unroll
        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
          for (int tb = 0; tb < 4; ++tb) acc[ta][tb] = MFMA_BF16(A[ta], Bv[tb], acc[ta][tb]);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
#pragma unroll
        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
          for (int tb = 0; tb < 4; ++tb) acc[ta][tb] = MFMA16(av[ks][ta], bb[ks][tb], acc[ta][tb]);
      }
    }
    f32x4 acc2[2][2];   // dW2 tiles [dt][tb] (MFMA path, D > 1, extras blocks only)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc2[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int ndt = Dp >> 4;
    if (do_dw2 && D > 1) {
      // all 16 LDS operands first, then the MFMAs (a read under a per-iteration `if` was waited for on the spot:
      // 16 exposed LDS latencies made these blocks the last of the kernel)
      float ad[16];
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) ad[ks] = dYs[AROW(ks) * DYA + l15];
      float hsel[16];
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) hsel[ks] = tb_own ? (float)hh[ks][1] : (float)hh[ks][0];
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) acc2[0][0] = MFMA16(ad[ks], hsel[ks], acc2[0][0]);
      if (ndt > 1) {
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) ad[ks] = dYs[AROW(ks) * DYA + 16 + l15];
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) acc2[1][0] = MFMA16(ad[ks], hsel[ks], acc2[1][0]);
      }
    }
    STAMP(p, 3);

    // ---- cross-wave reduction of the 32x64 tile through LDS, then coalesced store.  The extras (db1 / dW2 partial
    // sums of the 4 waves) are staged in the same pass, in the dLs region — only the loss-sum block (it == 2), which
    // has no extras, ever uses that region — so one barrier serves both reductions.
    // exA [4 waves x 4 lane groups][2 rows: db1, scalar dW2][32 cols]: every lane stores its own partial sums — the
    // sums over the lane groups g and over the waves are formed after the barrier, in the order ((g0+g1)+(g2+g3)) per
    // wave, ((w0+w1)+(w2+w3)) over the waves, i.e. the sums the two shuffle steps per value used to form before the
    // barrier (4 values x 2 dependent cross-lane steps: ~1.2 k cycles of every block that owns extras).
    // exB [4 waves][Dp rows][32 cols]: the MFMA tiles of a wide head's dW2.
    float* exA = dLs;
    float* exB = dLs + 1024;
    {
      float* myred = red + wave * 32 * T64_LD;
#pragma unroll
      for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int jl = 2 * (4 * g + reg) + ta;
          f32x4 v = (f32x4){acc[ta][0][reg], acc[ta][1][reg], acc[ta][2][reg], acc[ta][3][reg]};
          *(f32x4*)(myred + jl * T64_LD + 4 * l15) = v;
        }
    }
    if (extras) {
      float* mine = exA + (wave * 4 + g) * 64 + 2 * l15;
      *(f32x2*)mine = (f32x2){db1a[0], db1a[1]};
      if (D == 1) *(f32x2*)(mine + 32) = (f32x2){dw2a[0], dw2a[1]};
      if (D > 1 && do_dw2) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
          if (dt < ndt)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
              exB[(wave * 32 + 16 * dt + 4 * g + reg) * 32 + 2 * l15 + tb_own] = acc2[dt][0][reg];
      }
    }
    __syncthreads();
    {
      float* gw1 = slab + go.w1;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int f = tid + 256 * q;
        const int jl = f >> 4, i4 = f & 15;
        f32x4 s = *(const f32x4*)(red + jl * T64_LD + 4 * i4);
#pragma unroll
        for (int w = 1; w < 4; ++w) s += *(const f32x4*)(red + w * 32 * T64_LD + jl * T64_LD + 4 * i4);
        *(f32x4*)(gw1 + (j0 + jl) * HID + i0 + 4 * i4) = s;
      }
    }
    if (extras) {
      // (only the rows this block stores: row 0 = db1 costs 16 LDS reads per value, and a wave that holds row-0 AND row-1
      //  lanes runs both paths one after the other — 790 cycles at the end of the policy's dW2 blocks, the kernel's last
      //  blocks, which do not even own db1)
      const int e_lo = do_db1 ? 0 : 32;
      const int e_hi = do_dw2 ? (1 + D) * 32 : 32;
      for (int e = tid + e_lo; e < e_hi; e += 256) {
        const int rr = e >> 5, jj = e & 31;
        float s;
        if (rr == 0 || D == 1) {
          float wsum[4];
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            const float* a = exA + (w * 4) * 64 + rr * 32 + jj;
            wsum[w] = (a[0] + a[64]) + (a[128] + a[192]);
          }
          s = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
        } else {
          const float* b = exB + (rr - 1) * 32 + jj;
          s = (b[0] + b[1024]) + (b[2048] + b[3072]);
        }
        if (rr == 0) { if (do_db1) slab[go.b1 + j0 + jj] = s; }
        else if (do_dw2 && (D == 1 || (jj & 1) == tb_own)) slab[go.w2 + (rr - 1) * HID + j0 + jj] = s;
      }
    }
    STAMP(p, 4);
    RT_STAMP(p, 14, rt_entry_);
    RT_STAMP(p, 15, iql_realtime());
    return;
  }
#undef AROW

  // ===================== (b): dH0 / dW0 / db0 for one 32-row tile and 64-column slice =====================
  {
    const int lb = local - n_a;
    const int rt = lb >> (2 - bsl2);
    int i0 = ((lb & ((4 >> bsl2) - 1)) << bsl2) * 64;      // first (or only) column slice of this block
    const int row0 = rt * RT_ROWS;
    const int k0 = ao.k0;
    const int ld = h_ld;
    const int xoff = 0;                   // trainable nets read s or [s|a]: both start at column 0
    const float* w1 = np.w1;

    float* dH1s = smem;                              // [32][H0_LD]  (bf16 path: the same tile as bf16 [32][H0B_LD], below)
    __bf16* dH1b = (__bf16*)smem;
    float* red = dH1s + RT_ROWS * H0_LD;             // [4][32][T64_LD]
    float* dH0s = red + 4 * 32 * T64_LD;             // [32][T64_LD]
    float* dYs = dH0s + RT_ROWS * T64_LD;            // [32][DYLD]
    float* Xr = dYs + RT_ROWS * 36;                  // [32][ld] packed rows (parked late); 16-B aligned

    // ---- issue every global load of the block, first-needed first (vmcnt retires in issue order)
    RowIn in;
    const int prow = BROW(row0 + (tid & 31));
    const float lsr = pi_ls_issue_hot(q_params + ao.log_std, q_xb, h_pol, h_A);
    // the scalar nets' per-row loss inputs are consumed by the first 32 threads only: wave 0 alone loads them
    // (these loads head the in-order queue — issued by all four waves they delayed every load behind them);
    // the policy's own inputs follow below
#pragma unroll
    for (int i = 0; i < 6; ++i) in.h[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    in.r = 0.f; in.d = 0.f;
    if (wave == 0 && net != IQLHIP_NET_PI) row_issue_hot(q_heads, q_xb, h_ld, h_S, h_A, prow, in);
    // policy: the loss arithmetic of the 32 rows is spread over all 256 threads — thread (row tid>>3,
    // dims (tid&7) + 8c) — instead of 32 threads walking all dims while 224 wait at the barrier
    const int prl = tid >> 3, psub = tid & 7;
    const int prow8 = BROW(row0 + prl);
    f32x4 ph[3], php[4];
    float pac[4];
    if (net == IQLHIP_NET_PI) {
      const float* hsb = q_heads;
      const unsigned oh = (unsigned)prow8 * (unsigned)HEAD_LD;
      ph[0] = *(const f32x4*)(hsb + (oh + 4u)); ph[1] = *(const f32x4*)(hsb + (oh + 8u)); ph[2] = *(const f32x4*)(hsb + (oh + 12u));
      const float* arow = q_xb + (unsigned)(prow8 * h_ld + h_S);
      const f32x4* hp = (const f32x4*)(q_heads + MB * HEAD_LD + (unsigned)(prow8 * h_A * NSPLIT));
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        php[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        pac[c] = 0.f;
        if (c == 0 || 8 * c < h_A) {               // block-uniform: dims >= 8 only for wide action spaces
          const int dd = min(psub + 8 * c, h_A - 1);
          php[c] = hp[dd];
          pac[c] = arow[dd];
        }
      }
    }
    // W2 rows matching this thread's H1 columns (all threads use cols 4*(tid&63)): row 0 for the scalar
    // heads, rows 0..7 for the policy (issued now, ahead of the W1 stream; rows >= 8 are loaded later)
    const int j4 = tid & 63;
    const f32x4 w2v = *(const f32x4*)(w2 + 4 * j4);
    f32x4 w2v8[8];
    if (D > 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) w2v8[j] = *(const f32x4*)(w2 + min(j, D - 1) * HID + 4 * j4);
    }
    // H1 tile [32][256] as float4 f = tid + 256q: row f>>6, cols 4*(f&63)
    // (the first half of the tile here, the second half and the H0 mask BEHIND the loss arithmetic below: the wave is busy
    //  issuing loads for ~2 k cycles — the CU's one vector-memory pipe — while the per-row inputs requested first are back
    //  after half of that; the policy's (row, dim) terms then run inside the issue phase, cf. the dW1 blocks)
    typename Frag4<BF16>::type h1v[8];
#define H1V_LOAD(q0_, q1_)                                                                              \
    _Pragma("unroll") for (int q = (q0_); q < (q1_); ++q) {                                             \
      const int f = tid + 256 * q;                                                                      \
      const unsigned row = (unsigned)BROW(row0 + (f >> 6));   /* rows >= B: dY = 0 -> dH1 = 0 */        \
      h1v[q] = ld4<BF16>(H1g, row * (unsigned)HID + (unsigned)(4 * (f & 63)));                          \
    }
    H1V_LOAD(0, 4);
    // W1 fragments: k = j in [64*wave, +64), n = i0 + 4*l15 + t — requested after the dY barrier (below): 16 KiB
    // per wave of fragment-shaped loads take ~1.5 k cycles of the CU's vector-memory pipe to issue, which in front of
    // the loss arithmetic only delayed it; issued there they stream in under the dH1 tile phase
    // FULLB (bf16, large batches): the waves split the COLUMNS (64 each) instead of the k range — no cross-wave
    // reduction — and walk all 256 k in 8 blocks of 32; ALL 64 fragments of the wave (128 registers) are requested during
    // the dH1 tile phase: fetched two k-blocks ahead the product waited ~1 k cycles per k-block for them
    constexpr bool FULLB = BF16 && MULTI;
    typename Frag4<BF16>::type bw[FULLB ? 64 : 16];      // (bf16 path: np.w1 addresses the bf16 shadow of W1)
    // H0 mask slice [32][64] as float4 f = tid + 256q: row f>>4, cols i0 + 4*(f&15)
    typename Frag4<BF16>::type h0v[2];
    const int n_x = RT_ROWS * ld / 4;
    const int x_last = B * ld / 4 - 1;
    PIN_REST();
    const float dscale = (net == IQLHIP_NET_PI && p.drop_bits != nullptr) ? p.drop_scale : 1.f;

    const PiConst pc = pi_consts_hot(p, h_pol, net, lsr);
    if (net == IQLHIP_NET_PI) {
      const float tq = fminf(sum4(ph[1]), sum4(ph[2]));
      const float u = tq - sum4(ph[0]);
      const float w = fminf(expf(p.hy.beta * u), p.hy.exp_adv_max);
      const bool rvalid = (row0 + prl) < B;
      const bool gauss = (h_pol == IQLHIP_POLICY_GAUSSIAN);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int dd = psub + 8 * c;
        if (dd < Dp) {
          float dy = 0.f;
          const int ddc = min(dd, h_A - 1);
          const float ivar = __shfl(pc.ivar, ddc);        // lane ddc holds dim ddc's constants; whole wave active
          if (rvalid && dd < h_A) {
            const float mu = tanh_via_exp(sum4(php[c]));
            const float diff = pac[c] - mu;
            const float dmu = gauss ? (-(w * diff) * ivar) * p.inv_batch : (-2.f * w * diff) * p.inv_batch;
            dy = dmu * (1.f - mu * mu);
          }
          dYs[prl * DYLD + dd] = dy;
        }
      }
    } else if (tid < RT_ROWS) {
      const int row = row0 + tid;
      float la, lbv;
      float* dyrow = dYs + tid * DYLD;
      for (int dd = 0; dd < Dp; ++dd) dyrow[dd] = 0.f;
      if (row < B) row_finish(p, net, in, dyrow, la, lbv);
    }
    __builtin_amdgcn_sched_barrier(0);
    H1V_LOAD(4, 8);
#undef H1V_LOAD
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int f = tid + 256 * q;
      const unsigned row = (unsigned)BROW(row0 + (f >> 4));
      h0v[q] = ld4<BF16>(H0g, row * (unsigned)HID + (unsigned)(i0 + 4 * (f & 15)));
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    STAMP(p, 5);
    // (requested two at a time between the row groups of the dH1 tile below: the four waves' 64 KB take ~1 k cycles of
    //  the CU's 64 B/clk fill path, which the tile's arithmetic covers instead of waiting behind it)
    // W1 row (= k index j of dH0 = dH1 . W1) of fragment ks: fp32 k = 4 ks + g (+ 64 wave); bf16: the bf16 MFMA's native map,
    // 8 CONTIGUOUS k per lane group — k = 32 (ks >> 3) + 8 g + (ks & 7) — so that the dH1 operand is ONE 16-byte LDS read.
    // FULLB: fragment ks = 8 kb + e of k-block kb: row 32 kb + 8 g + e, columns 64 wave + 4 l15 ..
#define BWB_ROW(ks_) (FULLB ? (32 * ((ks_) >> 3) + 8 * g + ((ks_) & 7)) : (BF16 ? (64 * wave + 32 * ((ks_) >> 3) + 8 * g + ((ks_) & 7)) : (64 * wave + 4 * (ks_) + g)))
#define BWB_COL (FULLB ? (64 * wave + 4 * l15) : (i0 + 4 * l15))
#define BWB_LOAD(ks_) bw[ks_] = ld4<BF16>(w1, (unsigned)(BWB_ROW(ks_) * HID + BWB_COL))

    // dH1s[r][j] = (sum_dd dY[r][dd] W2[dd][j]) * (H1[r][j] > 0)
    if (D > 8) {
      // wide heads (policy with more than 8 action dims): 8 dims at a time, the chunk's 8 W2 rows loaded ONCE (the
      // next chunk's while this one is multiplied) and used for all 8 rows of the thread; dY rows are zero-filled
      // to Dp, so rows >= D of a chunk (clamped duplicates) add exact zeros — same sums, same order as per-row code
      f32x4 sacc[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) sacc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
      f32x4 wv[8], wn[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) wv[j] = w2v8[j];
      for (int d0 = 0; d0 < D; d0 += 8) {
        const bool more = (d0 + 8 < D);
        if (more) {
#pragma unroll
          for (int j = 0; j < 8; ++j) wn[j] = *(const f32x4*)(w2 + min(d0 + 8 + j, D - 1) * HID + 4 * j4);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int rl = (tid + 256 * q) >> 6;
          const f32x4 ya = *(const f32x4*)(dYs + rl * DYLD + d0), yb = *(const f32x4*)(dYs + rl * DYLD + d0 + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) sacc[q] += ya[j] * wv[j];
#pragma unroll
          for (int j = 0; j < 4; ++j) sacc[q] += yb[j] * wv[4 + j];
        }
        if (more) {
#pragma unroll
          for (int j = 0; j < 8; ++j) wv[j] = wn[j];
        }
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int rl = (tid + 256 * q) >> 6;
        f32x4 out;
#pragma unroll
        for (int e = 0; e < 4; ++e) out[e] = ((float)h1v[q][e] > 0.f) ? sacc[q][e] * dscale : 0.f;
        if constexpr (BF16) {
          bf16x4 ob_;
#pragma unroll
          for (int e = 0; e < 4; ++e) ob_[e] = (__bf16)out[e];
          *(bf16x4*)(dH1b + rl * H0B_LD + 4 * j4) = ob_;
        } else {
          *(f32x4*)(dH1s + rl * H0_LD + 4 * j4) = out;
        }
        if constexpr (FULLB) {
#pragma unroll
          for (int e = 0; e < 8; ++e) BWB_LOAD(8 * q + e);
        } else {
          BWB_LOAD(2 * q);
          BWB_LOAD(2 * q + 1);
        }
      }
    } else {
      // (the dY reads of the whole tile in one batch per net kind, then the arithmetic: with the kind's branch inside the row
      //  loop every row group was an LDS round trip of its own — read, wait, multiply, write)
      f32x4 sq[8];
      if (D == 1) {
        float dy1[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) dy1[q] = dYs[((tid + 256 * q) >> 6) * DYLD];
#pragma unroll
        for (int q = 0; q < 8; ++q) sq[q] = dy1[q] * w2v;
      } else {
        // dYs is zero-filled up to Dp >= 8 and w2v8[j >= D] repeats row D-1: unconditional float4 LDS reads, four row groups
        // at a time (registers)
#pragma unroll
        for (int hq = 0; hq < 2; ++hq) {
          f32x4 ya[4], yb[4];
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            const int rl = (tid + 256 * (4 * hq + qq)) >> 6;
            ya[qq] = *(const f32x4*)(dYs + rl * DYLD);
            yb[qq] = *(const f32x4*)(dYs + rl * DYLD + 4);
          }
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            f32x4 s = ya[qq][0] * w2v8[0];
#pragma unroll
            for (int j = 1; j < 4; ++j) s += ya[qq][j] * w2v8[j];
#pragma unroll
            for (int j = 0; j < 4; ++j) s += yb[qq][j] * w2v8[4 + j];
            sq[4 * hq + qq] = s;
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int rl = (tid + 256 * q) >> 6;
        const f32x4 s = sq[q];
        f32x4 out;
#pragma unroll
        for (int e = 0; e < 4; ++e) out[e] = ((float)h1v[q][e] > 0.f) ? s[e] * dscale : 0.f;
        if constexpr (BF16) {
          bf16x4 ob_;
#pragma unroll
          for (int e = 0; e < 4; ++e) ob_[e] = (__bf16)out[e];
          *(bf16x4*)(dH1b + rl * H0B_LD + 4 * j4) = ob_;
        } else {
          *(f32x4*)(dH1s + rl * H0_LD + 4 * j4) = out;
        }
        if constexpr (FULLB) {
#pragma unroll
          for (int e = 0; e < 8; ++e) BWB_LOAD(8 * q + e);
        } else {
          BWB_LOAD(2 * q);
          BWB_LOAD(2 * q + 1);
        }
      }
    }
    __syncthreads();
    STAMP(p, 6);
    if constexpr (FULLB) {
      // ===== bf16, large batches: the whole row tile in one pass (host: 4 slices per (b) block, i0 = 0).  Per slice
      // the old structure paid three barriers, a cross-wave reduction through LDS and a dozen dependent LDS round
      // trips for ~300 cycles of matrix work (8 k cycles per slice at 1 024 rows, profiles/r03_stamps_config5_1024_bf16.txt).
      // Here wave w owns columns [64 w, 64 w + 64) of dH0 = dH1 . W1 over ALL 256 k (8 k-blocks of 32, the W1 shadow's
      // fragments double-buffered in registers), masks them in registers, parks them TRANSPOSED (bf16 [col][row]) for the
      // dW0 product — whose operands then are one 16-byte LDS read each — and stores its 64 rows of [dW0 | db0].
      typename Frag4<true>::type hm[2][4];          // H0 mask in accumulator layout: rows 16 rt + 4 g + reg, cols 64 w + 4 l15 ..
#pragma unroll
      for (int rtl = 0; rtl < 2; ++rtl)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const unsigned row = (unsigned)BROW(row0 + 16 * rtl + 4 * g + reg);
          hm[rtl][reg] = ld4<true>(H0g, row * (unsigned)HID + (unsigned)(64 * wave + 4 * l15));
        }
      f32x4 xr[XR_MAX_F4];
      xr_load(xr, q_xb, row0 * ld / 4, n_x, x_last);
      bf16x8 Ad[2][8];
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) {
        Ad[0][kb] = *(const bf16x8*)(dH1b + l15 * H0B_LD + 32 * kb + 8 * g);
        Ad[1][kb] = *(const bf16x8*)(dH1b + (16 + l15) * H0B_LD + 32 * kb + 8 * g);
      }
      f32x4 acc[2][4];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) {
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) {
          bf16x8 Bv;
#pragma unroll
          for (int e = 0; e < 8; ++e) Bv[e] = bw[8 * kb + e][tb];
          acc[0][tb] = MFMA_BF16(Ad[0][kb], Bv, acc[0][tb]);
          acc[1][tb] = MFMA_BF16(Ad[1][kb], Bv, acc[1][tb]);
        }
      }
      STAMP(p, 7);
      constexpr int TLD = 40;                         // row stride of the transposed tile: 80 bytes, conflict-free 16-byte reads
      __bf16* dH0T = (__bf16*)red;                    // [256][TLD]
#pragma unroll
      for (int rtl = 0; rtl < 2; ++rtl)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) {
          bf16x4 o;
#pragma unroll
          for (int reg = 0; reg < 4; ++reg)
            o[reg] = (__bf16)(((float)hm[rtl][reg][tb] > 0.f) ? acc[rtl][tb][reg] * dscale : 0.f);    // rows >= B carry 0
          *(bf16x4*)(dH0T + (64 * wave + 4 * l15 + tb) * TLD + 16 * rtl + 4 * g) = o;
        }
      xr_store(xr, Xr, n_x);
      __syncthreads();
      STAMP(p, 8);
      // [dW0 | db0][i][kc] = sum_r dH0[r][i] [X | 1][r][kc]: A = [X | 1] (m = kc, k = row 8 g + e), B = dH0T (n = i, k = row)
      const int k1 = k0 + 1;
      const int nct = (k1 + 15) >> 4;
      bf16x8 Ax[9];
#pragma unroll
      for (int ct = 0; ct < 9; ++ct) {
        if (ct < nct) {
          const int kc = 16 * ct + l15;
          const int kcc = min(kc, k0 - 1);
          float a8[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float xa = Xr[(8 * g + e) * ld + xoff + kcc];
            a8[e] = (kc < k0) ? xa : ((kc == k0) ? 1.f : 0.f);      // ones column -> db0
          }
          Ax[ct] = pack8s(a8);
        }
      }
      float* slabB = p.sc.slab_b + p.sc.slab_b_off[net] + rt * (HID * k0 + HID);
      float* dstB = slabB + HID * k0;
      STAMP(p, 12);
#pragma unroll
      for (int itl = 0; itl < 4; ++itl) {
        const int il = 64 * wave + 16 * itl + l15;
        const bf16x8 Bd = *(const bf16x8*)(dH0T + il * TLD + 8 * g);
#pragma unroll
        for (int ct = 0; ct < 9; ++ct) {
          if (ct < nct) {
            const f32x4 r4 = MFMA_BF16(Ax[ct], Bd, ((f32x4){0.f, 0.f, 0.f, 0.f}));
            const int kc0 = 16 * ct + 4 * g;
            if (kc0 + 3 < k0) {
              *(f32x4u*)(slabB + (unsigned)(il * k0 + kc0)) = r4;
            } else {
#pragma unroll
              for (int reg = 0; reg < 4; ++reg) {
                const int kc = kc0 + reg;
                if (kc < k0) slabB[(unsigned)(il * k0 + kc)] = r4[reg];
                else if (kc == k0) dstB[il] = r4[reg];
              }
            }
          }
        }
      }
      STAMP(p, 9);
      RT_STAMP(p, 14, rt_entry_);
      RT_STAMP(p, 15, iql_realtime());
      return;
    }
    // the 32 packed rows, needed last (dW0): issued only now — the H1 / W2 registers are free again, the loads
    // queue behind the W1 fragments (so waiting for those does not wait for these) and the MFMA phase hides them
    f32x4 xr[XR_MAX_F4];
    xr_load(xr, q_xb, row0 * ld / 4, n_x, x_last);

    // ======== per column slice (one pass unless MULTI): dH0 slice, dW0 / db0 slice.  The next slice's W1 fragments are
    // requested into the registers this slice's MFMAs have just consumed, its H0 mask under the MFMA phase.
    for (int itn = 0;; ++itn) {
    const bool more = MULTI && (itn + 1 < (1 << bsl2));
    typename Frag4<BF16>::type h0n[2] = {h0v[0], h0v[1]};

    // dH0 partial over this wave's 64 j's: [32 rows][64 cols]
    {
      f32x4 acc[2][4];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if constexpr (BF16) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {     // k = j index: 64w + 32q + 8g + e, e = 0..7 — one 16-byte read per operand
          const bf16x8 A0 = *(const bf16x8*)(dH1b + l15 * H0B_LD + 64 * wave + 32 * q + 8 * g);
          const bf16x8 A1 = *(const bf16x8*)(dH1b + (16 + l15) * H0B_LD + 64 * wave + 32 * q + 8 * g);
#pragma unroll
          for (int tb = 0; tb < 4; ++tb) {
            bf16x8 Bv;                        // (the W1 shadow arrives as bf16: assembled, not converted)
#pragma unroll
            for (int e = 0; e < 8; ++e) Bv[e] = bw[8 * q + e][tb];
            acc[0][tb] = MFMA_BF16(A0, Bv, acc[0][tb]);
            acc[1][tb] = MFMA_BF16(A1, Bv, acc[1][tb]);
          }
          if (more) {
#pragma unroll
            for (int ks = 8 * q; ks < 8 * q + 8; ++ks)
              bw[ks] = ld4<BF16>(w1, (unsigned)(BWB_ROW(ks) * HID + i0 + 64 + 4 * l15));
          }
        }
      } else {
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
          const int kk = 64 * wave + 4 * ks + g;
          const float a0 = dH1s[l15 * H0_LD + kk];
          const float a1 = dH1s[(16 + l15) * H0_LD + kk];
#pragma unroll
          for (int tb = 0; tb < 4; ++tb) {
            acc[0][tb] = MFMA16(a0, bw[ks][tb], acc[0][tb]);
            acc[1][tb] = MFMA16(a1, bw[ks][tb], acc[1][tb]);
          }
          if ((ks & 3) == 3 && more) {
#pragma unroll
            for (int k2 = ks - 3; k2 <= ks; ++k2)
              bw[k2] = ld4<BF16>(w1, (unsigned)((64 * wave + 4 * k2 + g) * HID + i0 + 64 + 4 * l15));
          }
        }
      }
      if (more) {      // the next slice's H0 mask
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int f = tid + 256 * q;
          const unsigned row = (unsigned)BROW(row0 + (f >> 4));
          h0n[q] = ld4<BF16>(H0g, row * (unsigned)HID + (unsigned)(i0 + 64 + 4 * (f & 15)));
        }
      }
      float* myred = red + wave * 32 * T64_LD;
#pragma unroll
      for (int rtile = 0; rtile < 2; ++rtile)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int rl = 16 * rtile + 4 * g + reg;
          f32x4 v = (f32x4){acc[rtile][0][reg], acc[rtile][1][reg], acc[rtile][2][reg], acc[rtile][3][reg]};
          *(f32x4*)(myred + rl * T64_LD + 4 * l15) = v;
        }
    }
    // park the packed rows for the dW0 product
    if (itn == 0) xr_store(xr, Xr, n_x);
    __syncthreads();
    STAMP(p, 7);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int f = tid + 256 * q;
      const int rl = f >> 4, i4 = f & 15;
      f32x4 s = *(const f32x4*)(red + rl * T64_LD + 4 * i4);
#pragma unroll
      for (int w = 1; w < 4; ++w) s += *(const f32x4*)(red + w * 32 * T64_LD + rl * T64_LD + 4 * i4);
      f32x4 out;
#pragma unroll
      for (int e = 0; e < 4; ++e) out[e] = ((float)h0v[q][e] > 0.f) ? s[e] * dscale : 0.f;   // rows >= B carry s = 0
      *(f32x4*)(dH0s + rl * T64_LD + 4 * i4) = out;
    }
    __syncthreads();

    STAMP(p, 8);
    float* slabB = p.sc.slab_b + p.sc.slab_b_off[net] + rt * (HID * k0 + HID);
    // [dW0 | db0][i][kc] partial = sum_r dH0[r][i] * [X | 1][r][kc].  A = [X|1] (m = kc), B = dH0 (n = i):
    // a lane's 4 accumulator registers are 4 consecutive kc of one i.  This wave: i in [i0 + 16*wave, +16).
    {
      // (multi-slice blocks: this block's lane masks — kc < k0, kc == k0, kc0 + 3 < k0 per column tile and register — are
      //  loop-invariant; hipcc hoisted all ~80 of them, as 64-bit masks, in front of the slice loop and spilled 170-180
      //  SGPRs to keep them alive across it.  A per-iteration copy of k0 the compiler cannot see through keeps them where
      //  they are used.  The outer name is shadowed on purpose.)
      int k0_ = k0;
      if constexpr (MULTI) asm volatile("" : "+s"(k0_));
      const int k0 = k0_;
      const int k1 = k0 + 1;
      const int nct = (k1 + 15) >> 4;
      f32x4 acc[9];
#pragma unroll
      for (int ct = 0; ct < 9; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
      float bv[8];
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) bv[ks] = dH0s[(4 * ks + g) * T64_LD + 16 * wave + l15];
#pragma unroll
      for (int ct = 0; ct < 9; ++ct) {
        if (ct < nct) {
          const int kc = 16 * ct + l15;
          const int kcc = min(kc, k0 - 1);
          float xa[8];
#pragma unroll
          for (int ks = 0; ks < 8; ++ks) xa[ks] = Xr[(4 * ks + g) * ld + xoff + kcc];
          if (BF16) {      // the tile's 32 rows are ONE bf16 MFMA (lane group g holds rows 4 ks + g of both operands)
            float a8[8];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) a8[ks] = (kc < k0) ? xa[ks] : ((kc == k0) ? 1.f : 0.f);
            acc[ct] = MFMA_BF16(pack8s(a8), pack8s(bv), acc[ct]);
          } else {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
              const float a = (kc < k0) ? xa[ks] : ((kc == k0) ? 1.f : 0.f);   // ones column -> db0
              acc[ct] = MFMA16(a, bv[ks], acc[ct]);
            }
          }
        }
      }
      STAMP(p, 12);
      // straight from the accumulators into the row-tile slab: a lane's 4 registers of a tile are 4 consecutive kc of one
      // i, i.e. 16 contiguous bytes of the [i][kc] slab (4-byte aligned: k0 is odd as often as not) — one unaligned
      // 16-byte store where the whole run lies below k0, single words around the k0 column (= db0).  Staging the tile in
      // LDS for aligned float4 stores cost a barrier and two passes (1.8 k cycles of every (b) block's tail).
      {
        float* dstW = slabB + i0 * k0;
        float* dstB = slabB + HID * k0 + i0;
        const int il = 16 * wave + l15;
#pragma unroll
        for (int ct = 0; ct < 9; ++ct) {
          if (ct < nct) {
            const int kc0 = 16 * ct + 4 * g;
            if (kc0 + 3 < k0) {
              *(f32x4u*)(dstW + (unsigned)(il * k0 + kc0)) = acc[ct];
            } else {
#pragma unroll
              for (int reg = 0; reg < 4; ++reg) {
                const int kc = kc0 + reg;
                if (kc < k0) dstW[(unsigned)(il * k0 + kc)] = acc[ct][reg];
                else if (kc == k0) dstB[il] = acc[ct][reg];
              }
            }
          }
        }
      }
    }
    if (!more) break;
    h0v[0] = h0n[0]; h0v[1] = h0n[1];
    i0 += 64;
    }   // (red / dH0s of the next slice are written behind its own barriers: every thread has left this slice's dW0 reads)
    STAMP(p, 9);
    RT_STAMP(p, 14, rt_entry_);
    RT_STAMP(p, 15, iql_realtime());
  }
}

// ---------------------------------------------------------------------------
// bf16 OPERAND IMAGES of W1 and W0 (bf16 path; read by the large-batch forward, iqlhip_lb_kernels.h): the weights in the
// order the MFMA consumes them, so that a wave's fragment load is 1 KB of consecutive memory (with row-major weights the
// 16 lanes of a group read 16 different rows: the CU's vector-memory pipe handles about one (lane group, cache line)
// pair per cycle, and the forward's 56 fragment loads per wave cost ~14 k cycles that way).
//   fragment (slab w = unit >> 6, tile ct = (unit >> 4) & 3, k-block kb = k >> 5): 64 lanes x 8 elements,
//   lane = (unit & 15) + 16 ((k >> 3) & 3), element = k & 7
// One image per net slot (V, Q1, Q2, pi, target Q1, target Q2): [W1: 65 536 | W0: 256 x 32 ceil(k_in / 32), zero beyond k_in].
#define IMG_W0_OFF 65536
#define IMG_STRIDE (65536 + 256 * 128)
__device__ __forceinline__ unsigned img_w1_off(int unit, int k) {
  return (unsigned)((((unit >> 6) * 4 + ((unit >> 4) & 3)) * 8 + (k >> 5)) * 512 + ((unit & 15) + 16 * ((k >> 3) & 3)) * 8 + (k & 7));
}
__device__ __forceinline__ unsigned img_w0_off(int unit, int k, int nkb) {
  return (unsigned)(IMG_W0_OFF + (((unit >> 6) * 4 + ((unit >> 4) & 3)) * nkb + (k >> 5)) * 512 + ((unit & 15) + 16 * ((k >> 3) & 3)) * 8 + (k & 7));
}
// the four arena elements e .. e + 3 of a net (w1 / w0 offsets and k_in given) -> its image, if they are W1 or W0 elements
__device__ __forceinline__ void img_store4(__bf16* img, long long e, long long w1, long long w0, int k_in, const f32x4 v) {
  if (e >= w1 && e < w1 + 65536) {
    const int idx = (int)(e - w1);
    bf16x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = (__bf16)v[i];
    *(bf16x4*)(img + img_w1_off(idx >> 8, idx & 255)) = r;
  } else if (e >= w0 && e < w0 + 256 * k_in) {
    const int nkb = (k_in + 31) >> 5;
    // (row / column of element idx of the [256][k_in] matrix without an integer division: idx < 2^15, so the rounded
    //  quotient is at most one off — fixed up by the remainder's sign)
    const float inv_k = __builtin_amdgcn_rcpf((float)k_in);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = (int)(e - w0) + i;
      int unit = (int)((float)idx * inv_k);
      int k = idx - unit * k_in;
      if (k < 0) { unit -= 1; k += k_in; }
      if (k >= k_in) { unit += 1; k -= k_in; }
      img[img_w0_off(unit, k, nkb)] = (__bf16)v[i];
    }
  }
}

// ---------------------------------------------------------------------------
// Gradient assembly + Adam + Polyak.  Element e (float4 granularity) of the flat arena.
struct UpdParams {
  iqlhip_layout L;
  iqlhip_step_scalars sc;
  float tau, one_minus_tau;
  float* params;
  float* target;
  float* m;
  float* v;
  const float* slab_a;
  const float* slab_b;
  long long slab_b_off[4];
  const float* flat_grads;  // when non-null: gradient already summed (DP path), n_params + 4 words
  // direct-read exchange (xGMI peer-to-peer): n_peer > 0 -> the gradient is the sum, in rank order, of the n_peer flat
  // buffers peer_flat[0..n_peer) (this rank's own buffer included; the others are IPC-mapped peer memory, read with
  // system-scope loads).  Every rank forms the same sum in the same order: replicas stay bitwise equal.
  const float* peer_flat[IQLHIP_MAX_WORLD];
  int n_peer;
  // peer_direct: no flatten kernel ran — peer_flat[r] IS rank r's chunk slab (the backward wrote w1 / b1 / w2 / b2 /
  // log_std gradients straight into the exchange block; batches of <= 256 rows have ONE chunk slab, i.e. the gradient
  // itself), the w0 / b0 gradients are still per-row-tile partial slabs in peer_slab_b[r] (summed here, slab order
  // then rank order) and the loss sums are in peer_loss[r] ([4][64] like DevScratch::loss_parts)
  int peer_direct;
  const float* peer_slab_b[IQLHIP_MAX_WORLD];
  const float* peer_loss[IQLHIP_MAX_WORLD];
  float* loss_parts;
  float* losses;            // [4]
  float* losses_mirror;     // nullable: second copy of the three losses (host-mapped pinned words: iqlhip_online_step)
  float* loss_ring;         // nullable
  int ring_slot;
  int n_chunk, n_rt;        // chunk slabs (slab_a) and row-tile slabs (slab_b) the backward wrote
  int n_loss;               // entries of loss_parts per loss: 256-row chunks of the batch (large-batch backward: its row blocks)
  // large-batch backward (iqlhip_lb_kernels.h; LB instantiations only): slab_a holds the row-contraction products (w1, w0, b0,
  // the policy's w2; n_chunk chunk-group slabs), slab_x the row blocks' partial sums of everything else (b1, scalar w2, b2,
  // log_std; n_x slabs) — both laid out like the arena
  const float* slab_x;
  int n_x;
  int batch_rows;
  const iqlhip_step_scalars* sched;  // when non-null the scalars of this launch are sched[sched_idx]
  int sched_idx;                     // (hipGraph replay: kernel arguments are frozen, the table is not)
  // chunk replay: the loss-ring slot of this launch is ring_slot + ring_hdr[HDR_BASE] (the chunk's first step inside
  // the call; a captured chunk is replayed at different positions of the ring); null = ring_slot as given
  const unsigned long long* ring_hdr;
  // the LAST update of a chunk moves the chunk header on by the chunk's adv_k steps of adv_rows rows (thread 0 of
  // block 0, the only reader of the header in this kernel — every other reader belongs to an earlier or later
  // kernel), so the next chunk of the call starts without a host-side launch in between; null elsewhere
  unsigned long long* adv_hdr;
  int adv_k, adv_rows;
  // eager steps that return the losses to the host: after the three loss words have been written to losses_mirror
  // (host-mapped pinned memory) this host-mapped word receives done_val (release, system scope) — the host spins on it
  // instead of calling a HIP synchronise (measured 12-17 us of host time after the GPU has finished).  Everything else
  // a caller can observe of the step is ordered by the stream as before.  Null elsewhere.
  unsigned long long* done_flag;
  unsigned long long done_val;
  // bf16 path: bf16 shadows of the parameter and target arenas (same element offsets), written next to the fp32
  // masters so that the following forward / backward read their W1 fragments at half the bytes; null on the fp32 path
  __bf16* wsh;
  __bf16* tsh;
  __bf16* wimg;             // bf16 path: operand images [6][IMG_STRIDE] (slots V, Q1, Q2, pi, target Q1, target Q2); nullable
};

__device__ __forceinline__ int net_of(const iqlhip_layout& L, long long e) {
  int n = 0;
#pragma unroll
  for (int i = 1; i < 4; ++i) if (e >= L.net[i].seg_begin) n = i;
  return n;
}

// The layout words of a lane's net, WITHOUT indexing the kernel-argument block by the lane's net: `u.L.net[net]` with a
// per-lane index is a vector load from the argument block — a memory round trip that the gradient loads then wait for
// (they need these words for their address), i.e. a second dependent round trip in a kernel that otherwise has one.
// All four nets' words are uniform scalar loads; the lane selects.
struct NetWords { long long w0, b0; int k_in; long long slab_b_off; };
__device__ __forceinline__ NetWords net_words(const UpdParams& u, int net) {
  NetWords r;
  r.w0 = u.L.net[0].w0; r.b0 = u.L.net[0].b0; r.k_in = u.L.net[0].k_in; r.slab_b_off = u.slab_b_off[0];
#pragma unroll
  for (int i = 1; i < 4; ++i) {
    const bool is = (net == i);
    r.w0 = is ? u.L.net[i].w0 : r.w0;
    r.b0 = is ? u.L.net[i].b0 : r.b0;
    r.k_in = is ? u.L.net[i].k_in : r.k_in;
    r.slab_b_off = is ? u.slab_b_off[i] : r.slab_b_off;
  }
  return r;
}

template <bool LB = false>
__device__ __forceinline__ f32x4 slab_grad(const UpdParams& u, long long e, int net) {
  const NetWords nl = net_words(u, net);
  f32x4 gsum = (f32x4){0.f, 0.f, 0.f, 0.f};
  // slabs are read 8 at a time with unconditional (clamped) loads: a load inside a runtime-count loop
  // would be waited for individually — one dependent round trip per slab.
  const float* base;
  long long stride;
  int n;
  if (LB) {
    // [w1 | w0 | b0 | b1 | w2 | b2 | log_std]: the row contractions (w1, w0, the policy's w2) come from the chunk-group
    // slabs, every plain sum over rows (the rest) from the row blocks' slabs
    const long long w2b = nl.b0 + 2 * HID, b2b = w2b + (long long)HID * ((net == IQLHIP_NET_PI) ? u.L.net[IQLHIP_NET_PI].d_out : 1);
    // (the row blocks' slabs have been summed into chunk-group slab 0 by iql_bwd_gemm_kernel's reduction jobs)
    const bool in_x = (e >= nl.b0) && !(net == IQLHIP_NET_PI && e >= w2b && e < b2b);
    stride = u.L.n_params;
    base = u.slab_a + e;
    n = in_x ? 1 : u.n_chunk;
  } else if (e >= nl.w0 && e < nl.b0 + HID) {
    stride = (long long)HID * nl.k_in + HID;
    base = u.slab_b + nl.slab_b_off + (e - nl.w0);
    n = u.n_rt;
  } else {
    stride = u.L.n_params;
    base = u.slab_a + e;
    n = u.n_chunk;
  }
  if (n == 1) return *(const f32x4*)base;
  for (int r0 = 0; r0 < n; r0 += 8) {
    f32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = *(const f32x4*)(base + (long long)min(r0 + j, n - 1) * stride);
#pragma unroll
    for (int j = 0; j < 8; ++j) if (r0 + j < n) gsum += v[j];
  }
  return gsum;
}

__device__ __forceinline__ void loss_words(const UpdParams& u, float out[4]) {
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < u.n_loss; ++c)
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] += u.loss_parts[k * 64 + c];
  out[0] = s[0];   // sum_r w u^2
  out[1] = s[1];   // sum_r e1^2
  out[2] = s[2];   // sum_r e2^2
  out[3] = s[3];   // sum_r w bc
}

// `hdr` (nullable): device words {size, seed, offset, step0}: a captured graph is replayed with new values.
__global__ __launch_bounds__(256) void iql_dropmask_kernel(unsigned* bits, int n_words, unsigned thresh,
                                                           unsigned long long seed, unsigned long long step,
                                                           const unsigned long long* hdr, int k) {
  if (hdr) { seed = hdr[HDR_DROP_SEED]; step = hdr[HDR_DROP_STEP] + (unsigned long long)k; }
  dropmask_words(bits, n_words, thresh, seed, step, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
}

// First launch of an iqlhip_train_steps call (the only one that is not part of a chunk): block 0 publishes the call's
// header words; all blocks copy the call's scalar table from the library's pinned, host-mapped slot (read over PCIe,
// n_steps x 48 B) into its device copy, and — unless the previous call left them staged (B == 0) — gather the rows
// of the call's step 0 (indices drawn on the spot from the by-value header) and draw its dropout keep-bits.
// The block that finishes last acknowledges the table read in a host-mapped word (`ack`): the host reuses the pinned
// slot once it sees the call's number there — no event record in the stream, no HIP call on the host to test it.
__global__ __launch_bounds__(256) void iql_call_setup_kernel(unsigned long long* hdr, ChunkHdr h,
                                                             iqlhip_step_scalars* sched_call,
                                                             const iqlhip_step_scalars* sched_src, int n_steps,
                                                             const float* rows, long long ld, float* xb, int B,
                                                             unsigned* drop_dst, int drop_words, unsigned drop_thresh,
                                                             unsigned* arrivals, unsigned long long* ack,
                                                             unsigned long long ack_val) {
  if (blockIdx.x == 0 && threadIdx.x < HDR_WORDS) hdr[threadIdx.x] = h.w[threadIdx.x];
  const f32x4* s = (const f32x4*)sched_src;
  f32x4* d = (f32x4*)sched_call;
  const int n4 = n_steps * (int)(sizeof(iqlhip_step_scalars) / sizeof(f32x4));
  for (int i = (int)blockIdx.x * 256 + (int)threadIdx.x; i < n4; i += (int)gridDim.x * 256) d[i] = s[i];
  // (a thread's stores carry the data its loads returned: once every thread of every block is past this point the
  //  pinned slot has been read completely)
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned prev = __hip_atomic_fetch_add(arrivals, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev + 1u == gridDim.x) {
      __hip_atomic_store(arrivals, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(ack, ack_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  if (B > 0)
    gather_rows_drawn(rows, ld, xb, B, h.w[HDR_SEED], h.w[HDR_OFFSET], h.w[HDR_POS], h.w[HDR_SIZE],
                      (int)blockIdx.x * 256 + (int)threadIdx.x, (int)gridDim.x * 256);
  if (drop_dst)
    dropmask_words(drop_dst, drop_words, drop_thresh, h.w[HDR_DROP_SEED], h.w[HDR_DROP_STEP],
                   ((int)gridDim.x - 1 - (int)blockIdx.x) * 256 + (int)threadIdx.x, (int)gridDim.x * 256);
}

// bf16 shadows rebuilt from the fp32 masters (start of every library call on the bf16 path: the caller owns the
// masters and may have written them through its own tensors since the last update kernel ran).
__global__ __launch_bounds__(256) void iql_shadow_refresh_kernel(const float* params, const float* target, __bf16* wsh,
                                                                 __bf16* tsh, long long n_params, long long n_target,
                                                                 iqlhip_layout L, __bf16* wimg) {
  const long long e = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (e < n_params) {
    const f32x4 v = *(const f32x4*)(params + e);
    st4<true>((float*)wsh, (unsigned)e, v);
    if (wimg) {
      const int net = net_of(L, e);
      img_store4(wimg + (size_t)net * IMG_STRIDE, e, L.net[net].w1, L.net[net].w0, L.net[net].k_in, v);
    }
  }
  if (e < n_target) {
    const f32x4 v = *(const f32x4*)(target + e);
    st4<true>((float*)tsh, (unsigned)e, v);
    if (wimg) {
      const long long ea = e + L.target_src;       // the element's offset in the parameter arena: nets 1 (Q1), 2 (Q2)
      const int net = net_of(L, ea);
      img_store4(wimg + (size_t)(3 + net) * IMG_STRIDE, ea, L.net[net].w1, L.net[net].w0, L.net[net].k_in, v);
    }
  }
}

// Diagnostic (tools/gpu_call_overhead.py): a host-mapped word that says "everything queued before me has run".
__global__ void iql_debug_flag_kernel(unsigned long long* flag, unsigned long long v) {
  __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(256) void iql_gather_kernel(const float* rows, long long ld, const long long* idx,
                                                         float* xb, int n, long long n_rows) {
  gather_rows_flat(rows, ld, idx, xb, n, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256, n_rows);
}
// The same with the indices read straight from a pinned, host-mapped slot (ReplayBuffer.sample: no H2D copy, no
// event): every block first parks the indices of ITS rows in LDS, then the block that finishes last acknowledges the
// slot in a host-mapped word — the host reuses the slot once it sees the call's number there (cf. iql_call_setup_kernel).
// (n_rows: an index outside [0, n_rows) — the host checks them before the launch; this only guards a reused pinned slot —
//  is never dereferenced: its row is filled with NaN, as in gather_rows_flat)
__global__ __launch_bounds__(256) void iql_gather_hostidx_kernel(const float* rows, long long ld, const long long* idx_host,
                                                                 float* xb, int n, long long n_rows, unsigned* arrivals,
                                                                 unsigned long long* ack, unsigned long long ack_val) {
  __shared__ long long s_idx[260];
  const int q = (int)(ld >> 2);
  const int e0 = (int)blockIdx.x * 256;
  const int r_first = e0 / q;
  const int r_last = min((e0 + 255) / q, n - 1);
  if ((int)threadIdx.x <= r_last - r_first) s_idx[threadIdx.x] = idx_host[r_first + threadIdx.x];
  __syncthreads();                       // (the slot's words this block needs have been read)
  if (threadIdx.x == 0) {
    const unsigned prev = __hip_atomic_fetch_add(arrivals, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev + 1u == gridDim.x) {
      __hip_atomic_store(arrivals, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(ack, ack_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  const int e = e0 + (int)threadIdx.x;
  if (e < n * q) {
    const int r = e / q, c4 = e - r * q;
    const long long i = s_idx[r - r_first];
    f32x4 v;
    if (i < 0 || i >= n_rows) {
      const float nanv = __builtin_nanf("");
      v = (f32x4){nanv, nanv, nanv, nanv};
    } else {
      v = *(const f32x4*)(rows + i * ld + 4 * c4);
    }
    *(f32x4*)(xb + (long long)r * ld + 4 * c4) = v;
  }
}

// 16-byte accesses at SYSTEM scope (sc0 sc1): the load misses every cache level that is not coherent with another
// device's writes (IPC-mapped peer memory over xGMI), the store is written through to memory.  There is no 16-byte
// atomic, so these are the instructions the memory model uses for system-scope relaxed accesses, issued by hand.
// (load and wait are ONE asm statement: the compiler does not know that the load completes asynchronously and would
//  otherwise be free to copy the destination registers before a separate s_waitcnt has run)
__device__ __forceinline__ f32x4 load16_sys(const float* p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}
// eight such loads in flight together (one fabric round trip), then one wait
__device__ __forceinline__ void load16_sys_x8(f32x4 (&v)[8], const float* const (&p)[8]) {
  asm volatile(
      "global_load_dwordx4 %0, %8, off sc0 sc1\n\t"
      "global_load_dwordx4 %1, %9, off sc0 sc1\n\t"
      "global_load_dwordx4 %2, %10, off sc0 sc1\n\t"
      "global_load_dwordx4 %3, %11, off sc0 sc1\n\t"
      "global_load_dwordx4 %4, %12, off sc0 sc1\n\t"
      "global_load_dwordx4 %5, %13, off sc0 sc1\n\t"
      "global_load_dwordx4 %6, %14, off sc0 sc1\n\t"
      "global_load_dwordx4 %7, %15, off sc0 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
      : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7])
      : "memory");
}
__device__ __forceinline__ void store16_sys(float* p, f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Direct-read gradient exchange between the ranks of one node (one process per GPU, buffers mapped with hipIpc):
// after its flatten kernel (whose stores are complete at the kernel boundary) rank r stores the step number into ITS
// slot of every peer's flag block, then waits until every peer's slot of its OWN flag block has reached that number;
// the update kernel that follows reads all ranks' flat gradients.  Buffers alternate by step parity: rank r rewrites
// buffer b two steps later, i.e. after it has seen every peer's signal of the step in between, which a peer only
// sends after its own update (its reads of buffer b) has finished.
// The wait is bounded: a peer that never arrives (crashed rank) makes the lane give up after `timeout_ticks` of the
// 100 MHz wall clock, record the step in status[0] (sticky: later waits return at once) and let the stream drain —
// the host reports it (iqlhip_xch_status), the grid never hangs.
struct XchParams {
  unsigned long long* peer_flags[IQLHIP_MAX_WORLD];   // base of every rank's flag block [world][16] (own = local)
  unsigned long long* status;                          // [0] first timed-out step (0 = none), [1] polls (diagnostic)
  const unsigned long long* hdr;                       // HDR_XSTEP: steps exchanged before this chunk (chunk replay) ...
  unsigned long long xstep;                            // ... or, hdr == null, the same number as a kernel argument
  unsigned long long timeout_ticks;
  int rank, world;
};
__global__ __launch_bounds__(64) void iql_xch_signal_wait_kernel(XchParams x, int k) {
  const int t = threadIdx.x;
  const unsigned long long v = (x.hdr ? x.hdr[HDR_XSTEP] : x.xstep) + (unsigned long long)k + 1ull;
  if (t < x.world && t != x.rank) {
    // release at system scope, drained, THEN the flag (the explicit wait: hipcc may drop the one that belongs to the
    // fence when its own scoreboard is empty, and the flag must not overtake the write-back)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    wait_vm0();
    __hip_atomic_store(x.peer_flags[t] + 16 * x.rank, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long* mine = x.peer_flags[x.rank] + 16 * t;
    if (__hip_atomic_load(x.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0ull) {
      const unsigned long long t0 = wall_clock64();
      unsigned long long polls = 0;
      while (__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < v) {
        __builtin_amdgcn_s_sleep(4);
        if (((++polls) & 63ull) == 0ull && wall_clock64() - t0 > x.timeout_ticks) {
          __hip_atomic_store(x.status, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
    }
  }
}

// Writes the summed flat gradient (+ tail: value, q, actor loss contributions, spare) for the DP exchange.
// SYS: write-through system-scope stores (the buffer is read by peer GPUs).
template <bool SYS, bool LB = false>
__global__ __launch_bounds__(256) void iql_grad_flatten_kernel(UpdParams u, float* out) {
  const long long e = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (e < u.L.n_params) {
    const int net = net_of(u.L, e);
    const f32x4 gv = slab_grad<LB>(u, e, net);
    if (SYS) store16_sys(out + e, gv);
    else *(f32x4*)(out + e) = gv;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    float s[4];
    loss_words(u, s);
    const float ib = u.sc.inv_batch;
    const f32x4 tail = (f32x4){s[0] * ib, (s[1] * ib + s[2] * ib) * 0.5f, s[3] * ib, 0.f};
    if (SYS) store16_sys(out + u.L.n_params, tail);
    else *(f32x4*)(out + u.L.n_params) = tail;
  }
}

// FROM_TABLE: the per-step scalars come from the device table u.sched[u.sched_idx] (hipGraph replay: kernel
// arguments are frozen, the table is not); otherwise from the kernel argument u.sc.  Two instantiations
// rather than a run-time pointer select, which would turn every access into a flat load.
// PEER: the direct-read exchange variant (gradient = rank-ordered sum over UpdParams::peer_flat).
// Leading arguments (14 dwords, PRELOADED into SGPRs with the wave, cf. iql_bwd_kernel): the three state arenas, the chunk
// slab, the four segment starts and the end of the last segment as 32-bit element offsets, and q_flags: bit 0 = "the W1
// gradient of an element is the single word slab_a[e]" (one chunk slab, no exchange, not the large-batch form).  A block
// issues its m / v / p loads — and, for the 91 % of the elements that are W1, the gradient load — ~40 cycles after it starts
// instead of behind the ~600-cycle fetch of the by-value UpdParams, which now runs under those loads' latency.
#define UPD_EARLY_G 1u
template <bool FROM_TABLE, bool PEER, bool LB = false>
__global__ __launch_bounds__(256) void iql_update_kernel(float* q_p, float* q_m, float* q_v, const float* q_slab_a, unsigned q_s0,
                                                         unsigned q_s1, unsigned q_s2, unsigned q_s3, unsigned q_end,
                                                         unsigned q_flags, UpdParams u) {
  // XCD-affine element map: block (x = blockIdx & 7, q = blockIdx >> 3) — XCD x under the round-robin workgroup
  // dispatch — owns net x & 3, and of that net's arena segment the 64-float stripes of parity x >> 2: window q of 2 048
  // floats, 16 stripes of 16 threads.  The backward's blocks of net n run on XCDs n and n + 4 and a dW1 tile of column
  // parity h is written on XCD n + 4 h (W1 sits at the start of the segment, 4 stripes per row): the gradient is read
  // on the XCD that wrote it; the forward instances of net n (and the target copies') sit on XCDs n and n + 4 and read
  // the stripes their own XCD wrote, and the optimizer state is only ever touched by one XCD.
  const int ux = (int)(blockIdx.x & 7u), uq = (int)(blockIdx.x >> 3);
  const int net = ux & 3, uhalf = ux >> 2;
  // (segments are contiguous and 64-aligned: a net's segment ends where the next one begins, iqlhip_arena_layout)
  const long long seg_b = (long long)((net == 0) ? q_s0 : ((net == 1) ? q_s1 : ((net == 2) ? q_s2 : q_s3)));
  const long long seg_e = (long long)((net == 0) ? q_s1 : ((net == 1) ? q_s2 : ((net == 2) ? q_s3 : q_end)));
  const long long e = seg_b + (long long)uq * 2048 + (long long)((((int)threadIdx.x >> 4) * 2 + uhalf) * 64 + ((int)threadIdx.x & 15) * 4);
  if (e < seg_e) {
    // issue the state loads before the gradient sum so that all of them are in flight together — from the preloaded
    // arguments: nothing here waits for `u`
    f32x4 m = *(f32x4*)(q_m + e);
    f32x4 v = *(f32x4*)(q_v + e);
    f32x4 pw = *(f32x4*)(q_p + e);
    // W1 leads a segment (65 536 elements = 32 of the blocks' 2 048-float windows: the test is block-uniform)
    const bool early_g = !PEER && !LB && (q_flags & UPD_EARLY_G) != 0u && (e - seg_b) < 65536;
    f32x4 gr = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (early_g) gr = *(const f32x4*)(q_slab_a + e);
    __builtin_amdgcn_sched_barrier(0);      // (the loads above are issued BEFORE the argument fetch below is waited for)
    // every kernel-argument word the optimizer path uses, fetched in ONE batch of scalar loads (hipcc otherwise sinks
    // each load next to its first use: five dependent scalar-cache misses in front of the gradient loads).  ONE asm
    // statement for all of them: a volatile asm per word is ordered against the others and gets its own wait.
#define U64(x) ((unsigned long long)(x))
    asm volatile("" ::"s"(U64(u.L.net[0].seg_begin)), "s"(U64(u.L.net[1].seg_begin)), "s"(U64(u.L.net[2].seg_begin)),
                 "s"(U64(u.L.net[3].seg_begin)), "s"(U64(u.L.net[0].w0)), "s"(U64(u.L.net[1].w0)), "s"(U64(u.L.net[2].w0)),
                 "s"(U64(u.L.net[3].w0)), "s"(U64(u.L.net[0].b0)), "s"(U64(u.L.net[1].b0)), "s"(U64(u.L.net[2].b0)),
                 "s"(U64(u.L.net[3].b0)), "s"(u.L.net[0].k_in), "s"(u.L.net[1].k_in), "s"(u.L.net[2].k_in),
                 "s"(u.L.net[3].k_in), "s"(U64(u.slab_b_off[0])), "s"(U64(u.slab_b_off[1])), "s"(U64(u.slab_b_off[2])),
                 "s"(U64(u.slab_b_off[3])), "s"(U64(u.L.n_params)), "s"(U64(u.L.target_src)), "s"(U64((uintptr_t)u.params)),
                 "s"(U64((uintptr_t)u.target)), "s"(U64((uintptr_t)u.m)), "s"(U64((uintptr_t)u.v)),
                 "s"(U64((uintptr_t)u.slab_a)), "s"(U64((uintptr_t)u.slab_b)), "s"(U64((uintptr_t)u.flat_grads)),
                 "s"(U64((uintptr_t)u.sched)));
#undef U64

    const bool is_q = (net == IQLHIP_NET_Q1 || net == IQLHIP_NET_Q2);
    float* tp = u.target + (is_q ? (e - u.L.target_src) : 0);
    f32x4 t = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (is_q) t = *(f32x4*)tp;
    if (PEER) {
      // all ranks' contributions requested together (one fabric round trip), summed in rank order
      static_assert(IQLHIP_MAX_WORLD == 8, "load16_sys_x8");
      const NetWords nl = net_words(u, net);
      if (u.peer_direct && e >= nl.w0 && e < nl.b0 + HID) {
        // w0 / b0: every rank's <= 8 row-tile partial slabs, summed per rank in slab order (exactly slab_grad's sum,
        // i.e. what that rank's flatten kernel would have written), then over the ranks in rank order
        const long long stride = (long long)HID * nl.k_in + HID;
        const long long off = nl.slab_b_off + (e - nl.w0);
        gr = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < u.n_peer; ++r) {
          f32x4 pv[8];
          const float* pp[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) pp[j] = u.peer_slab_b[r] + off + (long long)min(j, u.n_rt - 1) * stride;
          load16_sys_x8(pv, pp);
          f32x4 gsum = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int j = 0; j < 8; ++j) if (j < u.n_rt) gsum += pv[j];
          gr = (r == 0) ? gsum : gr + gsum;
        }
      } else {
        f32x4 pv[8];
        const float* pp[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) pp[r] = u.peer_flat[min(r, u.n_peer - 1)] + e;
        load16_sys_x8(pv, pp);
        gr = pv[0];
#pragma unroll
        for (int r = 1; r < IQLHIP_MAX_WORLD; ++r) if (r < u.n_peer) gr += pv[r];
      }
    } else if (u.flat_grads) gr = *(const f32x4*)(u.flat_grads + e);
    else if (!early_g) gr = slab_grad<LB>(u, e, net);
    const int grp = (net == IQLHIP_NET_V) ? 0 : ((net == IQLHIP_NET_PI) ? 2 : 1);
    // (copy by value: a pointer that may address either the kernarg segment or global memory would make
    //  every access a flat load)
    iqlhip_step_scalars sc;
    if (FROM_TABLE) sc = u.sched[u.sched_idx];
    else sc = u.sc;
    const float gs = sc.grad_scale;
    const float step = -((grp == 0) ? sc.step_size[0] : ((grp == 1) ? sc.step_size[1] : sc.step_size[2]));
    const float bc2 = (grp == 0) ? sc.bc2_sqrt[0] : ((grp == 1) ? sc.bc2_sqrt[1] : sc.bc2_sqrt[2]);
    const float omb1 = sc.one_minus_beta1, b2 = sc.beta2, omb2 = sc.one_minus_beta2, eps = sc.eps;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      // (the fused multiply-adds are spelled out: left to -ffp-contract the four instantiations of this kernel are free
      //  to fuse differently, and the exchange variants must stay bitwise equal to the plain one)
      const float gk = (gs == 1.f) ? gr[k] : gr[k] * gs;
      m[k] = fmaf(omb1, gk - m[k], m[k]);
      v[k] = fmaf(omb2 * gk, gk, v[k] * b2);
      const float denom = sqrtf(v[k]) / bc2 + eps;
      pw[k] = fmaf(step, m[k] / denom, pw[k]);
    }
    *(f32x4*)(q_m + e) = m;
    *(f32x4*)(q_v + e) = v;
    *(f32x4*)(q_p + e) = pw;
    if (u.wsh) st4<true>((float*)u.wsh, (unsigned)e, pw);
    if (is_q) {
#pragma unroll
      for (int k = 0; k < 4; ++k) t[k] = fmaf(u.tau, pw[k], u.one_minus_tau * t[k]);
      *(f32x4*)tp = t;
      if (u.tsh) st4<true>((float*)u.tsh, (unsigned)(e - u.L.target_src), t);
    }
    if (LB && u.wimg) {      // large-batch bf16 path: the operand images of W1 / W0 (W1 leads a net's segment)
      const NetWords nw = net_words(u, net);
      img_store4(u.wimg + (size_t)net * IMG_STRIDE, e, seg_b, nw.w0, nw.k_in, pw);
      if (is_q) img_store4(u.wimg + (size_t)(3 + net) * IMG_STRIDE, e, seg_b, nw.w0, nw.k_in, t);      // (nets 1, 2 -> slots 4, 5)
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    float l[3];
    const float sc_ib = FROM_TABLE ? u.sched[u.sched_idx].inv_batch : u.sc.inv_batch;
    if (PEER && u.peer_direct) {
      // per rank the tail words its flatten kernel would have written (one chunk: loss_parts[k * 64]), summed in rank order
      const float ib = sc_ib;
      f32x4 t = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int r = 0; r < u.n_peer; ++r) {
        float s4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) s4[k] = 0.f + __hip_atomic_load(u.peer_loss[r] + k * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const f32x4 tr = (f32x4){s4[0] * ib, (s4[1] * ib + s4[2] * ib) * 0.5f, s4[3] * ib, 0.f};
        t = (r == 0) ? tr : t + tr;
      }
      l[0] = t[0]; l[1] = t[1]; l[2] = t[2];
    } else if (PEER) {
      f32x4 t = load16_sys(u.peer_flat[0] + u.L.n_params);
      for (int r = 1; r < u.n_peer; ++r) t += load16_sys(u.peer_flat[r] + u.L.n_params);
      l[0] = t[0]; l[1] = t[1]; l[2] = t[2];
    } else if (u.flat_grads) {
      l[0] = u.flat_grads[u.L.n_params + 0];
      l[1] = u.flat_grads[u.L.n_params + 1];
      l[2] = u.flat_grads[u.L.n_params + 2];
    } else {
      float s[4];
      loss_words(u, s);
      const float ib = 1.f / (float)u.batch_rows;
      l[0] = s[0] * ib;                         // mean(w u^2)                        iql.py:302
      l[1] = (s[1] * ib + s[2] * ib) * 0.5f;    // (mse(q1,y) + mse(q2,y)) / 2        iql.py:508
      l[2] = s[3] * ib;                         // mean(exp_adv * bc)                 iql.py:534
    }
    u.losses[0] = l[0]; u.losses[1] = l[1]; u.losses[2] = l[2];
    if (u.losses_mirror) { u.losses_mirror[0] = l[0]; u.losses_mirror[1] = l[1]; u.losses_mirror[2] = l[2]; }
    if (u.done_flag) __hip_atomic_store(u.done_flag, u.done_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (u.loss_ring) {
      const long long slot = (long long)u.ring_slot + (u.ring_hdr ? (long long)u.ring_hdr[HDR_BASE] : 0ll);
      *(float4*)(u.loss_ring + 4 * slot) = make_float4(l[0], l[1], l[2], 0.f);     // (host-mapped: one posted write)
    }
    if (u.adv_hdr) {      // the chunk is done: its successor finds its own per-launch values
      u.adv_hdr[HDR_POS] += (unsigned long long)u.adv_k * (unsigned long long)u.adv_rows;
      u.adv_hdr[HDR_DROP_STEP] += (unsigned long long)u.adv_k;
      u.adv_hdr[HDR_BASE] += (unsigned long long)u.adv_k;
      u.adv_hdr[HDR_XSTEP] += (unsigned long long)u.adv_k;
    }
  }
}

// Policy inference (GaussianPolicy.act / DeterministicPolicy.act, algorithms/finetune/iql.py:371-379, 404-413):
// states -> packed rows whose first S columns are the state (the rest zero), then iql_fwd_kernel with
// only_inst = 6, then this finish kernel over the policy head partials:
//   action = clamp(max_action * (tanh(pre) [+ exp(clamp(log_std)) * noise]), -max_action, +max_action)
__global__ void iql_pack_states_kernel(float* xb, int ld, int S, int n, const float* s, long long ld_s) {
  const int total = n * ld;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int i = e / ld;
    const int c = e - i * ld;
    xb[e] = (c < S) ? s[i * ld_s + c] : 0.f;
  }
}

// noise: caller-supplied N(0,1) values, or — rng_seed != 0 — drawn here: Philox4x32-10 keyed by the seed, counter
// (element, call), Box-Muller on two of its words (the draw of dist.sample(), iql.py:376, without a host-side
// random-number launch per env step).
__device__ __forceinline__ void actor_finish_elem(const float* heads_pi, int e, int A, float max_action, const float* log_std,
                                                  float ls_min, float ls_max, const float* noise, long long ld_noise,
                                                  unsigned long long rng_seed, unsigned long long rng_call, float* out,
                                                  long long ld_out) {
  const int row = e / A, dd = e - row * A;
  const f32x4 hp = *(const f32x4*)(heads_pi + (long long)e * NSPLIT);
  float a = tanh_via_exp(((hp[0] + hp[1]) + hp[2]) + hp[3]);      // same fixed order as the training step (sum4)
  if (noise != nullptr || rng_seed != 0ull) {
    const float sigma = (log_std != nullptr) ? expf(fminf(fmaxf(log_std[dd], ls_min), ls_max)) : 0.f;
    float z;
    if (noise != nullptr) {
      z = noise[row * ld_noise + dd];
    } else {
      uint32_t c[4] = {(uint32_t)e, (uint32_t)rng_call, (uint32_t)(rng_call >> 32), 0xAC7u};
      philox4x32_10(c, (uint32_t)rng_seed, (uint32_t)(rng_seed >> 32));
      const float u1 = ((float)(c[0] >> 8) + 0.5f) * (1.f / 16777216.f);       // (0, 1)
      const float u2 = ((float)(c[1] >> 8) + 0.5f) * (1.f / 16777216.f);
      z = sqrtf(-2.f * logf(u1)) * cosf(6.283185307179586f * u2);
    }
    a = a + sigma * z;
  }
  out[row * ld_out + dd] = fminf(fmaxf(a * max_action, -max_action), max_action);
}
// done_flag (nullable; one-block launches only): after every thread's store — ordered by the system-scope fence and
// the barrier — a host-mapped word receives done_val: the host of iqlhip_online_step spins on it instead of
// synchronising the stream.
__global__ void iql_actor_finish_kernel(const float* heads_pi, int n, int A, float max_action, const float* log_std,
                                        float ls_min, float ls_max, const float* noise, long long ld_noise,
                                        unsigned long long rng_seed, unsigned long long rng_call, float* out,
                                        long long ld_out, unsigned long long* done_flag, unsigned long long done_val) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n * A) actor_finish_elem(heads_pi, e, A, max_action, log_std, ls_min, ls_max, noise, ld_noise, rng_seed, rng_call, out, ld_out);
  if (done_flag) {
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(done_flag, done_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// Five row-major arrays (any strides) -> packed rows [s | a | s' | r | d | pad] of the staging batch.
__global__ void iql_pack_kernel(float* xb, int ld, int S, int A, int n, const float* s, long long ld_s, const float* a,
                                long long ld_a, const float* r, long long ld_r, const float* ns, long long ld_ns,
                                const float* d, long long ld_d) {
  const int total = n * ld;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int i = e / ld;
    const int c = e - i * ld;
    float v = 0.f;
    if (c < S) v = s[i * ld_s + c];
    else if (c < S + A) v = a[i * ld_a + (c - S)];
    else if (c < 2 * S + A) v = ns[i * ld_ns + (c - S - A)];
    else if (c == 2 * S + A) v = r[i * ld_r];
    else if (c == 2 * S + A + 1) v = d[i * ld_d];
    xb[e] = v;
  }
}

// ---------------------------------------------------------------------------
// Replay-buffer storage kernels.  Row layout: [s(S) | a(A) | s'(S) | r | d | pad].
__global__ void iql_rows_write_kernel(float* rows, long long ld, int S, int A, long long row0, long long n,
                                      const float* s, const float* a, const float* r, const float* ns,
                                      const float* d) {
  const int W = 2 * S + A + 2;
  const long long total = n * W;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const long long i = e / W;
    const int c = (int)(e - i * W);
    float v;
    if (c < S) v = s[i * S + c];
    else if (c < S + A) v = a[i * A + (c - S)];
    else if (c < 2 * S + A) v = ns[i * S + (c - S - A)];
    else if (c == 2 * S + A) v = r[i];
    else v = d[i];
    rows[(row0 + i) * ld + c] = v;
  }
}

__global__ void iql_rows_gather_kernel(const float* rows, long long ld, long long n_rows, int S, int A,
                                       const long long* idx, long long n, float* s, float* a, float* r, float* ns,
                                       float* d) {
  const int W = 2 * S + A + 2;
  const long long total = n * W;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const long long i = e / W;
    const int c = (int)(e - i * W);
    const long long ri = idx[i];
    const float v = (n_rows > 0 && (ri < 0 || ri >= n_rows)) ? __builtin_nanf("") : rows[ri * ld + c];     // (see gather_rows_flat)
    if (c < S) s[i * S + c] = v;
    else if (c < S + A) a[i * A + (c - S)] = v;
    else if (c < 2 * S + A) ns[i * S + (c - S - A)] = v;
    else if (c == 2 * S + A) r[i] = v;
    else d[i] = v;
  }
}

// Index draw: Philox4x32-10, counter = (offset + i/2), key = seed.
// idx[i] uniform over [0,size): 64 random bits, multiply-high (bias <= size / 2^64).
// `hdr` (nullable) = device words {size, seed, offset} so that a captured graph can be replayed
// with new values.
__global__ void iql_draw_indices_kernel(long long* idx, long long n, long long size, unsigned long long seed,
                                        unsigned long long offset, const unsigned long long* hdr) {
  if (hdr) { size = (long long)hdr[HDR_SIZE]; seed = hdr[HDR_SEED]; offset = hdr[HDR_OFFSET]; }
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (n + 1) / 2;
       i += (long long)gridDim.x * blockDim.x) {
    const unsigned long long ctr = offset + (unsigned long long)i;
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0x49514C48u /* "IQLH" */, 0u};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const unsigned long long r0 = ((unsigned long long)c[1] << 32) | c[0];
    const unsigned long long r1 = ((unsigned long long)c[3] << 32) | c[2];
    idx[2 * i] = (long long)__umul64hi(r0, (unsigned long long)size);
    if (2 * i + 1 < n) idx[2 * i + 1] = (long long)__umul64hi(r1, (unsigned long long)size);
  }
}

// ---------------------------------------------------------------------------
// Synthetic D4RL-shaped rows generated where they will live (bench data, SURVEY §8d's distributions, not a parity
// fixture): obs, next_obs ~ N(0,1); actions ~ U(-1,1) * 0.999; rewards ~ N(0,1) (antmaze flavour: -1 with p = 0.98,
// else 0); dones ~ Bernoulli(p_done).  One Philox4x32-10 block per element (key = seed, counter = element number):
// Box-Muller on two words for the normals.  At configs[3]'s 10 M rows the host generator of jsrl-corl_amd/synth.py takes
// ~30 s per rank (and a 1.7 GB upload); this takes milliseconds and every rank produces identical rows.
__global__ void iql_rows_fill_synth_kernel(float* rows, long long ld, int S, int A, long long row0, long long n,
                                           unsigned long long seed, float p_done, int antmaze) {
  const int W = 2 * S + A + 2;
  const long long total = n * W;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const long long i = e / W;
    const int c = (int)(e - i * W);
    const unsigned long long ctr = (unsigned long long)(row0 + i) * (unsigned long long)W + (unsigned long long)c;
    uint32_t k[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0x46494C4Cu /* "FILL" */, 0u};
    philox4x32_10(k, (uint32_t)seed, (uint32_t)(seed >> 32));
    const float u1 = ((float)(k[0] >> 8) + 0.5f) * (1.f / 16777216.f);       // (0, 1)
    const float u2 = ((float)(k[1] >> 8) + 0.5f) * (1.f / 16777216.f);
    float v;
    if (c < S || (c >= S + A && c < 2 * S + A)) v = sqrtf(-2.f * logf(u1)) * cosf(6.283185307179586f * u2);
    else if (c < S + A) v = (2.f * u1 - 1.f) * 0.999f;
    else if (c == 2 * S + A) v = antmaze ? ((u1 < 0.98f) ? -1.f : 0.f) : sqrtf(-2.f * logf(u1)) * cosf(6.283185307179586f * u2);
    else v = (u1 < p_done) ? 1.f : 0.f;
    rows[(row0 + i) * ld + c] = v;
  }
}

// ---------------------------------------------------------------------------
// Dataset ingest: column mean / std of n rows (compute_mean_std, algorithms/finetune/iql.py:77-80:
// mean = x.mean(0), std = x.std(0) + eps, population variance) and the in-place normalisation of the state columns of
// packed rows (normalize_states, :83-84: (x - mean) / std in fp32, IEEE division).  Sums are kept in float64 and
// combined in a fixed order (block partials, then one thread per column): deterministic, and closer to the exact
// value than numpy's float32 row-after-row sums — the parity statement is therefore a tolerance against numpy
// (tests: rel <= 1e-6 against a float64 numpy evaluation, <= 1e-3 against numpy's own float32 result at 1 M rows).
#define MS_BLOCKS 512
// pass 0: partial[b][c] = sum over the block's rows of x[r][c];  pass 1: sum of (x - mean[c])^2
template <int PASS>
__global__ __launch_bounds__(256) void iql_cols_moment_kernel(const float* x, long long ld, int ncols, long long n,
                                                              const double* mean, double* partial) {
  __shared__ double red[8][33];
  const int c_lane = threadIdx.x & 31, r_lane = threadIdx.x >> 5;     // 8 rows x 32 columns per pass
  for (int c0 = 0; c0 < ncols; c0 += 32) {
    const int c = c0 + c_lane;
    double acc = 0.0;
    if (c < ncols) {
      const double mu = PASS ? mean[c] : 0.0;
      for (long long r = (long long)blockIdx.x * 8 + r_lane; r < n; r += (long long)gridDim.x * 8) {
        const double v = (double)x[r * ld + c];
        acc += PASS ? (v - mu) * (v - mu) : v;
      }
    }
    red[r_lane][c_lane] = acc;
    __syncthreads();
    if (r_lane == 0 && c < ncols) {
      double s = red[0][c_lane];
#pragma unroll
      for (int k = 1; k < 8; ++k) s += red[k][c_lane];
      partial[(long long)blockIdx.x * ncols + c] = s;
    }
    __syncthreads();
  }
}
// one thread per column: fixed-order sum of the block partials -> mean (pass 0) or std + eps (pass 1)
template <int PASS>
__global__ void iql_cols_finish_kernel(const double* partial, int nblocks, int ncols, long long n, float eps,
                                       double* mean64, float* out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ncols) return;
  double s = 0.0;
  for (int b = 0; b < nblocks; ++b) s += partial[(long long)b * ncols + c];
  if (PASS == 0) {
    mean64[c] = s / (double)n;
    out[c] = (float)(s / (double)n);
  } else {
    out[c] = (float)sqrt(s / (double)n) + eps;       // np.std(float32 array) returns float32, then + eps in float32
  }
}

__global__ void iql_rows_normalize_kernel(float* rows, long long ld, int S, int A, long long row0, long long n,
                                          const float* mean, const float* stdv) {
  const long long total = n * 2 * S;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const long long i = e / (2 * S);
    const int c2 = (int)(e - i * 2 * S);
    const int c = (c2 < S) ? c2 : c2 - S;
    float* p = rows + (row0 + i) * ld + ((c2 < S) ? c : S + A + c);
    *p = (*p - mean[c]) / stdv[c];
  }
}

// ---------------------------------------------------------------------------
// One online iteration's buffer work in one launch (iqlhip_online_step): the new transition `row_host` (pinned,
// host-mapped, ld floats) is stored at ring row `pointer`, and the batch rows[idx_host[r]] (indices in pinned,
// host-mapped memory, exactly as np.random.randint drew them) are gathered into xb.  A sampled index equal to
// `pointer` reads the new row from `row_host` itself — the ring write of block 0 may not have happened yet.
__global__ __launch_bounds__(256) void iql_online_gather_kernel(float* rows, long long ld, long long pointer,
                                                                const float* row_host, const long long* idx_host,
                                                                float* xb, int n) {
  __shared__ long long s_idx[260];
  const int q = (int)(ld >> 2);
  const int e0 = (int)blockIdx.x * 256;
  const int r_first = e0 / q;
  const int r_last = min((e0 + 255) / q, n - 1);
  if ((int)threadIdx.x <= r_last - r_first) s_idx[threadIdx.x] = idx_host[r_first + threadIdx.x];
  if (blockIdx.x == 0 && (int)threadIdx.x < q)
    *(f32x4*)(rows + pointer * ld + 4 * threadIdx.x) = *(const f32x4*)(row_host + 4 * threadIdx.x);
  __syncthreads();
  const int e = e0 + (int)threadIdx.x;
  if (e < n * q) {
    const int r = e / q, c4 = e - r * q;
    const long long i = s_idx[r - r_first];
    const float* src = (i == pointer) ? row_host : rows + i * ld;
    *(f32x4*)(xb + (long long)r * ld + 4 * c4) = *(const f32x4*)(src + 4 * c4);
  }
}
