// iqlhip_kernels.h — device code of the IQL step for gfx950 (MI355X, CDNA4).
//
// One step = three launches (cut at every all-to-all seam):
//   iql_fwd_kernel     7 MLP instances x row-tiles(32 rows) x 4 column slices
//   iql_bwd_kernel     (a) dW1 tiles over a 256-row chunk, (b) dH0/dW0 per row-tile
//   iql_update_kernel  slab-sum of gradients + Adam (3 lr groups) + Polyak + losses
//
// All GEMMs use v_mfma_f32_16x16x4_f32 (exact fp32 fma chain).  Lane maps
// (wave64, l = lane, l15 = l&15, g = l>>4):
//   A operand: A[m=l15][k=g]      B operand: B[k=g][n=l15]
//   C/D:       D[m=4g+reg][n=l15] (reg = 0..3)
// "float4-along-k" trick: a lane loads 4 consecutive k of its row once and
// feeds element t to the t-th of 4 MFMAs — A and B use the same (g,t)->k map,
// so the 4 MFMAs cover 16 k exactly once.  "float4-along-n" trick: a lane
// loads 4 consecutive output columns and MFMA t produces columns {4n+t}.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "iqlhip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define HID 256
#define RT_ROWS 32          // rows per forward / (b) block
#define CHUNK_ROWS 256      // rows per (a) block chunk
#define H0_LD 260           // LDS row stride of a [rows][256] tile (16-B aligned, bank-shifted)
#define T64_LD 68           // LDS row stride of a [rows][64] tile
#define NSPLIT 4            // column slices of the hidden layer per row tile

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

struct DevBatch {
  const float *s, *a, *r, *ns, *d;
  long long ld_s, ld_a, ld_r, ld_ns, ld_d;
  const long long* idx;   // nullable
  int rows;
};

struct DevScratch {
  float* h0;        // [4][max_batch][256]  post-ReLU layer-0 activations of V(s),Q1,Q2,pi
  float* h1;        // [4][max_batch][256]
  float* heads;     // scalar instances: [6][NSPLIT][max_batch]; pi: [NSPLIT][max_batch][A] after that
  float* slab_a;    // [n_chunk_max][n_params]  chunk slabs: w1,b1,w2,b2,log_std grads
  float* slab_b;    // per net [n_rt_max][256*k_in+256]  row-tile slabs: w0,b0 grads
  float* loss_parts;// [4][n_chunk_max]: value, q1, q2(err^2 sums), actor
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
    if ((p).stamps && threadIdx.x == 0) {                                                  \
      (p).stamps[(long long)blockIdx.x * 16 + (i)] = t_;                                   \
      if ((i) == 0) (p).stamps[(long long)blockIdx.x * 16 + 15] = wall_clock64();          \
    }                                                                                      \
  } while (0)
#else
#define STAMP(p, i) do {} while (0)
#endif

struct StepParams {
  unsigned long long* stamps;
  iqlhip_layout L;
  iqlhip_hyper hy;
  const float* params;
  const float* target;
  DevScratch sc;
  DevBatch b;
  int S, A, policy;
  float inv_batch;
};

// ---------------------------------------------------------------------------
__device__ __forceinline__ int xld_for(int k0p) { return ((k0p + 29) / 32) * 32 + 2; }

// instance -> (net layout index, uses target arena, input kind, activation slot)
// input kind: 0 = s, 1 = s', 2 = [s|a]
__device__ __forceinline__ void inst_info(int inst, int& net, bool& tgt, int& kind, int& slot) {
  switch (inst) {
    case 0: net = IQLHIP_NET_V;  tgt = false; kind = 1; slot = -1; break;  // V(s')
    case 1: net = IQLHIP_NET_V;  tgt = false; kind = 0; slot = 0;  break;  // V(s)
    case 2: net = IQLHIP_NET_Q1; tgt = true;  kind = 2; slot = -1; break;  // Qt1
    case 3: net = IQLHIP_NET_Q2; tgt = true;  kind = 2; slot = -1; break;  // Qt2
    case 4: net = IQLHIP_NET_Q1; tgt = false; kind = 2; slot = 1;  break;  // Q1
    case 5: net = IQLHIP_NET_Q2; tgt = false; kind = 2; slot = 2;  break;  // Q2
    default: net = IQLHIP_NET_PI; tgt = false; kind = 0; slot = 3; break;  // pi
  }
}

__device__ __forceinline__ const float* net_base(const StepParams& p, bool tgt) {
  return tgt ? (p.target - p.L.target_src) : p.params;
}

__device__ __forceinline__ long long src_row(const DevBatch& b, int r) {
  return b.idx ? b.idx[r] : (long long)r;
}

// Gather RT_ROWS input rows of `kind` into Xs[32][xld] (zero padded).
__device__ __forceinline__ void gather_rows(const StepParams& p, int kind, int row0, int k0, int xld, float* Xs) {
  const int tid = threadIdx.x;
  const int rl = tid >> 3;
  const int row = row0 + rl;
  const bool valid = row < p.b.rows;
  const long long j = valid ? src_row(p.b, row) : 0;
  const float* base0;
  long long ld0;
  if (kind == 1) { base0 = p.b.ns; ld0 = p.b.ld_ns; } else { base0 = p.b.s; ld0 = p.b.ld_s; }
  const float* r0 = base0 + j * ld0;
  const float* r1 = p.b.a + j * p.b.ld_a;
  const int S = p.S;
  for (int c = (tid & 7); c < xld; c += 8) {
    float v = 0.f;
    if (valid && c < k0) v = (c < S) ? r0[c] : r1[c - S];
    Xs[rl * xld + c] = v;
  }
}

// ---------------------------------------------------------------------------
// Forward: block = (instance, row tile of 32 rows, column slice ns of 64 hidden-1 units).
// grid = 8 * n_rt * NSPLIT; blockIdx & 7 = instance (7 = idle) so that the
// blocks of one instance share an XCD and hence one L2 copy of its weights.
__global__ __launch_bounds__(256) void iql_fwd_kernel(StepParams p) {
  const int bid = blockIdx.x;
  const int inst = bid & 7;
  if (inst >= 7) return;
  const int rest = bid >> 3;
  const int ns = rest & (NSPLIT - 1);
  const int rt = rest >> 2;
  const int row0 = rt * RT_ROWS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;

  int net, kind, slot;
  bool tgt;
  inst_info(inst, net, tgt, kind, slot);
  const iqlhip_net_layout& nl = p.L.net[net];
  const float* base = net_base(p, tgt);
  const float* w0 = base + nl.w0;
  const float* b0 = base + nl.b0;
  const float* w1 = base + nl.w1;
  const float* b1 = base + nl.b1;
  const float* w2 = base + nl.w2;
  const int k0 = nl.k_in;
  const int k0p = (k0 + 3) & ~3;
  const int xld = xld_for(k0p);
  const int D = nl.d_out;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* H0s = smem;                         // [32][H0_LD]
  float* H1s = H0s + RT_ROWS * H0_LD;        // [32][T64_LD]
  float* Xs = H1s + RT_ROWS * T64_LD;        // [32][xld]

  STAMP(p, 0);
  // ---- prefetch this wave's W1 rows (16 output units x 256 k) as MFMA B fragments
  const int n1 = ns * 64 + wave * 16 + l15;  // hidden-1 unit of this lane
  f32x4 bw[16];
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) bw[ks] = *(const f32x4*)(w1 + (long long)n1 * HID + 16 * ks + 4 * g);

  gather_rows(p, kind, row0, k0, xld, Xs);
  __syncthreads();
  STAMP(p, 1);

  // ---- layer 0: this wave computes H0[32][64*wave .. +64)
  {
    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nks = k0p >> 2;
    const float* wrow[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) wrow[ct] = w0 + (long long)(wave * 64 + ct * 16 + l15) * k0;
    float bcur[4], bnxt[4];
    {
      const int k = g;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) bcur[ct] = (k < k0) ? wrow[ct][k] : 0.f;
    }
    for (int ks = 0; ks < nks; ++ks) {
      const int kn = 4 * (ks + 1) + g;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) bnxt[ct] = (kn < k0) ? wrow[ct][kn] : 0.f;
      const float a0 = Xs[l15 * xld + 4 * ks + g];
      const float a1 = Xs[(16 + l15) * xld + 4 * ks + g];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        acc[0][ct] = MFMA16(a0, bcur[ct], acc[0][ct]);
        acc[1][ct] = MFMA16(a1, bcur[ct], acc[1][ct]);
      }
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) bcur[ct] = bnxt[ct];
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const int n = wave * 64 + ct * 16 + l15;
      const float bias = b0[n];
#pragma unroll
      for (int rtile = 0; rtile < 2; ++rtile)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const float h = fmaxf(acc[rtile][ct][reg] + bias, 0.f);
          H0s[(rtile * 16 + 4 * g + reg) * H0_LD + n] = h;
        }
    }
  }
  __syncthreads();
  STAMP(p, 2);

  // save H0 columns [64*ns, +64) of the trainable instances for the backward pass
  if (slot >= 0) {
    float* dst = p.sc.h0 + (long long)slot * p.sc.max_batch * HID;
    const int rl = tid >> 3;
    const int row = row0 + rl;
    if (row < p.b.rows) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = ns * 64 + 4 * ((tid & 7) + 8 * j);
        *(f32x4*)(dst + (long long)row * HID + col) = *(const f32x4*)(H0s + rl * H0_LD + col);
      }
    }
  }

  // ---- layer 1: this wave computes H1[32][n1 tile of 16]
  {
    f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const f32x4 a0 = *(const f32x4*)(H0s + l15 * H0_LD + 16 * ks + 4 * g);
      const f32x4 a1 = *(const f32x4*)(H0s + (16 + l15) * H0_LD + 16 * ks + 4 * g);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc0 = MFMA16(a0[t], bw[ks][t], acc0);
        acc1 = MFMA16(a1[t], bw[ks][t], acc1);
      }
    }
    const float bias = b1[n1];
    const int cl = wave * 16 + l15;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      H1s[(4 * g + reg) * T64_LD + cl] = fmaxf(acc0[reg] + bias, 0.f);
      H1s[(16 + 4 * g + reg) * T64_LD + cl] = fmaxf(acc1[reg] + bias, 0.f);
    }
  }
  __syncthreads();
  STAMP(p, 3);

  {
    const int rl = tid >> 3;
    const int row = row0 + rl;
    const int sub = tid & 7;
    if (slot >= 0 && row < p.b.rows) {
      float* dst = p.sc.h1 + (long long)slot * p.sc.max_batch * HID;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int cl = 4 * (sub + 8 * j);
        *(f32x4*)(dst + (long long)row * HID + ns * 64 + cl) = *(const f32x4*)(H1s + rl * T64_LD + cl);
      }
    }
    // ---- head partial sums over this block's 64 hidden-1 units
    const int MB = p.sc.max_batch;
    for (int dd = 0; dd < D; ++dd) {
      const float* w2r = w2 + (long long)dd * HID + ns * 64;
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = sub + 8 * j;
        acc = fmaf(H1s[rl * T64_LD + c], w2r[c], acc);
      }
      acc += __shfl_xor(acc, 1);
      acc += __shfl_xor(acc, 2);
      acc += __shfl_xor(acc, 4);
      if (sub == 0 && row < p.b.rows) {
        if (inst < 6) p.sc.heads[((long long)inst * NSPLIT + ns) * MB + row] = acc;
        else p.sc.heads[(long long)6 * NSPLIT * MB + ((long long)ns * MB + row) * p.A + dd] = acc;
      }
    }
  }
  STAMP(p, 4);
}

// ---------------------------------------------------------------------------
// Head values (sum of the 4 column-slice partials in fixed order + bias).
__device__ __forceinline__ float head_scalar(const StepParams& p, int inst, int row, float bias) {
  const int MB = p.sc.max_batch;
  const float* h = p.sc.heads + (long long)inst * NSPLIT * MB + row;
  return ((h[0] + h[MB]) + h[2 * MB]) + h[3 * (long long)MB] + bias;
}
__device__ __forceinline__ float head_pi(const StepParams& p, int row, int dd, float bias) {
  const int MB = p.sc.max_batch;
  const float* h = p.sc.heads + (long long)6 * NSPLIT * MB + (long long)row * p.A + dd;
  const long long st = (long long)MB * p.A;
  return ((h[0] + h[st]) + h[2 * st]) + h[3 * st] + bias;
}

// Per-row loss terms and dL/d(head pre-activation) for `net`.
//   dy[0..D) written to dyrow (stride 1); returns the row's loss term(s):
//   V: w*u^2      Q1: e1^2 (lossB = e2^2)     pi: w*bc
// For the Gaussian policy dls[dd] receives w*(1 - diff^2/var) (d/dlog_std terms).
__device__ __forceinline__ void row_loss_grad(const StepParams& p, int net, int row, float* dyrow, float* dlsrow,
                                              float& lossA, float& lossB) {
  const float invB = p.inv_batch;
  lossA = 0.f;
  lossB = 0.f;
  const float bV = p.params[p.L.net[IQLHIP_NET_V].b2];
  if (net == IQLHIP_NET_V || net == IQLHIP_NET_PI) {
    const float bt1 = (p.target - p.L.target_src)[p.L.net[IQLHIP_NET_Q1].b2];
    const float bt2 = (p.target - p.L.target_src)[p.L.net[IQLHIP_NET_Q2].b2];
    const float tq = fminf(head_scalar(p, 2, row, bt1), head_scalar(p, 3, row, bt2));
    const float v = head_scalar(p, 1, row, bV);
    const float u = tq - v;
    if (net == IQLHIP_NET_V) {
      const float wgt = fabsf(p.hy.iql_tau - ((u < 0.f) ? 1.f : 0.f));
      lossA = wgt * u * u;
      dyrow[0] = (-2.f * wgt * u) * invB;
      return;
    }
    // policy
    const float w = fminf(expf(p.hy.beta * u), p.hy.exp_adv_max);
    const iqlhip_net_layout& nl = p.L.net[IQLHIP_NET_PI];
    const int A = p.A;
    const DevBatch& b = p.b;
    const float* arow = b.a + src_row(b, row) * b.ld_a;
    float bc = 0.f;
    for (int dd = 0; dd < A; ++dd) {
      const float mu = tanhf(head_pi(p, row, dd, p.params[nl.b2 + dd]));
      const float diff = arow[dd] - mu;
      float dmu;
      if (p.policy == IQLHIP_POLICY_GAUSSIAN) {
        const float lsr = p.params[nl.log_std + dd];
        const float ls = fminf(fmaxf(lsr, p.hy.log_std_min), p.hy.log_std_max);
        const float sig = expf(ls);
        const float var = sig * sig;
        bc += diff * diff / (2.f * var) + ls + 0.918938533204672742f;  // log(sqrt(2 pi))
        dmu = (-(w * diff) / var) * invB;
        if (dlsrow) dlsrow[dd] = w * (1.f - diff * diff / var);
      } else {
        bc += diff * diff;
        dmu = (-2.f * w * diff) * invB;
      }
      dyrow[dd] = dmu * (1.f - mu * mu);
    }
    lossA = w * bc;
    return;
  }
  // Q nets
  const DevBatch& b = p.b;
  const long long j = src_row(b, row);
  const float r = b.r[j * b.ld_r];
  const float d = b.d[j * b.ld_d];
  const float nv = head_scalar(p, 0, row, bV);
  const float y = r + ((1.f - d) * p.hy.discount) * nv;
  const float e1 = head_scalar(p, 4, row, p.params[p.L.net[IQLHIP_NET_Q1].b2]) - y;
  const float e2 = head_scalar(p, 5, row, p.params[p.L.net[IQLHIP_NET_Q2].b2]) - y;
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
__global__ __launch_bounds__(256) void iql_bwd_kernel(StepParams p, int n_chunk, int n_rt) {
  const int bid = blockIdx.x;
  const int x = bid & 7;
  const int net = x & 3;
  const int local = (bid >> 3) * 2 + (x >> 2);
  const int n_a = 32 * n_chunk;
  if (local >= n_a + 4 * n_rt) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int B = p.b.rows;
  const int MB = p.sc.max_batch;

  const iqlhip_net_layout& nl = p.L.net[net];
  const int D = nl.d_out;
  const int Dp = (D + 15) & ~15;      // 16 or 32
  const int DYLD = Dp + 1;
  const float* w2 = p.params + nl.w2;
  const float* H1g = p.sc.h1 + (long long)net * MB * HID;
  const float* H0g = p.sc.h0 + (long long)net * MB * HID;

  extern __shared__ __attribute__((aligned(16))) float smem[];
#ifdef IQL_STAMPS
  if (p.stamps) p.stamps += 2048 * 16;   // second half of the stamp buffer: the forward kernel owns the first
#endif
  STAMP(p, 0);

  if (local < n_a) {
    // ===================== (a): dW1[j-tile][i-tile] over one 256-row chunk =====================
    const int c = local >> 5;
    const int jt = (local >> 2) & 7;
    const int it = local & 3;
    const int j0 = jt * 32, i0 = it * 64;
    const int cbase = c * CHUNK_ROWS;
    float* red = smem;                               // [4][32][T64_LD]
    float* dYs = red + 4 * 32 * T64_LD;              // [256][DYLD]
    float* dLs = dYs + CHUNK_ROWS * DYLD;            // [256][DYLD]  (gaussian pi designated block only)
    float* W2s = dLs + CHUNK_ROWS * DYLD;            // [D][32]
    float* rsm = W2s + 32 * 32;                      // [64] small reductions
    const bool designated = (jt == 0 && it == 0);
    float* slab = p.sc.slab_a + (long long)c * p.L.n_params;

    // ---- prologue: dY for the 256 rows of the chunk (thread = row)
    {
      const int row = cbase + tid;
      float lossA = 0.f, lossB = 0.f;
      float* dyrow = dYs + tid * DYLD;
      for (int dd = 0; dd < Dp; ++dd) dyrow[dd] = 0.f;
      float* dlsrow = (designated && net == IQLHIP_NET_PI && p.policy == IQLHIP_POLICY_GAUSSIAN) ? (dLs + tid * DYLD) : nullptr;
      if (dlsrow) for (int dd = 0; dd < Dp; ++dd) dlsrow[dd] = 0.f;
      if (row < B) row_loss_grad(p, net, row, dyrow, dlsrow, lossA, lossB);
      for (int e = tid; e < D * 32; e += 256) {
        const int dd = e >> 5, jj = e & 31;
        W2s[dd * 32 + jj] = w2[(long long)dd * HID + j0 + jj];
      }
      if (designated) {
        // loss partial sums of this chunk
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
    if (designated) {
      // db2[dd] = sum_r dY[r][dd];  dlog_std[dd] = sum_r w (1 - diff^2/var) * inv_batch (inside clamp range only)
      for (int dd = wave; dd < D; dd += 4) {
        float s = 0.f, sl = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          s += dYs[(lane + 64 * q) * DYLD + dd];
          if (net == IQLHIP_NET_PI && p.policy == IQLHIP_POLICY_GAUSSIAN) sl += dLs[(lane + 64 * q) * DYLD + dd];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); sl += __shfl_xor(sl, o); }
        if (lane == 0) {
          slab[nl.b2 + dd] = s;
          if (net == IQLHIP_NET_PI && p.policy == IQLHIP_POLICY_GAUSSIAN) {
            const float lsr = p.params[nl.log_std + dd];
            const bool inside = (lsr >= p.hy.log_std_min) && (lsr <= p.hy.log_std_max);
            slab[nl.log_std + dd] = inside ? sl * p.inv_batch : 0.f;
          }
        }
      }
    }

    STAMP(p, 2);
    // ---- main loop: this wave reduces rows [cbase + 64*wave, +64)
    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 acc2[2][2];   // dW2 tiles [dt][tb] (MFMA path, D > 1)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc2[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float db1a[2] = {0.f, 0.f};
    float dw2a[2] = {0.f, 0.f};    // D == 1 VALU path
    const bool extras = (it == 0);
    const int ndt = Dp >> 4;

    f32x2 hh[16];
    f32x4 bb[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      // rows >= B are clamped to a valid row: their dY is 0, so they contribute nothing
      // (no per-load branch: a select around each load would serialise the prefetch)
      const int rl = 64 * wave + 4 * ks + g;
      const int row = min(cbase + rl, B - 1);
      hh[ks] = *(const f32x2*)(H1g + (long long)row * HID + j0 + 2 * l15);
      bb[ks] = *(const f32x4*)(H0g + (long long)row * HID + i0 + 4 * l15);
    }
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const int rl = 64 * wave + 4 * ks + g;
      float av[2];
      if (D == 1) {
        const float dy = dYs[rl * DYLD];
#pragma unroll
        for (int ta = 0; ta < 2; ++ta) av[ta] = (hh[ks][ta] > 0.f) ? dy * W2s[2 * l15 + ta] : 0.f;
        if (extras) {
          dw2a[0] = fmaf(dy, hh[ks][0], dw2a[0]);
          dw2a[1] = fmaf(dy, hh[ks][1], dw2a[1]);
        }
      } else {
        float s0 = 0.f, s1 = 0.f;
        for (int dd = 0; dd < D; ++dd) {
          const float dy = dYs[rl * DYLD + dd];
          s0 = fmaf(dy, W2s[dd * 32 + 2 * l15], s0);
          s1 = fmaf(dy, W2s[dd * 32 + 2 * l15 + 1], s1);
        }
        av[0] = (hh[ks][0] > 0.f) ? s0 : 0.f;
        av[1] = (hh[ks][1] > 0.f) ? s1 : 0.f;
        if (extras) {
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            if (dt < ndt) {
              const float ad = dYs[rl * DYLD + 16 * dt + l15];
              acc2[dt][0] = MFMA16(ad, hh[ks][0], acc2[dt][0]);
              acc2[dt][1] = MFMA16(ad, hh[ks][1], acc2[dt][1]);
            }
          }
        }
      }
      if (extras) { db1a[0] += av[0]; db1a[1] += av[1]; }
#pragma unroll
      for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) acc[ta][tb] = MFMA16(av[ta], bb[ks][tb], acc[ta][tb]);
    }

    STAMP(p, 3);
    // ---- cross-wave reduction of the 32x64 tile through LDS, then coalesced store
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
    __syncthreads();
    {
      float* gw1 = slab + nl.w1;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int f = tid + 256 * q;
        const int jl = f >> 4, i4 = f & 15;
        f32x4 s = *(const f32x4*)(red + jl * T64_LD + 4 * i4);
#pragma unroll
        for (int w = 1; w < 4; ++w) s += *(const f32x4*)(red + w * 32 * T64_LD + jl * T64_LD + 4 * i4);
        *(f32x4*)(gw1 + (long long)(j0 + jl) * HID + i0 + 4 * i4) = s;
      }
    }
    if (extras) {
      __syncthreads();
      // db1 and dW2: reduce over g (lanes with equal l15), then over waves via LDS
      float* ex = red;   // reuse: [4 waves][ (1 + Dp) rows ][32 cols]
#pragma unroll
      for (int ta = 0; ta < 2; ++ta) {
        float v = db1a[ta];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (g == 0) ex[(wave * 33 + 0) * 32 + 2 * l15 + ta] = v;
        if (D == 1) {
          float u = dw2a[ta];
          u += __shfl_xor(u, 16);
          u += __shfl_xor(u, 32);
          if (g == 0) ex[(wave * 33 + 1) * 32 + 2 * l15 + ta] = u;
        }
      }
      if (D > 1) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
          if (dt < ndt)
#pragma unroll
            for (int tb = 0; tb < 2; ++tb)
#pragma unroll
              for (int reg = 0; reg < 4; ++reg)
                ex[(wave * 33 + 1 + 16 * dt + 4 * g + reg) * 32 + 2 * l15 + tb] = acc2[dt][tb][reg];
      }
      __syncthreads();
      for (int e = tid; e < (1 + D) * 32; e += 256) {
        const int rr = e >> 5, jj = e & 31;
        const float s = (ex[(0 * 33 + rr) * 32 + jj] + ex[(1 * 33 + rr) * 32 + jj]) +
                        (ex[(2 * 33 + rr) * 32 + jj] + ex[(3 * 33 + rr) * 32 + jj]);
        if (rr == 0) slab[nl.b1 + j0 + jj] = s;
        else slab[nl.w2 + (long long)(rr - 1) * HID + j0 + jj] = s;
      }
    }
    STAMP(p, 4);
    return;
  }

  // ===================== (b): dH0 / dW0 / db0 for one 32-row tile and 64-column slice =====================
  {
    const int lb = local - n_a;
    const int rt = lb >> 2;
    const int is = lb & 3;
    const int i0 = is * 64;
    const int row0 = rt * RT_ROWS;
    const int k0 = nl.k_in;
    const int k0p = (k0 + 3) & ~3;
    const int xld = xld_for(k0p);
    const float* w1 = p.params + nl.w1;

    float* dH1s = smem;                              // [32][H0_LD]
    float* red = dH1s + RT_ROWS * H0_LD;             // [4][32][T64_LD]
    float* dH0s = red + 4 * 32 * T64_LD;             // [32][T64_LD]
    float* dYs = dH0s + RT_ROWS * T64_LD;            // [32][DYLD]
    float* Xs = dYs + RT_ROWS * 33;                  // [32][xld]

    // prefetch this wave's W1 fragments: k = j in [64*wave, +64), n = i0 + 4*l15 + t
    f32x4 bw[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
      bw[ks] = *(const f32x4*)(w1 + (long long)(64 * wave + 4 * ks + g) * HID + i0 + 4 * l15);

    const int kind = (net == IQLHIP_NET_Q1 || net == IQLHIP_NET_Q2) ? 2 : 0;
    gather_rows(p, kind, row0, k0, xld, Xs);
    if (tid < RT_ROWS) {
      const int row = row0 + tid;
      float la, lbv;
      float* dyrow = dYs + tid * DYLD;
      for (int dd = 0; dd < Dp; ++dd) dyrow[dd] = 0.f;
      if (row < B) row_loss_grad(p, net, row, dyrow, nullptr, la, lbv);
    }
    __syncthreads();
    STAMP(p, 5);

    // dH1s[r][j] = (sum_dd dY[r][dd] W2[dd][j]) * (H1[r][j] > 0)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int f = tid + 256 * q;      // float4 index in [32][64]
      const int rl = f >> 6, j4 = f & 63;
      const int row = min(row0 + rl, B - 1);   // rows >= B: dY = 0 -> dH1 = 0
      f32x4 out;
      const f32x4 h = *(const f32x4*)(H1g + (long long)row * HID + 4 * j4);
      f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int dd = 0; dd < D; ++dd) {
        const float dy = dYs[rl * DYLD + dd];
        const f32x4 wv = *(const f32x4*)(w2 + (long long)dd * HID + 4 * j4);
        s += dy * wv;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) out[e] = (h[e] > 0.f) ? s[e] : 0.f;
      *(f32x4*)(dH1s + rl * H0_LD + 4 * j4) = out;
    }
    __syncthreads();
    STAMP(p, 6);

    // dH0 partial over this wave's 64 j's: [32 rows][64 cols]
    {
      f32x4 acc[2][4];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
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
    __syncthreads();
    STAMP(p, 7);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int f = tid + 256 * q;
      const int rl = f >> 4, i4 = f & 15;
      const int row = min(row0 + rl, B - 1);   // rows >= B carry s = 0
      f32x4 s = *(const f32x4*)(red + rl * T64_LD + 4 * i4);
#pragma unroll
      for (int w = 1; w < 4; ++w) s += *(const f32x4*)(red + w * 32 * T64_LD + rl * T64_LD + 4 * i4);
      f32x4 out;
      const f32x4 h = *(const f32x4*)(H0g + (long long)row * HID + i0 + 4 * i4);
#pragma unroll
      for (int e = 0; e < 4; ++e) out[e] = (h[e] > 0.f) ? s[e] : 0.f;
      *(f32x4*)(dH0s + rl * T64_LD + 4 * i4) = out;
    }
    __syncthreads();

    STAMP(p, 8);
    float* slabB = p.sc.slab_b + p.sc.slab_b_off[net] + (long long)rt * (HID * k0 + HID);
    // dW0[i][kc] partial = sum_r dH0[r][i] X[r][kc];  this wave: i in [i0 + 16*wave, +16)
    {
      const int nct = (k0 + 15) >> 4;
      f32x4 acc[8];
#pragma unroll
      for (int ct = 0; ct < 8; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const int rl = 4 * ks + g;
        const float a = dH0s[rl * T64_LD + 16 * wave + l15];
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) {
          if (ct < nct) {
            const int kc = 16 * ct + l15;
            const float bv = (kc < xld) ? Xs[rl * xld + kc] : 0.f;
            acc[ct] = MFMA16(a, bv, acc[ct]);
          }
        }
      }
#pragma unroll
      for (int ct = 0; ct < 8; ++ct) {
        if (ct < nct) {
          const int kc = 16 * ct + l15;
          if (kc < k0) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
              const int i = i0 + 16 * wave + 4 * g + reg;
              slabB[(long long)i * k0 + kc] = acc[ct][reg];
            }
          }
        }
      }
    }
    if (tid < 64) {
      float s = 0.f;
#pragma unroll
      for (int rl = 0; rl < RT_ROWS; ++rl) s += dH0s[rl * T64_LD + tid];
      slabB[(long long)HID * k0 + i0 + tid] = s;
    }
    STAMP(p, 9);
  }
}

// ---------------------------------------------------------------------------
// Gradient assembly: element e (float4 granularity) of the flat arena.
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
  float* loss_parts;
  float* losses;            // [4]
  float* loss_ring;         // nullable
  int ring_slot;
  int n_chunk, n_rt;
  int batch_rows;
  const iqlhip_step_scalars* sched;  // when non-null the scalars of this launch are sched[sched_idx]
  int sched_idx;                     // (hipGraph replay: kernel arguments are frozen, the table is not)
};

__device__ __forceinline__ int net_of(const iqlhip_layout& L, long long e) {
  int n = 0;
#pragma unroll
  for (int i = 1; i < 4; ++i) if (e >= L.net[i].seg_begin) n = i;
  return n;
}

__device__ __forceinline__ f32x4 slab_grad(const UpdParams& u, long long e, int net) {
  const iqlhip_net_layout& nl = u.L.net[net];
  f32x4 gsum = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (e >= nl.w0 && e < nl.b0 + HID) {
    const long long len = (long long)HID * nl.k_in + HID;
    const float* base = u.slab_b + u.slab_b_off[net] + (e - nl.w0);
    for (int rt = 0; rt < u.n_rt; ++rt) gsum += *(const f32x4*)(base + rt * len);
  } else {
    for (int c = 0; c < u.n_chunk; ++c) gsum += *(const f32x4*)(u.slab_a + (long long)c * u.L.n_params + e);
  }
  return gsum;
}

__device__ __forceinline__ void loss_words(const UpdParams& u, float out[4]) {
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < u.n_chunk; ++c)
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] += u.loss_parts[k * 64 + c];
  out[0] = s[0];   // sum_r w u^2
  out[1] = s[1];   // sum_r e1^2
  out[2] = s[2];   // sum_r e2^2
  out[3] = s[3];   // sum_r w bc
}

// Writes the summed flat gradient (+ tail: value, q, actor loss contributions, spare) for the DP all-reduce.
__global__ __launch_bounds__(256) void iql_grad_flatten_kernel(UpdParams u, float* out) {
  const long long e = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (e < u.L.n_params) {
    const int net = net_of(u.L, e);
    *(f32x4*)(out + e) = slab_grad(u, e, net);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    float s[4];
    loss_words(u, s);
    const float ib = u.sc.inv_batch;
    out[u.L.n_params + 0] = s[0] * ib;
    out[u.L.n_params + 1] = (s[1] * ib + s[2] * ib) * 0.5f;
    out[u.L.n_params + 2] = s[3] * ib;
    out[u.L.n_params + 3] = 0.f;
  }
}

__global__ __launch_bounds__(256) void iql_update_kernel(UpdParams u) {
  const long long e = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (e < u.L.n_params) {
    const int net = net_of(u.L, e);
    f32x4 gr;
    if (u.flat_grads) gr = *(const f32x4*)(u.flat_grads + e);
    else gr = slab_grad(u, e, net);
    const int grp = (net == IQLHIP_NET_V) ? 0 : ((net == IQLHIP_NET_PI) ? 2 : 1);
    const iqlhip_step_scalars* scp = u.sched ? (u.sched + u.sched_idx) : &u.sc;
    const float gs = scp->grad_scale;
    const float step = -scp->step_size[grp];
    const float bc2 = scp->bc2_sqrt[grp];
    const float omb1 = scp->one_minus_beta1, b2 = scp->beta2, omb2 = scp->one_minus_beta2, eps = scp->eps;
    f32x4 m = *(f32x4*)(u.m + e);
    f32x4 v = *(f32x4*)(u.v + e);
    f32x4 pw = *(f32x4*)(u.params + e);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gk = (gs == 1.f) ? gr[k] : gr[k] * gs;
      m[k] = m[k] + omb1 * (gk - m[k]);
      v[k] = v[k] * b2 + (omb2 * gk) * gk;
      const float denom = sqrtf(v[k]) / bc2 + eps;
      pw[k] = pw[k] + step * (m[k] / denom);
    }
    *(f32x4*)(u.m + e) = m;
    *(f32x4*)(u.v + e) = v;
    *(f32x4*)(u.params + e) = pw;
    if (net == IQLHIP_NET_Q1 || net == IQLHIP_NET_Q2) {
      float* tp = u.target + (e - u.L.target_src);
      f32x4 t = *(f32x4*)tp;
#pragma unroll
      for (int k = 0; k < 4; ++k) t[k] = u.one_minus_tau * t[k] + u.tau * pw[k];
      *(f32x4*)tp = t;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    float l[3];
    if (u.flat_grads) {
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
    if (u.loss_ring) {
      float* rr = u.loss_ring + 4 * (long long)u.ring_slot;
      rr[0] = l[0]; rr[1] = l[1]; rr[2] = l[2]; rr[3] = 0.f;
    }
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

__global__ void iql_rows_gather_kernel(const float* rows, long long ld, int S, int A, const long long* idx,
                                       long long n, float* s, float* a, float* r, float* ns, float* d) {
  const int W = 2 * S + A + 2;
  const long long total = n * W;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (long long)gridDim.x * blockDim.x) {
    const long long i = e / W;
    const int c = (int)(e - i * W);
    const float v = rows[idx[i] * ld + c];
    if (c < S) s[i * S + c] = v;
    else if (c < S + A) a[i * A + (c - S)] = v;
    else if (c < 2 * S + A) ns[i * S + (c - S - A)] = v;
    else if (c == 2 * S + A) r[i] = v;
    else d[i] = v;
  }
}

// Philox4x32-10 (Salmon et al. 2011), counter = (offset + i/2), key = seed.
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

// idx[i] uniform over [0,size): 64 random bits, multiply-high (bias <= size / 2^64).
// `hdr` (nullable) = device words {size, seed, offset} so that a captured graph can be replayed
// with new values.
__global__ void iql_draw_indices_kernel(long long* idx, long long n, long long size, unsigned long long seed,
                                        unsigned long long offset, const unsigned long long* hdr) {
  if (hdr) { size = (long long)hdr[0]; seed = hdr[1]; offset = hdr[2]; }
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
