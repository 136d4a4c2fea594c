// iqlhip_lb_kernels.h — the LARGE-BATCH bf16 step (BASELINE configs[4]: obs 39 / act 28, 8 192 rows over 8 GPUs).
//
// The kernels of iqlhip_kernels.h are shaped for 256 rows: one launch chain per step, every block a single pass over
// a 32-row tile, weights streamed per block.  Above ~512 rows that shape re-reads W1 once per 32 rows and column
// slice, recomputes the policy's loss gradient in each of its 128 dW1 tiles and writes one gradient slab per 256-row
// chunk / 32-row tile (profiles/r03_stamps_config5_1024_bf16.txt).  Here, for the bf16 path and
// batches of more than 512 rows:
//
//   iql_fwd_lb_kernel   block = (instance, a STRIDED SET of 32-row tiles).  The wave keeps its 64 output units' W0 and W1
//                       rows as bf16 MFMA operands IN REGISTERS for the whole block (176 VGPRs) and walks its row tiles:
//                       X tile -> bf16 LDS image -> layer 0 (24 MFMAs) -> H0 tile (bf16 LDS) -> layer 1 (64 MFMAs, one
//                       pass over all 256 units, no column slices) -> heads.  The policy instance finishes its heads in
//                       the block (all four waves' partial sums meet in LDS) and writes, ONCE per (row, dim), the
//                       weight-free part of the loss gradient:   T = d(-log pi)/d(pre-tanh) / B,  G = 1 - q,  L = sum_d
//                       of the log-prob terms — the backward only multiplies them by the row's advantage weight w.
//   iql_bwd_lb_kernel   (a) blocks: a 32 x 64 tile of dW1 accumulated IN REGISTERS over a group of 256-row chunks (one
//                       slab per chunk group instead of one per chunk); dY = w * T is loaded, not recomputed; the policy's
//                       dH1 = dY . W2 and dW2 = dY^T . H1 run on the bf16 MFMA.
//                       (b) blocks: a strided set of 32-row tiles with the wave's 64 columns of W1 (all 256 k) in
//                       registers for the whole block; [dW0 | db0] is accumulated in registers over the block's row
//                       tiles (one slab per block instead of one per row tile).
//   iql_update_kernel   unchanged: it sums however many slabs the launch wrote.
//
// Lane maps (wave64, l15 = lane & 15, g = lane >> 4), v_mfma_f32_16x16x32_bf16:
//   A[m = l15][k = 8 g + e]   B[k = 8 g + e][n = l15]   D[m = 4 g + reg][n = l15]      (e = 0..7, reg = 0..3)
// Reference arithmetic replaced: algorithms/finetune/iql.py:314-344 (MLP), :482-540 (_update_v/_q/_policy).
#pragma once
#include "iqlhip_kernels.h"

#ifndef LB_SKIP
#define LB_SKIP 0           // timing experiments only (wrong results): bit 0 (b) no W1 loads, 1 (a) no H1 / H0 loads, 2 (b) no slab
#endif                      // store, 3 (a) no T loads, 4 (a) no head loads, 5 (b) no tile loads
#define LB_DYLD 40          // bf16 row stride of a [rows][32 dims] / [cols][32 rows] tile: 80 bytes, 16-byte aligned rows
#define LB_PLD 33           // fp32 row stride of the policy head partials [wave][32 rows][32 dims]
#define LB_HDLD 28          // fp32 row stride of a chunk's scalar head partials in LDS (24 + 4: 16-byte rows, banks spread)

struct LbArgs {
  float* pi_t;      // [max_batch][32]  T: dL_pi/d(pre-tanh) / w   (zero beyond action_dim)
  float* pi_g;      // [max_batch][32]  G: 1 - (a - mu)^2 / sigma^2 (gaussian; dlog_std terms / w)
  float* pi_l;      // [max_batch]      L: sum_d of the per-dim loss terms (actor loss of the row / w)
  int n_rt;         // 32-row tiles of the batch
  int n_chunk;      // 256-row chunks of the batch
  int nbi;          // forward: blocks per instance (even): block ib walks row tiles ib, ib + nbi, ...
  int nbb;          // backward: (b) blocks per net (even): block lb walks row tiles lb, lb + nbb, ...
  int cpb;          // backward: chunks per (a) block
  int n_cg;         // backward: chunk groups = ceil(n_chunk / cpb) = chunk slabs written
};

__device__ __forceinline__ bf16x4 cvt4(const f32x4 v) {
  bf16x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = (__bf16)v[i];
  return r;
}

// ---------------------------------------------------------------------------
// Forward.  grid = 8 x nbi; XCD map and instance pairs as iql_fwd_kernel's (XCDs n and n + 4 host the two instances that
// read net n's weights; block parity = row-tile parity, which the backward's (b) blocks of that parity read back).
// NKB: 32-wide k-blocks of the widest layer-0 input (S + A <= 128 -> <= 4).
template <int NKB>
__global__ __launch_bounds__(256) void iql_fwd_lb_kernel(StepParams p, LbArgs a) {
  constexpr int XLD = 32 * NKB + 8;       // bf16 row stride of the X tile
  __shared__ __attribute__((aligned(16))) __bf16 Xb[32 * XLD];
  __shared__ __attribute__((aligned(16))) __bf16 H0b[32 * H0B_LD];
  __shared__ __attribute__((aligned(16))) __bf16 H1b[32 * H0B_LD];
  __shared__ __attribute__((aligned(16))) float Pp[4 * 32 * LB_PLD];

  RT_ENTRY();
  const int bid = blockIdx.x;
  const int fx = bid & 7, fh = fx >> 2, fr = bid >> 3;
  constexpr unsigned FWD_PAIR_A = 0x6541u, FWD_PAIR_B = 0x7320u;      // (iql_fwd_kernel)
  const int inst = (int)((((fr & 1) ? FWD_PAIR_B : FWD_PAIR_A) >> (4 * (fx & 3))) & 7u);
  const int nbi = a.nbi;
  const int ib = (fr >> 1) * 2 + fh;
  if (inst >= 7) {
    if (p.g_work) idle_block_work(p.g_work, ib, nbi);
    return;
  }
  const int n_rt = a.n_rt;
  if (ib >= n_rt) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  // Which 64 units a wave owns, in which order it walks their four 16-unit tiles and W1's eight k-blocks, ROTATES with the
  // block: the blocks of an instance start together and would otherwise all miss on the same weight lines at the same
  // moment (every CU then sees the fabric's latency on every line; staggered, a line one block has fetched is an L2 hit
  // for the others).  wv = the wave's unit slab, CP(ct) = the unit tile held in register slot ct, KP(kb) likewise.
#ifdef LB_ROT
  const int wv = (wave + (ib >> 1)) & 3;
  const int crot = (ib >> 3) & 3, krot = (ib >> 1) & 7;
#else
  const int wv = wave;
  const int crot = 0, krot = 0;
#endif
#define CP(ct) (((ct) + crot) & 3)
#define KP(kb) (((kb) + krot) & 7)

  const NetPtrs np = p.inst[inst];
  const int xoff = p.xoff[inst], slot = p.slot[inst];
  const int k0 = np.k0, D = np.d;
  const int nkb = (k0 + 31) >> 5;
  const int ld = p.ld, B = p.rows, MB = p.sc.max_batch, S = p.S, A = p.A;
  const float* xb = p.xb;
  const bool is_pi = (inst == 6);
  const bool gauss = (p.policy == IQLHIP_POLICY_GAUSSIAN);
  const bool drop = is_pi && (p.drop_bits != nullptr);
  const unsigned xtotal = (unsigned)B * (unsigned)ld;

  // ---- the 32 packed rows of a tile: thread (row tid >> 3, float4 (tid & 7) + 8 q of the instance's input columns)
  const int xr = tid >> 3, xc = tid & 7;
  f32x4 xv[NKB];
  auto x_issue = [&](int rt) {
    const unsigned row = (unsigned)min(rt * RT_ROWS + xr, B - 1);
#pragma unroll
    for (int q = 0; q < NKB; ++q) {
      xv[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
      // (columns >= k0 are zeroed in x_store; the address is clamped into the row, and — the last floats of the batch —
      //  to the batch's last 16 bytes: the wanted elements then sit `sh` places up in the loaded vector.  No branch: a
      //  branch around a load makes the compiler wait for every load in flight.)
      const int c = min(4 * (xc + 8 * q), (k0 - 1) & ~3);
      const unsigned want = row * (unsigned)ld + (unsigned)(xoff + c);
      const unsigned idx = min(want, xtotal - 4u);
      const unsigned sh = want - idx;
      const f32x4 v = *(const f32x4u*)(xb + idx);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float a1 = (j + 1 < 4) ? v[(j + 1) & 3] : 0.f, a2 = (j + 2 < 4) ? v[(j + 2) & 3] : 0.f, a3 = (j + 3 < 4) ? v[(j + 3) & 3] : 0.f;
        xv[q][j] = (sh == 0u) ? v[j] : ((sh == 1u) ? a1 : ((sh == 2u) ? a2 : a3));
      }
    }
  };
  auto x_store = [&]() {
#pragma unroll
    for (int q = 0; q < NKB; ++q) {
      if (q < nkb) {
        const int c = 4 * (xc + 8 * q);
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (__bf16)((c + j < k0) ? xv[q][j] : 0.f);     // columns >= k0: other fields of the row
        *(bf16x4*)(Xb + xr * XLD + c) = o;
      }
    }
  };
  int rt = ib;
  STAMP_BASE(p, 0);
  STAMP(p, 0);
  x_issue(rt);

  // ---- the wave's operands for the whole block.  Layer 0: A = W0 rows of units 64 w + 16 ct + l15 (fp32 master, k = 32 kb
  // + 8 g + e; the last k-block reads past k0 into the next row / the bias that follows — finite values against zero X)
  // (k-blocks beyond the instance's own are loaded from its last one — never used: no branch around a load)
  f32x4 w0lo[4][NKB], w0hi[4][NKB];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#ifdef LB_HACK_FRAG      // timing only (wrong values): the loads of a fragment-major image
      const float* src = np.w0 + (unsigned)(((((wv * 4 + ct) * nkb + min(kb, nkb - 1)) * 64 + lane) * 8) % (256 * k0 - 8));
#else
      const float* src = np.w0 + (unsigned)((64 * wv + 16 * CP(ct) + l15) * k0 + 32 * min(kb, nkb - 1) + 8 * g);
#endif
      w0lo[ct][kb] = *(const f32x4u*)src;
      w0hi[ct][kb] = *(const f32x4u*)(src + 4);
    }
  f32x4 bias0[4], bias1[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    bias0[ct] = *(const f32x4*)(np.b0 + (unsigned)(64 * wv + 16 * CP(ct) + 4 * g));
    bias1[ct] = *(const f32x4*)(np.b1 + (unsigned)(64 * wv + 16 * CP(ct) + 4 * g));
  }
  // heads.  Scalar nets: the wave's 64 head weights in accumulator layout.  Policy: B operands of the head product,
  // k = unit 64 w + 16 (2 kb2 + (e >> 2)) + 4 g + (e & 3) — the units the lane's layer-1 accumulators hold — n = dim
  // (every instance issues the same loads — the scalar nets read their one W2 row where the policy reads its dims')
  f32x4 w2v[4];
  f32x4 w2lo[2][2], w2hi[2][2];
  const int ndt = is_pi ? ((D + 15) >> 4) : 0;
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) w2v[ct] = *(const f32x4*)(np.w2 + (unsigned)(64 * wv + 16 * CP(ct) + 4 * g));
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2) {
      const unsigned o = (unsigned)(min(16 * nt + l15, D - 1) * HID + 64 * wv + 4 * g);
      w2lo[nt][kb2] = *(const f32x4*)(np.w2 + o + (unsigned)(16 * CP(2 * kb2)));
      w2hi[nt][kb2] = *(const f32x4*)(np.w2 + o + (unsigned)(16 * CP(2 * kb2 + 1)));
    }
  const float b2s = np.b2[0];
  // policy: the thread's (row tid >> 3, dims (tid & 7) + 8 c) constants
  float b2v[4], lsr[4];
  const float* lsp = (is_pi && gauss) ? p.log_std : np.b2;      // (any valid address when unused)
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int d = min(xc + 8 * c, D - 1);
    b2v[c] = np.b2[d];
    lsr[c] = lsp[d];
  }
  // (requested last: the biggest and the last needed)
  // layer 1: A = W1 rows (bf16 shadow) of the same 64 units, all 256 k: 32 fragments of 8 contiguous k
  bf16x8 w1f[4][8];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int kb = 0; kb < 8; ++kb)
#ifdef LB_HACK_FRAG
      w1f[ct][kb] = *(const bf16x8*)((const __bf16*)np.w1 + (unsigned)((((wv * 4 + ct) * 8 + kb) * 64 + lane) * 8));
#else
      w1f[ct][kb] = *(const bf16x8*)((const __bf16*)np.w1 + (unsigned)((64 * wv + 16 * CP(ct) + l15) * HID + 32 * KP(kb) + 8 * g));
#endif
  bf16x8 w0f[4][NKB];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) w0f[ct][kb] = pack8(w0lo[ct][kb], w0hi[ct][kb]);
  bf16x8 w2f[2][2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2) w2f[nt][kb2] = pack8(w2lo[nt][kb2], w2hi[nt][kb2]);
  float ivar[4], lsc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    lsc[c] = (is_pi && gauss) ? fminf(fmaxf(lsr[c], p.hy.log_std_min), p.hy.log_std_max) : 0.f;
    const float sig = expf(lsc[c]);
    ivar[c] = 1.f / (sig * sig);
  }

  __bf16* h0g = (__bf16*)p.sc.h0;
  __bf16* h1g = (__bf16*)p.sc.h1;
  float* headsg = p.sc.heads;
  const float invB = p.inv_batch;

  STAMP(p, 1);
  for (; rt < n_rt; rt += nbi) {
    const int row0 = rt * RT_ROWS;
    const bool first = (rt == ib);
    // policy: this tile's actions and dropout keep-bits (in flight under layer 0)
    // (issued by every instance, from a harmless address where unused: a branch around a load would make the compiler
    //  wait for every load in flight — in the first tile that is the whole W1 stream)
    float pac[4];
    unsigned dm0[2][2], dm1[2][2];
    {
      const unsigned prow = (unsigned)min(row0 + xr, B - 1);
#pragma unroll
      for (int c = 0; c < 4; ++c) pac[c] = xb[prow * (unsigned)ld + (unsigned)(S + min(xc + 8 * c, A - 1))];
      const unsigned* dbits = drop ? p.drop_bits : (const unsigned*)xb;
      const unsigned mb_off = drop ? (unsigned)MB : 0u;
#pragma unroll
      for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
        for (int ws = 0; ws < 2; ++ws) {
          const unsigned mrow = (unsigned)min(row0 + 16 * r2 + l15, B - 1);
          dm0[r2][ws] = dbits[mrow * 8u + (unsigned)(2 * wv + ws)];
          dm1[r2][ws] = dbits[(mb_off + mrow) * 8u + (unsigned)(2 * wv + ws)];
        }
    }
    x_store();
    __syncthreads();
    if (first) STAMP(p, 2);
    if (rt + nbi < n_rt) x_issue(rt + nbi);       // the next tile's rows, under this tile's arithmetic

    // ---- layer 0: H0[32 rows][units 64 w ..]: A = W0 (m = unit), B = X (n = row)
    {
      f32x4 acc[2][4];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        if (kb < nkb) {
          const bf16x8 x0 = *(const bf16x8*)(Xb + l15 * XLD + 32 * kb + 8 * g);
          const bf16x8 x1 = *(const bf16x8*)(Xb + (16 + l15) * XLD + 32 * kb + 8 * g);
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) {
            acc[0][ct] = MFMA_BF16(w0f[ct][kb], x0, acc[0][ct]);
            acc[1][ct] = MFMA_BF16(w0f[ct][kb], x1, acc[1][ct]);
          }
        }
      }
#pragma unroll
      for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          f32x4 h;
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) h[reg] = fmaxf(acc[r2][ct][reg] + bias0[ct][reg], 0.f);
          if (drop) {      // units 64 wv + 16 CP(ct) + 4 g .. + 3 of row 16 r2 + l15
            const unsigned bits = ((CP(ct) >> 1) ? dm0[r2][1] : dm0[r2][0]) >> ((CP(ct) & 1) * 16 + 4 * g);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) h[reg] = ((bits >> reg) & 1u) ? h[reg] * p.drop_scale : 0.f;
          }
          *(bf16x4*)(H0b + (16 * r2 + l15) * H0B_LD + 64 * wv + 16 * CP(ct) + 4 * g) = cvt4(h);
        }
    }
    if (first) STAMP(p, 3);
    __syncthreads();
    if (first) STAMP(p, 4);
    // H0 of the trained instances -> memory (the backward's dW1 / mask operand)
    if (slot >= 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int f = tid + 256 * q;
        const int rl = f >> 5, col = 8 * (f & 31);
        if (row0 + rl < B)
          *(bf16x8*)(h0g + (unsigned)((slot * MB + row0 + rl) * HID + col)) = *(const bf16x8*)(H0b + rl * H0B_LD + col);
      }
    }
    // ---- layer 1: all 256 units in one pass (wave: its 64), operands of the whole tile read first
    f32x4 h1[2][4];
    {
      bf16x8 bq[2][8];
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) {
        bq[0][kb] = *(const bf16x8*)(H0b + l15 * H0B_LD + 32 * KP(kb) + 8 * g);
        bq[1][kb] = *(const bf16x8*)(H0b + (16 + l15) * H0B_LD + 32 * KP(kb) + 8 * g);
      }
      f32x4 acc[2][4];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < 8; ++kb)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          acc[0][ct] = MFMA_BF16(w1f[ct][kb], bq[0][kb], acc[0][ct]);
          acc[1][ct] = MFMA_BF16(w1f[ct][kb], bq[1][kb], acc[1][ct]);
        }
      if (first) STAMP(p, 5);
#pragma unroll
      for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) h1[r2][ct][reg] = fmaxf(acc[r2][ct][reg] + bias1[ct][reg], 0.f);
          if (drop) {
            const unsigned bits = ((CP(ct) >> 1) ? dm1[r2][1] : dm1[r2][0]) >> ((CP(ct) & 1) * 16 + 4 * g);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) h1[r2][ct][reg] = ((bits >> reg) & 1u) ? h1[r2][ct][reg] * p.drop_scale : 0.f;
          }
        }
    }
    if (slot >= 0) {
#pragma unroll
      for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
          *(bf16x4*)(H1b + (16 * r2 + l15) * H0B_LD + 64 * wv + 16 * CP(ct) + 4 * g) = cvt4(h1[r2][ct]);
    }
    if (!is_pi) {
      // scalar heads (fp32): the wave's 64-unit partial sum of each row = one of the four slice partials the backward adds
#pragma unroll
      for (int r2 = 0; r2 < 2; ++r2) {
        float s = 0.f;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) s = fmaf(h1[r2][ct][reg], w2v[ct][reg], s);
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (wave == 0) s += b2s;
        const int row = row0 + 16 * r2 + l15;
        if (g == 0 && row < B) headsg[(unsigned)(row * HEAD_LD + inst * NSPLIT + wave)] = s;
      }
    } else {
      // policy head on the bf16 MFMA: partial[row][dim] over the wave's 64 units; A = H1 straight from the accumulators
#pragma unroll
      for (int r2 = 0; r2 < 2; ++r2) {
        const bf16x8 hA0 = pack8(h1[r2][0], h1[r2][1]);
        const bf16x8 hA1 = pack8(h1[r2][2], h1[r2][3]);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          if (nt < ndt) {
            f32x4 pa = MFMA_BF16(hA0, w2f[nt][0], ((f32x4){0.f, 0.f, 0.f, 0.f}));
            pa = MFMA_BF16(hA1, w2f[nt][1], pa);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) Pp[(wave * 32 + 16 * r2 + 4 * g + reg) * LB_PLD + 16 * nt + l15] = pa[reg];
          }
        }
      }
    }
    if (first) STAMP(p, 6);
    __syncthreads();
    if (first) STAMP(p, 7);
    if (slot >= 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int f = tid + 256 * q;
        const int rl = f >> 5, col = 8 * (f & 31);
        if (row0 + rl < B)
          *(bf16x8*)(h1g + (unsigned)((slot * MB + row0 + rl) * HID + col)) = *(const bf16x8*)(H1b + rl * H0B_LD + col);
      }
    }
    if (is_pi) {
      // the policy's loss terms of (row xr, dims xc + 8 c), without the advantage weight (iql.py:519-534):
      //   gaussian: -log N(a; mu, sigma) = q / 2 + log sigma + log(2 pi) / 2,  q = (a - mu)^2 / sigma^2;  deterministic: (mu - a)^2
      const int row = row0 + xr;
      float lsum = 0.f;
      f32x4 tv, gv;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int d = xc + 8 * c;
        const int dc = min(d, 16 * ndt - 1);
        const float pre = (((Pp[(0 * 32 + xr) * LB_PLD + dc] + Pp[(1 * 32 + xr) * LB_PLD + dc]) + Pp[(2 * 32 + xr) * LB_PLD + dc]) +
                           Pp[(3 * 32 + xr) * LB_PLD + dc]) + b2v[c];
        const float mu = tanh_via_exp(pre);
        const float diff = pac[c] - mu;
        const float q = diff * diff * ivar[c];
        const bool live = d < A;
        const float l = gauss ? (0.5f * q + lsc[c] + 0.918938533204672742f) : diff * diff;
        const float dmu = gauss ? (-(diff * ivar[c])) * invB : (-2.f * diff) * invB;
        lsum += live ? l : 0.f;
        tv[c] = live ? dmu * (1.f - mu * mu) : 0.f;
        gv[c] = (live && gauss) ? (1.f - q) : 0.f;
      }
      lsum += __shfl_xor(lsum, 1);
      lsum += __shfl_xor(lsum, 2);
      lsum += __shfl_xor(lsum, 4);
      if (row < B) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          a.pi_t[(unsigned)(row * 32 + xc + 8 * c)] = tv[c];
          a.pi_g[(unsigned)(row * 32 + xc + 8 * c)] = gv[c];
        }
        if (xc == 0) a.pi_l[row] = lsum;
      }
    }
    if (first) STAMP(p, 8);
  }
  STAMP(p, 9);
  RT_STAMP(p, 14, rt_entry_);
  RT_STAMP(p, 15, iql_realtime());
#undef CP
#undef KP
}

// ---------------------------------------------------------------------------
// Backward.  grid = 8 x ceil((32 n_cg + nbb) / 2); blockIdx & 7 = x: net = x & 3, parity = x >> 2 (XCD n + 4 parity).  Within a
// net the (b) blocks come first (they are the long ones), then the dW1 tiles.
// NCT: 16-wide tiles of [dW0 | db0]'s kc range a (b) block accumulates in registers (k_in + 1 <= 16 NCT).
// Both kinds of block request the inputs of their NEXT chunk / row tile before they work on the current one (vmcnt
// retires in order, so the work never waits for the prefetch), and nothing is loaded behind a branch that the
// compiler would have to wait in front of.
// (Load shapes: the CU's vector-memory pipe handles about one (16-lane group, cache line) pair per cycle, so the 16 lanes
//  of a group should read one contiguous run.  The per-row head partials — 96 bytes per row, a row per lane: 12 lines per
//  group and instruction — are therefore fetched as a flat, fully coalesced copy and redistributed through LDS.)
struct LbChunkIn {        // what a dW1-tile block loads per 256-row chunk
  f32x4 hq[6];            // the chunk's scalar head partials [256 rows][24] as a flat copy: float4 tid + 256 q
  float r, d;             // Q nets: the thread's row's reward / done
  f32x4 tq[8];            // policy: T of (rows (tid + 256 q) >> 3, dims 4 ((tid + 256 q) & 7) ..)
  float lrow;             // policy, loss block: L of the thread's row
  bf16x2 hh[16];          // H1[row AROW(ks)][j0 + 2 l15 ..]
  bf16x4 bb[16];          // H0[row AROW(ks)][i0 + 4 l15 ..]
};
struct LbTileIn {         // what a (b) block loads per 32-row tile
  RowIn in;               // scalar nets, wave 0: the row's loss inputs
  f32x4 ph[3], t4;        // policy: V / Qt1 / Qt2 partials of row tid >> 3, T of dims 4 (tid & 7) ..
  bf16x8 h1q[4];          // the H1 tile [32][256] as a flat copy: 16 bytes tid + 256 q (-> LDS -> dH1's accumulator layout)
  bf16x4 hm[2][4];        // H0 in dH0's accumulator layout
  f32x4 xq[3];            // X of (row tid >> 3, columns 4 ((tid & 7) + 8 q) ..)
};
template <int NCT>
__global__ __launch_bounds__(256) void iql_bwd_lb_kernel(StepParams p, LbArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  RT_ENTRY();
  const int bid = blockIdx.x;
  const int x = bid & 7;
  const int net = x & 3;
  const int local_ = (bid >> 3) * 2 + (x >> 2);
  const int n_a = 32 * a.n_cg, n_b = a.nbb;
  if (local_ >= n_a + n_b) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int B = p.rows, MB = p.sc.max_batch, ld = p.ld;
  const NetPtrs np = p.net[net];
  const NetGrad go = p.go[net];
  const int D = np.d, k0 = np.k0;
  const bool is_pi = (net == IQLHIP_NET_PI);
  const bool gauss = (p.policy == IQLHIP_POLICY_GAUSSIAN);
  const float* w2 = np.w2;
  const float* H1g = (const float*)((const __bf16*)p.sc.h1 + net * MB * HID);      // (bf16 arrays, read through ld4 / ld2)
  const float* H0g = (const float*)((const __bf16*)p.sc.h0 + net * MB * HID);
  const float* heads = p.sc.heads;
  const float* xb = p.xb;
  const float dscale = (is_pi && p.drop_bits != nullptr) ? p.drop_scale : 1.f;
  const float invB = p.inv_batch;
#define LBROW(r) min((r), B - 1)
  STAMP_BASE(p, 2048 * 16);
  STAMP(p, 0);

  if (local_ >= n_b) {
    // ===================== (a): dW1[32 j][64 i] over the chunks of one chunk group =====================
    const int la = local_ - n_b;
    const int cg = la >> 5, jt = (la >> 2) & 7, it = la & 3;
    const int j0 = jt * 32, i0 = it * 64;
    float* red = smem;                                   // [4][32][T64_LD]
    __bf16* dYb = (__bf16*)(red + 4 * 32 * T64_LD);      // [256][LB_DYLD]  dY of the chunk (policy), rows x dims
    __bf16* dYT = dYb + CHUNK_ROWS * LB_DYLD;            // [32][H0B_LD]    the same, dims x rows (dW2 blocks)
    float* wS = (float*)(dYT + 32 * H0B_LD);             // [256] per-row weight (policy) / dy (scalar nets)
    float* hdS = wS + CHUNK_ROWS;                        // [256][LB_HDLD] the chunk's scalar head partials
    float* part = hdS + CHUNK_ROWS * LB_HDLD;            // [2][32][32] column-sum partials of dY and w G (designated block)
    float* exA = part + 2 * 32 * 32;                     // [16][64]
    float* rsm = exA + 16 * 64;                          // [64]
    float* exB = (float*)dYb;                            // [4][32][32] (after the last chunk: dYb is dead)
    const bool designated = (jt == 0 && it == 2);
    const bool loss_block = (jt == 1 && it == 2);
    const bool do_db1 = (it == 0);
    const bool do_dw2 = is_pi ? (it == 1 || it == 3) : (it == 0);
    const int tb_own = (it == 3) ? 1 : 0;
    const bool extras = do_db1 || do_dw2;
    const int ndt = (D + 15) >> 4;
    const bool is_q = (net == IQLHIP_NET_Q1 || net == IQLHIP_NET_Q2);
    float* slab = p.sc.slab_a + (long long)cg * p.n_params;
#define AROW(ks) (64 * wave + 16 * ((ks) >> 2) + 4 * g + ((ks) & 3))
    auto issue = [&](LbChunkIn& s, int c) {
      const int cbase = c * CHUNK_ROWS;
      {
        const unsigned hmax = (unsigned)B * 6u - 1u;      // last float4 of the batch's head partials
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          if (LB_SKIP & 16) s.hq[q] = (f32x4){0.1f, 0.2f, 0.3f, 0.4f};
          else s.hq[q] = *(const f32x4*)(heads + 4u * min((unsigned)cbase * 6u + (unsigned)(tid + 256 * q), hmax));
        }
        s.r = 0.f; s.d = 0.f;
        if (is_q) {
          const unsigned ox = (unsigned)LBROW(cbase + tid) * (unsigned)ld + (unsigned)(2 * p.S + p.A);
          s.r = xb[ox];
          s.d = xb[ox + 1u];
        }
      }
      if (is_pi) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int f = tid + 256 * q;
          if (LB_SKIP & 8) s.tq[q] = (f32x4){0.1f, 0.2f, 0.3f, 0.4f};
          else s.tq[q] = *(const f32x4*)(a.pi_t + (unsigned)(LBROW(cbase + (f >> 3)) * 32 + 4 * (f & 7)));
        }
        s.lrow = a.pi_l[LBROW(cbase + tid)];
      }
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const unsigned row = (unsigned)LBROW(cbase + AROW(ks));
        if (LB_SKIP & 2) {
          s.hh[ks] = (bf16x2){(__bf16)1.f, (__bf16)0.f};
          s.bb[ks] = (bf16x4){(__bf16)1.f, (__bf16)0.f, (__bf16)2.f, (__bf16)0.f};
        } else {
          s.hh[ks] = ld2<true>(H1g, row * (unsigned)HID + (unsigned)(j0 + 2 * l15));
          s.bb[ks] = ld4<true>(H0g, row * (unsigned)HID + (unsigned)(i0 + 4 * l15));
        }
      }
    };
    const int c0 = cg * a.cpb, c_end = min((cg + 1) * a.cpb, a.n_chunk);
    LbChunkIn SA, SB;
    if (!is_pi) {
#pragma unroll
      for (int q = 0; q < 8; ++q) { SA.tq[q] = (f32x4){0.f, 0.f, 0.f, 0.f}; SB.tq[q] = SA.tq[q]; }
      SA.lrow = 0.f; SB.lrow = 0.f;
    }
    issue(SA, c0);
    // W2 of this j tile, straight from memory in operand layout.  Scalar nets: the lane's two columns.  Policy: the B
    // operand of dH1 = dY . W2 (k = dim 8 g + e, n = j = 2 l15 + ta), in registers for the whole block
    f32x2 wq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) wq[e] = *(const f32x2*)(w2 + (unsigned)(min(8 * g + e, D - 1) * HID + j0 + 2 * l15));
    const float w2a = wq[0][0], w2b = wq[0][1];      // (D == 1: every e reads row 0)
    bf16x8 W2B[2];
#pragma unroll
    for (int ta = 0; ta < 2; ++ta) {
      float t8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) t8[e] = (8 * g + e < D) ? wq[e][ta] : 0.f;
      W2B[ta] = pack8s(t8);
    }
    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 acc2[2];      // policy dW2 tiles [dt] of the lane's column tb_own
    acc2[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc2[1] = acc2[0];
    float db1a[2] = {0.f, 0.f}, dw2a[2] = {0.f, 0.f};
    float tot_b2 = 0.f, tot_ls = 0.f;      // designated block: thread d (< 32) accumulates db2[d] / dlog_std[d] over the chunks
    STAMP(p, 1);

    auto compute = [&](LbChunkIn& s, int c) {
      const int cbase = c * CHUNK_ROWS;
      const bool first = (c == c0);
      const int prow = cbase + tid;
      // the policy's dlog_std terms: one block in 32 needs them — loaded here, not prefetched
      f32x4 gq[8];
      if (is_pi && designated && gauss) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int f = tid + 256 * q;
          gq[q] = *(const f32x4*)(a.pi_g + (unsigned)(LBROW(cbase + (f >> 3)) * 32 + 4 * (f & 7)));
        }
      }
      // ---- the flat copy of the head partials -> [row][24] in LDS -> the thread's own row
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        const int f = tid + 256 * q;
        *(f32x4*)(hdS + (f / 6) * LB_HDLD + 4 * (f % 6)) = s.hq[q];
      }
      __syncthreads();
      // ---- per row: dy of a scalar head, or the policy's advantage weight
      float lossA = 0.f, lossB = 0.f;
      {
        RowIn in;
#pragma unroll
        for (int i = 0; i < 6; ++i) in.h[i] = *(const f32x4*)(hdS + tid * LB_HDLD + 4 * i);
        in.r = s.r; in.d = s.d;
        float v = 0.f;
        if (prow < B) {
          if (!is_pi) {
            row_finish(p, net, in, &v, lossA, lossB);
          } else {
            const float tqv = fminf(sum4(in.h[2]), sum4(in.h[3]));
            const float u = tqv - sum4(in.h[1]);
            v = fminf(expf(p.hy.beta * u), p.hy.exp_adv_max);
            lossA = v * s.lrow;
          }
        }
        wS[tid] = v;
      }
      if (first) STAMP(p, 2);
      __syncthreads();
      if (first) STAMP(p, 3);
      if (loss_block) {
        const float sA = block_sum_256(lossA, rsm);
        if (net == IQLHIP_NET_V && tid == 0) p.sc.loss_parts[0 * 64 + c] = sA;
        if (net == IQLHIP_NET_PI && tid == 0) p.sc.loss_parts[3 * 64 + c] = sA;
        if (net == IQLHIP_NET_Q1) {
          const float sB = block_sum_256(lossB, rsm + 8);
          if (tid == 0) { p.sc.loss_parts[1 * 64 + c] = sA; p.sc.loss_parts[2 * 64 + c] = sB; }
        }
      }
      if (is_pi) {
        // dY = w T of the chunk -> bf16 [row][dim] (dH1's A operand) and, dW2 blocks, [dim][row] (dW2's A operand)
        f32x4 csum = (f32x4){0.f, 0.f, 0.f, 0.f}, gsum = csum;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int f = tid + 256 * q;
          const int r = f >> 3, c4 = f & 7;
          const float w = wS[r];
          const f32x4 v = s.tq[q] * w;
          *(bf16x4*)(dYb + r * LB_DYLD + 4 * c4) = cvt4(v);
          if (do_dw2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) dYT[(4 * c4 + j) * H0B_LD + r] = (__bf16)v[j];
          }
          if (designated) {
            csum += v;
            if (gauss) gsum += gq[q] * w;
          }
        }
        if (designated) {       // thread (row group tid >> 3, dims 4 (tid & 7) ..): its 8 rows' sums
          *(f32x4*)(part + (tid >> 3) * 32 + 4 * (tid & 7)) = csum;
          *(f32x4*)(part + 1024 + (tid >> 3) * 32 + 4 * (tid & 7)) = gsum;
        }
        __syncthreads();
        if (designated && tid < 32) {
          float sb = 0.f, sl = 0.f;
#pragma unroll 8
          for (int k = 0; k < 32; ++k) { sb += part[k * 32 + tid]; sl += part[1024 + k * 32 + tid]; }
          tot_b2 += sb;
          tot_ls += sl;
        }
      } else if (designated && wave == 0) {
        float sb = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) sb += wS[lane + 64 * q];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sb += __shfl_xor(sb, o);
        tot_b2 += sb;
      }
      if (first) STAMP(p, 4);
      // ---- operand phase: av[ks][ta] = dH1[row AROW(ks)][j0 + 2 l15 + ta]
      float av[16][2];
      if (!is_pi) {
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
          const float dy = wS[AROW(ks)];
          av[ks][0] = ((float)s.hh[ks][0] > 0.f) ? dy * w2a * dscale : 0.f;
          av[ks][1] = ((float)s.hh[ks][1] > 0.f) ? dy * w2b * dscale : 0.f;
          if (do_dw2) {
            dw2a[0] = fmaf(dy, (float)s.hh[ks][0], dw2a[0]);
            dw2a[1] = fmaf(dy, (float)s.hh[ks][1], dw2a[1]);
          }
        }
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const bf16x8 Ad = *(const bf16x8*)(dYb + (64 * wave + 16 * t + l15) * LB_DYLD + 8 * g);
          const f32x4 p0 = MFMA_BF16(Ad, W2B[0], ((f32x4){0.f, 0.f, 0.f, 0.f}));
          const f32x4 p1 = MFMA_BF16(Ad, W2B[1], ((f32x4){0.f, 0.f, 0.f, 0.f}));
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            av[4 * t + reg][0] = ((float)s.hh[4 * t + reg][0] > 0.f) ? p0[reg] * dscale : 0.f;
            av[4 * t + reg][1] = ((float)s.hh[4 * t + reg][1] > 0.f) ? p1[reg] * dscale : 0.f;
          }
        }
      }
      if (do_db1) {
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) { db1a[0] += av[ks][0]; db1a[1] += av[ks][1]; }
      }
      if (first) STAMP(p, 5);
      // ---- dW1 += dH1^T . H0 over the chunk's 256 rows: k index e of k-block q = row AROW(8 q + e) on both operands
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        bf16x8 Aop[2], Bv[4];
#pragma unroll
        for (int ta = 0; ta < 2; ++ta) {
          float t8[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) t8[e] = av[8 * q + e][ta];
          Aop[ta] = pack8s(t8);
        }
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
          for (int e = 0; e < 8; ++e) Bv[tb][e] = s.bb[8 * q + e][tb];
#pragma unroll
        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
          for (int tb = 0; tb < 4; ++tb) acc[ta][tb] = MFMA_BF16(Aop[ta], Bv[tb], acc[ta][tb]);
      }
      if (is_pi && do_dw2) {
        // dW2[dim][j] += dY^T . H1: A = dY^T (m = dim 16 dt + l15, k = row), B = H1 (n = the lane's column tb_own)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          bf16x8 Bh;
#pragma unroll
          for (int e = 0; e < 8; ++e) Bh[e] = s.hh[8 * q + e][tb_own];
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            if (dt < ndt) {
              const __bf16* src = dYT + (16 * dt + l15) * H0B_LD + 64 * wave + 32 * q + 4 * g;
              const bf16x8 Ay = cat8(*(const bf16x4*)src, *(const bf16x4*)(src + 16));
              acc2[dt] = MFMA_BF16(Ay, Bh, acc2[dt]);
            }
          }
        }
      }
      if (first) STAMP(p, 6);
      __syncthreads();      // every thread has left this chunk's LDS tiles
      if (first) STAMP(p, 7);
    };
    for (int c = c0; c < c_end; c += 2) {
      if (c + 1 < c_end) issue(SB, c + 1);
      compute(SA, c);
      if (c + 1 < c_end) {
        if (c + 2 < c_end) issue(SA, c + 2);
        compute(SB, c + 1);
      }
    }
    STAMP(p, 8);

    // ---- cross-wave reduction of the tile, extras, stores (as iql_bwd_kernel's (a) blocks)
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
      if (!is_pi) *(f32x2*)(mine + 32) = (f32x2){dw2a[0], dw2a[1]};
      if (is_pi && do_dw2) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
          if (dt < ndt)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
              exB[(wave * 32 + 16 * dt + 4 * g + reg) * 32 + 2 * l15 + tb_own] = acc2[dt][reg];
      }
    }
    __syncthreads();
    {
      float* gw1 = slab + go.w1;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int f = tid + 256 * q;
        const int jl = f >> 4, i4 = f & 15;
        f32x4 sv = *(const f32x4*)(red + jl * T64_LD + 4 * i4);
#pragma unroll
        for (int w = 1; w < 4; ++w) sv += *(const f32x4*)(red + w * 32 * T64_LD + jl * T64_LD + 4 * i4);
        *(f32x4*)(gw1 + (j0 + jl) * HID + i0 + 4 * i4) = sv;
      }
    }
    if (extras) {
      const int e_lo = do_db1 ? 0 : 32;
      const int e_hi = do_dw2 ? (1 + D) * 32 : 32;
      for (int e = tid + e_lo; e < e_hi; e += 256) {
        const int rr = e >> 5, jj = e & 31;
        float sv;
        if (rr == 0 || !is_pi) {
          float wsum[4];
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            const float* q = exA + (w * 4) * 64 + rr * 32 + jj;
            wsum[w] = (q[0] + q[64]) + (q[128] + q[192]);
          }
          sv = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
        } else {
          const float* b = exB + (rr - 1) * 32 + jj;
          sv = (b[0] + b[1024]) + (b[2048] + b[3072]);
        }
        if (rr == 0) { if (do_db1) slab[go.b1 + j0 + jj] = sv; }
        else if (do_dw2 && (!is_pi || (jj & 1) == tb_own)) slab[go.w2 + (rr - 1) * HID + j0 + jj] = sv;
      }
    }
    if (designated) {
      if (!is_pi) {
        if (tid == 0) slab[go.b2] = tot_b2;
      } else if (tid < D) {
        slab[go.b2 + tid] = tot_b2;
        if (gauss) {
          const float lsr = p.log_std[tid];
          const bool inside = (lsr >= p.hy.log_std_min) && (lsr <= p.hy.log_std_max);
          slab[go.log_std + tid] = inside ? tot_ls * invB : 0.f;
        }
      }
    }
    STAMP(p, 9);
    RT_STAMP(p, 14, rt_entry_);
    RT_STAMP(p, 15, iql_realtime());
    return;
  }
#undef AROW

  // ===================== (b): dH1 -> dH0 -> [dW0 | db0] over a strided set of 32-row tiles =====================
  {
    const int lb = local_;
    float* slabB = p.sc.slab_b + p.sc.slab_b_off[net] + (long long)lb * (HID * k0 + HID);
    if (lb >= a.n_rt) {      // (fewer row tiles than (b) blocks: this block's slab is still summed — it must hold zeros)
      for (int e = tid; e < HID * k0 + HID; e += 256) slabB[e] = 0.f;
      return;
    }
    __bf16* dH1b = (__bf16*)smem;                        // [32][H0B_LD]
    __bf16* dH0T = dH1b + 32 * H0B_LD;                   // [256][LB_DYLD]  masked dH0, transposed: [col][row]
    __bf16* XT = dH0T + 256 * LB_DYLD;                   // [16 NCT][LB_DYLD] [X | 1 | 0]^T: [kc][row]
    __bf16* dYb = XT + 16 * NCT * LB_DYLD;               // [32][LB_DYLD]  dY of the tile (policy)
    float* dys = (float*)(dYb + 32 * LB_DYLD);           // [32] dy of the tile (scalar nets)
    __bf16* H1t = (__bf16*)(dys + 32);                   // [32][H0B_LD] the H1 tile (mask of dH1)
    const int nct = (k0 + 1 + 15) >> 4;
    const int xr = tid >> 3, xc = tid & 7;
    auto issue = [&](LbTileIn& s, int rt) {
      const int row0 = rt * RT_ROWS;
      // (the policy's loads are issued by every net's blocks — from valid addresses — no branch in front of the big streams)
      if (wave == 0) row_issue(p, LBROW(row0 + (tid & 31)), s.in);
      const unsigned prow = (unsigned)LBROW(row0 + xr);
      const unsigned oh = prow * (unsigned)HEAD_LD;
      s.ph[0] = *(const f32x4*)(heads + (oh + 4u)); s.ph[1] = *(const f32x4*)(heads + (oh + 8u)); s.ph[2] = *(const f32x4*)(heads + (oh + 12u));
      s.t4 = *(const f32x4*)(a.pi_t + (prow * 32u + 4u * (unsigned)xc));
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int f = tid + 256 * q;
        s.h1q[q] = *(const bf16x8*)((const __bf16*)H1g + ((unsigned)LBROW(row0 + (f >> 5)) * (unsigned)HID + (unsigned)(8 * (f & 31))));
      }
#pragma unroll
      for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
          s.hm[r2][reg] = ld4<true>(H0g, (unsigned)LBROW(row0 + 16 * r2 + 4 * g + reg) * (unsigned)HID + (unsigned)(64 * wave + 4 * l15));
      const unsigned xrow = prow * (unsigned)ld;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        if (q < (16 * NCT + 31) / 32) {
          const int c = min(4 * (xc + 8 * q), (k0 - 1) & ~3);      // (c + 3 <= k0 + 2 < ld: inside the row; columns >= k0 unused)
          s.xq[q] = *(const f32x4*)(xb + (xrow + (unsigned)c));
        }
      }
    };
    LbTileIn T;
#pragma unroll
    for (int i = 0; i < 6; ++i) T.in.h[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    T.in.r = 0.f; T.in.d = 0.f;
    if (LB_SKIP & 32) {
      T.ph[0] = T.ph[1] = T.ph[2] = T.t4 = (f32x4){0.1f, 0.2f, 0.3f, 0.4f};
#pragma unroll
      for (int q = 0; q < 4; ++q) T.h1q[q] = (bf16x8){(__bf16)1.f, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < 8; ++i) T.hm[i >> 2][i & 3] = (bf16x4){(__bf16)1.f, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < 3; ++q) T.xq[q] = (f32x4){0.1f, 0.2f, 0.3f, 0.4f};
    } else
    issue(T, lb);
    // ---- the wave's operands for the whole block (requested BEHIND the first tile's inputs: the loss arithmetic and the
    // dH1 tile run under this stream)
    // W2: scalar nets — the lane's 16 columns j = 64 w + 16 jt + 4 g + reg; policy — A operand of dH1^T = W2^T . dY^T
    // (m = j = 64 w + 16 jt + l15, k = dim 8 g + e)
    f32x4 w2q[4];
    float w2s[4][8];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      w2q[jt] = *(const f32x4*)(w2 + (unsigned)(64 * wave + 16 * jt + 4 * g));
#pragma unroll
      for (int e = 0; e < 8; ++e) w2s[jt][e] = w2[(unsigned)(min(8 * g + e, D - 1) * HID + 64 * wave + 16 * jt + l15)];
    }
    // W1 (bf16 shadow): B operand of dH0 = dH1 . W1, k = j = 32 kb + 8 g + e, n = column 64 w + 4 l15 + tb
    bf16x4 bw[64];
#pragma unroll
    for (int ks = 0; ks < 64; ++ks) {
      if (LB_SKIP & 1) bw[ks] = (bf16x4){(__bf16)1.f, (__bf16)0.f, (__bf16)2.f, (__bf16)0.f};
      else bw[ks] = ld4<true>(np.w1, (unsigned)((32 * (ks >> 3) + 8 * g + (ks & 7)) * HID + 64 * wave + 4 * l15));
    }
    bf16x8 w2A[4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      float t8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) t8[e] = (8 * g + e < D) ? w2s[jt][e] : 0.f;
      w2A[jt] = pack8s(t8);
    }
    f32x4 accW[4][NCT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NCT; ++j) accW[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    STAMP(p, 1);

    for (int rt = lb; rt < a.n_rt; rt += a.nbb) {
      const int row0 = rt * RT_ROWS;
      const bool first = (rt == lb);
      // [X | 1 | 0]^T as bf16: thread (row xr, kc 4 (xc + 8 q) ..)
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        if (q < (16 * NCT + 31) / 32) {
          const int c = 4 * (xc + 8 * q);
          if (c < 16 * nct) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int kc = c + j;
              const float xvj = (kc < k0) ? T.xq[q][j] : ((kc == k0) ? 1.f : 0.f);      // ones column -> db0
              XT[kc * LB_DYLD + xr] = (__bf16)xvj;
            }
          }
        }
      }
      // the H1 tile -> LDS
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int f = tid + 256 * q;
        *(bf16x8*)(H1t + (f >> 5) * H0B_LD + 8 * (f & 31)) = T.h1q[q];
      }
      // ---- dy of the scalar heads / dY = w T of the policy
      if (!is_pi) {
        if (tid < RT_ROWS) {
          float v = 0.f, la_, lb_;
          if (row0 + tid < B) row_finish(p, net, T.in, &v, la_, lb_);
          dys[tid] = v;
        }
      } else {
        float w = 0.f;
        if (row0 + xr < B) {
          const float tqv = fminf(sum4(T.ph[1]), sum4(T.ph[2]));
          const float u = tqv - sum4(T.ph[0]);
          w = fminf(expf(p.hy.beta * u), p.hy.exp_adv_max);
        }
        *(bf16x4*)(dYb + xr * LB_DYLD + 4 * xc) = cvt4(T.t4 * w);
      }
      if (first) STAMP(p, 2);
      __syncthreads();
      if (first) STAMP(p, 3);
      // ---- dH1 tile [32][256] (bf16): (dY . W2) masked by H1 > 0; lane: rows 16 r2 + l15, columns 64 w + 16 jt + 4 g ..
#pragma unroll
      for (int r2 = 0; r2 < 2; ++r2) {
        bf16x8 Bd = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        float dyr = 0.f;
        if (is_pi) Bd = *(const bf16x8*)(dYb + (16 * r2 + l15) * LB_DYLD + 8 * g);
        else dyr = dys[16 * r2 + l15];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
          f32x4 pre;
          if (is_pi) pre = MFMA_BF16(w2A[jt], Bd, ((f32x4){0.f, 0.f, 0.f, 0.f}));
          else pre = w2q[jt] * dyr;
          const bf16x4 hmk = *(const bf16x4*)(H1t + (16 * r2 + l15) * H0B_LD + 64 * wave + 16 * jt + 4 * g);
          f32x4 o;
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) o[reg] = ((float)hmk[reg] > 0.f) ? pre[reg] * dscale : 0.f;
          *(bf16x4*)(dH1b + (16 * r2 + l15) * H0B_LD + 64 * wave + 16 * jt + 4 * g) = cvt4(o);
        }
      }
      __syncthreads();
      if (first) STAMP(p, 4);
      // ---- dH0 = dH1 . W1: the wave's 64 columns over all 256 k
      {
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
        for (int kb = 0; kb < 8; ++kb)
#pragma unroll
          for (int tb = 0; tb < 4; ++tb) {
            bf16x8 Bv;
#pragma unroll
            for (int e = 0; e < 8; ++e) Bv[e] = bw[8 * kb + e][tb];
            acc[0][tb] = MFMA_BF16(Ad[0][kb], Bv, acc[0][tb]);
            acc[1][tb] = MFMA_BF16(Ad[1][kb], Bv, acc[1][tb]);
          }
        if (first) STAMP(p, 5);
        // masked, transposed (bf16 [col][row]): the dW0 product's operand is then one 16-byte read
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
          for (int tb = 0; tb < 4; ++tb) {
            bf16x4 o;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
              o[reg] = (__bf16)(((float)T.hm[r2][reg][tb] > 0.f) ? acc[r2][tb][reg] * dscale : 0.f);      // rows >= B carry 0
            *(bf16x4*)(dH0T + (64 * wave + 4 * l15 + tb) * LB_DYLD + 16 * r2 + 4 * g) = o;
          }
      }
      // the next tile's inputs: everything of this tile's has been consumed
      if (!(LB_SKIP & 32) && rt + a.nbb < a.n_rt) issue(T, rt + a.nbb);
      __syncthreads();
      if (first) STAMP(p, 6);
      // ---- [dW0 | db0][i][kc] += sum_r dH0[r][i] [X | 1][r][kc]: A = [X | 1]^T (m = kc, k = row), B = dH0T (n = i)
      {
        bf16x8 Ax[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) Ax[ct] = *(const bf16x8*)(XT + (16 * min(ct, nct - 1) + l15) * LB_DYLD + 8 * g);
#pragma unroll
        for (int itl = 0; itl < 4; ++itl) {
          const bf16x8 Bd = *(const bf16x8*)(dH0T + (64 * wave + 16 * itl + l15) * LB_DYLD + 8 * g);
#pragma unroll
          for (int ct = 0; ct < NCT; ++ct)
            if (ct < nct) accW[itl][ct] = MFMA_BF16(Ax[ct], Bd, accW[itl][ct]);
        }
      }
      __syncthreads();      // the next tile rewrites XT / dYb / dH1b / dH0T
      if (first) STAMP(p, 7);
    }
    STAMP(p, 8);
    if (!(LB_SKIP & 4))
    // ---- the block's slab [i][kc], db0 behind it.  A lane's 4 registers are 4 consecutive kc of one i — 16 bytes at a
    // 4-byte-aligned address of a 268-byte row: as direct stores the 20 of them took 4-5 k cycles.  Each wave parks its 64
    // rows (contiguous in the slab) in LDS and copies them out as aligned, fully coalesced 16-byte stores.
    {
      float* park = smem + wave * (64 * k0);           // (16-byte aligned: 64 k0 floats; the loop's barrier freed the tiles)
      float* dstB = slabB + HID * k0;
#pragma unroll
      for (int itl = 0; itl < 4; ++itl) {
        const int il = 16 * itl + l15;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          if (ct < nct) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
              const int kc = 16 * ct + 4 * g + reg;
              if (kc < k0) park[il * k0 + kc] = accW[itl][ct][reg];
              else if (kc == k0) dstB[64 * wave + il] = accW[itl][ct][reg];
            }
          }
        }
      }
      float* dst = slabB + wave * (64 * k0);
      // (all the reads first, then the stores: a read -> store loop waits for each LDS round trip in turn)
      constexpr int NQ = (16 * (16 * NCT - 1) + 63) / 64;      // 16 k0 float4 per wave, k0 <= 16 NCT - 1
      f32x4 cv[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) cv[q] = *(const f32x4*)(park + 4 * min(lane + 64 * q, 16 * k0 - 1));
#pragma unroll
      for (int q = 0; q < NQ; ++q) if (lane + 64 * q < 16 * k0) *(f32x4*)(dst + 4 * (lane + 64 * q)) = cv[q];
    }
    STAMP(p, 9);
    RT_STAMP(p, 14, rt_entry_);
    RT_STAMP(p, 15, iql_realtime());
  }
#undef LBROW
}
