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
//   iql_bwd_rows_kernel block = (net, a strided set of 32-row tiles), the wave's 64 columns of W1 (all 256 k) in registers
//                       for the whole block.  Per tile: dY = w T (loaded, not recomputed) -> dH1 = (dY . W2) masked ->
//                       dH0 = (dH1 . W1) masked; dH1, dH0 (and the policy's dY) go to memory as bf16 rows, and everything
//                       that is a plain sum over the block's rows — db1, the scalar nets' dW2, db2, dlog_std, the loss
//                       sums — is accumulated in registers and written once per block.
//   iql_bwd_gemm_kernel every product that contracts over the batch rows — dW1 = dH1^T . H0, [dW0 | db0] = dH0^T . [X | 1],
//                       the policy's dW2 = dY^T . H1 — as ONE LDS-tiled bf16 GEMM: 64 x 64 output tiles, row-major
//                       operand tiles copied in with fully coalesced loads, operands transposed on the way out of LDS
//                       (ds_read_b64_tr_b16), split over chunk groups of rows.  ~64 registers: four blocks per CU.
//   iql_update_kernel   LB instantiation: sums the chunk-group slabs (w1, w0, b0, policy w2) and the row blocks' slabs.
// (A first version kept iql_bwd_kernel's shape — dW1 tiles that rebuild dH1 per chunk next to row-tile blocks, in one
//  launch; with one wave per SIMD its blocks were bound by their own instruction streams: with every load removed the
//  launch still took 14 of its 23 us at 1 024 rows.)
//
// Lane maps (wave64, l15 = lane & 15, g = lane >> 4), v_mfma_f32_16x16x32_bf16:
//   A[m = l15][k = 8 g + e]   B[k = 8 g + e][n = l15]   D[m = 4 g + reg][n = l15]      (e = 0..7, reg = 0..3)
// Reference arithmetic replaced: algorithms/finetune/iql.py:314-344 (MLP), :482-540 (_update_v/_q/_policy).
#pragma once
#include "iqlhip_kernels.h"

#ifndef LB_STAMP_TILE
#define LB_STAMP_TILE 0
#endif
#ifndef LB_SKIP
#define LB_SKIP 0           // timing experiments only (wrong results): bit 0 (b) no W1 loads, 1 (a) no H1 / H0 loads, 2 (b) no slab
#endif                      // store, 3 (a) no T loads, 4 (a) no head loads, 5 (b) no tile loads
#define LB_DYLD 40          // bf16 row stride of a [rows][32 dims] / [cols][32 rows] tile: 80 bytes, 16-byte aligned rows
#define LB_XLD 128          // bf16 row stride of the [s | a] copy of the batch (S + A <= 128)
#define LB_PLD 33           // fp32 row stride of the policy head partials [wave][32 rows][32 dims]
#define LB_HDLD 28          // fp32 row stride of a chunk's scalar head partials in LDS (24 + 4: 16-byte rows, banks spread)

struct LbArgs {
  float* pi_t;      // [max_batch][32]  T: dL_pi/d(pre-tanh) / w   (zero beyond action_dim)
  float* pi_g;      // [max_batch][32]  G: 1 - (a - mu)^2 / sigma^2 (gaussian; dlog_std terms / w)
  float* pi_l;      // [max_batch]      L: sum_d of the per-dim loss terms (actor loss of the row / w)
  int n_rt;         // 32-row tiles of the batch
  int n_chunk;      // 256-row chunks of the batch
  int nbi;          // forward: blocks per instance (even): block ib walks row tiles ib, ib + nbi, ...
  int nbb;          // backward: row blocks per net (even): block lb walks row tiles lb, lb + nbb, ...
  int cpb;          // backward: 256-row chunks per GEMM block
  int n_cg;         // backward: chunk groups = ceil(n_chunk / cpb) = chunk slabs written
  __bf16* dh1g;     // [4][max_batch][256] dL/d(pre-activation of layer 1), masked — rows of the row-contraction GEMMs
  __bf16* dh0g;     // [4][max_batch][256] the same of layer 0
  __bf16* dyg;      // [max_batch][32]     the policy's dY = w T
  __bf16* xbf;      // [max_batch][LB_XLD] the batch's [s | a] columns as bf16 (zero beyond S + A), written by the forward's Q1 blocks
  float* slab_x;    // [64][n_params]      per row block: partial sums of b1, scalar w2, b2, log_std gradients (arena layout)
  long long go_w0[4], go_b0[4];      // arena offsets of w0 / b0 per net (the others: StepParams::go)
  long long go_w1n[4];               // first arena element of each net's segment
  const __bf16* wimg;                // operand images of W1 / W0 (iqlhip_kernels.h), slots V, Q1, Q2, pi, target Q1, target Q2
  __bf16* w1t;                       // [4][65 536] W1 of the trained nets as the B operand of dH0 = dH1 . W1 (iql_w1t_build)
};

// Sum of v over the 16 lanes of a row (lanes 16 g .. 16 g + 15), left in every lane of the row: four DPP row rotations
// (row_ror:8 / 4 / 2 / 1) — vector-ALU instructions, no LDS round trip (__shfl_xor goes through ds_bpermute).
__device__ __forceinline__ float row16_sum(float v) {
#define ROR_ADD(n_) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + (n_), 0xF, 0xF, false))
  ROR_ADD(8); ROR_ADD(4); ROR_ADD(2); ROR_ADD(1);
#undef ROR_ADD
  return v;
}

__device__ __forceinline__ bf16x4 lds_tr4(const __bf16* ptr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)ptr);
}

__device__ __forceinline__ bf16x4 cvt4(const f32x4 v) {
  bf16x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = (__bf16)v[i];
  return r;
}

// ---------------------------------------------------------------------------
// W1 as the B operand of dH0 = dH1 . W1 (contraction over W1's ROW index j): fragment (slab w = i >> 6, tile tb = i & 3,
// k-block kb = j >> 5), lane = ((i >> 2) & 15) + 16 ((j >> 3) & 3), element = j & 7 — a lane's 8 elements are 8 consecutive
// j of one column i = 64 w + 4 l15 + tb, i.e. a TRANSPOSE of the row-major weights.  Built once per step from the bf16
// shadow by the forward's idle eighth (the update kernel's threads own 4 consecutive i of one j: from there it would be
// four scattered 2-byte stores per thread, ~1.7 us of the chip's vector-memory pipes), read by iql_bwd_rows_kernel.
// Unit u of 128: (net u >> 5, kb (u >> 2) & 7, w u & 3): 32 rows x 64 columns in, 4 fragments of 1 KB out.
__device__ __forceinline__ void iql_w1t_build(const StepParams& p, __bf16* w1t, __bf16* stage /* 2 048 elements of LDS */, int blk, int nblk) {
  const int tid = threadIdx.x;
  for (int u = blk; u < 128; u += nblk) {
    const int net = u >> 5, kb = (u >> 2) & 7, w = u & 3;
    const int jr = tid >> 3, c8 = 8 * (tid & 7);
    const bf16x8 v = *(const bf16x8*)((const __bf16*)p.net[net].w1 + (unsigned)((32 * kb + jr) * HID + 64 * w + c8));
    __syncthreads();      // (the previous unit's copy-out has read the stage)
#pragma unroll
    for (int x = 0; x < 8; ++x) {
      const int il = c8 + x;
      stage[(il & 3) * 512 + (((il >> 2) & 15) + 16 * (jr >> 3)) * 8 + (jr & 7)] = v[x];
    }
    __syncthreads();
    *(bf16x8*)(w1t + (unsigned)(net * 65536 + ((w * 4 + (tid >> 6)) * 8 + kb) * 512 + (tid & 63) * 8)) = *(const bf16x8*)(stage + tid * 8);
  }
  // the policy's W2 as the A operand of dH1^T = W2^T . dY^T (m = j = 64 w + 16 jt + l15, k = dim 8 g + e; zero beyond the
  // action dims), behind the four W1 images: 4 slabs x 4 tiles x 64 lanes x 8 — one 16-byte load per tile in the row kernel
  // instead of eight scalar gathers
  for (int w = blk; w < 4; w += nblk) {
    const int jt = tid >> 6, ln = tid & 63, D = p.net[IQLHIP_NET_PI].d;
    float t8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int dd = 8 * (ln >> 4) + e;
      const float v = p.net[IQLHIP_NET_PI].w2[(unsigned)(min(dd, D - 1) * HID + 64 * w + 16 * jt + (ln & 15))];
      t8[e] = (dd < D) ? v : 0.f;
    }
    *(bf16x8*)(w1t + (unsigned)(4 * 65536 + ((w * 4 + jt) * 64 + ln) * 8)) = pack8s(t8);
  }
}

// ---------------------------------------------------------------------------
// Forward.  grid = 8 x nbi; XCD map and instance pairs as iql_fwd_kernel's (XCDs n and n + 4 host the two instances that
// read net n's weights; block parity = row-tile parity, which the backward's (b) blocks of that parity read back).
// NKB: 32-wide k-blocks of the widest layer-0 input (S + A <= 128 -> <= 4).
// SPREAD (more than one row tile per block, i.e. above 1 024 rows): see pi_spread below.
template <int NKB, bool SPREAD = false>
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
  int inst = (int)((((fr & 1) ? FWD_PAIR_B : FWD_PAIR_A) >> (4 * (fx & 3))) & 7u);
  const int nbi = a.nbi;
  int ib = (fr >> 1) * 2 + fh;
  const int n_rt = a.n_rt;
  // The policy's blocks are the forward's longest (its loss terms: +2 k cycles per row tile), and they share their XCD pair
  // with the idle eighth, whose blocks are done after a few microseconds: at 8 192 rows the policy's 32 blocks ended the
  // launch at 38 us while every other block was done by 31.  SPREAD: of the T = ceil(n_rt / nbi) tiles a policy block would
  // walk, the idle block of the same index takes the last 3 T / 8 (none below 3 tiles per block: in graph chunks the idle
  // duties include the next step's gather, ~16 k cycles at 8 192 rows — with an even split the idle blocks became the
  // launch's last ones instead).  Same stride, same parity, hence the same XCD as the tiles' consumers; the idle duties
  // (nothing in this launch consumes them) run behind the block's tiles.
  int rt_hi = n_rt;                      // the block's tiles: ib, ib + nbi, ... below rt_hi
  const int ib_idle = ib;
  const bool idle_block = inst >= 7;
  if constexpr (SPREAD) {
    const int T = (n_rt + nbi - 1) / nbi;
    const int t_idle = (3 * T) >> 3;
    if (idle_block) {
      inst = 6;
      ib += nbi * (T - t_idle);          // (>= n_rt when it takes no tiles: the loop below does not run)
    } else if (inst == 6) {
      rt_hi = min(n_rt, ib + nbi * (T - t_idle));
    }
  } else {
    if (idle_block) {      // the idle eighth: W1 transposed for the backward; in graph chunks the NEXT step's bookkeeping
      iql_w1t_build(p, a.w1t, H0b, ib, nbi);
      if (p.g_work) idle_block_work(p.g_work, ib, nbi);
      return;
    }
    if (ib >= n_rt) return;
  }
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
  bf16x4 xo[NKB];         // the thread's bf16 X values of the current tile (the Q1 instance copies them out)
  const unsigned xtotal = (unsigned)B * (unsigned)ld;
  // every kernel-argument word the block uses, fetched in one batch now (hipcc sinks each scalar load next to its first
  // use: a dependent ~500-cycle fetch in front of each phase otherwise)
  __bf16* h0g = (__bf16*)p.sc.h0;
  __bf16* h1g = (__bf16*)p.sc.h1;
  float* headsg = p.sc.heads;
  const float invB = p.inv_batch;
  const float ls_min = p.hy.log_std_min, ls_max = p.hy.log_std_max, drop_scale = p.drop_scale;
  const unsigned* drop_bits = p.drop_bits;
  const float* log_std = p.log_std;
  float* pi_t = a.pi_t; float* pi_g = a.pi_g; float* pi_l = a.pi_l;
  __bf16* xbf = a.xbf;
  PIN_P(np.w0); PIN_P(np.b0); PIN_P(np.w1); PIN_P(np.b1); PIN_P(np.w2); PIN_P(np.b2);
  PIN_S(k0); PIN_S(D); PIN_S(xoff); PIN_S(slot); PIN_S(ld); PIN_S(B); PIN_S(MB); PIN_S(S); PIN_S(A); PIN_S(n_rt); PIN_S(nbi);
  PIN_P(xb); PIN_P(h0g); PIN_P(h1g); PIN_P(headsg); PIN_P(drop_bits); PIN_P(log_std); PIN_P(pi_t); PIN_P(pi_g); PIN_P(pi_l); PIN_P(xbf);
  PIN_S(invB); PIN_S(ls_min); PIN_S(ls_max); PIN_S(drop_scale);

  // ---- the 32 packed rows of a tile: thread (row tid >> 3, float4 (tid & 7) + 8 q of the instance's input columns)
  const int xr = tid >> 3, xc = tid & 7;
 f32x4 xv[NKB];
  unsigned xsh[NKB];      // (the last floats of the batch are read from its last 16 bytes: the wanted elements sit xsh places up)
  int x_row = 0;          // the batch row this thread's X values belong to (x_store)
  auto x_issue = [&](int rt) __attribute__((always_inline)) {
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
      xsh[q] = want - idx;
      xv[q] = *(const f32x4u*)(xb + idx);      // (ISSUE only: the shift is applied in x_store — used here, every load was waited for on the spot)
    }
  };
  auto x_store = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < NKB; ++q) {
      if (q < nkb) {
        const int c = 4 * (xc + 8 * q);
        const f32x4 v = xv[q];
        const unsigned sh = xsh[q];
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float a1 = (j + 1 < 4) ? v[(j + 1) & 3] : 0.f, a2 = (j + 2 < 4) ? v[(j + 2) & 3] : 0.f, a3 = (j + 3 < 4) ? v[(j + 3) & 3] : 0.f;
          const float xj = (sh == 0u) ? v[j] : ((sh == 1u) ? a1 : ((sh == 2u) ? a2 : a3));
          o[j] = (__bf16)((c + j < k0) ? xj : 0.f);     // columns >= k0: other fields of the row
        }
        *(bf16x4*)(Xb + xr * XLD + c) = o;
        xo[q] = o;
      }
    }
  };
  int rt = ib;
  STAMP_BASE(p, 0);
  STAMP(p, 0);
  x_issue(rt);

  // ---- the wave's operands for the whole block, from the net's operand image (fragment-major bf16: a fragment load is
  // 1 KB of consecutive memory).  Layer 0: A = W0 rows of units 64 w + 16 ct + l15, k = 32 kb + 8 g + e (zero beyond k0;
  // k-blocks beyond the instance's own are loaded from its last one — never used: no branch around a load)
  constexpr unsigned IMG_SLOT = 0x3215400u;       // instance -> image slot: V(s') V(s) Qt1 Qt2 Q1 Q2 pi -> 0 0 4 5 1 2 3
  const __bf16* img = a.wimg + (size_t)((IMG_SLOT >> (4 * inst)) & 7u) * IMG_STRIDE;
  bf16x8 w0f[4][NKB];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
      w0f[ct][kb] = *(const bf16x8*)(img + (unsigned)(IMG_W0_OFF + (((wv * 4 + CP(ct)) * nkb + min(kb, nkb - 1)) * 64 + lane) * 8));
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
  const float* lsp = (is_pi && gauss) ? log_std : np.b2;      // (any valid address when unused)
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int d = min(xc + 8 * c, D - 1);
    b2v[c] = np.b2[d];
    lsr[c] = lsp[d];
  }
  __builtin_amdgcn_sched_barrier(0);      // (the scheduler must not move the W1 stream in front of the small loads above)
  // (requested last: the biggest and the last needed)
  // layer 1: A = W1 rows (bf16 shadow) of the same 64 units, all 256 k: 32 fragments of 8 contiguous k
  bf16x8 w1f[4][8];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int kb = 0; kb < 8; ++kb)
      w1f[ct][kb] = *(const bf16x8*)(img + (unsigned)((((wv * 4 + CP(ct)) * 8 + KP(kb)) * 64 + lane) * 8));
  bf16x8 w2f[2][2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2) w2f[nt][kb2] = pack8(w2lo[nt][kb2], w2hi[nt][kb2]);
  float ivar[4], lsc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    // (computed by every instance, from whatever lsp addressed: with the value used only under `is_pi && gauss` hipcc sinks
    //  the LOAD into that branch — behind the W1 stream — and waits there for everything in flight)
    lsc[c] = fminf(fmaxf(lsr[c], ls_min), ls_max);
    const float sig = expf(lsc[c]);
    ivar[c] = 1.f / (sig * sig);
  }

  STAMP(p, 1);
  // (At the head of this loop, whose body re-issues the X prefetch, the compiler assumes those loads are nearly the last in
  //  flight: the wait in front of x_store() is vmcnt(8), i.e. on the way in from the prologue the first tile's layer 0 starts
  //  behind the whole W1 image.  Peeling the first tile — exact counts there — was measured: forward 9.36 -> 9.45 us at 1 024
  //  rows, 37.8 -> 38.8 at 8 192: the second copy of the body costs more than the overlap gives, profiles/r04_ab_lb_peel.txt.)
  for (; rt < rt_hi; rt += nbi) {
    const int row0 = rt * RT_ROWS;
    const bool first = (rt == ib + LB_STAMP_TILE * nbi);      // (the tile whose phases the diagnostic build stamps)
    // policy: this tile's actions and dropout keep-bits (in flight under layer 0)
    // (issued by every instance, from a harmless address where unused: a branch around a load would make the compiler
    //  wait for every load in flight — in the first tile that is the whole W1 stream)
    float pac[4];
    unsigned dm0[2][2], dm1[2][2];
    {
      const unsigned prow = (unsigned)min(row0 + xr, B - 1);
#pragma unroll
      for (int c = 0; c < 4; ++c) pac[c] = xb[prow * (unsigned)ld + (unsigned)(S + min(xc + 8 * c, A - 1))];
      const unsigned* dbits = drop ? drop_bits : (const unsigned*)xb;
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
    x_row = row0 + xr;
    x_store();
    __syncthreads();
    if (first) STAMP(p, 2);
    if (rt + nbi < rt_hi) x_issue(rt + nbi);       // the next tile's rows, under this tile's arithmetic

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
            for (int reg = 0; reg < 4; ++reg) h[reg] = ((bits >> reg) & 1u) ? h[reg] * drop_scale : 0.f;
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
            for (int reg = 0; reg < 4; ++reg) h1[r2][ct][reg] = ((bits >> reg) & 1u) ? h1[r2][ct][reg] * drop_scale : 0.f;
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
          pi_t[(unsigned)(row * 32 + xc + 8 * c)] = tv[c];
          pi_g[(unsigned)(row * 32 + xc + 8 * c)] = gv[c];
        }
        if (xc == 0) pi_l[row] = lsum;
      }
    }
    // the Q1 instance reads [s | a], i.e. every net's layer-0 input: its bf16 rows are the operand of dW0 = dH0^T . X
    // (stored last in the tile: a store in front of the barriers sits in front of every later load in the in-order vmcnt)
    if (inst == 4 && x_row < B) {
#pragma unroll
      for (int q = 0; q < NKB; ++q)
        if (q < nkb) *(bf16x4*)(xbf + (unsigned)(x_row * LB_XLD + 4 * (xc + 8 * q))) = xo[q];
    }
    if (first) STAMP(p, 8);
  }
  STAMP(p, 9);
  if constexpr (SPREAD) {
    if (idle_block) {
      __syncthreads();      // (every wave has left the last tile's LDS reads: H0b is the transpose's staging area)
      iql_w1t_build(p, a.w1t, H0b, ib_idle, nbi);
      if (p.g_work) idle_block_work(p.g_work, ib_idle, nbi);
    }
  }
  RT_STAMP(p, 14, rt_entry_);
  RT_STAMP(p, 15, iql_realtime());
#undef CP
#undef KP
}

// ---------------------------------------------------------------------------
// Backward, rows.  grid = 8 x nbb / 2; blockIdx & 7 = x: net = x & 3, block lb = 2 (blockIdx >> 3) + (x >> 2) on XCD n + 4 (lb & 1)
// — the XCD whose forward blocks wrote the H0 / H1 rows of tiles of that parity.
// (Load shapes: the CU's vector-memory pipe handles about one (16-lane group, cache line) pair per cycle, so the 16 lanes
//  of a group read one contiguous run wherever possible; tiles needed in an MFMA accumulator layout — the H1 mask — are
//  fetched as flat copies and redistributed through LDS.  Nothing is loaded behind a branch the compiler would wait at.)
struct LbTileIn {         // what a row block loads per 32-row tile
  RowIn in;               // scalar nets, wave 0: the row's loss inputs
  f32x4 ph[3], t4, g4;    // policy: V / Qt1 / Qt2 partials of row tid >> 3, T and G of dims 4 (tid & 7) ..
  float lrow;             // policy: L of row tid >> 3
  bf16x8 h1q[4];          // the H1 tile [32][256] as a flat copy: 16 bytes tid + 256 q (-> LDS -> dH1's accumulator layout)
  bf16x4 hm[2][4];        // H0 in dH0's accumulator layout
};
// CS ("column split", batches of at most 32 row tiles = 1 024 rows, where a block per tile leaves half the chip idle and
// a block is mostly the prologue that pulls the net's 128 KB W1 image through the CU's one vector-memory pipe): TWO blocks per
// row tile, on the XCD of the tile's parity.  Both build the whole dH1 tile (all 256 j are the contraction index); block ch
// then owns the dH0 columns [128 ch, 128 ch + 128) — half the W1 image, half the MFMAs, half the copy-out — wave w the combs
// tb0, tb0 + 1 of 64-column slab 2 ch + (w >> 1).  Everything that belongs to the tile as a whole (dY, the loss sums, db1 /
// db2 sums) is block 0's, the column sums db1 / dW2 are block 1's; the dH1 copy-out is shared by rows.  The two blocks share ONE slab of sums: block 1 writes
// its half of db0 into it and nothing else (64 slabs instead of 32 made the GEMM launch's reduction job 0.7 us longer).
template <bool CS>
__global__ __launch_bounds__(256) void iql_bwd_rows_kernel(StepParams p, LbArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  RT_ENTRY();
  const int bid = blockIdx.x;
  const int x = bid & 7;
  const int net = x & 3;
  const int lbq = bid >> 3, hx = x >> 2;
  const int ch = CS ? (lbq & 1) : 0;                        // the column half
  const int lb = (CS ? (lbq >> 1) : lbq) * 2 + hx;          // the slab of sums / loss-sum entry (CS: shared by the tile's two blocks)
  if (lb >= a.nbb) return;
  const int rt0 = lb;                                       // the block's first row tile ...
  const int rts = a.nbb;                                    // ... and the stride to its next one
  constexpr int NTB = CS ? 2 : 4;                           // column combs (n = 4 l15 + tb of a 64-column slab) per wave
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int wq = CS ? (2 * ch + (wave >> 1)) : wave;        // the 64-column slab of W1 / dH0 this wave works on
  const int tb0 = CS ? 2 * (wave & 1) : 0;                  // ... and its first column comb
  const int B = p.rows, MB = p.sc.max_batch;
  const NetPtrs np = p.net[net];
  const NetGrad go = p.go[net];
  const int D = np.d;
  const bool is_pi = (net == IQLHIP_NET_PI);
  const bool gauss = (p.policy == IQLHIP_POLICY_GAUSSIAN);
  const float* w2 = np.w2;
  const __bf16* H1g = (const __bf16*)p.sc.h1 + net * MB * HID;
  const __bf16* H0g = (const __bf16*)p.sc.h0 + net * MB * HID;
  __bf16* dH1g = a.dh1g + net * MB * HID;
  __bf16* dH0g = a.dh0g + net * MB * HID;
  const float* heads = p.sc.heads;
  const float dscale = (is_pi && p.drop_bits != nullptr) ? p.drop_scale : 1.f;
  const float invB = p.inv_batch;
  // every kernel-argument word the block uses, fetched in one batch now (cf. iql_fwd_lb_kernel)
  const float* pi_t = a.pi_t; const float* pi_g = a.pi_g; const float* pi_l = a.pi_l;
  __bf16* dyg = a.dyg;
  float* slabX = a.slab_x + (long long)lb * p.n_params;      // this block's slab of sums (arena layout)
  const long long go_b0 = a.go_b0[net];
  const float* log_std = p.log_std;
  const __bf16* w1t = a.w1t + net * 65536;
  const float ls_min = p.hy.log_std_min, ls_max = p.hy.log_std_max, beta = p.hy.beta, adv_max = p.hy.exp_adv_max;
  float* loss_parts = p.sc.loss_parts;
  const int n_rt = a.n_rt;
  PIN_P(pi_t); PIN_P(pi_g); PIN_P(pi_l); PIN_P(dyg); PIN_P(slabX); PIN_P(log_std); PIN_P(loss_parts); PIN_P(w2); PIN_P(w1t);
  PIN_P(H1g); PIN_P(H0g); PIN_P(dH1g); PIN_P(dH0g); PIN_P(heads); PIN_P(p.xb);
  PIN_S(go_b0); PIN_S(go.b1); PIN_S(go.w2); PIN_S(go.b2); PIN_S(go.log_std); PIN_S(ls_min); PIN_S(ls_max); PIN_S(beta); PIN_S(adv_max);
  PIN_S(dscale); PIN_S(invB); PIN_S(n_rt); PIN_S(rts); PIN_S(B); PIN_S(D); PIN_S(p.ld); PIN_S(p.S); PIN_S(p.A);
  PIN_S(p.hy.iql_tau); PIN_S(p.hy.discount);
#define LBROW(r) min((r), B - 1)
  STAMP_BASE(p, 2048 * 16);
  STAMP(p, 0);
  __bf16* dH1b = (__bf16*)smem;                        // [32][H0B_LD]  dH1 tile (dH0's A operand; copied out to dH1g)
  __bf16* H1t = dH1b + 32 * H0B_LD;                    // [32][H0B_LD]  H1 tile (mask of dH1; the scalar nets' dW2 operand)
  __bf16* dH0b = H1t + 32 * H0B_LD;                    // [32][H0B_LD]  dH0 tile on its way to dH0g
  __bf16* dYb = dH0b + 32 * H0B_LD;                    // [32][LB_DYLD] dY of the tile (policy)
  float* dys = (float*)(dYb + 32 * LB_DYLD);           // [32] dy of the tile (scalar nets)
  const int xr = tid >> 3, xc = tid & 7;
  auto issue = [&](LbTileIn& s, int rt) __attribute__((always_inline)) {
    const int row0 = rt * RT_ROWS;
    if (wave == 0) row_issue(p, LBROW(row0 + (tid & 31)), s.in);
    const unsigned prow = (unsigned)LBROW(row0 + xr);
    const unsigned oh = prow * (unsigned)HEAD_LD;
    s.ph[0] = *(const f32x4*)(heads + (oh + 4u)); s.ph[1] = *(const f32x4*)(heads + (oh + 8u)); s.ph[2] = *(const f32x4*)(heads + (oh + 12u));
    s.t4 = *(const f32x4*)(pi_t + (prow * 32u + 4u * (unsigned)xc));
    s.g4 = *(const f32x4*)(pi_g + (prow * 32u + 4u * (unsigned)xc));
    s.lrow = pi_l[prow];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int f = tid + 256 * q;
      s.h1q[q] = *(const bf16x8*)(H1g + ((unsigned)LBROW(row0 + (f >> 5)) * (unsigned)HID + (unsigned)(8 * (f & 31))));
    }
#pragma unroll
    for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg)
        s.hm[r2][reg] = *(const bf16x4*)(H0g + ((unsigned)LBROW(row0 + 16 * r2 + 4 * g + reg) * (unsigned)HID + (unsigned)(64 * wq + 4 * l15)));
  };
  LbTileIn T;
#pragma unroll
  for (int i = 0; i < 6; ++i) T.in.h[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  T.in.r = 0.f; T.in.d = 0.f;
  issue(T, rt0);
  // (the raw log_std of dim tid, for the block's last lines: loaded HERE, with the first batch — left at its use the
  //  compiler hoists the load in front of the tile loop and waits for everything in flight, the whole W1 stream, there)
  const float lsr_e = ((is_pi && gauss) ? log_std : w2)[min(tid, D - 1)];
  // (hipcc's scheduler moved the 44 weight loads below IN FRONT of these — vmcnt retires in order, so the first tile's loss
  //  arithmetic then waited for the whole 128 KB weight stream; nothing crosses this line)
  __builtin_amdgcn_sched_barrier(0);
  // ---- the wave's operands for the whole block (requested BEHIND the first tile's inputs: the loss arithmetic and the
  // dH1 tile run under this stream)
  // W2: scalar nets — the lane's 16 columns j = 64 w + 16 jt + 4 g + reg; policy — A operand of dH1^T = W2^T . dY^T
  // (m = j = 64 w + 16 jt + l15, k = dim 8 g + e)
  f32x4 w2q[4];
  bf16x8 w2A[4];      // (the image of the policy's W2, iql_w1t_build; loaded — and ignored — by the scalar nets' blocks too)
#pragma unroll
  for (int jt = 0; jt < 4; ++jt) {
    w2q[jt] = *(const f32x4*)(w2 + (unsigned)(64 * wave + 16 * jt + 4 * g));
    w2A[jt] = *(const bf16x8*)(a.w1t + (unsigned)(4 * 65536 + ((wave * 4 + jt) * 64 + lane) * 8));
  }
  // W1 as the B operand of dH0 = dH1 . W1 (k = j = 32 kb + 8 g + e, n = column 64 w + 4 l15 + tb), from the transposed
  // image the forward's idle blocks built (iql_w1t_build): a fragment load is 1 KB of consecutive memory
  bf16x8 bwf[NTB][8];
#pragma unroll
  for (int tb = 0; tb < NTB; ++tb)
#pragma unroll
    for (int kb = 0; kb < 8; ++kb)
      bwf[tb][kb] = *(const bf16x8*)(w1t + (unsigned)((((wq * 4 + tb0 + tb) * 8 + kb) * 64 + lane) * 8));
  // the block's COLUMN sums over its rows — db1 = sum_r dH1, the scalar nets' dW2 = sum_r dy H1, db0 = sum_r dH0 — on the
  // matrix cores: [1 .. 1] (or [dy]) . tile, the tile read transposed from the LDS image it sits in anyway
  // (ds_read_b64_tr_b16).  Every accumulator row then holds the same column sums: register 0 of lane l15 (any g) is column
  // 64 w + 16 t + l15's — no cross-lane reduction at the end, no per-tile vector-ALU adds (as per-lane partial sums with a
  // DPP / LDS reduction this cost ~4 k cycles at the end of every block).
  f32x4 sum1[4], sum2[4], sum0[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) { sum1[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; sum2[t] = sum1[t]; sum0[t] = sum1[t]; }
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.f;
  const int q4 = l15 >> 2, p4 = l15 & 3;
  // B operand of a column-sum product: rows 8 g .. 8 g + 7, column c0 + l15 of a [32][H0B_LD] bf16 tile
#define COLSUM_B(tile_, c0_) cat8(lds_tr4((tile_) + (8 * g + q4) * H0B_LD + (c0_) + 4 * p4), lds_tr4((tile_) + (8 * g + 4 + q4) * H0B_LD + (c0_) + 4 * p4))
  f32x4 pb2 = (f32x4){0.f, 0.f, 0.f, 0.f}, pls = pb2;      // policy: dims 4 xc .. over rows xr of the block's tiles
  float accA = 0.f, accB = 0.f, accb2 = 0.f;      // loss sums (scalar nets: threads < 32; policy: threads xc == 0), scalar db2
  // ---- the block's sums -> its slab (arena layout).  Run in FRONT of the last tile's dH0 copy-out (LDS region of its own):
  // behind it, the first registers it touches were the copy-out stores' sources, and the block waited ~4 k cycles for
  // those stores to complete before it even started — and then once more for its own stores at the end of the kernel.
  // (the loss sums by value: captured by reference in this closure AND in do_tile's, which calls it, they landed in scratch)
  auto block_sums = [&](const float accA, const float accB, const float accb2) __attribute__((always_inline)) {
  float* pip = (float*)(dys + 32);     // [2][4 waves][32] policy db2 / dlog_std partials
  float* rsm = pip + 2 * 4 * 32;       // [16]
  if (CS && ch == 1) {      // (block-uniform) the second block of a tile: db1 / the scalar dW2 (the column sums over the dH1 and
                            // H1 tiles, which it holds like block 0) and its 128 columns of db0, into the slab block 0 fills
    if (g == 0) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int col = 64 * wave + 16 * t + l15;
        slabX[go.b1 + col] = sum1[t][0];
        if (!is_pi) slabX[go.w2 + col] = sum2[t][0];
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) slabX[go_b0 + 128 + 32 * wave + 16 * t + l15] = sum0[t][0];
    }
    return;
  }
  if (g == 0) {
    if (!CS) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int col = 64 * wave + 16 * t + l15;
        slabX[go.b1 + col] = sum1[t][0];
        if (!is_pi) slabX[go.w2 + col] = sum2[t][0];
        slabX[go_b0 + col] = sum0[t][0];
      }
    }
    if (CS) {      // db0 of block 0's 128 columns (the other 128, db1 and the scalar dW2: block 1, above)
#pragma unroll
      for (int t = 0; t < 2; ++t) slabX[go_b0 + 32 * wave + 16 * t + l15] = sum0[t][0];
    }
  }
  if (is_pi) {
    // dims 4 xc .. of the thread, over the wave's 8 row groups (lanes 8 apart), then over the four waves through LDS
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int o = 8; o < 64; o <<= 1) { pb2[j] += __shfl_xor(pb2[j], o); pls[j] += __shfl_xor(pls[j], o); }
    }
    if (lane < 8) {
      *(f32x4*)(pip + wave * 32 + 4 * lane) = pb2;
      *(f32x4*)(pip + 128 + wave * 32 + 4 * lane) = pls;
    }
    const float sA = block_sum_256(accA, rsm);      // (its barriers also publish pip)
    if (tid == 0) loss_parts[3 * 64 + lb] = sA;
    if (tid < D) {
      const float sb = (pip[tid] + pip[32 + tid]) + (pip[64 + tid] + pip[96 + tid]);
      const float sl = (pip[128 + tid] + pip[160 + tid]) + (pip[192 + tid] + pip[224 + tid]);
      slabX[go.b2 + tid] = sb;
      if (gauss) {
        const bool inside = (lsr_e >= ls_min) && (lsr_e <= ls_max);
        slabX[go.log_std + tid] = inside ? sl * invB : 0.f;
      }
    }
  } else if (wave == 0) {
    float sA = accA, sB = accB, sb = accb2;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sA += __shfl_xor(sA, o); sB += __shfl_xor(sB, o); sb += __shfl_xor(sb, o); }
    if (lane == 0) {
      slabX[go.b2] = sb;
      if (net == IQLHIP_NET_V) loss_parts[0 * 64 + lb] = sA;
      if (net == IQLHIP_NET_Q1) { loss_parts[1 * 64 + lb] = sA; loss_parts[2 * 64 + lb] = sB; }
    }
  }
  };
  STAMP(p, 1);

  // One row tile.  CS: a block has exactly one, so there is no loop — and without the back edge the compiler counts the
  // loads in flight exactly (at the head of a loop whose body re-issues the tile inputs it assumes they are the last loads
  // issued, and the first tile's arithmetic starts behind the whole W1 stream): row kernel 7.74 -> 7.37 us at 1 024 rows.
  // (Peeling the first tile of the general loop the same way: 9.57 -> 9.37 us at 2 048 rows but 25.15 -> 25.47 at 8 192 — not
  //  done, profiles/r04_ab_lb_peel.txt.)
  auto do_tile = [&](const int rt) __attribute__((always_inline)) {
    // (no implicit fused multiply-adds in the tile's arithmetic: left to -ffp-contract the two instantiations contracted
    //  `r + c * nv` and `acc += w * l` differently, and one bf16 dH1 value in ~10^5 came out one ulp apart — the column-split
    //  form must reproduce the one-block form bit for bit, tests/test_hip_lb.py)
#pragma clang fp contract(off)
    const int row0 = rt * RT_ROWS;
    const bool first = (rt == rt0);
    const bool stamped = (rt == rt0 + LB_STAMP_TILE * rts);      // (the tile whose phases the diagnostic build stamps)
    // the H1 tile -> LDS
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int f = tid + 256 * q;
      *(bf16x8*)(H1t + (f >> 5) * H0B_LD + 8 * (f & 31)) = T.h1q[q];
    }
    // ---- dy of the scalar heads / dY = w T of the policy; the per-row sums
    // (this tile's contributions to the block's loss sums are collected in locals and added behind the two net kinds'
    //  branches, in ONE place: with an `acc += ...` at the end of each branch the optimiser sank the updates into the join
    //  behind a pointer phi — while the accumulators were still closure captures — and they stayed in scratch memory)
    float tA = 0.f, tB = 0.f, tb2 = 0.f;
    if (!is_pi) {
      if (tid < RT_ROWS) {
        // (row_finish's arithmetic restated by value: inside this closure its three by-address results landed in scratch)
        float v = 0.f, la_ = 0.f, lb_ = 0.f;
        if (row0 + tid < B) {
          if (net == IQLHIP_NET_V) {
            const float tq = fminf(sum4(T.in.h[2]), sum4(T.in.h[3]));
            const float u = tq - sum4(T.in.h[1]);
            const float wgt = fabsf(p.hy.iql_tau - ((u < 0.f) ? 1.f : 0.f));
            la_ = wgt * u * u;
            v = (-2.f * wgt * u) * invB;
          } else {
            const float nv = sum4(T.in.h[0]);
            const float y = T.in.r + ((1.f - T.in.d) * p.hy.discount) * nv;
            const float e1 = sum4(T.in.h[4]) - y;
            const float e2 = sum4(T.in.h[5]) - y;
            la_ = e1 * e1;
            lb_ = e2 * e2;
            v = ((net == IQLHIP_NET_Q1) ? e1 : e2) * invB;
          }
        }
        dys[tid] = v;
        tA = la_; tB = lb_; tb2 = v;
      }
    } else {
      float w = 0.f;
      if (row0 + xr < B) {
        const float tqv = fminf(sum4(T.ph[1]), sum4(T.ph[2]));
        const float u = tqv - sum4(T.ph[0]);
        w = fminf(expf(beta * u), adv_max);
      }
      const f32x4 dy4 = T.t4 * w;
      const bf16x4 dyb = cvt4(dy4);
      *(bf16x4*)(dYb + xr * LB_DYLD + 4 * xc) = dyb;
      if (row0 + xr < B && ch == 0) *(bf16x4*)(dyg + (unsigned)((row0 + xr) * 32 + 4 * xc)) = dyb;
      const f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f};
      pb2 += (ch == 0) ? dy4 : z4;
      pls += (ch == 0) ? T.g4 * w : z4;
      tA = (xc == 0) ? w * T.lrow : 0.f;
    }
    accA += (ch == 0) ? tA : 0.f;
    accB += (ch == 0) ? tB : 0.f;
    accb2 += (ch == 0) ? tb2 : 0.f;
    if (stamped) STAMP(p, 2);
    __syncthreads();
    if (stamped) STAMP(p, 3);
    // ---- dH1 tile [32][256]: (dY . W2) masked by H1 > 0; lane: rows 16 r2 + l15, columns 64 w + 16 jt + 4 g ..
    // (all eight mask reads first, the two net kinds in loops of their own: with the kind's branch inside the (r2, jt) loop
    //  every iteration was an LDS round trip of its own — read, wait, multiply, write — 2.7 k cycles for eight of them)
    {
      bf16x4 hmk[2][4];
#pragma unroll
      for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) hmk[r2][jt] = *(const bf16x4*)(H1t + (16 * r2 + l15) * H0B_LD + 64 * wave + 16 * jt + 4 * g);
      f32x4 pre[2][4];
      if (is_pi) {
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) {
          const bf16x8 Bd = *(const bf16x8*)(dYb + (16 * r2 + l15) * LB_DYLD + 8 * g);
#pragma unroll
          for (int jt = 0; jt < 4; ++jt) pre[r2][jt] = MFMA_BF16(w2A[jt], Bd, ((f32x4){0.f, 0.f, 0.f, 0.f}));
        }
      } else {
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) {
          const float dyr = dys[16 * r2 + l15];
#pragma unroll
          for (int jt = 0; jt < 4; ++jt) pre[r2][jt] = w2q[jt] * dyr;
        }
      }
#pragma unroll
      for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
          f32x4 o;
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const float h = (float)hmk[r2][jt][reg];
            o[reg] = (h > 0.f) ? pre[r2][jt][reg] * dscale : 0.f;
          }
          *(bf16x4*)(dH1b + (16 * r2 + l15) * H0B_LD + 64 * wave + 16 * jt + 4 * g) = cvt4(o);
        }
    }
    __syncthreads();
    if (stamped) STAMP(p, 4);
    // column sums of the dH1 tile; scalar nets: of dy . H1 (A = the tile's 32 dy, the same in every row m)
    {
      bf16x8 Ady = ones;
      if (!is_pi) {
        const f32x4 ya = *(const f32x4*)(dys + 8 * g), yb = *(const f32x4*)(dys + 8 * g + 4);
        Ady = pack8(ya, yb);
      }
      if (ch == (CS ? 1 : 0)) {      // (CS: block 1's — block 0 carries the loss sums and the policy's db2 / dlog_std)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          sum1[t] = MFMA_BF16(ones, COLSUM_B(dH1b, 64 * wave + 16 * t), sum1[t]);
          if (!is_pi) sum2[t] = MFMA_BF16(Ady, COLSUM_B(H1t, 64 * wave + 16 * t), sum2[t]);
        }
      }
    }
    // ---- dH0 = dH1 . W1: the wave's 64 columns over all 256 k
    {
      bf16x8 Ad[2][8];
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) {
        Ad[0][kb] = *(const bf16x8*)(dH1b + l15 * H0B_LD + 32 * kb + 8 * g);
        Ad[1][kb] = *(const bf16x8*)(dH1b + (16 + l15) * H0B_LD + 32 * kb + 8 * g);
      }
      f32x4 acc[2][NTB];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NTB; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < 8; ++kb)
#pragma unroll
        for (int tb = 0; tb < NTB; ++tb) {
          acc[0][tb] = MFMA_BF16(Ad[0][kb], bwf[tb][kb], acc[0][tb]);
          acc[1][tb] = MFMA_BF16(Ad[1][kb], bwf[tb][kb], acc[1][tb]);
        }
      if (stamped) STAMP(p, 5);
      // masked -> the row-major dH0 tile: a lane's column combs are consecutive columns of row 16 r2 + 4 g + reg
#pragma unroll
      for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          if constexpr (CS) {
            const bf16x4 hmk = T.hm[r2][reg];
            bf16x2 o;
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) {
              const float hv = (tb0 == 0) ? (float)hmk[tb] : (float)hmk[2 + tb];
              o[tb] = (__bf16)((hv > 0.f) ? acc[r2][tb][reg] * dscale : 0.f);      // rows >= B carry 0
            }
            *(bf16x2*)(dH0b + (16 * r2 + 4 * g + reg) * H0B_LD + 64 * wq + 4 * l15 + tb0) = o;
          } else {
            bf16x4 o;
#pragma unroll
            for (int tb = 0; tb < 4; ++tb) {
              const float v = ((float)T.hm[r2][reg][tb] > 0.f) ? acc[r2][tb][reg] * dscale : 0.f;      // rows >= B carry 0
              o[tb] = (__bf16)v;
            }
            *(bf16x4*)(dH0b + (16 * r2 + 4 * g + reg) * H0B_LD + 64 * wave + 4 * l15) = o;
          }
        }
    }
    // the next tile's inputs: everything of this tile's has been consumed
    if (!CS && rt + rts < n_rt) issue(T, rt + rts);      // (CS: one tile per block, nbb >= n_rt)
    __syncthreads();
    if (stamped) STAMP(p, 6);
#pragma unroll
    for (int t = 0; t < NTB; ++t)      // column sums of the dH0 tile (CS: of the block's half, two 16-column tiles per wave)
      sum0[t] = MFMA_BF16(ones, COLSUM_B(dH0b, CS ? (128 * ch + 32 * wave + 16 * t) : (64 * wave + 16 * t)), sum0[t]);
    if (rt + rts >= n_rt) block_sums(accA, accB, accb2);      // (the block's last tile)
    // dH1 / dH0 rows -> memory (operands of dW1 = dH1^T . H0, dW0 = dH0^T . X): the tile's last instructions — a register
    // that was a store's source is not reused before the store has completed (~4 k cycles)
#pragma unroll
    for (int q = 0; q < (CS ? 2 : 4); ++q) {      // (CS: rows 16 ch .. 16 ch + 15 of the dH1 tile both blocks hold)
      const int f = tid + 256 * q;
      const int rl = (CS ? 16 * ch : 0) + (f >> 5), col = 8 * (f & 31);
      if (row0 + rl < B) *(bf16x8*)(dH1g + (unsigned)((row0 + rl) * HID + col)) = *(const bf16x8*)(dH1b + rl * H0B_LD + col);
    }
#pragma unroll
    for (int q = 0; q < (CS ? 2 : 4); ++q) {      // (CS: the block's 128 columns of all 32 rows)
      const int f = tid + 256 * q;
      const int rl = CS ? (f >> 4) : (f >> 5), col = CS ? (128 * ch + 8 * (f & 15)) : (8 * (f & 31));
      if (row0 + rl < B) *(bf16x8*)(dH0g + (unsigned)((row0 + rl) * HID + col)) = *(const bf16x8*)(dH0b + rl * H0B_LD + col);
    }
    if (stamped) STAMP(p, 7);
    // (no barrier here: the next tile's first LDS writes — H1t, dYb / dys — were last read before the barrier above, its
    //  dH1b writes come behind its own first barrier, its dH0b writes behind its second)
  };
  if constexpr (CS) {
    if (rt0 < n_rt) do_tile(rt0);
  } else {
    for (int rt = rt0; rt < n_rt; rt += rts) do_tile(rt);
  }
  STAMP(p, 8);
  if (rt0 >= n_rt) block_sums(accA, accB, accb2);      // (a block without row tiles still owns a slab: zeros)
  STAMP(p, 9);
  RT_STAMP(p, 14, rt_entry_);
  RT_STAMP(p, 15, iql_realtime());
#undef LBROW
#undef COLSUM_B
}

// ---------------------------------------------------------------------------
// Backward, row contractions: C[m][n] = sum_r P[r][m0 + m] Q[r][n0 + n] over the rows of one chunk group, for
//   job 0..15   dW1[j][i]       P = dH1 (64 columns j0 ..), Q = H0 (64 columns i0 ..)
//   job 16..23  dW0[i][kc]      P = dH0 (64 columns i0 ..), Q = the batch's [s | a] columns as bf16 (64 columns kc0 ..)
//   job 24..27  policy dW2[d][j]   P = dY (32 columns), Q = H1 (64 columns j0 ..)
// All operands are bf16 row-major matrices over the batch rows, so every job is the same code: only base pointers, row
// strides and the output map differ (block-uniform scalars).
// grid = 8 x ceil(28 n_cg / 2): net = x & 3, local = 2 (blockIdx >> 3) + (x >> 2) = 28 cg + job; a dW1 tile's column-tile
// parity = (job & 1) = local & 1 = the XCD half whose update blocks read that stripe parity (iql_update_kernel).
// 64-row stages, double-buffered in LDS as row-major [64][LB_GLD] bf16 tiles; wave (wm, wn) owns a 32 x 32 quadrant =
// 2 x 2 MFMA tiles; both operands are read TRANSPOSED (ds_read_b64_tr_b16: per 16-lane group a 4-row x 16-column block,
// lane 4 q + p passes the address of (row q, columns 4 p ..) and receives column l15 of the 4 rows;
// tools/microbench/tr_read_check.hip).
#define LB_GLD 72
#define LB_NJOB 28          // per net and chunk group; job 28 exists once per net (chunk group 0 only): see below
__global__ __launch_bounds__(256) void iql_bwd_gemm_kernel(StepParams p, LbArgs a) {
  __shared__ __attribute__((aligned(16))) __bf16 Ps[2][64 * LB_GLD];
  __shared__ __attribute__((aligned(16))) __bf16 Qs[2][64 * LB_GLD];
  RT_ENTRY();
  const int bid = blockIdx.x;
  const int x = bid & 7;
  const int net = x & 3;
  const int local_ = (bid >> 3) * 2 + (x >> 2);
  const int cg = local_ / LB_NJOB, job = local_ - cg * LB_NJOB;
  if (cg >= a.n_cg) {
    // ---- one extra block per net (local = 28 n_cg): the row blocks' slabs of plain row sums (b0, b1, the scalar nets' w2,
    // b2, log_std — ~1 300 floats per net in nbb slabs) summed into chunk-group slab 0, so that the update kernel reads
    // them like everything else instead of walking up to 64 slabs on its own critical path
    if (local_ != LB_NJOB * a.n_cg) return;
    const long long x0 = a.go_b0[net];                                   // [b0 | b1 | w2 | b2 | log_std] is one run of the arena
    const long long x1 = (net < 3) ? a.go_w1n[net + 1] : p.n_params;     // ... up to the next net's segment
    const long long pw2 = p.go[net].w2, pb2 = p.go[net].b2;
    const bool is_pi_ = (net == IQLHIP_NET_PI);
    const int nbb = a.nbb;
    float* dst = p.sc.slab_a;
    for (long long e = x0 + 4 * (long long)threadIdx.x; e < x1; e += 4 * 256) {
      if (is_pi_ && e >= pw2 && e < pb2) continue;                        // (the policy's w2 is a row contraction: jobs 24..27)
      f32x4 sum = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int r0 = 0; r0 < nbb; r0 += 8) {
        f32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *(const f32x4*)(a.slab_x + (long long)min(r0 + j, nbb - 1) * p.n_params + e);
#pragma unroll
        for (int j = 0; j < 8; ++j) if (r0 + j < nbb) sum += v[j];
      }
      *(f32x4*)(dst + e) = sum;
    }
    // ... and the row blocks' loss sums (loss_parts[k][0 .. nbb)) into entry 0 of a second set of four rows, which is all
    // the update kernel then reads (its thread 0 otherwise walks nbb entries per loss on the kernel's critical path).
    // Wave 0 of the V block: loss 0; of the Q1 block: losses 1 and 2; of the policy block: loss 3.  Fixed order.
    if (threadIdx.x < 64) {
      float* lp = p.sc.loss_parts;
      const int lane = threadIdx.x;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool mine = (k == 0 && net == IQLHIP_NET_V) || ((k == 1 || k == 2) && net == IQLHIP_NET_Q1) || (k == 3 && net == IQLHIP_NET_PI);
        if (!mine) continue;
        float v = (lane < nbb) ? lp[k * 64 + lane] : 0.f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) lp[256 + k * 64] = v;
      }
    }
    return;
  }
  const bool is_pi = (net == IQLHIP_NET_PI);
  const int B = p.rows, MB = p.sc.max_batch;
  const int k0 = p.net[net].k0, D = p.net[net].d;
  const int kind = (job < 16) ? 0 : ((job < 24) ? 1 : 2);
  const int mt = (kind == 0) ? (job >> 2) : ((kind == 1) ? ((job - 16) >> 1) : 0);
  const int nt = (kind == 0) ? (job & 3) : ((kind == 1) ? ((job - 16) & 1) : (job - 24));
  if (kind == 2 && !is_pi) return;
  if (kind == 1 && 64 * nt >= k0) return;      // (no kc in this tile)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  // operands: base pointer (first column of the tile), row stride, real columns of the tile (beyond: zeros)
  const __bf16* Pg = (kind == 0) ? a.dh1g + net * MB * HID + 64 * mt : ((kind == 1) ? a.dh0g + net * MB * HID + 64 * mt : a.dyg);
  const int ldP = (kind == 2) ? 32 : HID;
  const int pcols = (kind == 2) ? 32 : 64;
  const __bf16* Qg = (kind == 0) ? (const __bf16*)p.sc.h0 + net * MB * HID + 64 * nt
                   : ((kind == 1) ? a.xbf + 64 * nt : (const __bf16*)p.sc.h1 + net * MB * HID + 64 * nt);
  const int ldQ = (kind == 1) ? LB_XLD : HID;
  // output map: C[m][n] -> slab[out0 + m * ldo + n] for m < m_lim, n < n_lim
  const long long out0 = (kind == 0) ? p.go[net].w1 + (long long)(64 * mt) * HID + 64 * nt
                       : ((kind == 1) ? a.go_w0[net] + (long long)(64 * mt) * k0 + 64 * nt : p.go[net].w2 + 64 * nt);
  const int ldo = (kind == 1) ? k0 : HID;
  const int m_lim = (kind == 2) ? D : 64;
  const int n_lim = (kind == 1) ? (k0 - 64 * nt) : 64;
  float* slab = p.sc.slab_a + (long long)cg * p.n_params + out0;
  PIN_P(Pg); PIN_P(Qg); PIN_P(slab); PIN_S(ldP); PIN_S(ldQ); PIN_S(pcols); PIN_S(ldo); PIN_S(m_lim); PIN_S(n_lim);
  const int r_begin = cg * a.cpb * CHUNK_ROWS;
  const int r_end = min(min((cg + 1) * a.cpb, a.n_chunk) * CHUNK_ROWS, (B + 63) & ~63);
  const int n_stage = (r_end - r_begin + 63) >> 6;
  // stage copy: thread -> rows (tid >> 3) + 32 i, 16-byte chunk tid & 7 of the tile's 128-byte rows (fully coalesced)
  const int sr = tid >> 3, sc8 = 8 * (tid & 7);
  const unsigned pc8 = (unsigned)min(sc8, pcols - 8);
  const bool p_live = sc8 < pcols;
  // Two register sets: the rows of stage s + 2 are requested while stage s is multiplied and stage s + 1 (requested one
  // iteration earlier) is parked in LDS — with one stage of look-ahead the ~700 cycles of a stage's arithmetic did not
  // cover the loads' latency (stamps: ~900 cycles of every stage waiting for them).
  bf16x8 plA[2], qlA[2], plB[2], qlB[2];
  STAMP_BASE(p, 3072 * 16);
  STAMP(p, 0);
#define GEMM_LOAD(pl_, ql_, s_)                                                                         \
  do {                                                                                                  \
    const int s__ = min((s_), n_stage - 1);      /* (beyond the last stage: re-read it, never used — no branch) */ \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                     \
      const unsigned row = (unsigned)min(r_begin + 64 * s__ + sr + 32 * i, B - 1);                      \
      pl_[i] = *(const bf16x8*)(Pg + (row * (unsigned)ldP + pc8));                                      \
      ql_[i] = *(const bf16x8*)(Qg + (row * (unsigned)ldQ + (unsigned)sc8));                            \
    }                                                                                                   \
  } while (0)
#define GEMM_STORE(pl_, ql_, s_, buf_)                                                                  \
  do {                                                                                                  \
    const bf16x8 z = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};                                                  \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                     \
      const bool live = (r_begin + 64 * (s_) + sr + 32 * i) < B;      /* rows beyond the batch: zeros */ \
      *(bf16x8*)(&Ps[buf_][(sr + 32 * i) * LB_GLD + sc8]) = (live && p_live) ? pl_[i] : z;              \
      *(bf16x8*)(&Qs[buf_][(sr + 32 * i) * LB_GLD + sc8]) = live ? ql_[i] : z;                          \
    }                                                                                                   \
  } while (0)
  // one stage from LDS buffer buf_: k-block kb (32 rows), column tile t (16 columns): rows 32 kb + 8 g + 4 h + q4, columns 16 t + 4 p4
#define GEMM_STAGE(buf_)                                                                                \
  do {                                                                                                  \
    _Pragma("unroll") for (int kb = 0; kb < 2; ++kb) {                                                  \
      bf16x8 Af[2], Bf[2];                                                                              \
      _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                   \
        const __bf16* pa = &Ps[buf_][(32 * kb + 8 * g + q4) * LB_GLD + 32 * wm + 16 * t + 4 * p4];      \
        const __bf16* pb = &Qs[buf_][(32 * kb + 8 * g + q4) * LB_GLD + 32 * wn + 16 * t + 4 * p4];      \
        Af[t] = cat8(lds_tr4(pa), lds_tr4(pa + 4 * LB_GLD));                                            \
        Bf[t] = cat8(lds_tr4(pb), lds_tr4(pb + 4 * LB_GLD));                                            \
      }                                                                                                 \
      _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                     \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] = MFMA_BF16(Af[i], Bf[j], acc[i][j]);   \
    }                                                                                                   \
  } while (0)
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int q4 = l15 >> 2, p4 = l15 & 3;
  GEMM_LOAD(plA, qlA, 0);
  GEMM_LOAD(plB, qlB, 1);
  GEMM_STORE(plA, qlA, 0, 0);
  __syncthreads();
  STAMP(p, 1);
  // (stages come in pairs: even stages live in LDS buffer 0 / register set A, odd ones in buffer 1 / set B.  The main loop
  //  runs while both look-ahead loads are inside the range — straight-line, no branch around a load; the last one to three
  //  stages follow without loads behind the range: the epilogue would wait for them)
  int s = 0;
  for (; s + 3 < n_stage; s += 2) {
    GEMM_LOAD(plA, qlA, s + 2);
    GEMM_STAGE(0);
    if (s == 0) STAMP(p, 4);
    GEMM_STORE(plB, qlB, s + 1, 1);
    if (s == 0) STAMP(p, 5);
    __syncthreads();
    if (s == 0) STAMP(p, 6);
    GEMM_LOAD(plB, qlB, s + 3);
    GEMM_STAGE(1);
    GEMM_STORE(plA, qlA, s + 2, 0);
    __syncthreads();
  }
  const int left = n_stage - s;      // 1, 2 or 3: stage s is in buffer 0, stage s + 1 (if any) in register set B
  if (left == 3) GEMM_LOAD(plA, qlA, s + 2);
  GEMM_STAGE(0);
  if (left >= 2) {
    GEMM_STORE(plB, qlB, s + 1, 1);
    __syncthreads();
    GEMM_STAGE(1);
    if (left == 3) {
      GEMM_STORE(plA, qlA, s + 2, 0);
      __syncthreads();
      GEMM_STAGE(0);
    }
  }
  STAMP(p, 2);
  // ---- the tile -> this chunk group's slab: D[m = 4 g + reg][n = l15] of tile (i, j) = C[32 wm + 16 i + 4 g + reg][32 wn + 16 j + l15]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int m = 32 * wm + 16 * i + 4 * g + reg, n = 32 * wn + 16 * j + l15;
        if (m < m_lim && n < n_lim) slab[m * ldo + n] = acc[i][j][reg];
      }
  STAMP(p, 3);
  RT_STAMP(p, 14, rt_entry_);
  RT_STAMP(p, 15, iql_realtime());
#undef GEMM_LOAD
#undef GEMM_STORE
#undef GEMM_STAGE
}
