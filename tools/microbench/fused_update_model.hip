// Model of "backward + optimizer update in one launch" against "two launches" on gfx950, to decide whether the step's
// third kernel boundary can be bought back with an in-kernel hand-off.  256 blocks = 4 nets x 64 (the backward's grid at
// 256 rows); a block "works" for a fixed time (scalar nets 6.0 us, the policy 7.0 us: the measured tails), writes 8 KB
// of gradient, then
//   two launches: kernel ends; an update kernel of 284 blocks reads gradient + m, v, p (+ target) and writes them back
//   fused:        the block's stores are written through (sc1), it signals its net's counter (non-returning atomic);
//                 scalar-net blocks wait for their net, update its 71 units, 71 of them then take the policy's units
//                 (state prefetched before the wait), policy blocks just leave
// Reported: us per iteration over 2 000 back-to-back iterations on one stream.
// Build: hipcc --offload-arch=gfx950 -O3 -o fused_update_model fused_update_model.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int UNITS = 71;            // 1 024-float units per net
constexpr int NETF = UNITS * 1024;   // floats per net
__device__ __forceinline__ unsigned long long memtime() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
__device__ __forceinline__ void work(unsigned long long cycles) {
  const unsigned long long t0 = memtime();
  while (memtime() - t0 < cycles) __builtin_amdgcn_s_sleep(1);
}
struct State { float *p, *m, *v, *t, *g; };
__device__ __forceinline__ void unit_math(f32x4& p, f32x4& m, f32x4& v, f32x4& t, const f32x4 g) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    m[k] = fmaf(0.1f, g[k] - m[k], m[k]);
    v[k] = fmaf(0.001f * g[k], g[k], v[k] * 0.999f);
    p[k] = fmaf(-3e-4f, m[k] / (sqrtf(v[k]) + 1e-8f), p[k]);
    t[k] = fmaf(0.005f, p[k], 0.995f * t[k]);
  }
}
// gradient: block b of net n writes floats [b * 1136, (b + 1) * 1136) of the net's 72 704 (64 x 1 136 = 72 704 = 71 x 1 024)
template <bool SC1>
__device__ __forceinline__ void write_grad(const State& s, int net, int b, int tid, float val) {
  float* g = s.g + (size_t)net * NETF + (size_t)b * 1136;
  for (int i = tid; i < 284; i += 256) {
    const f32x4 v = (f32x4){val, val, val, val};
    if (SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(g + 4 * i), "v"(v) : "memory");
    else *(f32x4*)(g + 4 * i) = v;
  }
}
__global__ __launch_bounds__(256) void k_work(State s, unsigned long long c_scalar, unsigned long long c_pi, float val) {
  const int net = blockIdx.x & 3, b = blockIdx.x >> 2;
  work(net == 3 ? c_pi : c_scalar);
  write_grad<false>(s, net, b, threadIdx.x, val);
}
__global__ __launch_bounds__(256) void k_update(State s) {
  const size_t e = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  f32x4 m = *(f32x4*)(s.m + e), v = *(f32x4*)(s.v + e), p = *(f32x4*)(s.p + e), t = *(f32x4*)(s.t + e);
  const f32x4 g = *(const f32x4*)(s.g + e);
  unit_math(p, m, v, t, g);
  *(f32x4*)(s.m + e) = m; *(f32x4*)(s.v + e) = v; *(f32x4*)(s.p + e) = p; *(f32x4*)(s.t + e) = t;
}
__device__ __forceinline__ bool wait_cnt(const unsigned* c, unsigned want) {
  unsigned long long spins = 0;
  while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
    if (++spins > 2000000ull) return false;
    __builtin_amdgcn_s_sleep(1);
  }
  return true;
}
__device__ __forceinline__ void fused_unit(const State& s, int net, int unit, int tid, const unsigned* cnt, unsigned want) {
  const size_t e = (size_t)net * NETF + (size_t)unit * 1024 + tid * 4;
  f32x4 m = *(f32x4*)(s.m + e), v = *(f32x4*)(s.v + e), p = *(f32x4*)(s.p + e), t = *(f32x4*)(s.t + e);   // in flight during the wait
  __shared__ int ok;
  if (tid == 0) ok = wait_cnt(cnt + net, want) ? 1 : 0;
  __syncthreads();
  if (!ok) return;
  f32x4 g;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(g) : "v"(s.g + e) : "memory");
  unit_math(p, m, v, t, g);
  *(f32x4*)(s.m + e) = m; *(f32x4*)(s.v + e) = v; *(f32x4*)(s.p + e) = p; *(f32x4*)(s.t + e) = t;
  __syncthreads();
}
__global__ __launch_bounds__(256) void k_fused(State s, unsigned* cnt, unsigned epoch, unsigned long long c_scalar,
                                               unsigned long long c_pi, float val) {
  const int net = blockIdx.x & 3, b = blockIdx.x >> 2, tid = threadIdx.x;
  work(net == 3 ? c_pi : c_scalar);
  write_grad<true>(s, net, b, tid, val);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) __hip_atomic_fetch_add(cnt + net, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (net == 3) return;
  const unsigned want = epoch * 64u;
  fused_unit(s, net, b, tid, cnt, want);
  if (b < UNITS - 64) fused_unit(s, net, 64 + b, tid, cnt, want);
  const int idx = net * 64 + b;            // 0..191 over the scalar nets' blocks, taken from the BACK: the blocks with one unit
  if (idx >= 192 - UNITS) fused_unit(s, 3, idx - (192 - UNITS), tid, cnt, want);
}
int main() {
  State s;
  const size_t n = (size_t)4 * NETF;
  hipMalloc(&s.p, n * 4); hipMalloc(&s.m, n * 4); hipMalloc(&s.v, n * 4); hipMalloc(&s.t, n * 4); hipMalloc(&s.g, n * 4);
  unsigned* cnt; hipMalloc(&cnt, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const unsigned long long cs = 6000 * 2.4, cp = 7000 * 2.4;      // ~6.0 / 7.0 us at 2.4 GHz (the printed times tell)
  for (int variant = 0; variant < 3; ++variant) {
    for (int rep = 0; rep < 3; ++rep) {
      hipMemset(s.p, 0, n * 4); hipMemset(s.m, 0, n * 4); hipMemset(s.v, 0, n * 4); hipMemset(s.t, 0, n * 4);
      hipMemset(cnt, 0, 16);
      hipDeviceSynchronize();
      const int iters = 2000;
      hipEventRecord(e0, 0);
      for (int i = 0; i < iters; ++i) {
        const float val = 1.f + (i & 7);
        if (variant == 0) { hipLaunchKernelGGL(k_work, dim3(256), dim3(256), 0, 0, s, cs, cp, val); }
        else if (variant == 1) {
          hipLaunchKernelGGL(k_work, dim3(256), dim3(256), 0, 0, s, cs, cp, val);
          hipLaunchKernelGGL(k_update, dim3(4 * UNITS), dim3(256), 0, 0, s);
        } else hipLaunchKernelGGL(k_fused, dim3(256), dim3(256), 0, 0, s, cnt, (unsigned)(i + 1), cs, cp, val);
      }
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      std::vector<float> hm(n), hp(n);
      hipMemcpy(hm.data(), s.m, n * 4, hipMemcpyDeviceToHost);
      hipMemcpy(hp.data(), s.p, n * 4, hipMemcpyDeviceToHost);
      double cm = 0, cpv = 0;
      for (size_t i = 0; i < n; ++i) { cm += hm[i]; cpv += hp[i]; }
      printf("%-34s rep %d: %7.2f us per iteration   (checksums m %.6e p %.6e)\n",
             variant == 0 ? "work kernel alone" : (variant == 1 ? "work + update, two launches" : "fused, one launch"), rep,
             ms / iters * 1e3, cm, cpv);
    }
  }
  return 0;
}
