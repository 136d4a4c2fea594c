// What an in-kernel hand-off between the blocks of ONE launch costs on gfx950 (8 XCDs, one non-coherent L2 each):
// every block writes 8 KB, signals on a device-scope counter, waits until all blocks have signalled, then reads the
// 8 KB another block (on another XCD) wrote and checks it.
//   mode 0  plain stores + release fence at agent scope (buffer_wbl2 sc1) / acquire fence (buffer_inv sc1) + plain loads
//   mode 1  write-through stores (sc1) + s_waitcnt vmcnt(0) / loads that bypass the non-coherent levels (sc1)
// Reported per phase: median / p90 / max cycles (s_memtime, 100 MHz-independent shader clock) over the 256 blocks.
// Build: hipcc --offload-arch=gfx950 -O3 -o fused_tail_latency fused_tail_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned long long memtime() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
template <int MODE>
__global__ __launch_bounds__(256) void k(float* buf, unsigned* cnt, unsigned long long* stamps, unsigned* bad, float tag) {
  const int b = blockIdx.x, tid = threadIdx.x, nb = gridDim.x;
  const unsigned long long t0 = memtime();
  float* mine = buf + (size_t)b * 2048;
  for (int q = 0; q < 2; ++q) {
    f32x4 v = (f32x4){tag + b, tag + tid, tag + q, 1.f};
    float* p = mine + (q * 256 + tid) * 4;
    if (MODE == 0) *(f32x4*)p = v;
    else asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  }
  if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const unsigned long long t1 = memtime();
  if (tid == 0) {
    unsigned got = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" ::"v"(got));
  }
  const unsigned long long t2 = memtime();
  if (tid == 0) {
    unsigned long long spins = 0;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nb && ++spins < 4000000ull) __builtin_amdgcn_s_sleep(2);
  }
  __syncthreads();
  const unsigned long long t3 = memtime();
  if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  const int o = (b + 37) % nb;     // (consecutive block indices go round the XCDs: +37 is another XCD)
  const float* other = buf + (size_t)o * 2048;
  unsigned wrong = 0;
  for (int q = 0; q < 2; ++q) {
    const float* p = other + (q * 256 + tid) * 4;
    f32x4 v;
    if (MODE == 0) v = *(const f32x4*)p;
    else asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    if (v[0] != tag + o || v[1] != tag + tid || v[2] != tag + q) wrong++;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t4 = memtime();
  if (wrong) atomicAdd(bad, wrong);
  if (tid == 0) {
    stamps[b * 4 + 0] = t1 - t0; stamps[b * 4 + 1] = t2 - t1; stamps[b * 4 + 2] = t3 - t2; stamps[b * 4 + 3] = t4 - t3;
  }
}
int main() {
  const int nb = 256;
  float* buf; unsigned *cnt, *bad; unsigned long long* st;
  hipMalloc(&buf, (size_t)nb * 2048 * 4); hipMalloc(&cnt, 4); hipMalloc(&bad, 4); hipMalloc(&st, nb * 4 * 8);
  hipMemset(bad, 0, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      float ms_sum = 0;
      const int iters = 200;
      for (int i = 0; i < iters; ++i) {
        hipMemsetAsync(cnt, 0, 4, 0);
        hipEventRecord(e0, 0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(nb), dim3(256), 0, 0, buf, cnt, st, bad, (float)(i + 1 + 1000 * rep));
        else hipLaunchKernelGGL(k<1>, dim3(nb), dim3(256), 0, 0, buf, cnt, st, bad, (float)(i + 1 + 1000 * rep));
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms_sum += ms;
      }
      std::vector<unsigned long long> h(nb * 4);
      hipMemcpy(h.data(), st, nb * 4 * 8, hipMemcpyDeviceToHost);
      unsigned hb; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
      const char* names[4] = {"store + make visible", "counter add (returning)", "wait for all blocks", "read the other block's 8 KB"};
      printf("mode %d (%s) rep %d: %.2f us per launch (event pair), wrong words so far %u\n", mode, mode ? "sc1 stores / sc1 loads" : "fences", rep, ms_sum / iters * 1e3, hb);
      for (int ph = 0; ph < 4; ++ph) {
        std::vector<unsigned long long> a;
        for (int b = 0; b < nb; ++b) a.push_back(h[b * 4 + ph]);
        std::sort(a.begin(), a.end());
        printf("    %-30s median %6llu  p90 %6llu  max %6llu cycles\n", names[ph], a[nb / 2], a[nb * 9 / 10], a[nb - 1]);
      }
    }
  }
  return 0;
}
