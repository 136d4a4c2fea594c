// How long does a block wait for its kernel arguments, and what does kernel-argument preloading (the first dwords of
// the explicit arguments delivered in user SGPRs at wave launch: -mllvm -amdgpu-kernarg-preload-count=N) buy?
// Three kernels that do the same thing — one dependent global load through a pointer, then a store — and differ in how
// the pointer reaches the wave:
//   byval    the pointer is the LAST field of a 1.1 KB by-value struct (the step kernels' StepParams shape)
//   preload  the pointer is the first explicit argument (preloaded when the file is built with the option)
//   indirect a preloaded pointer to a device-resident copy of the struct, the pointer read from there (s_load from
//            global memory that stays cached across launches, instead of the freshly written kernel-argument ring)
// Reported: s_memtime cycles from kernel entry to "pointer available" and to "dependent load returned", and the
// average launch time of 1 000 dependent back-to-back launches (graph-free, one stream).
// Build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=16 -o kernarg_latency kernarg_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

struct Big {
  unsigned long long pad[140];     // 1 120 bytes
  const float* src;
  float* dst;
  unsigned long long* stamps;
};

__device__ __forceinline__ unsigned long long memtime() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

__global__ __launch_bounds__(256) void k_byval(Big b) {
  const unsigned long long t0 = memtime();
  const float* s = b.src;
  asm volatile("" ::"s"((unsigned long long)(uintptr_t)s));     // the pointer is in SGPRs here
  const unsigned long long t1 = memtime();
  const float v = s[blockIdx.x * 256 + threadIdx.x];
  asm volatile("s_waitcnt vmcnt(0)" ::"v"(v) : "memory");
  const unsigned long long t2 = memtime();
  b.dst[blockIdx.x * 256 + threadIdx.x] = v + 1.f;
  if (threadIdx.x == 0) { b.stamps[blockIdx.x * 2] = t1 - t0; b.stamps[blockIdx.x * 2 + 1] = t2 - t0; }
}

__global__ __launch_bounds__(256) void k_preload(const float* s, float* dst, unsigned long long* stamps) {
  const unsigned long long t0 = memtime();
  asm volatile("" ::"s"((unsigned long long)(uintptr_t)s));
  const unsigned long long t1 = memtime();
  const float v = s[blockIdx.x * 256 + threadIdx.x];
  asm volatile("s_waitcnt vmcnt(0)" ::"v"(v) : "memory");
  const unsigned long long t2 = memtime();
  dst[blockIdx.x * 256 + threadIdx.x] = v + 1.f;
  if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = t2 - t0; }
}

__global__ __launch_bounds__(256) void k_indirect(const Big* __restrict__ pb, float* dst, unsigned long long* stamps) {
  const unsigned long long t0 = memtime();
  const float* s = pb->src;
  asm volatile("" ::"s"((unsigned long long)(uintptr_t)s));
  const unsigned long long t1 = memtime();
  const float v = s[blockIdx.x * 256 + threadIdx.x];
  asm volatile("s_waitcnt vmcnt(0)" ::"v"(v) : "memory");
  const unsigned long long t2 = memtime();
  dst[blockIdx.x * 256 + threadIdx.x] = v + 1.f;
  if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = t2 - t0; }
}

static void report(const char* name, unsigned long long* st_dev, float us) {
  std::vector<unsigned long long> h(512);
  hipMemcpy(h.data(), st_dev, 512 * 8, hipMemcpyDeviceToHost);
  std::vector<unsigned long long> a, b;
  for (int i = 0; i < 256; ++i) { a.push_back(h[2 * i]); b.push_back(h[2 * i + 1]); }
  std::sort(a.begin(), a.end());
  std::sort(b.begin(), b.end());
  printf("%-9s pointer available after median %5llu (p90 %5llu) cycles; dependent load returned after median %5llu (p90 %5llu); %.2f us per dependent launch\n",
         name, a[128], a[230], b[128], b[230], us);
}

int main() {
  float *src, *dst;
  unsigned long long* st;
  Big* bdev;
  hipMalloc(&src, 256 * 256 * 4);
  hipMalloc(&dst, 256 * 256 * 4);
  hipMalloc(&st, 512 * 8);
  hipMalloc(&bdev, sizeof(Big));
  hipMemset(src, 0, 256 * 256 * 4);
  Big b;
  for (auto& p : b.pad) p = 1;
  b.src = src; b.dst = dst; b.stamps = st;
  hipMemcpy(bdev, &b, sizeof b, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int variant = 0; variant < 3; ++variant) {
    for (int rep = 0; rep < 2; ++rep) {
      auto launch = [&]() {
        // the source is rewritten by the previous launch's destination role being swapped: keep the loads cold-ish
        if (variant == 0) hipLaunchKernelGGL(k_byval, dim3(256), dim3(256), 0, 0, b);
        else if (variant == 1) hipLaunchKernelGGL(k_preload, dim3(256), dim3(256), 0, 0, (const float*)src, dst, st);
        else hipLaunchKernelGGL(k_indirect, dim3(256), dim3(256), 0, 0, (const Big*)bdev, dst, st);
      };
      for (int w = 0; w < 20; ++w) launch();
      hipDeviceSynchronize();
      hipEventRecord(e0, 0);
      for (int w = 0; w < 1000; ++w) launch();
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      report(variant == 0 ? "byval" : (variant == 1 ? "preload" : "indirect"), st, ms);
    }
  }
  return 0;
}
