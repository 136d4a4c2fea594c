// Ground truth for the step's MFMA phases: how long do 128 back-to-back v_mfma_f32_16x16x4_f32 per wave take on
// this chip (one wave per SIMD, 256 blocks), with 8 / 2 / 1 independent accumulators, in s_memtime units and on the
// chip-wide 100 MHz clock?  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip
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
__device__ __forceinline__ unsigned long long realtime() {
  unsigned long long t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

template <int NACC, int REPS>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* stamps, float a0, float b0) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = a0 + threadIdx.x, b = b0 + threadIdx.x;
  __syncthreads();
  const unsigned long long r0 = realtime();
  const unsigned long long t0 = memtime();
#pragma unroll
  for (int r = 0; r < REPS; ++r) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a + (float)i, b, acc[i], 0, 0, 0);   // distinct operands: no CSE across accumulators
  }
  // force completion: read the accumulators
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  asm volatile("" ::"v"(s));
  const unsigned long long t1 = memtime();
  const unsigned long long r1 = realtime();
  if (threadIdx.x == 0) {
    stamps[blockIdx.x * 2] = t1 - t0;
    stamps[blockIdx.x * 2 + 1] = r1 - r0;
  }
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int REPS>
void run(const char* name) {
  float* out;
  unsigned long long* st;
  hipMalloc(&out, 256 * 256 * 4);
  hipMalloc(&st, 256 * 2 * 8);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<NACC, REPS>), dim3(256), dim3(256), 0, 0, out, st, 1.f, 2.f);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  for (int w = 0; w < 100; ++w) hipLaunchKernelGGL((k<NACC, REPS>), dim3(256), dim3(256), 0, 0, out, st, 1.f, 2.f);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(512);
  hipMemcpy(h.data(), st, 512 * 8, hipMemcpyDeviceToHost);
  std::vector<double> mt, rt;
  for (int i = 0; i < 256; ++i) { mt.push_back((double)h[2 * i]); rt.push_back((double)h[2 * i + 1]); }
  std::sort(mt.begin(), mt.end()); std::sort(rt.begin(), rt.end());
  const int n = NACC * REPS;
  printf("%-28s %4d MFMA/wave: s_memtime median %8.0f (%.1f per MFMA)  realtime median %.2f us (%.1f ns per MFMA)  kernel %.2f us\n",
         name, n, mt[128], mt[128] / n, rt[128] / 100.0, rt[128] * 10.0 / n, ms * 10.0);
  hipFree(out); hipFree(st);
}

int main() {
  run<8, 16>("8 accumulators");
  run<2, 64>("2 accumulators");
  run<1, 128>("1 accumulator (dependent)");
  run<8, 64>("8 accumulators, 512");
  run<8, 256>("8 accumulators, 2048");
  return 0;
}
