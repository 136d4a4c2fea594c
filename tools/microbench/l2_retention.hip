// Does an XCD's L2 keep lines across a kernel boundary?  Producer kernel P touches (reads or writes) a
// 256 KiB region per XCD group g = blockIdx & 7; consumer kernel C then reads region (g + shift) & 7 with
// all 32 blocks of group g (each block reads the whole 256 KiB as 16-B lane loads, like a weight fetch).
// shift = 0: the consumer's XCD group touched the lines itself just before; shift = 3: another group did.
// Build: hipcc --offload-arch=gfx950 -O3 -o l2_retention l2_retention.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int REGION_F4 = 256 * 1024 / 16;   // float4 per region

__global__ void producer(f32x4* buf, int write, float v) {
  const int g = blockIdx.x & 7, b = blockIdx.x >> 3;          // 32 blocks per group
  f32x4* r = buf + (size_t)g * REGION_F4;
  for (int i = b * 256 + threadIdx.x; i < REGION_F4; i += 32 * 256) {
    if (write) r[i] = (f32x4){v, v, v, v};
    else { f32x4 x = r[i]; if (x[0] == 12345.f) r[i] = x; }
  }
}
__global__ void consumer(const f32x4* buf, float* out, int shift) {
  const int g = blockIdx.x & 7;
  const f32x4* r = buf + (size_t)((g + shift) & 7) * REGION_F4;
  f32x4 acc = (f32x4){0, 0, 0, 0};
  for (int i = threadIdx.x; i < REGION_F4; i += 256) acc += r[i];   // whole region, every block
  if (acc[0] == 12345.f) out[blockIdx.x] = acc[1];
}
int main() {
  f32x4* buf; float* out;
  CHK(hipMalloc(&buf, 8 * (size_t)REGION_F4 * 16)); CHK(hipMalloc(&out, 4096));
  CHK(hipMemset(buf, 0, 8 * (size_t)REGION_F4 * 16));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  for (int write = 0; write < 2; ++write)
    for (int shift : {0, 3}) {
      float best = 1e9f, sum = 0;
      const int reps = 200;
      for (int it = 0; it < reps + 20; ++it) {
        hipLaunchKernelGGL(producer, dim3(256), dim3(256), 0, 0, buf, write, (float)it);
        CHK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(consumer, dim3(256), dim3(256), 0, 0, buf, out, shift);
        CHK(hipEventRecord(e1, 0));
        CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        if (it >= 20) { sum += ms; if (ms < best) best = ms; }
      }
      printf("producer %s, consumer reads group+%d : avg %.2f us  min %.2f us  (256 blocks x 256 KiB)\n",
             write ? "WRITES" : "reads ", shift, sum / reps * 1e3f, best * 1e3f);
    }
  // reference: the consumer alone, twice in a row (second launch: whatever survives from the first)
  {
    float sum = 0; const int reps = 200;
    for (int it = 0; it < reps; ++it) {
      hipLaunchKernelGGL(consumer, dim3(256), dim3(256), 0, 0, buf, out, 0);
      CHK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(consumer, dim3(256), dim3(256), 0, 0, buf, out, 0);
      CHK(hipEventRecord(e1, 0));
      CHK(hipEventSynchronize(e1));
      float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); sum += ms;
    }
    printf("consumer right after an identical consumer launch: avg %.2f us\n", sum / reps * 1e3f);
  }
  return 0;
}
