// Semantics check of ds_read_b64_tr_b16 (gfx950) as iqlhip_lb_kernels.h uses it: a [rows][72] bf16 tile, lane (l15 = 4 q + p,
// g) of a wave passes the address of (row r0 + 8 g + q, columns c0 + 4 p ..) and must receive column c0 + l15 of rows
// r0 + 8 g + 0..3 in its elements 0..3.  hipcc --offload-arch=gfx950 -O3 tr_read_check.hip -o tr_read_check
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, int mode) {
  __shared__ __attribute__((aligned(16))) __bf16 t[64 * 72];
  for (int i = threadIdx.x; i < 64 * 72; i += 64) t[i] = (__bf16)(float)(mode ? (i % 72) : (i / 72));
  __syncthreads();
  const int l = threadIdx.x, l15 = l & 15, g = l >> 4, q = l15 >> 2, pp = l15 & 3;
  const int r0 = 8, c0 = 16;
  bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(t + (r0 + 8 * g + q) * 72 + c0 + 4 * pp));
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = (float)v[e];
}
int main() {
  float* d; float h[256];
  hipMalloc(&d, sizeof h);
  int bad = 0;
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l)
      for (int e = 0; e < 4; ++e) {
        const int l15 = l & 15, g = l >> 4;
        const float want = mode ? (float)(16 + l15) : (float)(8 + 8 * g + e);
        if (h[l * 4 + e] != want) { if (bad < 8) printf("mode %d lane %d e %d: got %g want %g\n", mode, l, e, h[l * 4 + e], want); ++bad; }
      }
  }
  printf("tr_read_check: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
  return bad != 0;
}
