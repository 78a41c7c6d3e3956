// Probe: sustained v_mfma_f32_32x32x2_f32 rate on MI355X with no memory traffic -- the practical ceiling the
// contraction kernels are priced against (nominal 157.3 TFLOP/s = 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip && ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* clk) {
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float a0 = threadIdx.x * 1e-9f, a1 = a0 + 1e-9f, b0 = 1.f + blockIdx.x * 1e-9f, b1 = b0 + 1e-9f;
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
    }
  }
  const unsigned long long c1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}
int main() {
  float* out; unsigned long long* clk;
  (void)hipMalloc(&out, 256 * 8 * 256 * 4); (void)hipMalloc(&clk, 16);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int wps = 1; wps <= 8; wps *= 2) {          // waves per SIMD = workgroups per CU (4 waves each, one per SIMD)
    const int blocks = 256 * wps, iters = 40000 / wps;
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, clk);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      unsigned long long h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
      const double flop = (double)blocks * 4 * iters * 32 * 4096.0;
      printf("waves/SIMD %d  %.2f ms  %.1f TFLOP/s   clock64/wall_clock64 = %.3f (x100 MHz => %.0f MHz if wall clock is 100 MHz)\n",
             wps, ms, flop / ms / 1e9, (double)h[0] / (double)h[1], 100.0 * h[0] / h[1]);
    }
  }
  return 0;
}
