// Probe: how much do plain VALU / SALU instructions interleaved with v_mfma_f32_32x32x2_f32 cost on gfx950?
// Each wave runs groups of 32 MFMAs (one K tile of the contraction kernels) followed by NV integer VALU ops and NS SALU ops.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NV, int NS, int QUARTER>
__global__ __launch_bounds__(256) void k(float* out, int iters, int seed) {
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float a0 = threadIdx.x * 1e-9f, a1 = a0 + 1e-9f, b0 = 1.f + blockIdx.x * 1e-9f, b1 = b0 + 1e-9f;
  int v0 = threadIdx.x + seed, v1 = v0 * 3, v2 = v0 + 7, v3 = v0 ^ 5;
  int s0 = seed, s1 = seed + 1;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < NV / 4; ++u) {
      if (QUARTER) {
        asm volatile("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %2, %2, %3" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
        asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %3" : "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v0));
      } else {
        asm volatile("v_add_u32 %0, %0, %1\n v_xor_b32 %2, %2, %3\n v_add_u32 %1, %1, %2\n v_xor_b32 %3, %3, %0"
                     : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
      }
    }
#pragma unroll
    for (int u = 0; u < NS / 2; ++u) asm volatile("s_add_i32 %0, %0, %1\n s_xor_b32 %1, %1, %0" : "+s"(s0), "+s"(s1) : : "scc");
  }
  float s = (float)(v0 + v1 + v2 + v3 + s0 + s1);
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NV, int NS, int Q>
void run(float* out, const char* name) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int wps = 8; wps <= 8; wps *= 2) {
    const int blocks = 256 * wps, iters = 400;
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL((k<NV, NS, Q>), dim3(blocks), dim3(256), 0, 0, out, iters, rep);
      if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return; }
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("%-28s %d workgroups: %.1f TFLOP/s\n", name, blocks, (double)blocks * 4 * iters * 32 * 4096.0 / best / 1e9);
  }
}
int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  float* out; (void)hipMalloc(&out, 256 * 16 * 256 * 4);
  run<0, 0, 0>(out, "32 MFMA only");
  run<104, 0, 0>(out, "+104 VALU");
  run<104, 92, 0>(out, "+104 VALU +92 SALU");
  run<0, 92, 0>(out, "+92 SALU");
  run<32, 0, 1>(out, "+32 VALU (half v_mul_lo)");
  run<200, 0, 0>(out, "+200 VALU");
  return 0;
}
