// Probe: raw buffer load semantics on gfx950 (valid offsets, out-of-range offsets, descriptor flags).
#include <hip/hip_runtime.h>
#include <cstdio>
#define DP_RSRC_FLAGS 0x00020000
__global__ void k(const float* p, float* o, unsigned bytes) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, DP_RSRC_FLAGS);
  unsigned t = threadIdx.x;
  unsigned off = (t < 32) ? t * 4 : ((t < 48) ? 0x80000000u : (bytes + (t - 48) * 4));
  asm volatile("" : "+v"(off));
  o[t] = __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0);
}
int main() {
  float h[256]; for (int i = 0; i < 256; ++i) h[i] = 100.f + i;
  float *d, *o; hipMalloc(&d, sizeof(h)); hipMalloc(&o, 64 * 4);
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, 128u * 4u);
  float r[64]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  for (int i = 0; i < 64; ++i) printf("%d:%g ", i, r[i]);
  printf("\n");
  return 0;
}
