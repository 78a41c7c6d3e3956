// Probe: buffer_load ... lds semantics on gfx950 (out-of-range lanes, dword and dwordx4 forms).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void lds_void;
__global__ void k(const float* p, float* o, unsigned bytes) {
  __shared__ __attribute__((aligned(16))) float sm[64 + 256];
  for (int i = threadIdx.x; i < 320; i += 64) sm[i] = -7.f;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000);
  unsigned t = threadIdx.x;
  unsigned off = (t % 3 == 2) ? 0x80000000u : t * 4;            // every third lane out of range
  asm volatile("" : "+v"(off));
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)sm, 4, (int)off, 0, 0, 0);
  unsigned off4 = (t % 4 == 1) ? 0x80000000u : t * 16;
  asm volatile("" : "+v"(off4));
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)(sm + 64), 16, (int)off4, 0, 0, 0);
  __syncthreads();
  for (int i = threadIdx.x; i < 320; i += 64) o[i] = sm[i];
}
int main() {
  float h[512]; for (int i = 0; i < 512; ++i) h[i] = 100.f + i;
  float *d, *o; (void)hipMalloc(&d, sizeof(h)); (void)hipMalloc(&o, 320 * 4);
  (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, 512u * 4u);
  float r[320]; (void)hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  printf("dword: "); for (int i = 0; i < 12; ++i) printf("%g ", r[i]); printf("\n");
  printf("x4   : "); for (int i = 0; i < 24; ++i) printf("%g ", r[64 + i]); printf("\n");
  return 0;
}
