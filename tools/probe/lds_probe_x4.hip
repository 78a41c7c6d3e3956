// Probe: buffer_load_dwordx4 ... lds on gfx950 with addresses that are 4-byte but not 16-byte aligned, and with a buffer
// whose end cuts through a lane's 16 bytes (is the range check per dword or per lane?).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void lds_void;
__global__ void k(const float* p, float* o, unsigned bytes, unsigned shift_bytes) {
  __shared__ __attribute__((aligned(16))) float sm[256];
  for (int i = threadIdx.x; i < 256; i += 64) sm[i] = -7.f;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000);
  unsigned off4 = threadIdx.x * 16 + shift_bytes;
  asm volatile("" : "+v"(off4));
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)sm, 16, (int)off4, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) o[i] = sm[i];
}
int main() {
  float h[512]; for (int i = 0; i < 512; ++i) h[i] = 100.f + i;
  float *d, *o; (void)hipMalloc(&d, sizeof(h)); (void)hipMalloc(&o, 256 * 4);
  (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  for (unsigned shift = 0; shift <= 12; shift += 4) {
    unsigned bytes = 64 * 16 - 8;                       // the last lane's 16 bytes straddle the end of the buffer when shift > 0 ...
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, bytes, shift);
    hipError_t e = hipDeviceSynchronize();
    float r[256]; (void)hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) {
      unsigned byte = i * 4 + shift;
      float want = (byte + 4 <= bytes) ? 100.f + byte / 4 : 0.f;          // per-dword range check expected
      if (r[i] != want) ++bad;
    }
    printf("shift %2u bytes: err %d, mismatches vs per-dword model %d; lane0: %g %g %g %g  last lanes: ", shift, (int)e, bad, r[0], r[1], r[2], r[3]);
    for (int i = 244; i < 256; ++i) printf("%g ", r[i]);
    printf("\n");
  }
  return 0;
}
