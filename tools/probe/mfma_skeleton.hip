// Probe: the K loop of conv_gemm_fast_kernel rebuilt piece by piece (no address logic, no epilogue) to price each
// ingredient next to the matrix pipe:  MFMA only -> + LDS fragment reads -> + barrier per K tile -> + global->LDS DMA.
// 5 workgroups of 4 waves per CU (32 KB LDS each), as the real kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void;
template <bool LDSR, bool BAR, bool DMA, bool STREAM = false, bool SAMEA = false>
__global__ __launch_bounds__(256, 4) void k(const float* __restrict__ src, float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float smem[2 * 4096];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 8192; i += 256) smem[i] = 1e-6f * i;
  __syncthreads();
  f32x16 acc[2][2];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int li = lane & 31, lk = lane >> 5;
  const float* fragA = smem + lk * 128 + (wave >> 1) * 32 + li;
  const float* fragB = smem + 2048 + lk * 128 + (wave & 1) * 32 + li;
  // STREAM: every workgroup walks its own 144 x 8 KB slice of a 1.2 GB buffer (B tiles straight from HBM, like the real
  // kernel's activations); otherwise all reads hit a 4 MB cache-resident window.
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, STREAM ? 0x7fffffff : (1 << 24), 0x00020000);
  unsigned voff4 = (unsigned)(tid * 16), voff = (unsigned)(tid * 4);
  float ra = 1e-9f * tid, rb = 1.f;
  for (int it = 0; it < iters; ++it) {
    const int buf = it & 1;
    if (DMA) {
      float* As = smem + (buf ^ 1) * 4096 + wave * 256;
      const unsigned soff = STREAM ? (unsigned)(((blockIdx.x & 1023) * 144 + (it % 144)) * 8192u)
                                   : (unsigned)(((it * 37 + blockIdx.x) & 255) * 16384);
      const unsigned soffA = SAMEA ? (unsigned)((it % 144) * 8192u) : soff;   // SAMEA: every workgroup reads the same A tile
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        unsigned o = voff4 + j * 4096; asm volatile("" : "+v"(o));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(As + 1024 * j), 16, (int)o, (int)soffA, 0, 0);
      }
      float* Bs = smem + (buf ^ 1) * 4096 + 2048 + (wave & 1) * 64 + (wave >> 1) * 128;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        unsigned o = voff + j * 1024; asm volatile("" : "+v"(o));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(Bs + 256 * j), 4, (int)o, (int)soff, 0, 0);
      }
    }
    const float* Af = fragA + buf * 4096;
    const float* Bf = fragB + buf * 4096;
    float a[2][2], b[2][2];
    if (LDSR) { for (int t = 0; t < 2; ++t) { a[0][t] = Af[64 * t]; b[0][t] = Bf[64 * t]; } }
    else { a[0][0] = ra; a[0][1] = rb; b[0][0] = rb; b[0][1] = ra; a[1][0] = rb; a[1][1] = ra; b[1][0] = ra; b[1][1] = rb; }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int cur = ks & 1;
      if (LDSR && ks + 1 < 8)
        for (int t = 0; t < 2; ++t) { a[cur ^ 1][t] = Af[(ks + 1) * 256 + 64 * t]; b[cur ^ 1][t] = Bf[(ks + 1) * 256 + 64 * t]; }
      __builtin_amdgcn_sched_barrier(0);
      for (int tm = 0; tm < 2; ++tm) for (int tn = 0; tn < 2; ++tn)
        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][tm], b[cur][tn], acc[tm][tn], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (BAR) __syncthreads();
  }
  float s = 0.f;
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
  out[blockIdx.x * 256 + tid] = s;
}
template <bool L, bool B, bool D, bool S = false, bool SA = false>
void run(const float* src, float* out, const char* name, int blocks) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 144;
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<L, B, D, S, SA>), dim3(blocks), dim3(256), 0, 0, src, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  printf("%-44s %5d workgroups x %d K tiles: %.3f ms  %.1f TFLOP/s\n", name, blocks, iters, best,
         (double)blocks * 4 * iters * 32 * 4096.0 / best / 1e9);
}
int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  float *src, *out; (void)hipMalloc(&src, 1300u << 20); (void)hipMemset(src, 0, 1300u << 20); (void)hipMalloc(&out, 8192 * 256 * 4);
  for (int blocks : {1024, 2048, 8192}) {
    run<false, false, false>(src, out, "MFMA only", blocks);
    run<true, false, false>(src, out, "+ LDS fragment reads", blocks);
    run<true, true, false>(src, out, "+ LDS reads + barrier", blocks);
    run<true, true, true>(src, out, "+ LDS reads + barrier + global->LDS DMA", blocks);
    run<false, true, true>(src, out, "barrier + DMA, no LDS reads", blocks);
    run<true, true, true, true>(src, out, "+ LDS + barrier + DMA streaming from HBM", blocks);
    run<true, true, true, true, true>(src, out, "  same, A tile shared by all workgroups", blocks);
  }
  return 0;
}
