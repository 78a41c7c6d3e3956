// Common device helpers for the Diff-Pruning MI355X (gfx950 / CDNA4) kernels.
// Wavefront = 64 lanes; everything here is written for gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/dp_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DP_WAVE 64

// Geometry of a (possibly strided / upsampled / zero-padded) 2-D gather: see dp_conv_geom in include/dp_hip.h.
// Shared by the implicit-GEMM convolution kernels (forward, dgrad and wgrad all read their activation
// operand through it).
typedef dp_conv_geom ConvGeom;

__device__ __forceinline__ bool dp_gather(const ConvGeom& g, int ho, int wo, int ky, int kx, int& off) {
    int hn = ho * g.stride + ky - g.pad_t;
    int wn = wo * g.stride + kx - g.pad_l;
    bool v = true;
    if (g.sden == 2) {
        v = (((hn | wn) & 1) == 0);
        hn >>= 1;
        wn >>= 1;
    }
    v = v && ((unsigned)hn < (unsigned)g.Hv) && ((unsigned)wn < (unsigned)g.Wv);
    off = (hn >> g.ups) * g.Ws + (wn >> g.ups);
    return v;
}

__device__ __forceinline__ float dp_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float dp_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Block-wide sum for blockDim.x == 256 (4 waves).  `sm` must hold >= 4 floats.  All threads get the result.
__device__ __forceinline__ float dp_block_sum_256(float v, float* sm) {
    v = dp_wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[w] = v;
    __syncthreads();
    return (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// sigmoid(x) as v_exp_f32 + v_rcp_f32 (1 ulp each) instead of expf + an IEEE division (~10 instructions): the GroupNorm kernels
// spent about a third of their time in this arithmetic.  [measured, round 5, one box, both builds: GroupNorm forward 1.50 -> 1.36 ms
// and backward 2.75 -> 2.62 ms per headline timestep (tools/bench_gn.py), headline 61.26 -> 60.81 ms, bedroom-256 47.77 -> 47.26 ms;
// forward 256 ch @ 16 x 16 26.7 -> 22.6 us = 5.9 TB/s of algorithmic bytes.]  -DDP_IEEE_SIGMOID builds the exact form.
#ifndef DP_IEEE_SIGMOID
#define DP_FAST_SIGMOID 1
#endif
__device__ __forceinline__ float dp_sigmoid(float x) {
#ifdef DP_FAST_SIGMOID
    return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
#else
    return 1.0f / (1.0f + expf(-x));
#endif
}

#ifdef DP_FAST_SIGMOID
__device__ __forceinline__ float dp_silu(float x) { return x * dp_sigmoid(x); }
#else
__device__ __forceinline__ float dp_silu(float x) { return x / (1.0f + expf(-x)); }
#endif

// d/dx silu(x) = s * (1 + x * (1 - s)),  s = sigmoid(x)
__device__ __forceinline__ float dp_silu_grad(float x) {
    const float s = dp_sigmoid(x);
    return s * (1.0f + x * (1.0f - s));
}

// ------------------------------------------------------------------------------------------------
// Dropout masks: counter-based Philox4x32-10 (Salmon et al., SC'11; the Random123 constants), so the backward pass
// REGENERATES the mask of the forward pass from (seed, site, step, element index) instead of storing it, and a CPU
// restatement (oracle/philox_ref.py) reproduces every mask bit-for-bit.
//   counter = (idx4 lo, idx4 hi, site, step), key = (seed lo, seed hi); element idx takes output word idx & 3;
//   keep iff (word >> 8) >= thr24  (thr24 = ceil(p * 2^24));  kept values are scaled by 1 / (1 - p).
// idx is the LOGICAL element index ((n_global * C + c) * HW + hw), independent of strides and of the rank's shard.
// ------------------------------------------------------------------------------------------------
struct DpDrop {
    unsigned thr24;            // 0 = dropout disabled
    float scale;               // 1 / (1 - p)
    unsigned seed_lo, seed_hi, site, step;
    long long n_off;           // global index of this shard's first image
    const unsigned* step_ptr;  // device counter read instead of `step` when set (replayed finetune steps)
};

__device__ __forceinline__ uint4 dp_philox4x32_10(uint4 c, unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const unsigned hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = make_uint4(hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

// multipliers (0 or scale) of the 4 elements idx .. idx+3, idx % 4 == 0
__device__ __forceinline__ float4 dp_drop4(const DpDrop& d, long long idx) {
    const unsigned long long q = (unsigned long long)idx >> 2;
    const unsigned step = d.step_ptr ? *d.step_ptr : d.step;          // wave-uniform address: one scalar load
    const uint4 r = dp_philox4x32_10(make_uint4((unsigned)q, (unsigned)(q >> 32), d.site, step), d.seed_lo, d.seed_hi);
    return make_float4((r.x >> 8) >= d.thr24 ? d.scale : 0.f, (r.y >> 8) >= d.thr24 ? d.scale : 0.f,
                       (r.z >> 8) >= d.thr24 ? d.scale : 0.f, (r.w >> 8) >= d.thr24 ? d.scale : 0.f);
}

__device__ __forceinline__ float dp_drop1(const DpDrop& d, long long idx) {
    const float4 m = dp_drop4(d, idx & ~3ll);
    const int j = (int)(idx & 3);
    return j == 0 ? m.x : j == 1 ? m.y : j == 2 ? m.z : m.w;
}

static inline DpDrop dp_drop_host(const dp_dropout* d) {
    DpDrop r{};
    if (d && d->thr24) {
        r.thr24 = d->thr24;
        r.scale = d->scale;
        r.seed_lo = (unsigned)(d->seed & 0xffffffffull);
        r.seed_hi = (unsigned)(d->seed >> 32);
        r.site = d->site;
        r.step = d->step;
        r.n_off = d->n_off;
        r.step_ptr = d->step_dev;
    }
    return r;
}

#define DP_LAUNCH_CHECK() ((int)hipGetLastError())
// every kernel launch of the library goes through DP_LAUNCH: dp_launch_count() reports launches per step in bench.py
extern unsigned long long dp_launches;
#define DP_LAUNCH(...) do { ++dp_launches; hipLaunchKernelGGL(__VA_ARGS__); } while (0)
