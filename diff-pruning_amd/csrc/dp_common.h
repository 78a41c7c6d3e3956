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

__device__ __forceinline__ float dp_silu(float x) { return x / (1.0f + expf(-x)); }

// d/dx silu(x) = s * (1 + x * (1 - s)),  s = sigmoid(x)
__device__ __forceinline__ float dp_silu_grad(float x) {
    const float s = 1.0f / (1.0f + expf(-x));
    return s * (1.0f + x * (1.0f - s));
}

#define DP_LAUNCH_CHECK() ((int)hipGetLastError())
