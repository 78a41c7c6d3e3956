// Fused |param * grad| channel reductions feeding the pruner (Taylor importance).
// HBM-bound: each call streams one weight and its accumulated gradient exactly once (8 bytes per
// parameter) and emits one score per channel.  Weight viewed as [R][C][T] (T = kernel taps, 1 for
// Linear / GroupNorm):
//   dim 0 (out-channel member): out[r] = sum_{c,t} f(w g)   -- rows are contiguous: one workgroup per row
//   dim 1 (in-channel member) : out[c] = sum_{r,t} f(w g)   -- one workgroup per column block, lanes walk (c,t)
// Reference arithmetic: ddpm_exp/torch_pruning/importance.py:375-434.
#include "dp_common.h"

__device__ __forceinline__ float wg_f(float w, float g, int mode) {
    const float p = w * g;
    if (mode == 0) { const float a = fabsf(p); return a * a; }   // (w*dw).abs().pow(2)
    if (mode == 1) return fabsf(p);
    if (mode == 4) return g * g;                                 // Fisher: dw.pow(2)
    return p;                                                   // mode 2: signed, abs after the sum; mode 5: signed
}

__global__ __launch_bounds__(256) void wg_rows_kernel(const float* __restrict__ w, const float* __restrict__ g, int R,
                                                      long long inner, int mode, float* __restrict__ out, int accumulate) {
    __shared__ float red[4];
    const int r = blockIdx.x;
    const float* wr = w + (long long)r * inner;
    const float* gr = g + (long long)r * inner;
    float s = 0.f;
    for (long long i = threadIdx.x; i < inner; i += 256) s += wg_f(wr[i], gr[i], mode);
    s = dp_block_sum_256(s, red);
    if (threadIdx.x == 0) {
        if (mode == 2) s = fabsf(s);
        out[r] = accumulate ? out[r] + s : s;
    }
}

// T > 1: per-(c,t) column sums go to a scratch row first, then taps are folded.
__global__ __launch_bounds__(256) void wg_cols_ct_kernel(const float* __restrict__ w, const float* __restrict__ g, int R,
                                                         long long CT, int mode, float* __restrict__ colsum) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const long long col = (long long)blockIdx.x * 64 + lane;
    float s = 0.f;
    if (col < CT) {
        for (int r = wave; r < R; r += 4) s += wg_f(w[(long long)r * CT + col], g[(long long)r * CT + col], mode);
    }
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && col < CT) colsum[col] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

__global__ void wg_fold_taps_kernel(const float* __restrict__ colsum, int C, int T, int mode, float* __restrict__ out,
                                    int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += colsum[(long long)c * T + t];
    if (mode == 2) s = fabsf(s);
    out[c] = accumulate ? out[c] + s : s;
}

__global__ void wg_gn_kernel(const float* __restrict__ w, const float* __restrict__ g, int R, float* __restrict__ out,
                             int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R) return;
    const float v = fabsf(w[i] * g[i]);
    out[i] = accumulate ? out[i] + v : v;
}

extern "C" int dp_wg_reduce(const float* w, const float* g, int R, int C, int T, int dim, int mode, float* out,
                            int accumulate, float* scratch, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (R <= 0 || C <= 0 || T <= 0) return 0;
    if (mode == 3) {
        DP_LAUNCH(wg_gn_kernel, dim3((R + 255) / 256), dim3(256), 0, st, w, g, R, out, accumulate);
        return DP_LAUNCH_CHECK();
    }
    if (dim == 0) {
        DP_LAUNCH(wg_rows_kernel, dim3(R), dim3(256), 0, st, w, g, R, (long long)C * T, mode, out, accumulate);
        return DP_LAUNCH_CHECK();
    }
    // dim 1: per-(c,t) column sums into `scratch` (C*T floats), then fold the T taps of each channel
    if (!scratch) return (int)hipErrorInvalidValue;
    const long long CT = (long long)C * T;
    DP_LAUNCH(wg_cols_ct_kernel, dim3((unsigned)((CT + 63) / 64)), dim3(256), 0, st, w, g, R, CT, mode, scratch);
    int e = DP_LAUNCH_CHECK();
    if (e) return e;
    DP_LAUNCH(wg_fold_taps_kernel, dim3((C + 255) / 256), dim3(256), 0, st, scratch, C, T, mode, out, accumulate);
    return DP_LAUNCH_CHECK();
}

// dst[i] += src[idx[i]]   (member sums are folded into the group score at the member's channel positions)
__global__ void gather_add_kernel(const float* __restrict__ src, const int64_t* __restrict__ idx, int n, float* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[idx[i]];
}
extern "C" int dp_gather_add(const float* src, const int64_t* idx, int n, float* dst, void* stream) {
    if (n <= 0) return 0;
    DP_LAUNCH(gather_add_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, idx, n, dst);
    return DP_LAUNCH_CHECK();
}
