// GroupNorm (+SiLU) forward / backward over channel-major (NCHW, explicit image stride) activations.
// HBM-bound: one workgroup per (image, group); the group's chunk (cpg*HW contiguous floats per source)
// is held in registers when it fits (<= 8192 elements: every CIFAR-32 shape), otherwise streamed
// three times (mean, variance, apply) with the re-reads served by L2.  Reductions: wave shuffles then
// a fixed-order 4-way LDS combine, so results are run-to-run deterministic.
// The input may be a *virtual concat* of two tensors along channels (up-block torch.cat, unet_2d_blocks.py:2035).
#include <cstdlib>
#include "dp_common.h"

#define GN_CACHE 32          // elements per thread kept in registers (256 threads -> 8192 per chunk)
#define GN_MAXCPG 64

struct GnSrc {
    const float* x1; const float* x2; int c_split; long long s1, s2;
};

__device__ __forceinline__ const float* gn_chan_ptr(const GnSrc& s, int n, int c, int HW) {
    return (c < s.c_split) ? s.x1 + (long long)n * s.s1 + (long long)c * HW
                           : s.x2 + (long long)n * s.s2 + (long long)(c - s.c_split) * HW;
}

__global__ __launch_bounds__(256) void gn_fwd_kernel(GnSrc src, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     int C, int HW, int G, float eps, int silu, float* __restrict__ y,
                                                     long long y_img_stride, float* __restrict__ stats, DpDrop drop) {
    __shared__ float red[4];
    const int n = blockIdx.x / G;
    const int g = blockIdx.x - n * G;
    const int cpg = C / G;
    const int cnt = cpg * HW;
    const long long didx0 = ((drop.n_off + n) * C + (long long)g * cpg) * HW;      // logical index of the group's first element
    const int tid = threadIdx.x;
    const bool cached = cnt <= 256 * GN_CACHE;
    const int c_base = g * cpg;

    float xr[GN_CACHE];
    float s = 0.f;
    if (cached) {
#pragma unroll
        for (int i = 0; i < GN_CACHE; ++i) {
            const int e = tid + 256 * i;
            float v = 0.f;
            if (e < cnt) {
                const int cl = e / HW;
                v = gn_chan_ptr(src, n, c_base + cl, HW)[e - cl * HW];
            }
            xr[i] = v;
            s += v;
        }
    } else {
        for (int e = tid; e < cnt; e += 256) {
            const int cl = e / HW;
            s += gn_chan_ptr(src, n, c_base + cl, HW)[e - cl * HW];
        }
    }
    const float mean = dp_block_sum_256(s, red) / (float)cnt;
    float q = 0.f;
    if (cached) {
#pragma unroll
        for (int i = 0; i < GN_CACHE; ++i) {
            const int e = tid + 256 * i;
            const float d = xr[i] - mean;
            if (e < cnt) q += d * d;
        }
    } else {
        for (int e = tid; e < cnt; e += 256) {
            const int cl = e / HW;
            const float d = gn_chan_ptr(src, n, c_base + cl, HW)[e - cl * HW] - mean;
            q += d * d;
        }
    }
    const float var = dp_block_sum_256(q, red) / (float)cnt;
    const float rstd = 1.0f / sqrtf(var + eps);
    if (tid == 0) {
        stats[(long long)blockIdx.x * 2 + 0] = mean;
        stats[(long long)blockIdx.x * 2 + 1] = rstd;
    }
    float* yb = y + (long long)n * y_img_stride + (long long)c_base * HW;
    if (cached) {
#pragma unroll
        for (int i = 0; i < GN_CACHE; ++i) {
            const int e = tid + 256 * i;
            if (e < cnt) {
                const int c = c_base + e / HW;
                float v = (xr[i] - mean) * rstd * gamma[c] + beta[c];
                if (silu) v = dp_silu(v);
                if (drop.thr24) v *= dp_drop1(drop, didx0 + e);
                yb[e] = v;
            }
        }
    } else {
        for (int e = tid; e < cnt; e += 256) {
            const int cl = e / HW;
            const int c = c_base + cl;
            float v = (gn_chan_ptr(src, n, c, HW)[e - cl * HW] - mean) * rstd * gamma[c] + beta[c];
            if (silu) v = dp_silu(v);
            if (drop.thr24) v *= dp_drop1(drop, didx0 + e);
            yb[e] = v;
        }
    }
}

// float4 variant (HW % 4 == 0, 16-byte aligned planes): a float4 never straddles a channel.  NI = cached float4 per thread (1, 2,
// 4, 8: groups of up to 1024 / 2048 / 4096 / 8192 elements; no dead iterations), 0 = the streaming form for larger groups.
template <int NI>
__global__ __launch_bounds__(256) void gn_fwd_vec4_kernel(GnSrc src, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int C, int HW, int G, float eps,
                                                          int silu, float* __restrict__ y, long long y_img_stride,
                                                          float* __restrict__ stats, DpDrop drop) {
    __shared__ float red[4];
    const int n = blockIdx.x / G;
    const int g = blockIdx.x - n * G;
    const int cpg = C / G;
    const int cnt4 = cpg * HW / 4;
    const long long didx0 = ((drop.n_off + n) * C + (long long)g * cpg) * HW;
    const int HW4 = HW / 4;
    const int tid = threadIdx.x;
    constexpr bool cached = NI > 0;
    constexpr int NC = NI > 0 ? NI : 1;
    const int c_base = g * cpg;
    float4 xr[NC];
    float gac[NC], bec[NC];          // gamma / beta of the channel each cached float4 belongs to (round 5: loaded
    float s = 0.f;                                       // with the data, not one dependent L2 round trip per store in the apply loop)
    if (cached) {
#pragma unroll
        for (int i = 0; i < NC; ++i) {         // branch-free: lanes past the end read the group's last float4, zeroed below
            const int e = tid + 256 * i;
            const int ec = e < cnt4 ? e : cnt4 - 1;
            const int cl = ec / HW4;
            xr[i] = reinterpret_cast<const float4*>(gn_chan_ptr(src, n, c_base + cl, HW))[ec - cl * HW4];
            gac[i] = gamma[c_base + cl];
            bec[i] = beta[c_base + cl];
        }
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            if (tid + 256 * i >= cnt4) xr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 v = xr[i];
            s += (v.x + v.y) + (v.z + v.w);
        }
    } else {
        for (int e = tid; e < cnt4; e += 256) {
            const int cl = e / HW4;
            const float4 v = reinterpret_cast<const float4*>(gn_chan_ptr(src, n, c_base + cl, HW))[e - cl * HW4];
            s += (v.x + v.y) + (v.z + v.w);
        }
    }
    const float mean = dp_block_sum_256(s, red) / (float)(cnt4 * 4);
    float q = 0.f;
    if (cached) {
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int e = tid + 256 * i;
            if (e < cnt4) {
                const float4 v = xr[i];
                const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
                q += (a * a + b * b) + (c * c + d * d);
            }
        }
    } else {
        for (int e = tid; e < cnt4; e += 256) {
            const int cl = e / HW4;
            const float4 v = reinterpret_cast<const float4*>(gn_chan_ptr(src, n, c_base + cl, HW))[e - cl * HW4];
            const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
            q += (a * a + b * b) + (c * c + d * d);
        }
    }
    const float var = dp_block_sum_256(q, red) / (float)(cnt4 * 4);
    const float rstd = 1.0f / sqrtf(var + eps);
    if (tid == 0) {
        stats[(long long)blockIdx.x * 2 + 0] = mean;
        stats[(long long)blockIdx.x * 2 + 1] = rstd;
    }
    float4* yb = reinterpret_cast<float4*>(y + (long long)n * y_img_stride + (long long)c_base * HW);
    auto apply = [&](float4 v, float ga, float be, int e4) {
        float4 o;
        o.x = (v.x - mean) * rstd * ga + be;
        o.y = (v.y - mean) * rstd * ga + be;
        o.z = (v.z - mean) * rstd * ga + be;
        o.w = (v.w - mean) * rstd * ga + be;
        if (silu) { o.x = dp_silu(o.x); o.y = dp_silu(o.y); o.z = dp_silu(o.z); o.w = dp_silu(o.w); }
        if (drop.thr24) { const float4 m = dp_drop4(drop, didx0 + 4ll * e4); o.x *= m.x; o.y *= m.y; o.z *= m.z; o.w *= m.w; }
        return o;
    };
    if (cached) {
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int e = tid + 256 * i;
            if (e < cnt4) yb[e] = apply(xr[i], gac[i], bec[i], e);
        }
    } else {
        for (int e = tid; e < cnt4; e += 256) {
            const int cl = e / HW4;
            yb[e] = apply(reinterpret_cast<const float4*>(gn_chan_ptr(src, n, c_base + cl, HW))[e - cl * HW4], gamma[c_base + cl],
                          beta[c_base + cl], e);
        }
    }
}

// One WAVEFRONT per (image, group) for groups of up to 2048 elements (the 16 x 16 / 8 x 8 / 4 x 4 layers of the CIFAR UNet): no
// LDS, no barrier -- the reductions are wave shuffles -- and the four groups of a workgroup run independently, so a CU holds up
// to 32 groups in different phases instead of 8 workgroups that each stall at two barriers.  [measured, round 3,
// tools/bench_gn.py, B = 256: 256 ch @ 8x8 23.3 -> 13.0 us, 256 ch @ 4x4 18.9 -> 12.5 us forward; groups of 4096 elements are
// FASTER on the workgroup kernel (54 vs 59 us) and stay there.]  Lane l holds float4 elements l, l + 64, ... of the group chunk
// (coalesced 1 KB wave loads); same two-pass mean / variance as the workgroup kernels (shuffle tree instead of the LDS combine).
#define GN_WAVE_NV 8
// NV = float4 steps a lane takes (1, 2, 4 or 8: groups of up to 256 / 512 / 1024 / 2048 elements), so the 8 x 8 and 4 x 4 layers do
// not walk dead iterations.  Loads are branch-free (lanes past the group's end read its last element and are masked afterwards):
// hipcc does not hoist a load out of an exec-masked region, and a load per `if (live)` is a memory round trip per iteration.
template <int NV>
__global__ __launch_bounds__(256) void gn_fwd_wave_kernel(GnSrc src, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int C, int HW, int G, float eps,
                                                          int silu, float* __restrict__ y, long long y_img_stride,
                                                          float* __restrict__ stats, DpDrop drop, int NG) {
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= NG) return;
    const int lane = threadIdx.x & 63;
    const int n = wid / G;
    const int g = wid - n * G;
    const int cpg = C / G;
    const int HW4 = HW / 4;
    const int cnt4 = cpg * HW4;
    const int c_base = g * cpg;
    const long long didx0 = ((drop.n_off + n) * C + (long long)g * cpg) * HW;
    float4 xr[NV];
    float ga[NV], be[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int e = lane + 64 * i;
        const int ec = e < cnt4 ? e : cnt4 - 1;
        const int cl = ec / HW4;
        xr[i] = reinterpret_cast<const float4*>(gn_chan_ptr(src, n, c_base + cl, HW))[ec - cl * HW4];
        ga[i] = gamma[c_base + cl];
        be[i] = beta[c_base + cl];
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (lane + 64 * i >= cnt4) xr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 v = xr[i];
        s += (v.x + v.y) + (v.z + v.w);
    }
    const float mean = dp_wave_sum(s) / (float)(cnt4 * 4);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (lane + 64 * i < cnt4) {
            const float4 v = xr[i];
            const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
            q += (a * a + b * b) + (c * c + d * d);
        }
    }
    const float var = dp_wave_sum(q) / (float)(cnt4 * 4);
    const float rstd = 1.0f / sqrtf(var + eps);
    if (lane == 0) {
        stats[(long long)wid * 2 + 0] = mean;
        stats[(long long)wid * 2 + 1] = rstd;
    }
    float4* yb = reinterpret_cast<float4*>(y + (long long)n * y_img_stride + (long long)c_base * HW);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int e = lane + 64 * i;
        if (e < cnt4) {
            const float4 v = xr[i];
            float4 o;
            o.x = (v.x - mean) * rstd * ga[i] + be[i];
            o.y = (v.y - mean) * rstd * ga[i] + be[i];
            o.z = (v.z - mean) * rstd * ga[i] + be[i];
            o.w = (v.w - mean) * rstd * ga[i] + be[i];
            if (silu) { o.x = dp_silu(o.x); o.y = dp_silu(o.y); o.z = dp_silu(o.z); o.w = dp_silu(o.w); }
            if (drop.thr24) { const float4 m = dp_drop4(drop, didx0 + 4ll * e); o.x *= m.x; o.y *= m.y; o.z *= m.z; o.w *= m.w; }
            yb[e] = o;
        }
    }
}

extern "C" int dp_groupnorm_silu_fwd(const float* x1, const float* x2, int c_split, long long x1_img_stride,
                                     long long x2_img_stride, const float* gamma, const float* beta, int N, int C, int HW,
                                     int G, float eps, int silu, float* y, long long y_img_stride, float* stats,
                                     const dp_dropout* drop, void* stream) {
    if (N <= 0 || C <= 0) return 0;
    if (C % G) return (int)hipErrorInvalidValue;
    const DpDrop dd = dp_drop_host(drop);
    GnSrc s{x1, x2, x2 ? c_split : C, x1_img_stride, x2_img_stride};
    const bool vec4 = (HW % 4 == 0) && (x1_img_stride % 4 == 0) && (x2_img_stride % 4 == 0) && (y_img_stride % 4 == 0) &&
                      (((uintptr_t)x1 | (uintptr_t)x2 | (uintptr_t)y) % 16 == 0);
    static const bool no_wave = getenv("DP_NO_GN_WAVE") != nullptr;
    if (vec4 && !no_wave && (long long)(C / G) * HW <= 256 * GN_WAVE_NV && N * G >= 1024)      // <= 512 float4 per group
    {
        const int cnt4 = (C / G) * (HW / 4);
#define GN_FWD_WAVE(NV_) DP_LAUNCH((gn_fwd_wave_kernel<NV_>), dim3((N * G + 3) / 4), dim3(256), 0, (hipStream_t)stream, s, gamma, \
                                   beta, C, HW, G, eps, silu, y, y_img_stride, stats, dd, N * G)
        if (cnt4 <= 64) GN_FWD_WAVE(1);
        else if (cnt4 <= 128) GN_FWD_WAVE(2);
        else if (cnt4 <= 256) GN_FWD_WAVE(4);
        else GN_FWD_WAVE(8);
#undef GN_FWD_WAVE
    }
    else if (vec4) {
        const long long cnt4 = (long long)(C / G) * (HW / 4);
#define GN_FWD_V4(NI_) DP_LAUNCH((gn_fwd_vec4_kernel<NI_>), dim3(N * G), dim3(256), 0, (hipStream_t)stream, s, gamma, beta, C, HW, G, \
                                 eps, silu, y, y_img_stride, stats, dd)
        if (cnt4 <= 256) GN_FWD_V4(1);
        else if (cnt4 <= 512) GN_FWD_V4(2);
        else if (cnt4 <= 1024) GN_FWD_V4(4);
        else if (cnt4 <= 2048) GN_FWD_V4(8);
        else GN_FWD_V4(0);
#undef GN_FWD_V4
    }
    else
        DP_LAUNCH(gn_fwd_kernel, dim3(N * G), dim3(256), 0, (hipStream_t)stream, s, gamma, beta, C, HW, G, eps, silu,
                           y, y_img_stride, stats, dd);
    return DP_LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------
// backward
//   yhat = xhat*gamma + beta;  dy = silu ? dz * silu'(yhat) : dz
//   S1_c = sum_hw dy, S2_c = sum_hw dy*xhat          -> pws (per image, reduced over images later)
//   a = sum_c gamma_c S1_c, b = sum_c gamma_c S2_c, M = cpg*HW
//   dx = rstd * (gamma_c*dy - a/M - xhat*b/M)  (+ add1) (+ add2)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_bwd_kernel(GnSrc src, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const float* __restrict__ stats, const float* __restrict__ dz,
                                                     long long dz_img_stride, int C, int HW, int G, int silu,
                                                     float* __restrict__ dx, long long dx_img_stride,
                                                     const float* __restrict__ add1, long long add1_s,
                                                     const float* __restrict__ add2, long long add2_s,
                                                     float* __restrict__ pws, DpDrop drop, float* __restrict__ rows) {
    __shared__ float s1[GN_MAXCPG], s2[GN_MAXCPG];
    const int n = blockIdx.x / G;
    const int g = blockIdx.x - n * G;
    const int cpg = C / G;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int c_base = g * cpg;
    const float mean = stats[(long long)blockIdx.x * 2 + 0];
    const float rstd = stats[(long long)blockIdx.x * 2 + 1];
    const float* dzb = dz + (long long)n * dz_img_stride;
    const long long didx0 = ((drop.n_off + n) * C + c_base) * (long long)HW;

    // pass 1: per-channel sums, one wavefront per channel
    for (int cl = wave; cl < cpg; cl += 4) {
        const int c = c_base + cl;
        const float* xp = gn_chan_ptr(src, n, c, HW);
        const float* dp = dzb + (long long)c * HW;
        const float ga = gamma[c], be = beta[c];
        float a1 = 0.f, a2 = 0.f;
        for (int i = lane; i < HW; i += 64) {
            const float xh = (xp[i] - mean) * rstd;
            float d = dp[i];
            if (drop.thr24) d *= dp_drop1(drop, didx0 + (long long)cl * HW + i);
            if (silu) d *= dp_silu_grad(xh * ga + be);
            a1 += d;
            a2 += d * xh;
        }
        a1 = dp_wave_sum(a1);
        a2 = dp_wave_sum(a2);
        if (lane == 0) {
            s1[cl] = a1;
            s2[cl] = a2;
            pws[((long long)n * C + c) * 2 + 0] = a1;
            pws[((long long)n * C + c) * 2 + 1] = a2;
        }
    }
    __syncthreads();
    float a = 0.f, b = 0.f;
    for (int cl = 0; cl < cpg; ++cl) {
        const float ga = gamma[c_base + cl];
        a += ga * s1[cl];
        b += ga * s2[cl];
    }
    const float invM = 1.0f / (float)(cpg * HW);
    a *= invM;
    b *= invM;

    // pass 2: one wavefront per channel (as pass 1), so that the per-(image, channel) sum of the OUTPUT -- the bias / time-
    // embedding-projection gradient rows of the layer below -- falls out of a wave reduction (rows, optional)
    float* dxb = dx + (long long)n * dx_img_stride + (long long)c_base * HW;
    const float* a1b = add1 ? add1 + (long long)n * add1_s + (long long)c_base * HW : nullptr;
    const float* a2b = add2 ? add2 + (long long)n * add2_s + (long long)c_base * HW : nullptr;
    for (int cl = wave; cl < cpg; cl += 4) {
        const int c = c_base + cl;
        const float* xp = gn_chan_ptr(src, n, c, HW);
        const float ga = gamma[c], be = beta[c];
        float rs = 0.f;
        for (int i = lane; i < HW; i += 64) {
            const int e = cl * HW + i;
            const float xh = (xp[i] - mean) * rstd;
            float d = dzb[(long long)c_base * HW + e];
            if (drop.thr24) d *= dp_drop1(drop, didx0 + e);
            if (silu) d *= dp_silu_grad(xh * ga + be);
            float v = rstd * (ga * d - a - xh * b);
            if (a1b) v += a1b[e];
            if (a2b) v += a2b[e];
            dxb[e] = v;
            rs += v;
        }
        if (rows) {
            rs = dp_wave_sum(rs);
            if (lane == 0) rows[(long long)n * C + c] = rs;
        }
    }
}

__global__ __launch_bounds__(256) void gn_bwd_vec4_kernel(GnSrc src, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ stats,
                                                          const float* __restrict__ dz, long long dz_img_stride, int C,
                                                          int HW, int G, int silu, float* __restrict__ dx,
                                                          long long dx_img_stride, const float* __restrict__ add1,
                                                          long long add1_s, const float* __restrict__ add2, long long add2_s,
                                                          float* __restrict__ pws, DpDrop drop, float* __restrict__ rows) {
    __shared__ float s1[GN_MAXCPG], s2[GN_MAXCPG];
    const int n = blockIdx.x / G;
    const int g = blockIdx.x - n * G;
    const int cpg = C / G;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int c_base = g * cpg;
    const int HW4 = HW / 4;
    const float mean = stats[(long long)blockIdx.x * 2 + 0];
    const float rstd = stats[(long long)blockIdx.x * 2 + 1];
    const float* dzb = dz + (long long)n * dz_img_stride;
    const long long didx0 = ((drop.n_off + n) * C + c_base) * (long long)HW;
    auto dyv = [&](float4 xv, float4 dv, float ga, float be, float4& xh, long long e4) {      // e4: float4 index in the group chunk
        xh.x = (xv.x - mean) * rstd; xh.y = (xv.y - mean) * rstd; xh.z = (xv.z - mean) * rstd; xh.w = (xv.w - mean) * rstd;
        if (drop.thr24) { const float4 m = dp_drop4(drop, didx0 + 4 * e4); dv.x *= m.x; dv.y *= m.y; dv.z *= m.z; dv.w *= m.w; }
        if (silu) {
            dv.x *= dp_silu_grad(xh.x * ga + be); dv.y *= dp_silu_grad(xh.y * ga + be);
            dv.z *= dp_silu_grad(xh.z * ga + be); dv.w *= dp_silu_grad(xh.w * ga + be);
        }
        return dv;
    };
    for (int cl = wave; cl < cpg; cl += 4) {
        const int c = c_base + cl;
        const float4* xp = reinterpret_cast<const float4*>(gn_chan_ptr(src, n, c, HW));
        const float4* dp = reinterpret_cast<const float4*>(dzb + (long long)c * HW);
        const float ga = gamma[c], be = beta[c];
        float a1 = 0.f, a2 = 0.f;
        for (int i = lane; i < HW4; i += 64) {
            float4 xh;
            const float4 d = dyv(xp[i], dp[i], ga, be, xh, (long long)cl * HW4 + i);
            a1 += (d.x + d.y) + (d.z + d.w);
            a2 += (d.x * xh.x + d.y * xh.y) + (d.z * xh.z + d.w * xh.w);
        }
        a1 = dp_wave_sum(a1);
        a2 = dp_wave_sum(a2);
        if (lane == 0) {
            s1[cl] = a1;
            s2[cl] = a2;
            pws[((long long)n * C + c) * 2 + 0] = a1;
            pws[((long long)n * C + c) * 2 + 1] = a2;
        }
    }
    __syncthreads();
    float a = 0.f, b = 0.f;
    for (int cl = 0; cl < cpg; ++cl) {
        const float ga = gamma[c_base + cl];
        a += ga * s1[cl];
        b += ga * s2[cl];
    }
    const float invM = 1.0f / (float)(cpg * HW);
    a *= invM;
    b *= invM;
    // pass 2: one wavefront per channel PAIR (cl, cl + 4): both channels' loads are in flight together (a single channel per
    // iteration left one 16-byte load per lane outstanding: 88 instead of 69 us per launch); the per-(image, channel) sums of
    // the output (rows, optional; see gn_bwd_kernel) fall out of a wave reduction
    float4* dxb = reinterpret_cast<float4*>(dx + (long long)n * dx_img_stride + (long long)c_base * HW);
    const float4* dzc = reinterpret_cast<const float4*>(dzb + (long long)c_base * HW);
    const float4* a1b = add1 ? reinterpret_cast<const float4*>(add1 + (long long)n * add1_s + (long long)c_base * HW) : nullptr;
    const float4* a2b = add2 ? reinterpret_cast<const float4*>(add2 + (long long)n * add2_s + (long long)c_base * HW) : nullptr;
    auto finish = [&](float4 xv, float4 dv, float4 t1, float4 t2, float ga, float be, long long e) {
        float4 xh;
        const float4 d = dyv(xv, dv, ga, be, xh, e);
        float4 v;
        v.x = rstd * (ga * d.x - a - xh.x * b) + t1.x + t2.x;
        v.y = rstd * (ga * d.y - a - xh.y * b) + t1.y + t2.y;
        v.z = rstd * (ga * d.z - a - xh.z * b) + t1.z + t2.z;
        v.w = rstd * (ga * d.w - a - xh.w * b) + t1.w + t2.w;
        return v;
    };
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int cl = wave; cl < cpg; cl += 8) {
        const int cB = cl + 4 < cpg ? cl + 4 : cl;                    // second channel of the pair (== cl when there is none)
        const bool two = cl + 4 < cpg;
        const float4* xpA = reinterpret_cast<const float4*>(gn_chan_ptr(src, n, c_base + cl, HW));
        const float4* xpB = reinterpret_cast<const float4*>(gn_chan_ptr(src, n, c_base + cB, HW));
        const float gaA = gamma[c_base + cl], beA = beta[c_base + cl], gaB = gamma[c_base + cB], beB = beta[c_base + cB];
        float rA = 0.f, rB = 0.f;
        for (int i = lane; i < HW4; i += 64) {
            const int eA = cl * HW4 + i, eB = cB * HW4 + i;
            const float4 xa = xpA[i], da = dzc[eA], xb = xpB[i], db = dzc[eB];
            const float4 pa = a1b ? a1b[eA] : zero4, qa = a2b ? a2b[eA] : zero4;
            const float4 pb = a1b ? a1b[eB] : zero4, qb = a2b ? a2b[eB] : zero4;
            const float4 va = finish(xa, da, pa, qa, gaA, beA, eA);
            dxb[eA] = va;
            rA += (va.x + va.y) + (va.z + va.w);
            if (two) {
                const float4 vb = finish(xb, db, pb, qb, gaB, beB, eB);
                dxb[eB] = vb;
                rB += (vb.x + vb.y) + (vb.z + vb.w);
            }
        }
        if (rows) {
            rA = dp_wave_sum(rA);
            rB = dp_wave_sum(rB);
            if (lane == 0) {
                rows[(long long)n * C + c_base + cl] = rA;
                if (two) rows[(long long)n * C + c_base + cB] = rB;
            }
        }
    }
}

// gn_bwd_vec4_kernel with x-hat and dy KEPT IN REGISTERS between the two passes (round 4): groups of up to 8 channels x 1024
// pixels (every 32 x 32 layer of the CIFAR UNet incl. the 256-channel concat inputs).  A wavefront owns channels wave and wave + 4
// in BOTH passes, lane l the float4 elements l, l + 64, l + 128, l + 192 of a channel plane -- the assignment and the summation
// order of gn_bwd_vec4_kernel, so the per-channel sums, the group terms and dx are the same expressions on the same values; what
// goes away is the second pass's re-read of x and dz (through L2, behind the barrier) and its second evaluation of silu' (two
// exponentials and a division per element).
// Round 5: loads branch-free and batched (hipcc keeps a load inside its `if (i < HW4)` region and waits for it there: the ISA
// read two loads per wait in pass 1 and eight scalarized dword loads per store in pass 2); the addends' presence is a template
// parameter and their loads go out BEFORE the barrier, so their latency hides behind the wait for the other wavefronts' sums.
template <int NCH, bool A1, bool A2>
__global__ __launch_bounds__(256) void gn_bwd_vec4c_kernel(GnSrc src, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ stats,
                                                           const float* __restrict__ dz, long long dz_img_stride, int C,
                                                           int HW, int G, int silu, float* __restrict__ dx,
                                                           long long dx_img_stride, const float* __restrict__ add1,
                                                           long long add1_s, const float* __restrict__ add2, long long add2_s,
                                                           float* __restrict__ pws, DpDrop drop, float* __restrict__ rows) {
    __shared__ float s1[GN_MAXCPG], s2[GN_MAXCPG];
    const int n = blockIdx.x / G;
    const int g = blockIdx.x - n * G;
    const int cpg = C / G;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int c_base = g * cpg;
    const int HW4 = HW / 4;
    const float mean = stats[(long long)blockIdx.x * 2 + 0];
    const float rstd = stats[(long long)blockIdx.x * 2 + 1];
    const float* dzb = dz + (long long)n * dz_img_stride;
    const long long didx0 = ((drop.n_off + n) * C + c_base) * (long long)HW;
    float4 xh[NCH][4], dd[NCH][4];
    float gak[NCH], bek[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {                             // every load of the workgroup's pass 1 in flight before the first use
        const int cl = wave + 4 * k;
        const int c = c_base + (cl < cpg ? cl : cpg - 1);
        const float4* xp = reinterpret_cast<const float4*>(gn_chan_ptr(src, n, c, HW));
        const float4* dp = reinterpret_cast<const float4*>(dzb + (long long)c * HW);
        gak[k] = gamma[c];
        bek[k] = beta[c];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = lane + 64 * j;
            const int ic = i < HW4 ? i : HW4 - 1;
            xh[k][j] = xp[ic];
            dd[k][j] = dp[ic];
        }
    }
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int cl = wave + 4 * k;
        if (cl < cpg) {
            const int c = c_base + cl;
            const float ga = gak[k], be = bek[k];
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = lane + 64 * j;
                if (i < HW4) {
                    const float4 xv = xh[k][j];
                    float4 h, d = dd[k][j];
                    h.x = (xv.x - mean) * rstd; h.y = (xv.y - mean) * rstd; h.z = (xv.z - mean) * rstd; h.w = (xv.w - mean) * rstd;
                    if (drop.thr24) {
                        const float4 m = dp_drop4(drop, didx0 + 4 * ((long long)cl * HW4 + i));
                        d.x *= m.x; d.y *= m.y; d.z *= m.z; d.w *= m.w;
                    }
                    if (silu) {
                        d.x *= dp_silu_grad(h.x * ga + be); d.y *= dp_silu_grad(h.y * ga + be);
                        d.z *= dp_silu_grad(h.z * ga + be); d.w *= dp_silu_grad(h.w * ga + be);
                    }
                    xh[k][j] = h;
                    dd[k][j] = d;
                    a1 += (d.x + d.y) + (d.z + d.w);
                    a2 += (d.x * h.x + d.y * h.y) + (d.z * h.z + d.w * h.w);
                }
            }
            a1 = dp_wave_sum(a1);
            a2 = dp_wave_sum(a2);
            if (lane == 0) {
                s1[cl] = a1;
                s2[cl] = a2;
                pws[((long long)n * C + c) * 2 + 0] = a1;
                pws[((long long)n * C + c) * 2 + 1] = a2;
            }
        }
    }
    const float4* a1b = A1 ? reinterpret_cast<const float4*>(add1 + (long long)n * add1_s + (long long)c_base * HW) : nullptr;
    const float4* a2b = A2 ? reinterpret_cast<const float4*>(add2 + (long long)n * add2_s + (long long)c_base * HW) : nullptr;
    float4 t1[NCH][4], t2[NCH][4];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int cl = wave + 4 * k;
        const int clc = cl < cpg ? cl : cpg - 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = lane + 64 * j;
            const int ec = clc * HW4 + (i < HW4 ? i : HW4 - 1);
            if (A1) t1[k][j] = a1b[ec];
            if (A2) t2[k][j] = a2b[ec];
        }
    }
    __syncthreads();
    float a = 0.f, b = 0.f;
    for (int cl = 0; cl < cpg; ++cl) {
        const float ga = gamma[c_base + cl];
        a += ga * s1[cl];
        b += ga * s2[cl];
    }
    const float invM = 1.0f / (float)(cpg * HW);
    a *= invM;
    b *= invM;
    float4* dxb = reinterpret_cast<float4*>(dx + (long long)n * dx_img_stride + (long long)c_base * HW);
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int cl = wave + 4 * k;
        if (cl < cpg) {
            const float ga = gak[k];
            float r = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = lane + 64 * j;
                if (i < HW4) {
                    const int e = cl * HW4 + i;
                    const float4 u1 = A1 ? t1[k][j] : zero4, u2 = A2 ? t2[k][j] : zero4;
                    const float4 d = dd[k][j], h = xh[k][j];
                    float4 v;
                    v.x = rstd * (ga * d.x - a - h.x * b) + u1.x + u2.x;
                    v.y = rstd * (ga * d.y - a - h.y * b) + u1.y + u2.y;
                    v.z = rstd * (ga * d.z - a - h.z * b) + u1.z + u2.z;
                    v.w = rstd * (ga * d.w - a - h.w * b) + u1.w + u2.w;
                    dxb[e] = v;
                    r += (v.x + v.y) + (v.z + v.w);
                }
            }
            if (rows) {
                r = dp_wave_sum(r);
                if (lane == 0) rows[(long long)n * C + c_base + cl] = r;
            }
        }
    }
}

// Backward, one wavefront per (image, group): HW / 4 is a power of two <= 64, so every wave-wide float4 step covers 64 / HW4
// whole channels and the per-channel sums are segmented xor-shuffle reductions over HW4 lanes.  x-hat and dy stay in registers
// between the two phases (the workgroup kernel re-reads them through L2 behind a barrier).  Fixed shuffle order: deterministic.
// Round 5: the ISA of the round-3 kernel read `load, s_waitcnt vmcnt(0), load, s_waitcnt vmcnt(0), ...` -- hipcc does not hoist a
// load out of an `if (live)` region, and the `ptr ? ptr[e] : 0` form of the optional addends became four dword loads each --
// i.e. ~25 serialized memory round trips per wavefront (1.6 TB/s, 20 % of the HBM roofline by the PMC bytes).  Now: NV (float4 steps
// per lane) and the presence of the addends are template parameters, every load is branch-free (lanes past the end of the group
// read its last element and are zeroed afterwards) and issued before the first use; the segmented reductions of all steps advance
// together (2 NV independent shuffles per level instead of NV dependent chains).  Same assignment, expressions and summation order.
template <int NV, bool A1, bool A2>
__global__ __launch_bounds__(256) void gn_bwd_wave_kernel(GnSrc src, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ stats,
                                                          const float* __restrict__ dz, long long dz_img_stride, int C,
                                                          int HW, int G, int silu, float* __restrict__ dx,
                                                          long long dx_img_stride, const float* __restrict__ add1,
                                                          long long add1_s, const float* __restrict__ add2, long long add2_s,
                                                          float* __restrict__ pws, DpDrop drop, float* __restrict__ rows, int NG) {
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= NG) return;
    const int lane = threadIdx.x & 63;
    const int n = wid / G;
    const int g = wid - n * G;
    const int cpg = C / G;
    const int HW4 = HW / 4;
    const int lg4 = 31 - __builtin_clz(HW4);
    const int cnt4 = cpg * HW4;
    const int c_base = g * cpg;
    const float mean = stats[(long long)wid * 2 + 0];
    const float rstd = stats[(long long)wid * 2 + 1];
    const long long didx0 = ((drop.n_off + n) * C + c_base) * (long long)HW;
    const float4* dzc = reinterpret_cast<const float4*>(dz + (long long)n * dz_img_stride + (long long)c_base * HW);
    const float4* a1b = A1 ? reinterpret_cast<const float4*>(add1 + (long long)n * add1_s + (long long)c_base * HW) : nullptr;
    const float4* a2b = A2 ? reinterpret_cast<const float4*>(add2 + (long long)n * add2_s + (long long)c_base * HW) : nullptr;
    const bool leader = (lane & (HW4 - 1)) == 0;
    constexpr bool EARLY = NV <= 4;              // the addends' loads go out with the operands' (registers allow it up to 4 steps)
    float4 xh[NV], dv[NV], t1[NV], t2[NV];
    float gam[NV], bet[NV];
    // ---- phase 0: every load of the wavefront in flight
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int e = lane + 64 * i;
        const int ec = e < cnt4 ? e : cnt4 - 1;
        const int cl = ec >> lg4;
        const int c = c_base + cl;
        xh[i] = reinterpret_cast<const float4*>(gn_chan_ptr(src, n, c, HW))[ec - (cl << lg4)];
        dv[i] = dzc[ec];
        gam[i] = gamma[c];
        bet[i] = beta[c];
        if (A1 && EARLY) t1[i] = a1b[ec];
        if (A2 && EARLY) t2[i] = a2b[ec];
    }
    // ---- phase 1: dy, x-hat, per-channel sums
    float s1[NV], s2[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int e = lane + 64 * i;
        const bool live = e < cnt4;
        const float ga = gam[i], be = bet[i];
        const float4 xv = xh[i];
        float4 d = dv[i];
        float4 h;
        h.x = (xv.x - mean) * rstd; h.y = (xv.y - mean) * rstd; h.z = (xv.z - mean) * rstd; h.w = (xv.w - mean) * rstd;
        if (drop.thr24) { const float4 m = dp_drop4(drop, didx0 + 4ll * (live ? e : cnt4 - 1)); d.x *= m.x; d.y *= m.y; d.z *= m.z; d.w *= m.w; }
        if (silu) {
            d.x *= dp_silu_grad(h.x * ga + be); d.y *= dp_silu_grad(h.y * ga + be);
            d.z *= dp_silu_grad(h.z * ga + be); d.w *= dp_silu_grad(h.w * ga + be);
        }
        if (!live) { h = make_float4(0.f, 0.f, 0.f, 0.f); d = h; }
        xh[i] = h; dv[i] = d;
        s1[i] = (d.x + d.y) + (d.z + d.w);
        s2[i] = (d.x * h.x + d.y * h.y) + (d.z * h.z + d.w * h.w);
    }
    for (int o = HW4 >> 1; o > 0; o >>= 1) {                         // per-channel sums: segments of HW4 lanes, all steps together
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            s1[i] += __shfl_xor(s1[i], o, 64);
            s2[i] += __shfl_xor(s2[i], o, 64);
        }
    }
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int e = lane + 64 * i;
        if (e < cnt4 && leader) {
            const int c = c_base + (e >> lg4);
            pws[((long long)n * C + c) * 2 + 0] = s1[i];
            pws[((long long)n * C + c) * 2 + 1] = s2[i];
            a += gam[i] * s1[i];
            b += gam[i] * s2[i];
        }
    }
    if (!EARLY) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = lane + 64 * i;
            const int ec = e < cnt4 ? e : cnt4 - 1;
            if (A1) t1[i] = a1b[ec];
            if (A2) t2[i] = a2b[ec];
        }
    }
    const float invM = 1.0f / (float)(cpg * HW);
    a = dp_wave_sum(a) * invM;
    b = dp_wave_sum(b) * invM;
    // ---- phase 2: dx (+ addends), its per-channel sums
    float4* dxb = reinterpret_cast<float4*>(dx + (long long)n * dx_img_stride + (long long)c_base * HW);
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float r[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int e = lane + 64 * i;
        const bool live = e < cnt4;
        const float4 u1 = A1 ? t1[i] : zero4, u2 = A2 ? t2[i] : zero4;
        const float ga = gam[i];
        const float4 d = dv[i], h = xh[i];
        float4 v;
        v.x = rstd * (ga * d.x - a - h.x * b) + u1.x + u2.x;
        v.y = rstd * (ga * d.y - a - h.y * b) + u1.y + u2.y;
        v.z = rstd * (ga * d.z - a - h.z * b) + u1.z + u2.z;
        v.w = rstd * (ga * d.w - a - h.w * b) + u1.w + u2.w;
        if (live) dxb[e] = v;
        r[i] = live ? (v.x + v.y) + (v.z + v.w) : 0.f;
    }
    if (rows) {
        for (int o = HW4 >> 1; o > 0; o >>= 1) {
#pragma unroll
            for (int i = 0; i < NV; ++i) r[i] += __shfl_xor(r[i], o, 64);
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = lane + 64 * i;
            if (e < cnt4 && leader) rows[(long long)n * C + c_base + (e >> lg4)] = r[i];
        }
    }
}

extern "C" int dp_groupnorm_silu_bwd(const float* x1, const float* x2, int c_split, long long x1_img_stride,
                                     long long x2_img_stride, const float* gamma, const float* beta, const float* stats,
                                     const float* dz, long long dz_img_stride, int N, int C, int HW, int G, int silu,
                                     float* dx, long long dx_img_stride, const float* add1, long long add1_img_stride,
                                     const float* add2, long long add2_img_stride, float* pws, const dp_dropout* drop,
                                     float* rows, void* stream) {
    if (N <= 0 || C <= 0) return 0;
    if (C % G || C / G > GN_MAXCPG) return (int)hipErrorInvalidValue;
    const DpDrop dd = dp_drop_host(drop);
    GnSrc s{x1, x2, x2 ? c_split : C, x1_img_stride, x2_img_stride};
    const bool vec4 = (HW % 4 == 0) && ((x1_img_stride | x2_img_stride | dz_img_stride | dx_img_stride | add1_img_stride |
                                         add2_img_stride) % 4 == 0) &&
                      (((uintptr_t)x1 | (uintptr_t)x2 | (uintptr_t)dz | (uintptr_t)dx | (uintptr_t)add1 | (uintptr_t)add2) % 16 == 0);
    static const bool no_wave = getenv("DP_NO_GN_WAVE") != nullptr;
    const int HW4 = HW / 4;
    if (vec4 && !no_wave && HW4 >= 1 && HW4 <= 64 && (HW4 & (HW4 - 1)) == 0 && (long long)(C / G) * HW <= 256 * GN_WAVE_NV &&
        N * G >= 1024)
    {
        const int cnt4 = (C / G) * HW4;
#define GN_BWD_WAVE(NV_, A1_, A2_) DP_LAUNCH((gn_bwd_wave_kernel<NV_, A1_, A2_>), dim3((N * G + 3) / 4), dim3(256), 0,       \
                                             (hipStream_t)stream, s, gamma, beta, stats, dz, dz_img_stride, C, HW, G, silu, dx,  \
                                             dx_img_stride, add1, add1_img_stride, add2, add2_img_stride, pws, dd, rows, N * G)
#define GN_BWD_WAVE_A(NV_) do { if (add1 && add2) GN_BWD_WAVE(NV_, true, true); else if (add1) GN_BWD_WAVE(NV_, true, false);   \
                                else if (add2) GN_BWD_WAVE(NV_, false, true); else GN_BWD_WAVE(NV_, false, false); } while (0)
        if (cnt4 <= 64) GN_BWD_WAVE_A(1);
        else if (cnt4 <= 128) GN_BWD_WAVE_A(2);
        else if (cnt4 <= 256) GN_BWD_WAVE_A(4);
        else GN_BWD_WAVE_A(8);
#undef GN_BWD_WAVE_A
#undef GN_BWD_WAVE
    }
    else if (vec4 && HW4 <= 256 && C / G <= 8 && !getenv("DP_NO_GN_CACHE")) {
#define GN_BWD_C(NCH_, A1_, A2_) DP_LAUNCH((gn_bwd_vec4c_kernel<NCH_, A1_, A2_>), dim3(N * G), dim3(256), 0, (hipStream_t)stream, s,  \
                                           gamma, beta, stats, dz, dz_img_stride, C, HW, G, silu, dx, dx_img_stride, add1,           \
                                           add1_img_stride, add2, add2_img_stride, pws, dd, rows)
#define GN_BWD_C_A(NCH_) do { if (add1 && add2) GN_BWD_C(NCH_, true, true); else if (add1) GN_BWD_C(NCH_, true, false);              \
                              else if (add2) GN_BWD_C(NCH_, false, true); else GN_BWD_C(NCH_, false, false); } while (0)
        if (C / G <= 4) GN_BWD_C_A(1);
        else GN_BWD_C_A(2);
#undef GN_BWD_C_A
#undef GN_BWD_C
    } else if (vec4)
        DP_LAUNCH(gn_bwd_vec4_kernel, dim3(N * G), dim3(256), 0, (hipStream_t)stream, s, gamma, beta, stats, dz,
                           dz_img_stride, C, HW, G, silu, dx, dx_img_stride, add1, add1_img_stride, add2, add2_img_stride,
                           pws, dd, rows);
    else
        DP_LAUNCH(gn_bwd_kernel, dim3(N * G), dim3(256), 0, (hipStream_t)stream, s, gamma, beta, stats, dz,
                           dz_img_stride, C, HW, G, silu, dx, dx_img_stride, add1, add1_img_stride, add2, add2_img_stride,
                           pws, dd, rows);
    return DP_LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------
// Split GroupNorm for FEW, LARGE groups (256x256 images at batch 4: N*G = 128 groups of 1 MB -- one workgroup per group
// leaves half of the CUs idle and the rest latency bound).  Work unit = one slice of one channel plane:
//   forward : part  -> (slice mean, slice M2)            combine (Chan's parallel variance, fixed order) -> stats
//             apply -> y = silu?((x - mean) * rstd * gamma + beta)
//   backward: part  -> (sum dy, sum dy*xhat) per slice   combine -> pws[n][c] and the group terms a, b
//             apply -> dx
// All vec4 (HW % 4 == 0, 16-byte aligned planes); slice length = HW / S floats, a multiple of 4.
// ------------------------------------------------------------------------------------------------
// Round 5: the slice (<= 1024 float4 for every slicing ops._gn_slices chooses) is read ONCE, four branch-free 16-byte loads per
// thread in flight together (the loop form kept one load per wavefront outstanding and re-read the slice for the variance).
__global__ __launch_bounds__(256) void gn_split_part_fwd_kernel(GnSrc src, int C, int HW, int S, float* __restrict__ part) {
    __shared__ float red[4];
    const int nc = blockIdx.x, sl = blockIdx.y;
    const int n = nc / C, c = nc - n * C;
    const int len4 = HW / S / 4;
    const float4* xp = reinterpret_cast<const float4*>(gn_chan_ptr(src, n, c, HW)) + (long long)sl * len4;
    float s = 0.f, q = 0.f, mean;
    if (len4 <= 1024) {
        float4 xr[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = threadIdx.x + 256 * k;
            xr[k] = xp[i < len4 ? i : len4 - 1];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (threadIdx.x + 256 * k >= len4) xr[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            s += (xr[k].x + xr[k].y) + (xr[k].z + xr[k].w);
        }
        mean = dp_block_sum_256(s, red) / (float)(len4 * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (threadIdx.x + 256 * k < len4) {
                const float a = xr[k].x - mean, b = xr[k].y - mean, cc = xr[k].z - mean, d = xr[k].w - mean;
                q += (a * a + b * b) + (cc * cc + d * d);
            }
        }
    } else {
        for (int i = threadIdx.x; i < len4; i += 256) { const float4 v = xp[i]; s += (v.x + v.y) + (v.z + v.w); }
        mean = dp_block_sum_256(s, red) / (float)(len4 * 4);
        for (int i = threadIdx.x; i < len4; i += 256) {
            const float4 v = xp[i];
            const float a = v.x - mean, b = v.y - mean, cc = v.z - mean, d = v.w - mean;
            q += (a * a + b * b) + (cc * cc + d * d);
        }
    }
    q = dp_block_sum_256(q, red);
    if (threadIdx.x == 0) {
        part[((long long)nc * S + sl) * 2 + 0] = mean;
        part[((long long)nc * S + sl) * 2 + 1] = q;
    }
}

// one thread per (n, group): combine the cpg*S equally sized slices in ascending order
__global__ void gn_split_combine_fwd_kernel(const float* __restrict__ part, int NG, int cpg, int S, int HW, float eps,
                                            float* __restrict__ stats) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NG) return;
    const float* p = part + (long long)i * cpg * S * 2;
    const float cnt = (float)(HW / S);
    float mean = p[0], m2 = p[1], k = 1.f;
    for (int j = 1; j < cpg * S; ++j) {
        const float d = p[2 * j] - mean;
        k += 1.f;
        mean += d / k;
        m2 += p[2 * j + 1] + d * d * cnt * (k - 1.f) / k;
    }
    stats[(long long)i * 2 + 0] = mean;
    stats[(long long)i * 2 + 1] = 1.0f / sqrtf(m2 / (cnt * k) + eps);
}

__global__ __launch_bounds__(256) void gn_split_apply_fwd_kernel(GnSrc src, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, int C, int HW, int G, int S,
                                                                 int silu, const float* __restrict__ stats,
                                                                 float* __restrict__ y, long long y_img_stride, DpDrop drop) {
    const int nc = blockIdx.x, sl = blockIdx.y;
    const int n = nc / C, c = nc - n * C;
    const long long didx0 = ((drop.n_off + n) * C + c) * (long long)HW + (long long)sl * (HW / S);
    const int g = c / (C / G);
    const float mean = stats[((long long)n * G + g) * 2 + 0], rstd = stats[((long long)n * G + g) * 2 + 1];
    const float ga = gamma[c], be = beta[c];
    const int len4 = HW / S / 4;
    const float4* xp = reinterpret_cast<const float4*>(gn_chan_ptr(src, n, c, HW)) + (long long)sl * len4;
    float4* yp = reinterpret_cast<float4*>(y + (long long)n * y_img_stride + (long long)c * HW) + (long long)sl * len4;
    for (int i0 = threadIdx.x; i0 < len4; i0 += 1024) {          // four branch-free loads in flight per thread
        float4 xr[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + 256 * k;
            xr[k] = xp[i < len4 ? i : len4 - 1];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + 256 * k;
            const float4 v = xr[k];
            float4 o;
            o.x = (v.x - mean) * rstd * ga + be; o.y = (v.y - mean) * rstd * ga + be;
            o.z = (v.z - mean) * rstd * ga + be; o.w = (v.w - mean) * rstd * ga + be;
            if (silu) { o.x = dp_silu(o.x); o.y = dp_silu(o.y); o.z = dp_silu(o.z); o.w = dp_silu(o.w); }
            if (drop.thr24) { const float4 m = dp_drop4(drop, didx0 + 4ll * i); o.x *= m.x; o.y *= m.y; o.z *= m.z; o.w *= m.w; }
            if (i < len4) yp[i] = o;
        }
    }
}

extern "C" int dp_groupnorm_silu_fwd_split(const float* x1, const float* x2, int c_split, long long x1_img_stride,
                                           long long x2_img_stride, const float* gamma, const float* beta, int N, int C, int HW,
                                           int G, float eps, int silu, float* y, long long y_img_stride, float* stats,
                                           int slices, float* ws, const dp_dropout* drop, void* stream) {
    if (N <= 0 || C <= 0) return 0;
    const DpDrop dd = dp_drop_host(drop);
    if (C % G || slices <= 0 || HW % (4 * slices) || !ws) return (int)hipErrorInvalidValue;
    if (((x1_img_stride | x2_img_stride | y_img_stride) % 4) || (((uintptr_t)x1 | (uintptr_t)x2 | (uintptr_t)y) % 16))
        return (int)hipErrorInvalidValue;
    GnSrc s{x1, x2, x2 ? c_split : C, x1_img_stride, x2_img_stride};
    hipStream_t st = (hipStream_t)stream;
    DP_LAUNCH(gn_split_part_fwd_kernel, dim3(N * C, slices), dim3(256), 0, st, s, C, HW, slices, ws);
    DP_LAUNCH(gn_split_combine_fwd_kernel, dim3((N * G + 63) / 64), dim3(64), 0, st, ws, N * G, C / G, slices, HW, eps,
                       stats);
    DP_LAUNCH(gn_split_apply_fwd_kernel, dim3(N * C, slices), dim3(256), 0, st, s, gamma, beta, C, HW, G, slices, silu,
                       stats, y, y_img_stride, dd);
    return DP_LAUNCH_CHECK();
}

__global__ __launch_bounds__(256) void gn_split_part_bwd_kernel(GnSrc src, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, const float* __restrict__ stats,
                                                                const float* __restrict__ dz, long long dz_img_stride, int C,
                                                                int HW, int G, int S, int silu, float* __restrict__ part,
                                                                DpDrop drop) {
    __shared__ float red[4];
    const int nc = blockIdx.x, sl = blockIdx.y;
    const int n = nc / C, c = nc - n * C;
    const long long didx0 = ((drop.n_off + n) * C + c) * (long long)HW + (long long)sl * (HW / S);
    const int g = c / (C / G);
    const float mean = stats[((long long)n * G + g) * 2 + 0], rstd = stats[((long long)n * G + g) * 2 + 1];
    const float ga = gamma[c], be = beta[c];
    const int len4 = HW / S / 4;
    const float4* xp = reinterpret_cast<const float4*>(gn_chan_ptr(src, n, c, HW)) + (long long)sl * len4;
    const float4* dp = reinterpret_cast<const float4*>(dz + (long long)n * dz_img_stride + (long long)c * HW) + (long long)sl * len4;
    float a1 = 0.f, a2 = 0.f;
    for (int i0 = threadIdx.x; i0 < len4; i0 += 1024) {          // eight branch-free loads in flight per thread (round 5)
        float4 xr[4], dr[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + 256 * k;
            const int ic = i < len4 ? i : len4 - 1;
            xr[k] = xp[ic];
            dr[k] = dp[ic];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + 256 * k;
            if (i < len4) {
                const float4 xv = xr[k];
                float4 d = dr[k];
                const float hx = (xv.x - mean) * rstd, hy = (xv.y - mean) * rstd, hz = (xv.z - mean) * rstd, hw = (xv.w - mean) * rstd;
                if (drop.thr24) { const float4 m = dp_drop4(drop, didx0 + 4ll * i); d.x *= m.x; d.y *= m.y; d.z *= m.z; d.w *= m.w; }
                if (silu) {
                    d.x *= dp_silu_grad(hx * ga + be); d.y *= dp_silu_grad(hy * ga + be);
                    d.z *= dp_silu_grad(hz * ga + be); d.w *= dp_silu_grad(hw * ga + be);
                }
                a1 += (d.x + d.y) + (d.z + d.w);
                a2 += (d.x * hx + d.y * hy) + (d.z * hz + d.w * hw);
            }
        }
    }
    a1 = dp_block_sum_256(a1, red);
    a2 = dp_block_sum_256(a2, red);
    if (threadIdx.x == 0) {
        part[((long long)nc * S + sl) * 2 + 0] = a1;
        part[((long long)nc * S + sl) * 2 + 1] = a2;
    }
}

// one thread per (n, group): channel sums -> pws, group terms a/M, b/M -> ab[(n*G+g)*2]
__global__ void gn_split_combine_bwd_kernel(const float* __restrict__ part, const float* __restrict__ gamma, int NG, int G,
                                            int C, int S, int HW, float* __restrict__ pws, float* __restrict__ ab) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NG) return;
    const int cpg = C / G;
    const int n = i / G, g = i - n * G;
    float a = 0.f, b = 0.f;
    for (int cl = 0; cl < cpg; ++cl) {
        const int c = g * cpg + cl;
        const float* p = part + ((long long)n * C + c) * S * 2;
        float s1 = 0.f, s2 = 0.f;
        for (int j = 0; j < S; ++j) { s1 += p[2 * j]; s2 += p[2 * j + 1]; }
        pws[((long long)n * C + c) * 2 + 0] = s1;
        pws[((long long)n * C + c) * 2 + 1] = s2;
        a += gamma[c] * s1;
        b += gamma[c] * s2;
    }
    const float invM = 1.0f / (float)(cpg * HW);
    ab[(long long)i * 2 + 0] = a * invM;
    ab[(long long)i * 2 + 1] = b * invM;
}

template <bool A1, bool A2>
__global__ __launch_bounds__(256) void gn_split_apply_bwd_kernel(GnSrc src, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, const float* __restrict__ stats,
                                                                 const float* __restrict__ ab, const float* __restrict__ dz,
                                                                 long long dz_img_stride, int C, int HW, int G, int S, int silu,
                                                                 float* __restrict__ dx, long long dx_img_stride,
                                                                 const float* __restrict__ add1, long long add1_s,
                                                                 const float* __restrict__ add2, long long add2_s,
                                                                 DpDrop drop) {
    const int nc = blockIdx.x, sl = blockIdx.y;
    const int n = nc / C, c = nc - n * C;
    const long long didx0 = ((drop.n_off + n) * C + c) * (long long)HW + (long long)sl * (HW / S);
    const int g = c / (C / G);
    const float mean = stats[((long long)n * G + g) * 2 + 0], rstd = stats[((long long)n * G + g) * 2 + 1];
    const float a = ab[((long long)n * G + g) * 2 + 0], b = ab[((long long)n * G + g) * 2 + 1];
    const float ga = gamma[c], be = beta[c];
    const int len4 = HW / S / 4;
    const long long off4 = (long long)sl * len4;
    const float4* xp = reinterpret_cast<const float4*>(gn_chan_ptr(src, n, c, HW)) + off4;
    const float4* dp = reinterpret_cast<const float4*>(dz + (long long)n * dz_img_stride + (long long)c * HW) + off4;
    float4* op = reinterpret_cast<float4*>(dx + (long long)n * dx_img_stride + (long long)c * HW) + off4;
    const float4* a1p = A1 ? reinterpret_cast<const float4*>(add1 + (long long)n * add1_s + (long long)c * HW) + off4 : nullptr;
    const float4* a2p = A2 ? reinterpret_cast<const float4*>(add2 + (long long)n * add2_s + (long long)c * HW) + off4 : nullptr;
    // round 5: two slice positions per thread and round, every operand of both (x, dz, addends: up to eight 16-byte loads) in
    // flight before the first use; the loop form waited for x / dz, then for add1, then for add2 -- three round trips per position
    for (int i0 = threadIdx.x; i0 < len4; i0 += 512) {
        float4 xr[2], dr[2], t1[2], t2[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = i0 + 256 * k;
            const int ic = i < len4 ? i : len4 - 1;
            xr[k] = xp[ic];
            dr[k] = dp[ic];
            if (A1) t1[k] = a1p[ic];
            if (A2) t2[k] = a2p[ic];
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = i0 + 256 * k;
            const float4 xv = xr[k];
            float4 d = dr[k];
            const float hx = (xv.x - mean) * rstd, hy = (xv.y - mean) * rstd, hz = (xv.z - mean) * rstd, hw = (xv.w - mean) * rstd;
            if (drop.thr24) { const float4 m = dp_drop4(drop, didx0 + 4ll * i); d.x *= m.x; d.y *= m.y; d.z *= m.z; d.w *= m.w; }
            if (silu) {
                d.x *= dp_silu_grad(hx * ga + be); d.y *= dp_silu_grad(hy * ga + be);
                d.z *= dp_silu_grad(hz * ga + be); d.w *= dp_silu_grad(hw * ga + be);
            }
            float4 v;
            v.x = rstd * (ga * d.x - a - hx * b); v.y = rstd * (ga * d.y - a - hy * b);
            v.z = rstd * (ga * d.z - a - hz * b); v.w = rstd * (ga * d.w - a - hw * b);
            if (A1) { const float4 t = t1[k]; v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
            if (A2) { const float4 t = t2[k]; v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
            if (i < len4) op[i] = v;
        }
    }
}

extern "C" int dp_groupnorm_silu_bwd_split(const float* x1, const float* x2, int c_split, long long x1_img_stride,
                                           long long x2_img_stride, const float* gamma, const float* beta, const float* stats,
                                           const float* dz, long long dz_img_stride, int N, int C, int HW, int G, int silu,
                                           float* dx, long long dx_img_stride, const float* add1, long long add1_img_stride,
                                           const float* add2, long long add2_img_stride, float* pws, int slices, float* ws,
                                           const dp_dropout* drop, void* stream) {
    if (N <= 0 || C <= 0) return 0;
    const DpDrop dd = dp_drop_host(drop);
    if (C % G || slices <= 0 || HW % (4 * slices) || !ws) return (int)hipErrorInvalidValue;
    if (((x1_img_stride | x2_img_stride | dz_img_stride | dx_img_stride | add1_img_stride | add2_img_stride) % 4) ||
        (((uintptr_t)x1 | (uintptr_t)x2 | (uintptr_t)dz | (uintptr_t)dx | (uintptr_t)add1 | (uintptr_t)add2) % 16))
        return (int)hipErrorInvalidValue;
    GnSrc s{x1, x2, x2 ? c_split : C, x1_img_stride, x2_img_stride};
    hipStream_t st = (hipStream_t)stream;
    float* ab = ws + (long long)N * C * slices * 2;               // ws: [N*C*slices*2] partials, then [N*G*2] group terms
    DP_LAUNCH(gn_split_part_bwd_kernel, dim3(N * C, slices), dim3(256), 0, st, s, gamma, beta, stats, dz, dz_img_stride,
                       C, HW, G, slices, silu, ws, dd);
    DP_LAUNCH(gn_split_combine_bwd_kernel, dim3((N * G + 63) / 64), dim3(64), 0, st, ws, gamma, N * G, G, C, slices, HW,
                       pws, ab);
#define GN_SPLIT_APPLY(A1_, A2_) DP_LAUNCH((gn_split_apply_bwd_kernel<A1_, A2_>), dim3(N * C, slices), dim3(256), 0, st, s, gamma,  \
                                           beta, stats, ab, dz, dz_img_stride, C, HW, G, slices, silu, dx, dx_img_stride, add1,    \
                                           add1_img_stride, add2, add2_img_stride, dd)
    if (add1 && add2) GN_SPLIT_APPLY(true, true);
    else if (add1) GN_SPLIT_APPLY(true, false);
    else if (add2) GN_SPLIT_APPLY(false, true);
    else GN_SPLIT_APPLY(false, false);
#undef GN_SPLIT_APPLY
    return DP_LAUNCH_CHECK();
}

// out[c] (+)= sum_n ws[(n*C + c)*wstride + woff].  64 channels per workgroup (coalesced across lanes), the rows are
// split over the 4 wavefronts and combined through LDS in a fixed order (deterministic).
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ ws, int N, int C, int wstride, int woff,
                                                     float* __restrict__ out, int accumulate) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (c < C) {
        // 4 independent partial sums keep 4 loads in flight per lane (the loop is latency-, not bandwidth-bound)
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int n = wave;
        for (; n + 12 < N; n += 16) {
            s0 += ws[((long long)n * C + c) * wstride + woff];
            s1 += ws[((long long)(n + 4) * C + c) * wstride + woff];
            s2 += ws[((long long)(n + 8) * C + c) * wstride + woff];
            s3 += ws[((long long)(n + 12) * C + c) * wstride + woff];
        }
        for (; n < N; n += 4) s0 += ws[((long long)n * C + c) * wstride + woff];
        s = (s0 + s1) + (s2 + s3);
    }
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && c < C) {
        const float t = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
        out[c] = accumulate ? out[c] + t : t;
    }
}

extern "C" int dp_colsum_accum(const float* ws, int N, int C, int wstride, int woff, float* out, int accumulate,
                               void* stream) {
    if (C <= 0) return 0;
    DP_LAUNCH(colsum_kernel, dim3((C + 63) / 64), dim3(256), 0, (hipStream_t)stream, ws, N, C, wstride, woff, out,
                       accumulate);
    return DP_LAUNCH_CHECK();
}

// Many column sums in ONE launch: the bias / GroupNorm-parameter gradients of a whole backward pass (~215 sums per CIFAR step,
// each a few microseconds of work behind a kernel boundary) are queued by the host and reduced together at the end of the
// pass.  Same per-item arithmetic and summation order as colsum_kernel (bit-identical results).
#define DP_COLSUM_BATCH 80
struct ColsumBatch {
    int n;
    int blk_start[DP_COLSUM_BATCH + 1];
    dp_colsum_item e[DP_COLSUM_BATCH];
};

__global__ __launch_bounds__(256) void colsum_batch_kernel(const ColsumBatch b) {
    __shared__ float part[4][64];
    int i = 0;
    while (i + 1 < b.n && (int)blockIdx.x >= b.blk_start[i + 1]) ++i;           // block-uniform scan over <= 80 entries
    const dp_colsum_item& it = b.e[i];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int c = ((int)blockIdx.x - b.blk_start[i]) * 64 + lane;
    const int N = it.N, C = it.C, wstride = it.wstride, woff = it.woff;
    const int ld = it.ld ? it.ld : C;                      // row pitch in (n, c) elements: a column slice of a wider [N][ld] matrix
    const float* __restrict__ ws = it.src;
    float s = 0.f;
    if (c < C) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int n = wave;
        for (; n + 12 < N; n += 16) {
            s0 += ws[((long long)n * ld + c) * wstride + woff];
            s1 += ws[((long long)(n + 4) * ld + c) * wstride + woff];
            s2 += ws[((long long)(n + 8) * ld + c) * wstride + woff];
            s3 += ws[((long long)(n + 12) * ld + c) * wstride + woff];
        }
        for (; n < N; n += 4) s0 += ws[((long long)n * ld + c) * wstride + woff];
        s = (s0 + s1) + (s2 + s3);
    }
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && c < C) {
        const float t = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
        it.dst[c] = it.accumulate ? it.dst[c] + t : t;
    }
}

extern "C" int dp_colsum_accum_batch(const dp_colsum_item* items, int n, void* stream) {
    // Two items of ONE launch must not share a destination: different workgroups would read-modify-write it unordered.
    // A repeated destination therefore closes the launch (launches are stream-ordered, so the sums stay deterministic).
    int i = 0;
    while (i < n) {
        ColsumBatch b;
        b.n = 0;
        int blocks = 0;
        for (; i < n && b.n < DP_COLSUM_BATCH; ++i) {          // `i` is the consumed index: empty items do not shift the window
            if (items[i].C <= 0) continue;
            bool dup = false;
            for (int j = 0; j < b.n && !dup; ++j) dup = (b.e[j].dst == items[i].dst);
            if (dup) break;
            b.blk_start[b.n] = blocks;
            b.e[b.n] = items[i];
            blocks += (items[i].C + 63) / 64;
            ++b.n;
        }
        b.blk_start[b.n] = blocks;
        if (!b.n) continue;
        DP_LAUNCH(colsum_batch_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, b);
    }
    return DP_LAUNCH_CHECK();
}

// rows[n*C + c] = sum_hw x[n*img_stride + c*HW + hw].  One wavefront per (n,c) plane for small planes, one workgroup per
// plane from 4096 pixels on (256x256 feature maps: 65536 pixels per plane would be 1024 dependent adds per lane);
// 16-byte loads and 4 independent partial sums per lane when the planes are aligned.  Fixed summation order.
template <bool VEC4>
__device__ __forceinline__ float rowsum_lane(const float* __restrict__ p, int HW, int t, int nt) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (VEC4) {
        const float4* p4 = reinterpret_cast<const float4*>(p);
        for (int i = t; i < HW / 4; i += nt) { const float4 v = p4[i]; s0 += v.x; s1 += v.y; s2 += v.z; s3 += v.w; }
    } else {
        for (int i = t; i < HW; i += nt) s0 += p[i];
    }
    return (s0 + s1) + (s2 + s3);
}

template <bool VEC4>
__global__ __launch_bounds__(256) void rowsum_kernel(const float* __restrict__ x, long long img_stride, int N, int C, int HW,
                                                     float* __restrict__ rows) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long long)N * C) return;
    const int n = (int)(row / C);
    const int c = (int)(row - (long long)n * C);
    float s = rowsum_lane<VEC4>(x + (long long)n * img_stride + (long long)c * HW, HW, threadIdx.x & 63, 64);
    s = dp_wave_sum(s);
    if ((threadIdx.x & 63) == 0) rows[row] = s;
}

template <bool VEC4>
__global__ __launch_bounds__(256) void rowsum_plane_kernel(const float* __restrict__ x, long long img_stride, int C, int HW,
                                                           float* __restrict__ rows) {
    __shared__ float red[4];
    const int n = blockIdx.x / C;
    const int c = blockIdx.x - n * C;
    float s = rowsum_lane<VEC4>(x + (long long)n * img_stride + (long long)c * HW, HW, threadIdx.x, 256);
    s = dp_block_sum_256(s, red);
    if (threadIdx.x == 0) rows[blockIdx.x] = s;
}

extern "C" int dp_rowsum_nc(const float* x, long long img_stride, int N, int C, int HW, float* rows, void* stream) {
    const long long nrows = (long long)N * C;
    if (nrows <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const bool vec4 = (HW % 4 == 0) && (img_stride % 4 == 0) && ((uintptr_t)x % 16 == 0);
    if (HW >= 4096) {
        if (vec4) DP_LAUNCH(rowsum_plane_kernel<true>, dim3((unsigned)nrows), dim3(256), 0, st, x, img_stride, C, HW, rows);
        else      DP_LAUNCH(rowsum_plane_kernel<false>, dim3((unsigned)nrows), dim3(256), 0, st, x, img_stride, C, HW, rows);
    } else {
        const dim3 grid((unsigned)((nrows + 3) / 4));
        if (vec4) DP_LAUNCH(rowsum_kernel<true>, grid, dim3(256), 0, st, x, img_stride, N, C, HW, rows);
        else      DP_LAUNCH(rowsum_kernel<false>, grid, dim3(256), 0, st, x, img_stride, N, C, HW, rows);
    }
    return DP_LAUNCH_CHECK();
}
