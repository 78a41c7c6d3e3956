// 3x3 stride-1 'same' convolution as a one-dimensional Winograd F(4, 3) implicit GEMM -- for the NO-GRAD forwards only.
//
// Round-4 verdict, item 5 (go / no-go).  F(2, 3) along W (winograd.hip) executes 2/3 of the direct form's multiplies; F(4, 3)
// executes 1/2: per (channel c, kernel row ky) and output QUAD (4p .. 4p+3), with d_j = x[c][y + ky - 1][4p + j - 1], j = 0..5, and
// g_k = w[m][c][ky][k] (Lavin & Gray 2015, the F(4, 3) matrices):
//     v0 = 4 d0 - 5 d2 + d4            u0 = g0 / 4
//     v1 = -4 (d1 + d2) + (d3 + d4)    u1 = -(g0 + g1 + g2) / 6
//     v2 =  4 (d1 - d2) - (d3 - d4)    u2 = -(g0 - g1 + g2) / 6
//     v3 = -2 d1 - d2 + 2 d3 + d4      u3 = g0 / 24 + g1 / 12 + g2 / 6
//     v4 =  2 d1 - d2 - 2 d3 + d4      u4 = g0 / 24 - g1 / 12 + g2 / 6
//     v5 = 4 d1 - 5 d3 + d5            u5 = g2
//     M_q += u_q v_q;   y0 = M0 + M1 + M2 + M3 + M4,  y1 = M1 - M2 + 2 M3 - 2 M4,  y2 = M1 + M2 + 4 M3 + 4 M4,
//                       y3 = M1 - M2 + 8 M3 - 8 M4 + M5
//   6 multiplies for 4 outputs x 3 taps (F(2, 3): 8).  Its 1/6 and 1/24 coefficients cost about half a decimal digit against F(2, 3)
//   (fp32 error vs fp64 ~1e-6 instead of ~3e-7), which the 2e-5 decision margin of the LDM masks does not leave room for in SCORED
//   gradients -- but the DDIM / DDPM sampling forwards and the 240 CFG forwards of an LDM importance step (93 % of config C5) feed no
//   mask decision directly: their bar is the image tolerance of the sampling fixtures.  The engines route only save=False forwards
//   here (ops.WINO43), and only the forward operand exists (no input-gradient flavour).
// Same machinery as conv_wino_kernel: ONE raw input image per (channel chunk, kernel row) in LDS serves all six positions
// (aligned 16-byte LDS-DMA loads), the input transform happens at fragment time (a ds_read_b128 + two ds_read_b32, 12 VALU per
// 6 MFMAs), the six position accumulators of a quad sit in one lane and register index, the quad leaves as ONE 16-byte store.
// Workgroup = 4 waves (2 x 2), each 32 rows x 32 quads x 6 positions (96 accumulator registers): 64 output channels x 256 pixels;
// K tile = 8 channels (A 12 KB + B 8 KB per stage: 40 KB double-buffered, three workgroups per CU).
#include <cstdlib>
#include "dp_common.h"

#define DPQ_RSRC_FLAGS 0x00020000
#define DPQ_OOB 0x80000000u
typedef __attribute__((address_space(3))) void dpq_lds_void;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t dpq_rsrc(const float* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, DPQ_RSRC_FLAGS);
}

__global__ __launch_bounds__(256, 3) void conv_wino43_kernel(const dp_conv_gemm_params p) {
    constexpr int BK = 8, BM = 64, BQ = 64, BN = 4 * BQ;   // 64 rows x 64 quads (256 pixels)
    constexpr int A_SZ = 6 * BK * BM;                      // [pos][k][m]
    constexpr int B_SZ = BK * BN;                          // [k][pixel]
    constexpr int STAGE = A_SZ + B_SZ;
    constexpr int NJA = A_SZ / 4 / 256;                    // 16-byte A loads per lane and K tile: 3
    constexpr int NJB = B_SZ / 4 / 256;                    // 16-byte B loads per lane and K tile: 2
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE + 4];      // + 4: d5 of the last quad of the last row reads one past

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    // the gy row tiles of one pixel tile back to back on one XCD (they read the same input tile): see conv_wino_kernel
    int bxx = blockIdx.x, byy = blockIdx.y;
    if ((gridDim.x & 7) == 0) {
        const int b = blockIdx.x + gridDim.x * blockIdx.y;
        const int xcd = b & 7, slot = b >> 3;
        const int gy = gridDim.y;
        const int grp = slot / gy;
        byy = slot - grp * gy;
        bxx = grp * 8 + xcd;
    }
    const int m0 = byy * BM;
    const int n0 = bxx * BN;

    const dp_conv_geom& g = p.g;
    const int W = g.Wo, H = g.Ho, HW = H * W;
    const int C = p.C;
    const int C1 = p.X2 ? g.c_split : C;
    const int nch = C / BK;
    const int nIterAll = 3 * nch;
    const bool ksplit = p.ksplit > 1;
    const int per = ksplit ? (nIterAll + p.ksplit - 1) / p.ksplit : nIterAll;
    const int it0 = ksplit ? (int)blockIdx.z * per : 0;
    const int nIter = ksplit ? max(0, min(per, nIterAll - it0)) : nIterAll;

    // ---- A loader: float4 element e = tid + 256 j of [pos][k][m/4]
    unsigned a_voff[NJA];
#pragma unroll
    for (int j = 0; j < NJA; ++j) {
        const int e = tid + 256 * j;
        const int pos = e / (BK * (BM / 4)), k = (e / (BM / 4)) % BK, m = m0 + 4 * (e % (BM / 4));
        a_voff[j] = (m < p.lda) ? (unsigned)(((pos * C + k) * p.lda + m) * 4) : DPQ_OOB;
    }
    const __amdgpu_buffer_rsrc_t rA = dpq_rsrc(p.A, p.a_bytes);
    // ---- B loader: lane = 4-pixel group of the 256-pixel row, wave + 4 j = channel row of the K tile
    unsigned x_pix1, x_pix2, vrow = 0;
    {
        const int gp = n0 + 4 * lane;
        const bool gv = gp < p.NPIX;
        const int pp = gv ? gp : 0;
        const int img = pp / HW, r = pp - img * HW;
        const int ho = r / W, wo = r - ho * W;
        const unsigned lin = (unsigned)(ho * W + wo);
        x_pix1 = (unsigned)((long long)img * g.x1_img_stride) + lin;
        x_pix2 = (unsigned)((long long)img * g.x2_img_stride) + lin;
        if (gv)
            for (int ky = 0; ky < 3; ++ky)
                if ((unsigned)(ho + ky - 1) < (unsigned)H) vrow |= 1u << ky;
    }
    unsigned b_voff1[NJB], b_voff2[NJB];
#pragma unroll
    for (int j = 0; j < NJB; ++j) {
        const int row = 4 * j + wave;
        b_voff1[j] = (x_pix1 + (unsigned)(row * HW)) * 4u;
        b_voff2[j] = (x_pix2 + (unsigned)(row * HW)) * 4u;
    }
    // descriptor bases one image row back: the scalar offset of kernel row ky is ky * W * 4 >= 0
    const __amdgpu_buffer_rsrc_t r1 = dpq_rsrc(p.X1 - W, p.x1_bytes + 4u * (unsigned)W);
    const __amdgpu_buffer_rsrc_t r2 = dpq_rsrc((p.X2 ? p.X2 : p.X1) - W, (p.X2 ? p.x2_bytes : p.x1_bytes) + 4u * (unsigned)W);

    float* const ldsA = smem + 4 * (wave * 64);                     // + buf*STAGE + 1024*j
    float* const ldsB = smem + A_SZ + wave * BN;                    // + buf*STAGE + 4*j*BN
    const unsigned a_ky_step = (unsigned)(6 * C) * (unsigned)p.lda * 4u;

    int ch = it0 / 3, ky = it0 - 3 * (it0 / 3);
    auto dma_tile = [&](int buf) {
        const unsigned a_soff = (unsigned)ky * a_ky_step + (unsigned)(ch * BK) * (unsigned)p.lda * 4u;
#pragma unroll
        for (int j = 0; j < NJA; ++j) {
            unsigned o = a_voff[j];
            asm volatile("" : "+v"(o));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (dpq_lds_void*)(ldsA + buf * STAGE + 1024 * j), 16, (int)o, (int)a_soff, 0, 0);
        }
        const int c0 = ch * BK;
        const bool first = c0 < C1;
        const unsigned b_soff = (unsigned)(((first ? c0 : c0 - C1) * HW + ky * W) * 4);
        const bool tv = (vrow >> ky) & 1u;
#pragma unroll
        for (int j = 0; j < NJB; ++j) {
            unsigned o = tv ? (first ? b_voff1[j] : b_voff2[j]) : DPQ_OOB;
            asm volatile("" : "+v"(o));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(first ? r1 : r2, (dpq_lds_void*)(ldsB + buf * STAGE + 4 * j * BN), 16, (int)o,
                                                     (int)b_soff, 0, 0);
        }
    };
    auto advance = [&]() {
        if (++ky == 3) { ky = 0; ++ch; }
    };

    f32x16 acc[6];
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

    // ---- fragment addressing: lane = (k half, column); A row wr*32 + li of position q, B quad wc*32 + li
    const int li = lane & 31, lk = lane >> 5;
    const float* fragA = smem + lk * BM + wr * 32 + li;                      // + (q*BK + 2*ks)*BM
    const float* fragB = smem + A_SZ + lk * BN + 4 * (wc * 32 + li);         // + 2*ks*BN; d1..d4 at [0..3], d0 at [-1], d5 at [4]
    // left / right zero padding: pixel 4p - 1 (4p + 4) lies outside the image row (a 256-pixel tile starts and ends on a row boundary)
    const int px = n0 + 4 * (wc * 32 + li);
    const int wo_p = px % W;
    const bool pad_l = wo_p == 0, pad_r = wo_p + 4 == W;

    if (nIter > 0) {
        dma_tile(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    for (int it = 0; it < nIter; ++it) {
        const int buf = it & 1;
        if (it + 1 < nIter) advance();
        const float* Af = fragA + buf * STAGE;
        const float* Bf = fragB + buf * STAGE;
        float a[2][6], d[2][6];
        auto frag = [&](int ks, float (&fa)[6], float (&fd)[6]) {
#pragma unroll
            for (int q = 0; q < 6; ++q) fa[q] = Af[(q * BK + 2 * ks) * BM];
            // plain float reads (hipcc merges the four aligned ones): a vector-typed LDS access makes the waitcnt pass treat the read as
            // aliasing the LDS-DMA writes and drain vmcnt(0) right behind the prefetch
            fd[1] = Bf[2 * ks * BN];
            fd[2] = Bf[2 * ks * BN + 1];
            fd[3] = Bf[2 * ks * BN + 2];
            fd[4] = Bf[2 * ks * BN + 3];
            fd[0] = Bf[2 * ks * BN - 1];
            fd[5] = Bf[2 * ks * BN + 4];
        };
        float v[2][6];
        auto xform = [&](const float (&fd)[6], float (&fv)[6]) {
            const float d0 = pad_l ? 0.f : fd[0], d5 = pad_r ? 0.f : fd[5];
            const float d1 = fd[1], d2 = fd[2], d3 = fd[3], d4 = fd[4];
            const float t1 = d4 - 4.f * d2, t2 = d3 - 4.f * d1;
            const float t3 = d4 - d2, t4 = 2.f * (d3 - d1);
            fv[0] = 4.f * d0 + (d4 - 5.f * d2);
            fv[1] = t1 + t2;
            fv[2] = t1 - t2;
            fv[3] = t3 + t4;
            fv[4] = t3 - t4;
            fv[5] = 4.f * d1 + (d5 - 5.f * d3);
        };
        frag(0, a[0], d[0]);
        xform(d[0], v[0]);
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
            const int cur = ks & 1;
            if (ks + 1 < BK / 2) frag(ks + 1, a[cur ^ 1], d[cur ^ 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 6; ++q)
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][q], v[cur][q], acc[q], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 1 < BK / 2) { xform(d[cur ^ 1], v[cur ^ 1]); __builtin_amdgcn_sched_barrier(0); }
            if (ks == 0) { dma_tile(buf ^ 1); __builtin_amdgcn_sched_barrier(0); }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- output transform + epilogue: col j = lane & 31 -> quad, row i = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    if (px >= p.NPIX) return;
    auto outq = [&](int r) {
        const float m1 = acc[1][r], m2 = acc[2][r], m3 = acc[3][r], m4 = acc[4][r];
        const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
        return make_float4((acc[0][r] + s12) + s34, d12 + 2.f * d34, s12 + 4.f * s34, (d12 + 8.f * d34) + acc[5][r]);
    };
    if (ksplit) {
        float* wsb = p.ws + (long long)blockIdx.z * p.M * p.NPIX + px;
        const int mbs = m0 + wr * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mbs + (r & 3) + 8 * (r >> 2);
            if (m >= p.M) continue;
            *reinterpret_cast<float4*>(wsb + (long long)m * p.NPIX) = outq(r);
        }
        return;
    }
    const int img = px / HW, r_in = px - img * HW;
    float* optr = p.out + (long long)img * p.o_img_stride + r_in;
    const float* rptr = p.res ? p.res + (long long)img * p.r_img_stride + r_in : nullptr;
    const float* tptr = p.tadd ? p.tadd + (long long)img * p.tadd_stride : nullptr;
#pragma unroll
    for (int h = 0; h < 4; ++h) {                 // four rows at a time, every operand load of the four in flight before the first use
        int mc[4];
        float tb[4], tt[4];
        float4 tr[4], tp[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = 4 * h + q;
            const int m = m0 + wr * 32 + 4 * (lane >> 5) + (r & 3) + 8 * (r >> 2);
            mc[q] = m < p.M ? m : p.M - 1;
            tb[q] = tt[q] = 0.f;
            tr[q] = tp[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (p.bias) {
#pragma unroll
            for (int q = 0; q < 4; ++q) tb[q] = p.bias[mc[q]];
        }
        if (tptr) {
#pragma unroll
            for (int q = 0; q < 4; ++q) tt[q] = tptr[mc[q]];
        }
        if (rptr) {
#pragma unroll
            for (int q = 0; q < 4; ++q) tr[q] = *reinterpret_cast<const float4*>(rptr + (long long)mc[q] * HW);
        }
        if (p.accumulate) {
#pragma unroll
            for (int q = 0; q < 4; ++q) tp[q] = *reinterpret_cast<const float4*>(optr + (long long)mc[q] * HW);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = 4 * h + q;
            const int m = m0 + wr * 32 + 4 * (lane >> 5) + (r & 3) + 8 * (r >> 2);
            float4 y = outq(r);
            y.x *= p.alpha; y.y *= p.alpha; y.z *= p.alpha; y.w *= p.alpha;
            if (p.bias) { y.x += tb[q]; y.y += tb[q]; y.z += tb[q]; y.w += tb[q]; }
            if (tptr) { y.x += tt[q]; y.y += tt[q]; y.z += tt[q]; y.w += tt[q]; }
            if (rptr) { y.x += tr[q].x; y.y += tr[q].y; y.z += tr[q].z; y.w += tr[q].w; }
            y.x *= p.post_scale; y.y *= p.post_scale; y.z *= p.post_scale; y.w *= p.post_scale;
            if (p.act == 1) { y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f); }
            if (p.accumulate) { y.x += tp[q].x; y.y += tp[q].y; y.z += tp[q].z; y.w += tp[q].w; }
            if (m < p.M) *reinterpret_cast<float4*>(optr + (long long)m * HW) = y;
        }
    }
}

// Shapes the kernel takes: 3x3, stride 1, pad 1, no upsampling, W a power of two in 4 .. 256 (it divides the 256-pixel tile, quads and
// 16-byte loads are aligned), channel counts (per concat source) in multiples of 8, pixel count and image planes in multiples of 4.
static bool wino43_ok(const dp_conv_gemm_params& p) {
    const dp_conv_geom& g = p.g;
    if (p.a_kc || p.ntaps != 9 || g.kw != 3 || g.stride != 1 || g.sden != 1 || g.ups || g.pad_t != 1 || g.pad_l != 1) return false;
    if (g.Ho != g.Hs || g.Wo != g.Ws || g.Hs != g.Hv || g.Ws != g.Wv || p.batches > 1 || (p.ksplit > 1 && !p.ws)) return false;
    const int W = g.Wo;
    if (W < 4 || W > 256 || (W & (W - 1))) return false;
    if ((p.lda & 3) || ((g.Ho * g.Wo) & 3) || (p.NPIX & 3)) return false;
    if ((g.x1_img_stride & 3) || (p.X2 && (g.x2_img_stride & 3)) || (p.o_img_stride & 3) || (p.res && (p.r_img_stride & 3))) return false;
    if (((uintptr_t)p.X1 | (uintptr_t)p.X2 | (uintptr_t)p.out | (uintptr_t)p.res | (uintptr_t)p.ws) & 15) return false;
    if ((unsigned long long)p.x1_bytes + 4ull * W >= 0x80000000ull || (p.X2 && (unsigned long long)p.x2_bytes + 4ull * W >= 0x80000000ull)) return false;
    const int C1 = p.X2 ? g.c_split : p.C;
    return p.C % 8 == 0 && C1 % 8 == 0;
}

extern "C" int dp_conv_wino43_supported(const dp_conv_gemm_params* p) { return wino43_ok(*p) ? 1 : 0; }
extern "C" int dp_conv_splitk_epilogue(const dp_conv_gemm_params* p, void* stream);      // gemm.hip

// p as for dp_conv_wino, with A = dp_pack_weight_wino43's operand (a_bytes = 18 * C * lda * 4).
extern "C" int dp_conv_wino43(const dp_conv_gemm_params* pp, void* stream) {
    const dp_conv_gemm_params& p = *pp;
    if (p.M <= 0 || p.NPIX <= 0) return 0;
    if (!wino43_ok(p)) return (int)hipErrorInvalidValue;
    dim3 grid((p.NPIX + 255) / 256, (p.M + 63) / 64, p.ksplit > 1 ? p.ksplit : 1);
    DP_LAUNCH(conv_wino43_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    const int e = DP_LAUNCH_CHECK();
    if (e || p.ksplit <= 1) return e;
    return dp_conv_splitk_epilogue(pp, stream);
}

// U[(ky*6 + pos)*Ci + c][ld] (m-contiguous, like dp_pack_weight_wino's operand) from a [Co, Ci, 3, 3] weight: the six F(4, 3)
// combinations of the kernel row (forward flavour only).
__global__ __launch_bounds__(256) void pack_weight_wino43_kernel(const float* __restrict__ Wt, int Co, int Ci, float* __restrict__ dst, int ld) {
    const long long total = 18ll * Ci * ld;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int m = (int)(i % ld);
        const long long rr = i / ld;
        const int k = (int)(rr % Ci);
        const int kp = (int)(rr / Ci);
        const int ky = kp / 6, pos = kp - 6 * ky;
        float v = 0.f;
        if (m < Co) {
            const float* w = Wt + ((long long)m * Ci + k) * 9 + ky * 3;
            const float g0 = w[0], g1 = w[1], g2 = w[2];
            v = pos == 0 ? g0 * 0.25f
              : pos == 1 ? ((g0 + g1) + g2) * (-1.0f / 6.0f)
              : pos == 2 ? ((g0 - g1) + g2) * (-1.0f / 6.0f)
              : pos == 3 ? (g0 * (1.0f / 24.0f) + g1 * (1.0f / 12.0f)) + g2 * (1.0f / 6.0f)
              : pos == 4 ? (g0 * (1.0f / 24.0f) - g1 * (1.0f / 12.0f)) + g2 * (1.0f / 6.0f)
              : g2;
        }
        dst[i] = v;
    }
}

extern "C" int dp_pack_weight_wino43(const float* W, int Co, int Ci, float* dst, int ld, void* stream) {
    const long long total = 18ll * Ci * ld;
    if (total <= 0) return 0;
    long long nb = (total + 255) / 256;
    if (nb > 4096) nb = 4096;
    DP_LAUNCH(pack_weight_wino43_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, W, Co, Ci, dst, ld);
    return DP_LAUNCH_CHECK();
}
