// Weight gradient of the 3x3 / stride-1 / pad-1 convolutions as the TWO-DIMENSIONAL transposed Winograd algorithm F(3x3, 2x2) on the
// fp32 matrix cores (round 6): 16 multiplies per (2x2 block of output gradients, m, c) instead of 36 -- 4/9 of the direct form, 2/3 of
// csrc/winograd.hip's one-dimensional F(3, 2) kernel.
//     A_t = Ta dy Ta^T (4x4 from the 2x2 block dy of output gradients),  Ta = [1 0; 1 1; 1 -1; 0 -1]
//     V   = B^T d B    (4x4 from the 4x4 input patch d, as in csrc/winograd2d.hip)
//     G[i][j][m][c] += A_t[i][j](m, tile) V[i][j](c, tile)   over all tiles of all images
//     dW[a][b] = sum_ij To[a][i] To[b][j] G[i][j],   To = [1 .5 .5 0; 0 .5 -.5 0; 0 .5 .5 1]
// One 256-thread workgroup = a 64 (m) x 32 (c) tile of ALL 16 position matrices over a range of K tiles (split-K over the pixels, like
// nt_gemm_fast / wgrad_wino); wavefront i owns row i of the position matrix (4 positions x 2 row blocks x 16 = 128 accumulators), so
// both operand transforms need only the two non-zero rows of Ta / B^T row i.  One K tile = 16 tiles = 64 contiguous output pixels
// (64 / W whole image rows): the dy tile is [m][64 pixels], the input tile [c][64 + 2 W pixels] (one image row of halo above and
// below; rows outside the image are out-of-range DMA offsets = zeros), both pixel-contiguous in global memory and in LDS (row pitch
// + 4 floats: the 32 lanes of a k half -- 32 different m or c -- then fall on 16 different 4-bank groups).  W is a template parameter
// (8, 16, 32): every fragment offset is an instruction immediate.  Horizontal zero padding = per-lane selects on the first / last
// tile of an image row.  The epilogue folds the 16 position matrices into the nine taps -- columns inside the wavefront, rows through
// LDS, one row block per pass -- and writes the same tap-major split-K partials as wgrad_wino (dp_splitk_reduce_taps sums them).
#include <cstdlib>
#include "dp_common.h"

#define DPG2_RSRC_FLAGS 0x00020000
#define DPG2_OOB 0x80000000u
typedef __attribute__((address_space(3))) void dpg2_lds_void;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t dpg2_rsrc(const float* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, DPG2_RSRC_FLAGS);
}

// TAIL (M % 64 in 1 .. 32, e.g. the pruned models' 96-wide layers): the workgroups of the last row tile have no second row block of
// output channels -- they run the K loop without its dy fragments, transforms and MFMAs.
template <int LW, bool TAIL>
__device__ __forceinline__ void wgrad_wino2d_body(const dp_nt_gemm_params& p) {
    constexpr int W = 1 << LW, TC = W / 2, LTC = LW - 1;
    constexpr int PA = 68;                             // floats per dy row in LDS: 64 pixels + 4
    constexpr int NXC = 16 + W / 2;                    // 16-byte chunks of an input row: 64 + 2 W pixels
    constexpr int PB = 4 * NXC + 4;                    // floats per input row in LDS
    constexpr int A_SLOTS = 64 * 17, B_SLOTS = 32 * (NXC + 1);       // 16-byte slots incl. the padding slot of every row
    constexpr int NA = (A_SLOTS + 255) / 256, NB = (B_SLOTS + 255) / 256;
    constexpr int A_SZ = NA * 1024, B_SZ = NB * 1024;  // floats, whole wave instructions
    constexpr int STAGE = A_SZ + B_SZ;
    __shared__ __attribute__((aligned(16))) float smem[(2 * STAGE > 12288) ? 2 * STAGE : 12288];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);             // = row i of the position matrix
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 32;
    const int split = blockIdx.z;
    const dp_conv_geom& g = p.g;
    const int H = g.Ho, HW = H * W;
    const int lhw = 31 - __builtin_clz((unsigned)HW);
    const int C1 = p.X2 ? g.c_split : p.NCOLS;
    const bool src1 = n0 < C1;                                             // the whole column tile lies in one concat source
    const float* Xs = src1 ? p.X1 : p.X2;
    const int ncs = src1 ? C1 : p.NCOLS - C1;
    const int cb = src1 ? n0 : n0 - C1;
    const long long xis = src1 ? g.x1_img_stride : g.x2_img_stride;
    const int T = p.P >> 6;                                                // K tiles of 64 pixels
    const int tps = p.p_per_split >> 6;
    const int t0 = split * tps, t1 = min(t0 + tps, T);
    const int nIter = t1 - t0;

    // ---- loaders: slot q = 256 j + tid; A: row q / 17, chunk q % 17 (16 = padding); B: row q / (NXC + 1), chunk q % (NXC + 1)
    const __amdgpu_buffer_rsrc_t rA = dpg2_rsrc(p.A, p.a_bytes);
    const __amdgpu_buffer_rsrc_t rB = dpg2_rsrc(Xs, src1 ? p.x1_bytes : p.x2_bytes);
    unsigned a_c[NA], b_c[NB];
    int b_dr[NB];                                       // image-row offset of the chunk inside the input tile (row 0 = one above the K tile)
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int q = 256 * j + tid;
        const int row = q / 17, s = q - 17 * row;
        a_c[j] = (q < A_SLOTS && s < 16 && m0 + row < p.M) ? (unsigned)(((m0 + row) * HW + 4 * s) * 4) : DPG2_OOB;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int q = 256 * j + tid;
        const int row = q / (NXC + 1), s = q - (NXC + 1) * row;
        b_c[j] = (q < B_SLOTS && s < NXC && cb + row < ncs) ? (unsigned)(((cb + row) * HW + 4 * s) * 4) : DPG2_OOB;
        b_dr[j] = (4 * s) >> LW;
    }
    float* const ldsA = smem + 4 * (wave * 64);                      // + buf*STAGE + 1024 j   (slot q at float 4 q)
    float* const ldsB = smem + A_SZ + 4 * (wave * 64);

    auto dma_tile = [&](int t, int buf) {
        const int pb = 64 * t;
        const int img = pb >> lhw, r = pb & (HW - 1);
        const int R0 = r >> LW;                                      // first image row of the K tile
        const unsigned a_s = (unsigned)((long long)img * p.a_img_stride + r) * 4u;
        const long long b_base = (long long)img * xis + r - W;       // pixel (R0 - 1, 0) of the image; may be negative for R0 = 0
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            unsigned oa = a_c[j] != DPG2_OOB ? a_c[j] + a_s : DPG2_OOB;
            asm volatile("" : "+v"(oa));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (dpg2_lds_void*)(ldsA + buf * STAGE + 1024 * j), 16, (int)oa, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const bool v = b_c[j] != DPG2_OOB && (unsigned)(R0 - 1 + b_dr[j]) < (unsigned)H;
            unsigned ob = v ? (unsigned)((long long)b_c[j] + b_base * 4) : DPG2_OOB;
            asm volatile("" : "+v"(ob));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (dpg2_lds_void*)(ldsB + buf * STAGE + 1024 * j), 16, (int)ob, 0, 0, 0);
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][t][r] = 0.f;

    // ---- fragments: lane = (k half lk, row li); tile 2 ks + lk of the K tile -> tile row tr = tile >> LTC, tile column tc = tile & (TC - 1)
    const int li = lane & 31, lk = lane >> 5;
    const float* fragA = smem + li * PA + 2 * lk;                          // + 32 t * PA + (2 tr + p) W + 2 tc0
    const float* fragB = smem + A_SZ + li * PB + 2 * lk;                   // + (2 tr + r) W + 2 tc0 - 1 + x
    // Ta row `wave` as (s0, s1): dy0 s0 + dy1 s1;  B^T row `wave`: patch rows (ra, rb), second one with sign sgn
    const float s0 = wave == 3 ? 0.f : 1.f, s1 = wave == 0 ? 0.f : wave == 1 ? 1.f : -1.f;
    const int ra = wave == 0 ? 0 : wave == 2 ? 2 : 1;
    const int rb = wave == 0 ? 2 : wave == 1 ? 2 : wave == 2 ? 1 : 3;
    const float sgn = wave == 1 ? 1.f : -1.f;
    const bool lk0 = lk == 0, lk1 = lk == 1;

    if (nIter > 0) {
        dma_tile(t0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (TAIL && m0 + 32 >= p.M) {
#define G2_NT 1
#include "wgrad2d_kloop.inc"
#undef G2_NT
        } else {
#define G2_NT 2
#include "wgrad2d_kloop.inc"
#undef G2_NT
        }
    }

    // ---- epilogue: 16 position matrices -> 9 taps.  Columns inside the wavefront: Hb[0] = G0 + (G1 + G2)/2, Hb[1] = (G1 - G2)/2,
    //      Hb[2] = (G1 + G2)/2 + G3 of row i; rows through LDS, one row block per pass: dW[0][b] = H_0 + (H_1 + H_2)/2,
    //      dW[1][b] = (H_1 - H_2)/2, dW[2][b] = (H_1 + H_2)/2 + H_3.  hbuf[(i*3 + b)*16 + r][lane]: 48 KB.
    float* hbuf = smem;
    const int o_cs = p.o_col_stride ? p.o_col_stride : p.ntaps;
    const long long o_ts = p.o_tap_stride ? p.o_tap_stride : 1;
    float* __restrict__ outb = p.out + (long long)split * p.o_bs;
    const int col = n0 + li;
    const bool col_ok = col < p.NCOLS && !(src1 && col >= C1);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        if (t) __syncthreads();                          // the previous pass's reads are done
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float hs = 0.5f * (acc[1][t][r] + acc[2][t][r]);
            hbuf[((wave * 3 + 0) * 16 + r) * 64 + lane] = acc[0][t][r] + hs;
            hbuf[((wave * 3 + 1) * 16 + r) * 64 + lane] = 0.5f * (acc[1][t][r] - acc[2][t][r]);
            hbuf[((wave * 3 + 2) * 16 + r) * 64 + lane] = hs + acc[3][t][r];
        }
        __syncthreads();
        if (!col_ok) continue;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const int r = 4 * wave + q4;                 // wavefront w finishes registers 4 w .. 4 w + 3 of the row block
            const int m = m0 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * lk;
            float h[4][3];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int b = 0; b < 3; ++b) h[i][b] = hbuf[((i * 3 + b) * 16 + r) * 64 + lane];
            if (m >= p.M) continue;
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const float hs = 0.5f * (h[1][b] + h[2][b]);
                const float w3[3] = {h[0][b] + hs, 0.5f * (h[1][b] - h[2][b]), hs + h[3][b]};
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    float* o = outb + (long long)(3 * a + b) * o_ts + (long long)m * p.ldo + (long long)col * o_cs;
                    float v = p.alpha * w3[a];
                    if (p.accumulate) v += *o;
                    *o = v;
                }
            }
        }
    }
}

template <int LW>
__global__ __launch_bounds__(256, 2) void wgrad_wino2d_kernel(const dp_nt_gemm_params p) {
    wgrad_wino2d_body<LW, false>(p);
}
template <int LW>
__global__ __launch_bounds__(256, 2) void wgrad_wino2d_tail_kernel(const dp_nt_gemm_params p) {
    wgrad_wino2d_body<LW, true>(p);
}

// Shapes: 3x3 / stride 1 / pad 1, W in {8, 16, 32}, H even, H*W a power of two >= 64 (a K tile of 64 pixels never leaves its image and
// starts on an even image row), P and p_per_split multiples of 64, a concat boundary on a multiple of 32 channels.
static bool wgrad_wino2d_ok(const dp_nt_gemm_params& p) {
    const dp_conv_geom& g = p.g;
    if (p.batched || p.merge || p.col_bias || p.ntaps != 9 || g.kw != 3 || g.stride != 1 || g.sden != 1 || g.ups) return false;
    if (g.pad_t != 1 || g.pad_l != 1 || g.Ho != g.Hs || g.Wo != g.Ws || g.Hs != g.Hv || g.Ws != g.Wv) return false;
    const int W = g.Wo, HW = g.Ho * g.Wo;
    if ((W != 8 && W != 16 && W != 32) || (HW & (HW - 1)) || HW < 64 || (g.Ho & 1)) return false;
    if ((p.P % 64) || (p.p_per_split % 64) || p.p_per_split <= 0) return false;
    if ((unsigned long long)p.x1_bytes >= 0x80000000ull || (p.X2 && (unsigned long long)p.x2_bytes >= 0x80000000ull)) return false;
    if (p.X2 && (g.c_split % 32)) return false;
    return true;
}

extern "C" int dp_wgrad_wino2d_supported(const dp_nt_gemm_params* p) { return wgrad_wino2d_ok(*p) ? 1 : 0; }

// p as for dp_wgrad_wino (A = dy, X1 / X2 = the convolution input, ntaps = 9, tap-major split-K partials); the split-K range is counted
// in K tiles of 64 pixels: splits * p_per_split must cover P.
extern "C" int dp_wgrad_wino2d(const dp_nt_gemm_params* pp, void* stream) {
    const dp_nt_gemm_params& p = *pp;
    if (p.M <= 0 || p.NCOLS <= 0) return 0;
    if (!wgrad_wino2d_ok(p) || p.splits <= 0 || (long long)p.splits * p.p_per_split < (long long)p.P) return (int)hipErrorInvalidValue;
    const int C1 = p.X2 ? p.g.c_split : p.NCOLS;
    dim3 grid((C1 + 31) / 32 + (p.X2 ? (p.NCOLS - C1 + 31) / 32 : 0), (p.M + 63) / 64, p.splits);
    hipStream_t st = (hipStream_t)stream;
    static const bool tail_off = [] { const char* e = getenv("DP_WINO2D_TAIL"); return e && atoi(e) == 0; }();
    const bool tail = !tail_off && (p.M & 63) >= 1 && (p.M & 63) <= 32;       // the last row tile holds one row block only
    if (tail) {
        if (p.g.Wo == 32)      DP_LAUNCH((wgrad_wino2d_tail_kernel<5>), grid, dim3(256), 0, st, p);
        else if (p.g.Wo == 16) DP_LAUNCH((wgrad_wino2d_tail_kernel<4>), grid, dim3(256), 0, st, p);
        else                   DP_LAUNCH((wgrad_wino2d_tail_kernel<3>), grid, dim3(256), 0, st, p);
        return DP_LAUNCH_CHECK();
    }
    if (p.g.Wo == 32)      DP_LAUNCH((wgrad_wino2d_kernel<5>), grid, dim3(256), 0, st, p);
    else if (p.g.Wo == 16) DP_LAUNCH((wgrad_wino2d_kernel<4>), grid, dim3(256), 0, st, p);
    else                   DP_LAUNCH((wgrad_wino2d_kernel<3>), grid, dim3(256), 0, st, p);
    return DP_LAUNCH_CHECK();
}
