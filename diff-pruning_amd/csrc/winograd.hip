// 3x3 stride-1 'same' convolution as a ONE-DIMENSIONAL Winograd F(2, 3) implicit GEMM on the fp32 matrix cores.
//
// Round-3 verdict, item 5 (go / no-go): the direct implicit GEMM has plateaued at 0.79-0.81 of a roofline that is the fp32
// MFMA rate itself; the lever left is executing fewer multiplies.  The 2-D F(2x2, 3x3) form (16/36 of the MACs) needs 16
// accumulator sets per output tile -- no workgroup tile worth an MFMA fits the register file -- or a 4x activation round trip
// through HBM that makes the 128-channel layers HBM-bound.  This is the form that fits the machine: F(2, 3) along W only.
//   per (channel c, kernel row ky), output pair (2p, 2p+1), d_j = x[c][y + ky - 1][2p + j - 1], g_k = w[m][c][ky][k]:
//     M0 += (d0 - d2) g0,  M1 += (d1 + d2) (g0 + g1 + g2)/2,  M2 += (d2 - d1) (g0 - g1 + g2)/2,  M3 += (d1 - d3) g2
//     y[2p] = M0 + M1 + M2,  y[2p + 1] = M1 - M2 - M3
//   4 multiplies for 2 outputs x 3 taps: 2/3 of the MACs (12 "taps" per channel on HALF the columns instead of 9).
// Why it maps well:
//   * ONE raw input image [BK channels][128 pixels] per (channel chunk, ky) in LDS serves all four positions -- they differ only
//     in the LDS read offset -- so the B tile is fetched 3 times per channel chunk instead of 9, with 16-byte LDS-DMA loads that
//     are always aligned (no horizontal tap shift, no border fix-up in LDS);
//   * the input transform happens at FRAGMENT time: a lane reads d0..d3 of its (k, pair) (one ds_read_b64 + two ds_read_b32) and
//     forms the four B operands with 4 VALU subtractions / additions per 4 MFMAs; left / right zero padding = two per-lane
//     constant selects (a 128-pixel tile always starts and ends on an image-row boundary: W divides 128);
//   * the four position accumulators of an output pair live in the same lane and register index, so the output transform is
//     32 VALU adds in the epilogue and the two pixels of a pair leave as ONE 8-byte store.
// Weights are packed once per sweep as U[ky][pos][c][m] (dp_pack_weight_wino).  Same epilogue operands as dp_conv_gemm.
// fp32 everywhere; the result differs from the direct form by re-association (pre-summed taps, signed sums of inputs): ~1e-6.
#include <cstdlib>
#include <type_traits>
#include "dp_common.h"

#define DPW_RSRC_FLAGS 0x00020000
#define DPW_OOB 0x80000000u
typedef __attribute__((address_space(3))) void dpw_lds_void;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t dpw_rsrc(const float* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, DPW_RSRC_FLAGS);
}

// 256-thread workgroups, every wave 32 rows x 32 pairs x 4 positions.  WR = 2: waves 2 x 2, 64 output channels x 64 pairs (128
// pixels); WR = 1: waves 1 x 4, 32 output channels x 128 pairs (256 pixels) -- for row counts such as 96 or 288 (pruned widths),
// which fill 64-row tiles to 75 / 90 %.
// TM = 2 (with WR = 2, BK = 8): every wave 64 rows x 32 pairs -- 128-row workgroup tiles, ONE transformed B fragment feeds two row
// tiles (half the LDS reads and transform VALU per MFMA, half the B traffic per row), 128 accumulator registers (occupancy 3).
template <int BK, int WR, int TM = 1>
__global__ __launch_bounds__(256, (BK == 16 || TM == 2) ? 3 : 4) void conv_wino_kernel(const dp_conv_gemm_params p) {
    constexpr int BM = 32 * WR * TM, BP = 32 * (4 / WR), BN = 2 * BP;
    constexpr int G4 = BN / 4, RPW = 64 / G4;          // lanes per pixel row of the B tile, rows per wave instruction (2 / 1)
    constexpr int A_SZ = 4 * BK * BM;                  // [pos][k][m]
    constexpr int B_SZ = BK * BN;                      // [k][pixel]
    constexpr int STAGE = A_SZ + B_SZ;
    constexpr int NJA = A_SZ / 4 / 256;                // 16-byte A loads per lane and K tile (4 / 2)
    constexpr int NJB = B_SZ / 4 / 256;                // 16-byte B loads per lane and K tile (2 / 1)
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE + 4];      // + 4: d3 of the last pair of the last row reads one past

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = (WR == 2) ? (wave >> 1) : 0, wc = (WR == 2) ? (wave & 1) : wave;
    // The gy row tiles of one pixel tile read the same input tile: with the plain (x, y) order they are gx workgroups apart and run
    // on whatever XCD that lands on (FETCH_SIZE 1.7x the operands, profiles/round4_pmc_bench_traffic.json).  Remap (grids whose x
    // extent is a multiple of 8): linear id b -> XCD b % 8, slot b / 8; slot = (pixel-tile group) * gy + row tile, so the row tiles
    // of pixel tile 8 g + xcd run back to back on one XCD and the second to gy-th read the input from that L2.
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
    // split-K (small grids): workgroup z reduces K tiles [it0, it0 + nIter) and writes its output-transformed partial sums to
    // ws[z][m][pix]; dp_conv_splitk_epilogue sums them in ascending z and applies the epilogue (the transform is linear)
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
        a_voff[j] = (m < p.lda) ? (unsigned)(((pos * C + k) * p.lda + m) * 4) : DPW_OOB;
    }
    const __amdgpu_buffer_rsrc_t rA = dpw_rsrc(p.A, p.a_bytes);
    // ---- B loader: lane = (row of the wave's pair of rows, 4-pixel group)
    const int g4 = lane % G4, rsub = lane / G4;
    unsigned x_pix1, x_pix2, vrow = 0;
    {
        const int gp = n0 + 4 * g4;
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
        const int row = 4 * RPW * j + RPW * wave + rsub;
        b_voff1[j] = (x_pix1 + (unsigned)(row * HW)) * 4u;
        b_voff2[j] = (x_pix2 + (unsigned)(row * HW)) * 4u;
    }
    // descriptor bases one image row back: the scalar offset of kernel row ky is ky * W * 4 >= 0
    const __amdgpu_buffer_rsrc_t r1 = dpw_rsrc(p.X1 - W, p.x1_bytes + 4u * (unsigned)W);
    const __amdgpu_buffer_rsrc_t r2 = dpw_rsrc((p.X2 ? p.X2 : p.X1) - W, (p.X2 ? p.x2_bytes : p.x1_bytes) + 4u * (unsigned)W);

    float* const ldsA = smem + 4 * (wave * 64);                     // + buf*STAGE + 1024*j
    float* const ldsB = smem + A_SZ + RPW * wave * BN;              // + buf*STAGE + 4*RPW*j*BN
    const unsigned a_ky_step = (unsigned)(4 * C) * (unsigned)p.lda * 4u;

    int ch = it0 / 3, ky = it0 - 3 * (it0 / 3);
    auto dma_tile = [&](int buf) {
        const unsigned a_soff = (unsigned)ky * a_ky_step + (unsigned)(ch * BK) * (unsigned)p.lda * 4u;
#pragma unroll
        for (int j = 0; j < NJA; ++j) {
            unsigned o = a_voff[j];
            asm volatile("" : "+v"(o));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (dpw_lds_void*)(ldsA + buf * STAGE + 1024 * j), 16, (int)o, (int)a_soff, 0, 0);
        }
        const int c0 = ch * BK;
        const bool first = c0 < C1;
        const unsigned b_soff = (unsigned)(((first ? c0 : c0 - C1) * HW + ky * W) * 4);
        const bool tv = (vrow >> ky) & 1u;
#pragma unroll
        for (int j = 0; j < NJB; ++j) {
            unsigned o = tv ? (first ? b_voff1[j] : b_voff2[j]) : DPW_OOB;
            asm volatile("" : "+v"(o));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(first ? r1 : r2, (dpw_lds_void*)(ldsB + buf * STAGE + 4 * RPW * j * BN), 16, (int)o,
                                                     (int)b_soff, 0, 0);
        }
    };
    auto advance = [&]() {
        if (++ky == 3) { ky = 0; ++ch; }
    };

    f32x16 acc[TM][4];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][q][r] = 0.f;

    // ---- fragment addressing: lane = (k half, column); A row wr*32*TM + 32 t + li of position q, B pair wc*32 + li
    const int li = lane & 31, lk = lane >> 5;
    const float* fragA = smem + lk * BM + wr * (32 * TM) + li;               // + (q*BK + 2*ks)*BM + 32 t
    const float* fragB = smem + A_SZ + lk * BN + 2 * (wc * 32 + li);         // + 2*ks*BN; d1 d2 at [0..1], d0 at [-1], d3 at [2]
    // left / right zero padding: pixel 2p - 1 (2p + 2) lies outside the image row
    const int px = n0 + 2 * (wc * 32 + li);
    const int wo_p = px % W;
    const bool pad_l = wo_p == 0, pad_r = wo_p + 2 == W;

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
        float a[2][TM * 4], d[2][4];
        auto frag = [&](int ks, float (&fa)[TM * 4], float (&fd)[4]) {
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) fa[4 * t + q] = Af[(q * BK + 2 * ks) * BM + 32 * t];
            // plain float reads (hipcc pairs them into one ds_read2_b32): a float2-typed LDS access makes the waitcnt pass
            // treat the read as aliasing the LDS-DMA writes and drain vmcnt(0) right behind the prefetch
            fd[1] = Bf[2 * ks * BN];
            fd[2] = Bf[2 * ks * BN + 1];
            fd[0] = Bf[2 * ks * BN - 1];
            fd[3] = Bf[2 * ks * BN + 2];
        };
        // The input transform of k-step ks + 1 is issued BEHIND the four MFMAs of k-step ks (DPW_VPIPE, default): a transform
        // directly in front of its MFMA costs a VALU -> MFMA operand stall (s_nop 1) per matrix instruction.
        float v[2][4];
        auto xform = [&](const float (&fd)[4], float (&fv)[4]) {
            const float d0 = pad_l ? 0.f : fd[0], d3 = pad_r ? 0.f : fd[3];
            fv[0] = d0 - fd[2]; fv[1] = fd[1] + fd[2]; fv[2] = fd[2] - fd[1]; fv[3] = fd[1] - d3;
        };
        frag(0, a[0], d[0]);
#ifndef DPW_NO_VPIPE
        xform(d[0], v[0]);
#endif
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
            const int cur = ks & 1;
            if (ks + 1 < BK / 2) frag(ks + 1, a[cur ^ 1], d[cur ^ 1]);
            __builtin_amdgcn_sched_barrier(0);
#ifdef DPW_NO_VPIPE
            xform(d[cur], v[cur]);
#endif
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    acc[t][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][4 * t + q], v[cur][q], acc[t][q], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#ifndef DPW_NO_VPIPE
            if (ks + 1 < BK / 2) { xform(d[cur ^ 1], v[cur ^ 1]); __builtin_amdgcn_sched_barrier(0); }
#endif
            if (ks == 1) { dma_tile(buf ^ 1); __builtin_amdgcn_sched_barrier(0); }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- output transform + epilogue: col j = lane & 31 -> pair, row i = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    if (px >= p.NPIX) return;
    if (ksplit) {
        float* wsb = p.ws + (long long)blockIdx.z * p.M * p.NPIX + px;
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            const int mbs = m0 + wr * (32 * TM) + 32 * t + 4 * (lane >> 5);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mbs + (r & 3) + 8 * (r >> 2);
                if (m >= p.M) continue;
                *reinterpret_cast<float2*>(wsb + (long long)m * p.NPIX) =
                    make_float2((acc[t][0][r] + acc[t][1][r]) + acc[t][2][r], (acc[t][1][r] - acc[t][2][r]) - acc[t][3][r]);
            }
        }
        return;
    }
    const int img = px / HW, r_in = px - img * HW;
    float* optr = p.out + (long long)img * p.o_img_stride + r_in;
    const float* rptr = p.res ? p.res + (long long)img * p.r_img_stride + r_in : nullptr;
    const float* tptr = p.tadd ? p.tadd + (long long)img * p.tadd_stride : nullptr;
    // Eight rows at a time, and EVERY operand load of the eight (bias, per-image addend, residual, the accumulate read: up to 32
    // loads) in flight before the first use.  Round 4's form -- a null test, a load and a wait per operand and row -- read on the ISA
    // as 64 dependent memory round trips per wavefront behind a K loop of ~25 us (hipcc neither hoists a load out of its `if` nor
    // over the store in front of it).  Rows >= M of a partial tile load from the last row and skip the store.  Same expressions
    // and order per element.
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        int mc[8];
        float tb[8], tt[8];
        float2 tr[8], tp[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = 8 * h + q;
            const int m = m0 + wr * (32 * TM) + 32 * t + 4 * (lane >> 5) + (r & 3) + 8 * (r >> 2);
            mc[q] = m < p.M ? m : p.M - 1;
            tb[q] = tt[q] = 0.f;
            tr[q] = tp[q] = make_float2(0.f, 0.f);
        }
        if (p.bias) {
#pragma unroll
            for (int q = 0; q < 8; ++q) tb[q] = p.bias[mc[q]];
        }
        if (tptr) {
#pragma unroll
            for (int q = 0; q < 8; ++q) tt[q] = tptr[mc[q]];
        }
        if (rptr) {
#pragma unroll
            for (int q = 0; q < 8; ++q) tr[q] = *reinterpret_cast<const float2*>(rptr + (long long)mc[q] * HW);
        }
        if (p.accumulate) {
#pragma unroll
            for (int q = 0; q < 8; ++q) tp[q] = *reinterpret_cast<const float2*>(optr + (long long)mc[q] * HW);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = 8 * h + q;
            const int m = m0 + wr * (32 * TM) + 32 * t + 4 * (lane >> 5) + (r & 3) + 8 * (r >> 2);
            float y0 = p.alpha * ((acc[t][0][r] + acc[t][1][r]) + acc[t][2][r]);
            float y1 = p.alpha * ((acc[t][1][r] - acc[t][2][r]) - acc[t][3][r]);
            if (p.bias) { y0 += tb[q]; y1 += tb[q]; }
            if (tptr) { y0 += tt[q]; y1 += tt[q]; }
            if (rptr) { y0 += tr[q].x; y1 += tr[q].y; }
            y0 *= p.post_scale;
            y1 *= p.post_scale;
            if (p.act == 1) { y0 = fmaxf(y0, 0.f); y1 = fmaxf(y1, 0.f); }
            if (p.accumulate) { y0 += tp[q].x; y1 += tp[q].y; }
            if (m < p.M) *reinterpret_cast<float2*>(optr + (long long)m * HW) = make_float2(y0, y1);
        }
    }
}

// Shapes the kernel takes: 3x3, stride 1, pad 1, no upsampling, W a power of two in 4 .. 256 (so it divides the 128- / 256-pixel tile
// and the 16-byte loads are aligned), channel counts (per concat source) in whole K chunks, 8-byte aligned image planes.
static int wino_bk(const dp_conv_gemm_params& p) {
    const dp_conv_geom& g = p.g;
    if (p.a_kc || p.ntaps != 9 || g.kw != 3 || g.stride != 1 || g.sden != 1 || g.ups || g.pad_t != 1 || g.pad_l != 1) return 0;
    if (g.Ho != g.Hs || g.Wo != g.Ws || g.Hs != g.Hv || g.Ws != g.Wv || p.batches > 1 || (p.ksplit > 1 && (!p.ws || (p.NPIX & 1)))) return 0;
    const int W = g.Wo;
    if (W < 4 || W > 256 || (W & (W - 1))) return 0;        // W must divide the pixel tile (128, or 256 with the 32-row tiles)
    if ((p.lda & 3) || ((g.Ho * g.Wo) & 1)) return 0;
    // the X descriptors start one image row in front of the tensor (num_records = extent + 4 W): the out-of-range marker 0x80000000
    // must stay outside them, i.e. an activation within 4 W bytes of 2 GiB keeps the direct kernel  [advisor, round 4]
    if ((unsigned long long)p.x1_bytes + 4ull * W >= 0x80000000ull || (p.X2 && (unsigned long long)p.x2_bytes + 4ull * W >= 0x80000000ull)) return 0;
    const int C1 = p.X2 ? g.c_split : p.C;
    if (p.C % 16 == 0 && C1 % 16 == 0) return 16;
    if (p.C % 8 == 0 && C1 % 8 == 0) return 8;
    return 0;
}

extern "C" int dp_conv_wino_supported(const dp_conv_gemm_params* p) { return wino_bk(*p); }
extern "C" int dp_conv_splitk_epilogue(const dp_conv_gemm_params* p, void* stream);      // gemm.hip

extern "C" int dp_conv_wino(const dp_conv_gemm_params* pp, void* stream) {
    const dp_conv_gemm_params& p = *pp;
    if (p.M <= 0 || p.NPIX <= 0) return 0;
    int bk = wino_bk(p);
    if (!bk) return (int)hipErrorInvalidValue;
    // 32-row tiles when they pad fewer rows than 64-row tiles (M = 96, 288, ...).  Settled A/Bs, no run-time switch any more (DESIGN.md
    // section 5): 8-channel chunks where 16 apply (round 4: +-2 % by shape; round 5 per shape, profiles/round5_winograd_gate.txt: 16
    // equal or better everywhere), the 64-row WAVE tile <8, 2, 2> (round 4: +2 % on the largest grids, -10 ... -35 % on the small ones).
    const bool wr1 = p.g.Wo > 128 || ((p.M + 31) / 32) * 32 < ((p.M + 63) / 64) * 64;
    hipStream_t st = (hipStream_t)stream;
    if (wr1) {
        dim3 grid((p.NPIX + 255) / 256, (p.M + 31) / 32, p.ksplit > 1 ? p.ksplit : 1);
        if (bk == 16) DP_LAUNCH((conv_wino_kernel<16, 1>), grid, dim3(256), 0, st, p);
        else          DP_LAUNCH((conv_wino_kernel<8, 1>), grid, dim3(256), 0, st, p);
    } else {
        dim3 grid((p.NPIX + 127) / 128, (p.M + 63) / 64, p.ksplit > 1 ? p.ksplit : 1);
        if (bk == 16) DP_LAUNCH((conv_wino_kernel<16, 2>), grid, dim3(256), 0, st, p);
        else          DP_LAUNCH((conv_wino_kernel<8, 2>), grid, dim3(256), 0, st, p);
    }
    const int e = DP_LAUNCH_CHECK();
    if (e || p.ksplit <= 1) return e;
    return dp_conv_splitk_epilogue(pp, stream);
}

// U[(ky*4 + pos)*K + k][ld] from a torch [Co][Ci][3][3] weight.  mode 0 (forward): K = Ci, columns m = co, taps as stored;
// mode 1 (input gradient): K = Co, columns m = ci, both tap axes flipped.
__global__ __launch_bounds__(256) void pack_weight_wino_kernel(const float* __restrict__ Wt, int Co, int Ci, int mode,
                                                               float* __restrict__ dst, int ld) {
    const int K = mode == 0 ? Ci : Co, Mv = mode == 0 ? Co : Ci;
    const long long total = 12ll * K * ld;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int m = (int)(i % ld);
        const long long rk = i / ld;
        const int k = (int)(rk % K);
        const int kp = (int)(rk / K);
        const int ky = kp >> 2, pos = kp & 3;
        float v = 0.f;
        if (m < Mv) {
            const float* w = mode == 0 ? Wt + ((long long)m * Ci + k) * 9 + ky * 3 : Wt + ((long long)k * Ci + m) * 9 + (2 - ky) * 3;
            const float g0 = mode == 0 ? w[0] : w[2], g1 = w[1], g2 = mode == 0 ? w[2] : w[0];
            v = pos == 0 ? g0 : pos == 1 ? ((g0 + g1) + g2) * 0.5f : pos == 2 ? ((g0 - g1) + g2) * 0.5f : g2;
        }
        dst[i] = v;
    }
}

extern "C" int dp_pack_weight_wino(const float* W, int Co, int Ci, int mode, float* dst, int ld, void* stream) {
    const long long total = 12ll * (mode == 0 ? Ci : Co) * ld;
    if (total <= 0) return 0;
    long long nb = (total + 255) / 256;
    if (nb > 4096) nb = 4096;
    DP_LAUNCH(pack_weight_wino_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, W, Co, Ci, mode, dst, ld);
    return DP_LAUNCH_CHECK();
}

// ================================================================================================================================
// Weight gradient of the same convolutions, as the transposed F(2, 3) algorithm (F(3, 2): 3 taps from output pairs): per kernel
// row ky, 4 multiplies per (output pair, m, c) instead of 6:
//     a = (dy0, dy0 + dy1, dy0 - dy1, -dy1),  v = (d0 - d2, d1 + d2, d2 - d1, d1 - d3)   [d_j = x[c][y + ky - 1][2p + j - 1]]
//     G_q[m][c] = sum over pairs a_q v_q;    dW[ky][0] = G0 + (G1 + G2)/2,  dW[ky][1] = (G1 - G2)/2,  dW[ky][2] = (G1 + G2)/2 + G3
// i.e. 12 contractions over HALF the pixels per channel pair instead of 9 over all of them: 2/3 of the MACs.
// One workgroup = one 64 x 64 (m, c) tile of one kernel row ky over one pixel range (split-K over pixels like nt_gemm_fast, same
// tap-major partials, same reduction launch).  Both operands are pixel-contiguous rows; both transforms happen at fragment time
// (3 + 4 VALU per 4 MFMAs).  K tile = 16 pairs.  Its window of 34 pixels (pairs at pixels b+2 .. b+33, halo b+1 and b+34) is
// covered by NINE aligned 16-byte slots b .. b+35 when the tiling is shifted by two pixels (b = 32 t - 32, t = 0 .. P/32): aligned
// slots never straddle an image row, so vertical padding is a per-slot out-of-range offset, and nine slots per row put the 32
// rows a wavefront reads on 8 different 16-byte bank groups (a dense 8-slot row would put them on one).  Horizontal padding and
// the pairs outside [0, P) are per-lane multipliers (0 / 1) on d0 / d3 and zeroed A slots.
// ================================================================================================================================
// WRxWC wavefronts per workgroup, each a 32 x 32 (m, c) tile with its four position accumulators: 2 x 2 (64 x 64 tile, 256 threads)
// or 3 x 3 (96 x 96, 576 threads) for the 96-multiples of pruned models, which fill 64-wide tiles to 56 - 75 %.
template <int WR, int WC>
__global__ __launch_bounds__(64 * WR * WC, 4) void wgrad_wino_kernel(const dp_nt_gemm_params p) {
    constexpr int NT = 64 * WR * WC, BMt = 32 * WR, BNt = 32 * WC;
    constexpr int RS = 36;                              // floats per LDS row: 9 slots of 4 pixels
    constexpr int SA = BMt * 9, SB = BNt * 9;           // 16-byte slots per operand tile
    constexpr int RA = (SA + 63) / 64 * 64, RB = (SB + 63) / 64 * 64;      // ... rounded to whole wave instructions (zero-filled tail)
    constexpr int OP_SZ = 4 * RA;                       // floats in front of the B tile
    constexpr int STAGE = 4 * (RA + RB);
    constexpr int NA = (SA + NT - 1) / NT, NB = (SB + NT - 1) / NT;        // load instructions per lane and K tile
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave - WC * wr;
    // Workgroup b is dispatched to XCD b % 8 and every XCD has its own L2 (see nt_gemm_fast_kernel): p.xcd gives XCD x a contiguous
    // run of the split-major order, so the gx * gy * 3 workgroups that stream ONE pixel range share one L2's copy of dy and x.
    int bx = blockIdx.x, by = blockIdx.y, split, ky;
    if (p.xcd) {
        const int gxy = gridDim.x * gridDim.y;
        const int per_split = gxy * 3;
        const int nwg = per_split * p.splits;
        const int b = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const int q8 = nwg >> 3, r8 = nwg & 7, xcd = b & 7, k8 = b >> 3;
        const int v = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k8;
        split = __builtin_amdgcn_readfirstlane(v / per_split);
        const int rem = v - split * per_split;
        ky = __builtin_amdgcn_readfirstlane(rem / gxy);
        const int rem2 = rem - ky * gxy;
        by = __builtin_amdgcn_readfirstlane(rem2 / (int)gridDim.x);
        bx = rem2 - by * (int)gridDim.x;
    } else {
        split = blockIdx.z / 3;
        ky = blockIdx.z - 3 * split;
    }
    const int m0 = by * BMt, n0 = bx * BNt;

    const dp_conv_geom& g = p.g;
    const int W = g.Wo, H = g.Ho, HW = H * W;
    const int lw = 31 - __builtin_clz((unsigned)W), lhw = 31 - __builtin_clz((unsigned)HW);
    const int C1 = p.X2 ? g.c_split : p.NCOLS;
    const bool src1 = n0 < C1;                                   // the whole column tile lies in one concat source
    const float* Xs = src1 ? p.X1 : p.X2;
    const int ncs = src1 ? C1 : p.NCOLS - C1;                    // channels of that source
    const int cb = src1 ? n0 : n0 - C1;                          // first channel of the tile inside its source
    const long long xis = src1 ? g.x1_img_stride : g.x2_img_stride;
    // K tiles of this split: t in [t0, t1), window base pixel b = 32 t - 32
    const int T = p.P / 32 + 1;
    const int tps = p.p_per_split / 32;
    const int t0 = split * tps, t1 = min(t0 + tps, T);
    const int nIter = t1 - t0;

    // ---- loaders: slot q = NT j + tid of the A tile (j < NA) and of the B tile (j < NB); q -> (row q / 9, slot q % 9).  The last
    //      instruction of an operand is issued by the waves that still hold slots of it (wave-uniform branch).
    const __amdgpu_buffer_rsrc_t rA = dpw_rsrc(p.A, p.a_bytes);
    const __amdgpu_buffer_rsrc_t rB = dpw_rsrc(Xs - W, (src1 ? p.x1_bytes : p.x2_bytes) + 4u * (unsigned)W);   // ky row shift >= 0
    constexpr int NL = (NA > NB) ? NA : NB;
    unsigned a_c[NL], b_c[NL];         // constant part of the per-lane byte offset, DPW_OOB for rows / slots outside the tile
    bool is8[NL];                      // slot 8: the first four pixels of the NEXT 32-pixel block (its own image / row / validity)
    int dho[NL];                       // image-row offset of the slot inside the block (W < 32)
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        const int q = NT * j + tid;
        const int row = q / 9, s = q - 9 * row;
        is8[j] = s == 8;
        dho[j] = (s == 8) ? 0 : ((4 * s) >> lw);
        const int po = (s == 8) ? 0 : 4 * s;                      // pixel offset inside its 32-pixel block
        a_c[j] = (q < SA && m0 + row < p.M) ? (unsigned)(((m0 + row) * HW + po) * 4) : DPW_OOB;
        b_c[j] = (q < SB && cb + row < ncs) ? (unsigned)(((cb + row) * HW + po) * 4) : DPW_OOB;
    }
    float* const ldsA = smem + 4 * (wave * 64);                  // + buf*STAGE + 4 NT j  (slot q at float 4 q)
    float* const ldsB = smem + OP_SZ + 4 * (wave * 64);

    auto dma_tile = [&](int t, int buf) {
        const int b = 32 * t - 32;                               // block 0: pixels b .. b+31 (slots 0..7), block 1: b+32 .. (slot 8)
        const bool v0 = t >= 1, v1 = t < T - 1;                   // blocks inside [0, P)
        const int bb0 = v0 ? b : 0, bb1 = v1 ? b + 32 : 0;
        const int img0 = bb0 >> lhw, r0 = bb0 & (HW - 1), img1 = bb1 >> lhw, r1 = bb1 & (HW - 1);
        const unsigned a_s0 = (unsigned)((long long)img0 * p.a_img_stride + r0) * 4u;
        const unsigned a_s1 = (unsigned)((long long)img1 * p.a_img_stride + r1) * 4u;
        const unsigned b_s0 = (unsigned)((long long)img0 * xis + r0 + ky * W) * 4u;
        const unsigned b_s1 = (unsigned)((long long)img1 * xis + r1 + ky * W) * 4u;
        const int ho0 = r0 >> lw, ho1 = r1 >> lw;
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const bool vv = is8[j] ? v1 : v0;
            if (j < NA && NT * j + 64 * wave < RA) {
                unsigned oa = (vv && a_c[j] != DPW_OOB) ? a_c[j] + (is8[j] ? a_s1 : a_s0) : DPW_OOB;
                asm volatile("" : "+v"(oa));
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (dpw_lds_void*)(ldsA + buf * STAGE + 4 * NT * j), 16, (int)oa, 0, 0, 0);
            }
            if (j < NB && NT * j + 64 * wave < RB) {
                const int ho = (is8[j] ? ho1 : ho0 + dho[j]) + ky - 1;
                unsigned ob = (vv && (unsigned)ho < (unsigned)H && b_c[j] != DPW_OOB) ? b_c[j] + (is8[j] ? b_s1 : b_s0) : DPW_OOB;
                asm volatile("" : "+v"(ob));
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (dpw_lds_void*)(ldsB + buf * STAGE + 4 * NT * j), 16, (int)ob, 0, 0, 0);
            }
        }
    };

    f32x16 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

    // ---- fragments: lane = (k half lk, row li); pair j = 2 ks + lk of the tile sits at floats 2 j + 2 .. (dy0 dy1) / 2 j + 1 .. 2 j + 4 (d0..d3)
    const int li = lane & 31, lk = lane >> 5;
    const float* fragA = smem + (wr * 32 + li) * RS + 2 * lk + 2;
    const float* fragB = smem + OP_SZ + (wc * 32 + li) * RS + 2 * lk + 1;
    // horizontal padding: multipliers on d0 (pair starts an image row) and d3 (pair ends one).  The pair's pixel is
    // b + 2 + 2 j with b a multiple of 32: for W <= 32 a per-lane constant; for W >= 64 only the last two pairs of a tile can touch
    // a row boundary, and only in the tile that ends the row (decided per tile).
    float fl[8], fr[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const int px = 2 + 2 * (2 * ks + lk);
        const bool small = W <= 32;
        fl[ks] = (small && (px & (W - 1)) == 0) ? 0.f : 1.f;
        fr[ks] = (small && ((px + 2) & (W - 1)) == 0) ? 0.f : 1.f;
    }
    auto row_end_flags = [&](int t) {                            // W >= 64: pair 15 starts a row / pair 14 ends one iff b + 32 = 0 (mod W)
        if (W >= 64) {
            const bool e = ((32 * t) & (W - 1)) == 0;
            fl[7] = (e && lk == 1) ? 0.f : 1.f;
            fr[7] = (e && lk == 0) ? 0.f : 1.f;
        }
    };

    if (nIter > 0) {
        dma_tile(t0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int it = 0; it < nIter; ++it) {
            const int buf = it & 1;
            row_end_flags(t0 + it);
            const float* Af = fragA + buf * STAGE;
            const float* Bf = fragB + buf * STAGE;
            float a[2][2], d[2][4];
            auto frag = [&](int ks, float (&fa)[2], float (&fd)[4]) {
                fa[0] = Af[4 * ks];
                fa[1] = Af[4 * ks + 1];
                fd[0] = Bf[4 * ks];
                fd[1] = Bf[4 * ks + 1];
                fd[2] = Bf[4 * ks + 2];
                fd[3] = Bf[4 * ks + 3];
            };
            float ua[2][4], ub[2][4];                   // transformed operands, one k-step ahead of the MFMAs (see conv_wino_kernel)
            auto xform = [&](int ks, const float (&fa)[2], const float (&fd)[4], float (&xa)[4], float (&xb)[4]) {
                const float y0 = fa[0], y1 = fa[1];
                const float d0 = fd[0] * fl[ks], d1 = fd[1], d2 = fd[2], d3 = fd[3] * fr[ks];
                xa[0] = y0; xa[1] = y0 + y1; xa[2] = y0 - y1; xa[3] = -y1;
                xb[0] = d0 - d2; xb[1] = d1 + d2; xb[2] = d2 - d1; xb[3] = d1 - d3;
            };
            frag(0, a[0], d[0]);
#ifndef DPW_NO_VPIPE
            xform(0, a[0], d[0], ua[0], ub[0]);
#endif
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const int cur = ks & 1;
                if (ks + 1 < 8) frag(ks + 1, a[cur ^ 1], d[cur ^ 1]);
                __builtin_amdgcn_sched_barrier(0);
#ifdef DPW_NO_VPIPE
                xform(ks, a[cur], d[cur], ua[cur], ub[cur]);
#endif
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[cur][0], ub[cur][0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[cur][1], ub[cur][1], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[cur][2], ub[cur][2], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(ua[cur][3], ub[cur][3], acc[3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#ifndef DPW_NO_VPIPE
                if (ks + 1 < 8) { xform(ks + 1, a[cur ^ 1], d[cur ^ 1], ua[cur ^ 1], ub[cur ^ 1]); __builtin_amdgcn_sched_barrier(0); }
#endif
                if (ks == 1) { dma_tile(t0 + (it + 1 < nIter ? it + 1 : it), buf ^ 1); __builtin_amdgcn_sched_barrier(0); }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }

    // ---- epilogue: G -> the three taps of kernel row ky; output element (m, c, tap) as in nt_gemm_fast
    const int o_cs = p.o_col_stride ? p.o_col_stride : p.ntaps;
    const long long o_ts = p.o_tap_stride ? p.o_tap_stride : 1;
    float* __restrict__ outb = p.out + (long long)split * p.o_bs;
    const int col = n0 + wc * 32 + (lane & 31);
    if (col >= p.NCOLS || (src1 && col >= C1)) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m >= p.M) continue;
        const float hs = 0.5f * (acc[1][r] + acc[2][r]);
        const float t[3] = {acc[0][r] + hs, 0.5f * (acc[1][r] - acc[2][r]), hs + acc[3][r]};
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            float* o = outb + (long long)(3 * ky + kx) * o_ts + (long long)m * p.ldo + (long long)col * o_cs;
            float v = p.alpha * t[kx];
            if (p.accumulate) v += *o;
            *o = v;
        }
    }
}

static bool wgrad_wino_ok(const dp_nt_gemm_params& p) {
    const dp_conv_geom& g = p.g;
    if (p.batched || p.merge || p.col_bias || p.ntaps != 9 || g.kw != 3 || g.stride != 1 || g.sden != 1 || g.ups) return false;
    if (g.pad_t != 1 || g.pad_l != 1 || g.Ho != g.Hs || g.Wo != g.Ws || g.Hs != g.Hv || g.Ws != g.Wv) return false;
    const int W = g.Wo, HW = g.Ho * g.Wo;
    if (W < 8 || W > 256 || (W & (W - 1)) || (HW & (HW - 1)) || HW < 64) return false;
    if ((p.P % 32) || (p.p_per_split % 32) || p.p_per_split <= 0) return false;
    // same rule as wino_bk: the input descriptor is one image row longer than the tensor and must not reach the 0x80000000 marker
    if ((unsigned long long)p.x1_bytes + 4ull * W >= 0x80000000ull || (p.X2 && (unsigned long long)p.x2_bytes + 4ull * W >= 0x80000000ull)) return false;
    const int bn = p.tile == 3 ? 96 : 64;
    if (p.X2 && (g.c_split % bn)) return false;
    return true;
}

extern "C" int dp_wgrad_wino_supported(const dp_nt_gemm_params* p) { return wgrad_wino_ok(*p) ? 1 : 0; }

// p as for dp_nt_gemm's weight-gradient launches (A = dy, X1 / X2 = the convolution input, ntaps = 9); the split-K range is counted
// in K tiles of 32 pixels over P/32 + 1 tiles: splits * p_per_split must cover P + 32 pixels.  tile = 3: 96 x 96 tiles (9 waves).
extern "C" int dp_wgrad_wino(const dp_nt_gemm_params* pp, void* stream) {
    const dp_nt_gemm_params& p = *pp;
    if (p.M <= 0 || p.NCOLS <= 0) return 0;
    if (!wgrad_wino_ok(p) || p.splits <= 0 || (long long)p.splits * p.p_per_split < (long long)p.P + 32) return (int)hipErrorInvalidValue;
    const int C1 = p.X2 ? p.g.c_split : p.NCOLS;
    const int bt = p.tile == 3 ? 96 : 64;
    dim3 grid((C1 + bt - 1) / bt + (p.X2 ? (p.NCOLS - C1 + bt - 1) / bt : 0), (p.M + bt - 1) / bt, 3 * p.splits);
    if (p.tile == 3) DP_LAUNCH((wgrad_wino_kernel<3, 3>), grid, dim3(576), 0, (hipStream_t)stream, p);
    else             DP_LAUNCH((wgrad_wino_kernel<2, 2>), grid, dim3(256), 0, (hipStream_t)stream, p);
    return DP_LAUNCH_CHECK();
}
