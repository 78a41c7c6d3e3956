// 3x3 stride-1 'same' convolution as a TWO-DIMENSIONAL Winograd F(2x2, 3x3) implicit GEMM on the fp32 matrix cores (round 6).
//
// csrc/winograd.hip runs F(2, 3) along W only (12 multiplies per output pair and channel instead of 18: 2/3 of the direct form) and
// says why the 2-D form did not fit: 16 position accumulators per output tile.  They do fit when the POSITIONS are spread over the
// wavefronts instead of the rows: F(2x2, 3x3) does 16 multiplies per 2x2 output tile and channel instead of 36 -- 4/9 of the direct
// form, 2/3 of F(2, 3) --
//     U = G g G^T (4x4 per (m, c), packed once: dp_pack_weight_wino2d),  V = B^T d B (4x4 per (c, tile), d = the 4x4 input patch),
//     M[i][j] += U[i][j] V[i][j] over c,   Y = A^T M A (2x2 outputs),
//     B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1],  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1],  A^T = [1 1 1 0; 0 1 -1 -1].
// One 256-thread workgroup = 64 output channels x 32 tiles (128 output pixels: 128 / W whole image rows) x 16 positions.
// Wavefront i owns ROW i of the position matrix -- positions (i, 0..3) -- for all 64 rows x 32 tiles: 4 positions x 2 row blocks x
// 16 = 128 accumulator registers, two workgroups per CU.  Why this mapping:
//   * row i of V needs only TWO of the four patch rows (B^T row i has two non-zeros): a lane reads the 2 x 2 raw pixels of its (channel,
//     tile) that no other tile owns from LDS (two conflict-free 8-byte reads), takes the two halo columns of the row combination
//     from its neighbouring lanes (two DPP wave shifts) and forms its four B operands with 6 VALU adds and two border selects per
//     8 MFMAs;
//   * the raw input tile in LDS is [channel][4 patch rows][tile row][W]: vertical zero padding and image boundaries are out-of-range
//     byte offsets of the LDS-DMA loads (the hardware writes zeros), horizontal padding two per-lane selects;
//   * one K tile (8 channels) carries all 16 positions: there is no kernel-row loop, 32 MFMAs per wavefront between barriers;
//   * the output transform is 2 x 2 signed sums inside the wavefront (columns) and one exchange through LDS between the four
//     wavefronts (rows): 64 KB written and read once per workgroup, against ~1.3 MB read during its K loop.
// Same parameter block, epilogue operands and split-K contract as dp_conv_wino.  fp32 everywhere; the result differs from the direct
// form by re-association (~1e-6 of the output scale).
#include <cstdlib>
#include "dp_common.h"

#define DPW2_RSRC_FLAGS 0x00020000
#define DPW2_OOB 0x80000000u
typedef __attribute__((address_space(3))) void dpw2_lds_void;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t dpw2_rsrc(const float* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, DPW2_RSRC_FLAGS);
}

// One 8-byte LDS read as inline asm: the waitcnt pass treats a 64-bit-typed LDS load as aliasing the LDS-DMA writes of the prefetch
// and drains vmcnt(0) in front of it (on the ISA, round 6: in the middle of every K tile of the 8-channel variant); it does not look
// into inline asm.  The asm's result is pending on lgkmcnt: dpw2_lds_wait() before its first use.
template <int OFF>
__device__ __forceinline__ float2 dpw2_lds_read_b64(unsigned addr) {
    unsigned long long v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return __builtin_bit_cast(float2, v);
}
__device__ __forceinline__ void dpw2_lds_wait(float2& a, float2& b) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b) : : "memory");
}
__device__ __forceinline__ unsigned dpw2_lds_addr(const float* p) {
    return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const float*)p;
}

namespace {
constexpr int W2_BM = 64;                      // output channels per workgroup
constexpr int W2_BT = 32;                      // 2x2 tiles per workgroup (128 output pixels)
}

// W2_BK channels per K tile: 8 (32 MFMAs per wavefront between barriers, 80 KB of LDS: two workgroups per CU) or 4 (16 MFMAs, 40 KB:
// three workgroups per CU when the registers allow it)
// SEG (images wider than 64 pixels): a block is ONE tile row of 32 tiles = a 2-row x 64-column segment; the two halo columns of its
// 4-row patch belong to the neighbouring segments (or are the image border) and come through a small extra tile [k][patch row][side][4].
// TAIL (M % 64 in 1 .. 32, e.g. the pruned models' 96-wide layers): the workgroups of the last row tile have no second row block --
// they run the K loop without its MFMAs and A fragments (half the matrix work of that tile instead of multiplying zeros).
// BMT: output channels per workgroup, 64 (two row blocks of 32: 128 accumulators) or 32 (one block: 64 accumulators, 93 registers and 32 KB
// of LDS = FIVE workgroups per CU).  The 32-row form re-loads and re-transforms the input tile for every 32 rows and is 5-7 % slower than the
// 64-row form on full tiles, but on the 65 .. 96-row layers of the pruned models (one full + one half-empty 64-row tile) it is 17-19 %
// faster than even the tail instantiation: its short K tiles (8 MFMAs per wavefront) are covered by five resident workgroups instead of
// three [measured, round 6, profiles/round6_wino2d_m32.txt].
template <int W2_BK, int OCC, bool SEG, bool TAIL, int BMT = 64>
__device__ __forceinline__ void conv_wino2d_body(const dp_conv_gemm_params& p) {
    constexpr int W2_BM = BMT;                     // (shadows the namespace constant)
    constexpr int NBLK = BMT / 32;                 // row blocks of 32 output channels
    constexpr int W2_A_SZ = 16 * W2_BK * W2_BM;    // [pos][k][m] floats (32 KB at BK = 8)
    constexpr int W2_B_SZ = W2_BK * 64 * 4;        // [k][4 patch rows][tile row][W] floats, tile rows x W = 64 (8 KB at BK = 8)
    constexpr int W2_H_SZ = SEG ? 256 : 0;         // halo columns of a segment: [k][patch row][left | right][4 floats], one wave instruction
    constexpr int W2_STAGE = W2_A_SZ + W2_B_SZ + W2_H_SZ;
    // the epilogue's exchange buffer reuses this memory: 64 KB at once when the K loop's buffers hold it, else 32 KB in two passes
    __shared__ __attribute__((aligned(16))) float smem[(2 * W2_STAGE > 8192) ? 2 * W2_STAGE : 8192];       // (8192 floats: one row block's exchange)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);             // = row i of the position matrix
    // row tiles of one pixel block back to back on one XCD (they read the same input block): see conv_wino_kernel
    int bxx = blockIdx.x, byy = blockIdx.y;
    if (!(gridDim.x & 7)) {
        const int b = blockIdx.x + gridDim.x * blockIdx.y;
        const int xcd = b & 7, slot = b >> 3;
        const int gy = gridDim.y;
        const int grp = slot / gy;
        byy = slot - grp * gy;
        bxx = grp * 8 + xcd;
    }
    const int m0 = byy * W2_BM;
    const dp_conv_geom& g = p.g;
    const int W = g.Wo, H = g.Ho, HW = H * W;
    const int lw = 31 - __clz(W);                      // W is a power of two
    const int TC = W >> 1;                             // tiles per image row
    const int C = p.C;
    const int C1 = p.X2 ? g.c_split : C;
    const int nIterAll = C / W2_BK;
    const bool ksplit = p.ksplit > 1;
    const int per = ksplit ? (nIterAll + p.ksplit - 1) / p.ksplit : nIterAll;
    const int it0 = ksplit ? (int)blockIdx.z * per : 0;
    const int nIter = ksplit ? max(0, min(per, nIterAll - it0)) : nIterAll;
    // first global image row (over all images) of this block (even) and, for segments, its first column
    const int row0 = SEG ? 2 * (bxx >> (lw - 6)) : (bxx * 128) >> lw;
    const int x0 = SEG ? (bxx & ((1 << (lw - 6)) - 1)) << 6 : 0;
    const int rows_all = p.NPIX >> lw;                 // N * H

    // ---- A loader: 16-byte chunk e = tid + 256 j of [pos 16][k 8][m 64]
    constexpr int NJA = W2_BK * W2_BM / 64;         // 16-byte chunks per thread: 16 positions x BK channels x BM / 4 chunks over 256 threads
    constexpr int MCH = W2_BM / 4;                  // chunks per (position, channel) row
    unsigned a_voff[NJA];
#pragma unroll
    for (int j = 0; j < NJA; ++j) {
        const int e = tid + 256 * j;
        const int pos = e / (MCH * W2_BK), k = (e / MCH) % W2_BK, m = m0 + 4 * (e % MCH);
        a_voff[j] = (m < p.lda) ? (unsigned)(((pos * C + k) * p.lda + m) * 4) : DPW2_OOB;
    }
    const __amdgpu_buffer_rsrc_t rA = dpw2_rsrc(p.A, p.a_bytes);
    // ---- B loader: 16-byte chunk e = tid + 256 j of [k][patch row 4][tile row][W / 4]: 64 chunks per channel
    constexpr int NJB = W2_BK / 4;
    unsigned b_voff1[NJB], b_voff2[NJB];
#pragma unroll
    for (int j = 0; j < NJB; ++j) {
        const int e = tid + 256 * j;
        const int k = e >> 6, within = e & 63;
        const int r = within >> 4;                     // 16 chunks per patch row: (tile rows) x (W / 4)
        const int rem = within & 15;
        const int tr = SEG ? 0 : rem >> (lw - 2), cx = SEG ? rem : rem & ((W >> 2) - 1);
        const int rg = row0 + 2 * tr;                  // global row of the tile row's first output row
        const int img = rg / H, y = rg - img * H + r - 1;
        const bool v = rg < rows_all && (unsigned)y < (unsigned)H;
        const unsigned lin = (unsigned)(k * HW + y * W + x0 + 4 * cx);
        b_voff1[j] = v ? ((unsigned)((long long)img * g.x1_img_stride) + lin) * 4u : DPW2_OOB;
        b_voff2[j] = v ? ((unsigned)((long long)img * g.x2_img_stride) + lin) * 4u : DPW2_OOB;
    }
    // ---- halo loader (segments): chunk e = tid < 8 BK of [k][patch row][side]: the 4 pixels left of / right of the segment
    unsigned h_voff1 = DPW2_OOB, h_voff2 = DPW2_OOB;
    if (SEG && lane < 8 * W2_BK) {
        const int k = lane >> 3, r = (lane >> 1) & 3, side = lane & 1;
        const int img = row0 / H, y = row0 - img * H + r - 1;
        const int col = side ? x0 + 64 : x0 - 4;
        const bool v = row0 < rows_all && (unsigned)y < (unsigned)H && (unsigned)col < (unsigned)W;      // outside the image: zeros
        const unsigned lin = (unsigned)(k * HW + y * W + col);
        h_voff1 = v ? ((unsigned)((long long)img * g.x1_img_stride) + lin) * 4u : DPW2_OOB;
        h_voff2 = v ? ((unsigned)((long long)img * g.x2_img_stride) + lin) * 4u : DPW2_OOB;
    }
    const __amdgpu_buffer_rsrc_t r1 = dpw2_rsrc(p.X1, p.x1_bytes);
    const __amdgpu_buffer_rsrc_t r2 = dpw2_rsrc(p.X2 ? p.X2 : p.X1, p.X2 ? p.x2_bytes : p.x1_bytes);

    float* const ldsA = smem + 4 * (wave * 64);                       // + buf*STAGE + 1024*j   (chunk e -> float 4 e)
    float* const ldsB = smem + W2_A_SZ + 4 * (wave * 64);             // + buf*STAGE + 1024*j

    auto dma_tile = [&](int buf, int ch) {
        const unsigned a_soff = (unsigned)(ch * W2_BK) * (unsigned)p.lda * 4u;
#pragma unroll
        for (int j = 0; j < NJA; ++j) {
            unsigned o = a_voff[j];
            asm volatile("" : "+v"(o));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (dpw2_lds_void*)(ldsA + buf * W2_STAGE + 1024 * j), 16, (int)o, (int)a_soff, 0, 0);
        }
        const int c0 = ch * W2_BK;
        const bool first = c0 < C1;
        const unsigned b_soff = (unsigned)((first ? c0 : c0 - C1) * HW * 4);
#pragma unroll
        for (int j = 0; j < NJB; ++j) {
            unsigned o = first ? b_voff1[j] : b_voff2[j];
            asm volatile("" : "+v"(o));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(first ? r1 : r2, (dpw2_lds_void*)(ldsB + buf * W2_STAGE + 1024 * j), 16, (int)o,
                                                     (int)b_soff, 0, 0);
        }
        if (SEG) {
            // branch-free: every wavefront issues the same 64-lane load (lanes >= 8 BK are out of range and zero the padding of the
            // halo tile; the four copies write the same bytes) -- an exec-masked or one-wavefront load would split the K loop's block
            unsigned o = first ? h_voff1 : h_voff2;
            asm volatile("" : "+v"(o));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(first ? r1 : r2, (dpw2_lds_void*)(smem + W2_A_SZ + W2_B_SZ + buf * W2_STAGE), 16,
                                                     (int)o, (int)b_soff, 0, 0);
        }
    };

    f32x16 acc[4][NBLK];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < NBLK; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][t][r] = 0.f;

    // ---- fragment addressing: lane = (k parity, li); A row 32 t + li of position (wave, j); B tile li
    const int li = lane & 31, lk = lane >> 5;
    const float* fragA = smem + ((wave * 4) * W2_BK + lk) * W2_BM + li;           // + (j*BK + 2 ks)*BM + 32 t
    const int tr_l = SEG ? 0 : li >> (lw - 1), tc_l = SEG ? li : li & (TC - 1);      // TC = W / 2 is a power of two
    // the two patch rows of B^T row `wave`: (0, 2) -, (1, 2) +, (2, 1) -, (1, 3) -
    const int ra = wave == 0 ? 0 : wave == 2 ? 2 : 1;
    const int rb = wave == 0 ? 2 : wave == 1 ? 2 : wave == 2 ? 1 : 3;
    const float sgn = wave == 1 ? 1.f : -1.f;
    // [k][patch row][tile row][W]: the 32 lanes of a k half read 8 bytes each at 32 different 8-byte bank pairs (tile rows are W
    // floats apart and (tile rows) x W = 64 = the bank row): conflict-free ds_read_b64
    const float* fragB1 = smem + W2_A_SZ + lk * 256 + ra * 64 + (tr_l << lw) + 2 * tc_l;      // + 2 ks * 256; cols 0, 1 of the tile
    const float* fragB2 = smem + W2_A_SZ + lk * 256 + rb * 64 + (tr_l << lw) + 2 * tc_l;
    const bool pad_l = tc_l == 0, pad_r = tc_l == (SEG ? 31 : TC - 1);
    // segments: the halo tile's column x0 - 1 (fourth float of the left chunk) and x0 + 64 (first of the right one), rows ra / rb
    const float* fragH = smem + W2_A_SZ + W2_B_SZ + lk * 32;            // + 2 ks * 32 + (row*2 + side)*4 + {3 | 0}

    int ch = it0;
    if (nIter > 0) {
        dma_tile(0, ch);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (NBLK == 1 || (TAIL && m0 + 32 >= p.M)) {
#define W2_NT 1
#include "winograd2d_kloop.inc"
#undef W2_NT
    } else if (NBLK == 2) {
#define W2_NT 2
#include "winograd2d_kloop.inc"
#undef W2_NT
    }

    // ---- output transform.  Columns inside the wavefront: Z[i][0] = M[i][0] + M[i][1] + M[i][2], Z[i][1] = M[i][1] - M[i][2] - M[i][3];
    //      rows across the wavefronts through LDS: Y[0][q] = Z[0][q] + Z[1][q] + Z[2][q], Y[1][q] = Z[1][q] - Z[2][q] - Z[3][q].
    //      The K loop's buffers are free (every wavefront is past its last barrier).  PASSES = 1: both row blocks at once, 64 KB,
    //      wavefront w finishes row block w >> 1, registers 8 (w & 1) .. + 7.  PASSES = 2 (the 40 KB variant): one row block per pass,
    //      32 KB, wavefront w finishes registers 4 w .. 4 w + 3 of it.
    constexpr int PASSES = (NBLK == 1 || 2 * W2_STAGE >= 16384) ? 1 : 2;
    constexpr int BPP = NBLK / PASSES;                 // row blocks exchanged per pass
    constexpr int NR = 4 * BPP;                        // accumulator registers (= output rows) a wavefront finishes per pass
    float* zbuf = smem;
    // the lane's tile coordinates once more, from a laundered lane id: nothing of the epilogue's addressing stays live across the K
    // loop (the 40 KB variant runs at the 168-register cap of three workgroups per CU and would spill it)
    int li_e = threadIdx.x & 31;
    asm volatile("" : "+v"(li_e));
    const int tr_e = SEG ? 0 : li_e >> (lw - 1), tc_e = SEG ? li_e : li_e & (TC - 1);
    const int rg = row0 + 2 * tr_e;                    // global row of this lane's tile
    const bool tile_ok = rg < rows_all;
    int H_e = H;
    asm volatile("" : "+s"(H_e));                      // (... including the reciprocal of the division by H)
    const int img = tile_ok ? rg / H_e : 0, y = tile_ok ? rg - img * H_e : 0;
    const int r_in = y * W + x0 + 2 * tc_e;
    float* optr = p.out + (long long)img * p.o_img_stride + r_in;
    const float* rptr = p.res ? p.res + (long long)img * p.r_img_stride + r_in : nullptr;
    const float* tptr = p.tadd ? p.tadd + (long long)img * p.tadd_stride : nullptr;
    float* wsb = ksplit ? p.ws + (long long)blockIdx.z * p.M * p.NPIX + (long long)img * HW + r_in : nullptr;
#pragma unroll
    for (int pass = 0; pass < PASSES; ++pass) {
        if (pass) __syncthreads();                     // the previous pass's reads are done
#pragma unroll
        for (int t = 0; t < NBLK; ++t) {
            if (PASSES == 2 && t != pass) continue;
            const int tz = PASSES == 2 ? 0 : t;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                zbuf[(((wave * 2 + 0) * BPP + tz) * 16 + r) * 64 + lane] = (acc[0][t][r] + acc[1][t][r]) + acc[2][t][r];
                zbuf[(((wave * 2 + 1) * BPP + tz) * 16 + r) * 64 + lane] = (acc[1][t][r] - acc[2][t][r]) - acc[3][t][r];
            }
        }
        __syncthreads();
        if (!tile_ok) continue;
        const int t_o = PASSES == 2 ? pass : (wave * NR) >> 4;           // (two blocks at once: wave >> 1, registers 8 (wave & 1) ..)
        const int tz_o = PASSES == 2 ? 0 : t_o;
        const int r_o = (wave * NR) & 15;
        if (ksplit) {
#pragma unroll
            for (int q8 = 0; q8 < NR; ++q8) {
                const int r = r_o + q8;
                const int m = m0 + 32 * t_o + 4 * lk + (r & 3) + 8 * (r >> 2);
                float z[4][2];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int q = 0; q < 2; ++q) z[i][q] = zbuf[(((i * 2 + q) * BPP + tz_o) * 16 + r) * 64 + lane];
                if (m >= p.M) continue;
                float* o = wsb + (long long)m * p.NPIX;
                *reinterpret_cast<float2*>(o) = make_float2((z[0][0] + z[1][0]) + z[2][0], (z[0][1] + z[1][1]) + z[2][1]);
                *reinterpret_cast<float2*>(o + W) = make_float2((z[1][0] - z[2][0]) - z[3][0], (z[1][1] - z[2][1]) - z[3][1]);
            }
            continue;
        }
        // all NR rows' operand loads in flight before the first use (clamped row index; rows >= M skip the store)
        int mc[NR];
        float tb[NR], tt[NR];
        float2 tr0[NR], tr1[NR], tp0[NR], tp1[NR];
#pragma unroll
        for (int q8 = 0; q8 < NR; ++q8) {
            const int r = r_o + q8;
            const int m = m0 + 32 * t_o + 4 * lk + (r & 3) + 8 * (r >> 2);
            mc[q8] = m < p.M ? m : p.M - 1;
            tb[q8] = tt[q8] = 0.f;
            tr0[q8] = tr1[q8] = tp0[q8] = tp1[q8] = make_float2(0.f, 0.f);
        }
        if (p.bias) {
#pragma unroll
            for (int q8 = 0; q8 < NR; ++q8) tb[q8] = p.bias[mc[q8]];
        }
        if (tptr) {
#pragma unroll
            for (int q8 = 0; q8 < NR; ++q8) tt[q8] = tptr[mc[q8]];
        }
        if (rptr) {
#pragma unroll
            for (int q8 = 0; q8 < NR; ++q8) {
                tr0[q8] = *reinterpret_cast<const float2*>(rptr + (long long)mc[q8] * HW);
                tr1[q8] = *reinterpret_cast<const float2*>(rptr + (long long)mc[q8] * HW + W);
            }
        }
        if (p.accumulate) {
#pragma unroll
            for (int q8 = 0; q8 < NR; ++q8) {
                tp0[q8] = *reinterpret_cast<const float2*>(optr + (long long)mc[q8] * HW);
                tp1[q8] = *reinterpret_cast<const float2*>(optr + (long long)mc[q8] * HW + W);
            }
        }
#pragma unroll
        for (int q8 = 0; q8 < NR; ++q8) {
            const int r = r_o + q8;
            const int m = m0 + 32 * t_o + 4 * lk + (r & 3) + 8 * (r >> 2);
            float z[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int q = 0; q < 2; ++q) z[i][q] = zbuf[(((i * 2 + q) * BPP + tz_o) * 16 + r) * 64 + lane];
            float y00 = p.alpha * ((z[0][0] + z[1][0]) + z[2][0]), y01 = p.alpha * ((z[0][1] + z[1][1]) + z[2][1]);
            float y10 = p.alpha * ((z[1][0] - z[2][0]) - z[3][0]), y11 = p.alpha * ((z[1][1] - z[2][1]) - z[3][1]);
            if (p.bias) { y00 += tb[q8]; y01 += tb[q8]; y10 += tb[q8]; y11 += tb[q8]; }
            if (tptr) { y00 += tt[q8]; y01 += tt[q8]; y10 += tt[q8]; y11 += tt[q8]; }
            if (rptr) { y00 += tr0[q8].x; y01 += tr0[q8].y; y10 += tr1[q8].x; y11 += tr1[q8].y; }
            y00 *= p.post_scale; y01 *= p.post_scale; y10 *= p.post_scale; y11 *= p.post_scale;
            if (p.act == 1) { y00 = fmaxf(y00, 0.f); y01 = fmaxf(y01, 0.f); y10 = fmaxf(y10, 0.f); y11 = fmaxf(y11, 0.f); }
            if (p.accumulate) { y00 += tp0[q8].x; y01 += tp0[q8].y; y10 += tp1[q8].x; y11 += tp1[q8].y; }
            if (m < p.M) {
                *reinterpret_cast<float2*>(optr + (long long)m * HW) = make_float2(y00, y01);
                *reinterpret_cast<float2*>(optr + (long long)m * HW + W) = make_float2(y10, y11);
            }
        }
    }
}

template <int W2_BK, int OCC, bool SEG>
__global__ __launch_bounds__(256, OCC) void conv_wino2d_kernel(const dp_conv_gemm_params p) {
    conv_wino2d_body<W2_BK, OCC, SEG, false>(p);
}
template <int W2_BK, int OCC, bool SEG>
__global__ __launch_bounds__(256, OCC) void conv_wino2d_tail_kernel(const dp_conv_gemm_params p) {
    conv_wino2d_body<W2_BK, OCC, SEG, true>(p);
}
template <int W2_BK, int OCC, bool SEG>
__global__ __launch_bounds__(256, OCC) void conv_wino2d_m32_kernel(const dp_conv_gemm_params p) {
    conv_wino2d_body<W2_BK, OCC, SEG, false, 32>(p);
}

// Shapes the kernel takes: 3x3, stride 1, pad 1, no upsampling, W a power of two in 4 .. 256 (128 output pixels = whole image rows of
// whole tile rows up to 64 pixels; one 2 x 64 segment of a tile row beyond), H even, channel counts (per concat source) in multiples of 8, 8-byte aligned image planes.
static bool wino2d_ok(const dp_conv_gemm_params& p) {
    const dp_conv_geom& g = p.g;
    if (p.a_kc || p.ntaps != 9 || g.kw != 3 || g.stride != 1 || g.sden != 1 || g.ups || g.pad_t != 1 || g.pad_l != 1) return false;
    if (g.Ho != g.Hs || g.Wo != g.Ws || g.Hs != g.Hv || g.Ws != g.Wv || p.batches > 1 || (p.ksplit > 1 && !p.ws)) return false;
    const int W = g.Wo, H = g.Ho;
    if (W < 4 || W > 256 || (W & (W - 1)) || (H & 1)) return false;
    if ((p.lda & 3) || p.NPIX % (H * W)) return false;
    if ((g.x1_img_stride & 1) || (p.X2 && (g.x2_img_stride & 1)) || (p.o_img_stride & 1) || (p.res && (p.r_img_stride & 1))) return false;
    if ((unsigned long long)p.x1_bytes >= 0x80000000ull || (p.X2 && (unsigned long long)p.x2_bytes >= 0x80000000ull)) return false;
    const int C1 = p.X2 ? g.c_split : p.C;
    return p.C % 8 == 0 && C1 % 8 == 0 && p.C >= 8;
}

extern "C" int dp_conv_wino2d_supported(const dp_conv_gemm_params* p) { return wino2d_ok(*p) ? 1 : 0; }
extern "C" int dp_conv_splitk_epilogue(const dp_conv_gemm_params* p, void* stream);      // gemm.hip

extern "C" int dp_conv_wino2d(const dp_conv_gemm_params* pp, void* stream) {
    const dp_conv_gemm_params& p = *pp;
    if (p.M <= 0 || p.NPIX <= 0) return 0;
    if (!wino2d_ok(p)) return (int)hipErrorInvalidValue;
    dim3 grid((p.NPIX + 127) / 128, (p.M + W2_BM - 1) / W2_BM, p.ksplit > 1 ? p.ksplit : 1);
    // 32-row tiles for the big grids of layers with at most 96 output rows that leave a half-empty 64-row tile (DP_WINO2D_M32=0: never,
    // =2: every big grid -- measurements only)
    static const int m32 = [] { const char* e = getenv("DP_WINO2D_M32"); return e ? atoi(e) : 1; }();
    if (p.g.Wo <= 64 && p.ksplit <= 1 && (long long)grid.x * grid.y > 512 &&
        (m32 == 2 || (m32 == 1 && p.M <= 96 && (p.M & 63) >= 1 && (p.M & 63) <= 32))) {
        grid.y = (p.M + 31) / 32;
        DP_LAUNCH((conv_wino2d_m32_kernel<4, 5, false>), grid, dim3(256), 0, (hipStream_t)stream, p);
        return DP_LAUNCH_CHECK();
    }
    // K tiles of 4 channels (40 KB of LDS, 164 VGPRs: three workgroups per CU) for grids beyond one round of two per CU, K tiles
    // of 8 (80 KB, two per CU, half the barriers) for the small and the split-K grids.  [measured, round 6,
    // profiles/round6_wino2d_variants.txt, batch 256: 256 -> 256 @ 16 x 16 0.329 -> 0.307 ms, 128 -> 128 @ 32 x 32 0.367 -> 0.350,
    // 384 -> 384 @ 32 x 32 (12 latents) 0.187 -> 0.167; 256 -> 256 @ 8 x 8 0.097 -> 0.100, @ 4 x 4 (split-K) 0.063 -> 0.069]
    static const int forced = [] { const char* e = getenv("DP_WINO2D_VARIANT"); return e ? atoi(e) : -1; }();
    const long long wgs = (long long)grid.x * grid.y * grid.z;
    const int variant = forced >= 0 ? forced : ((p.ksplit <= 1 && wgs > 512) ? 1 : 0);     // 512 = one round of two per CU
    static const bool tail_off = [] { const char* e = getenv("DP_WINO2D_TAIL"); return e && atoi(e) == 0; }();
    const bool tail = !tail_off && (p.M & 63) >= 1 && (p.M & 63) <= 32;       // the last row tile holds one row block only
    if (p.g.Wo > 64) {                       // 2 x 64-pixel segments of two image rows: 4-channel K tiles (42 KB of LDS) at whatever
        (void)variant;                       // occupancy ~180 registers allow (the halo operands do not fit under the 168 of three per CU)
        if (tail) DP_LAUNCH((conv_wino2d_tail_kernel<4, 2, true>), grid, dim3(256), 0, (hipStream_t)stream, p);
        else      DP_LAUNCH((conv_wino2d_kernel<4, 2, true>), grid, dim3(256), 0, (hipStream_t)stream, p);
    } else if (variant == 1) {
        if (tail) DP_LAUNCH((conv_wino2d_tail_kernel<4, 3, false>), grid, dim3(256), 0, (hipStream_t)stream, p);
        else      DP_LAUNCH((conv_wino2d_kernel<4, 3, false>), grid, dim3(256), 0, (hipStream_t)stream, p);
    } else {
        if (tail) DP_LAUNCH((conv_wino2d_tail_kernel<8, 2, false>), grid, dim3(256), 0, (hipStream_t)stream, p);
        else      DP_LAUNCH((conv_wino2d_kernel<8, 2, false>), grid, dim3(256), 0, (hipStream_t)stream, p);
    }
    const int e = DP_LAUNCH_CHECK();
    if (e || p.ksplit <= 1) return e;
    return dp_conv_splitk_epilogue(pp, stream);
}

// U[(pos*K + k)][ld], pos = 4 i + j, U = G g G^T, from a torch [Co][Ci][3][3] weight.  mode 0 (forward): K = Ci, columns m = co, taps
// as stored; mode 1 (input gradient): K = Co, columns m = ci, both tap axes flipped.
__global__ __launch_bounds__(256) void pack_weight_wino2d_kernel(const float* __restrict__ Wt, int Co, int Ci, int mode,
                                                                 float* __restrict__ dst, int ld) {
    const int K = mode == 0 ? Ci : Co, Mv = mode == 0 ? Co : Ci;
    const long long total = 16ll * K * ld;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int m = (int)(idx % ld);
        const long long rk = idx / ld;
        const int k = (int)(rk % K);
        const int pos = (int)(rk / K);
        const int i = pos >> 2, j = pos & 3;
        float v = 0.f;
        if (m < Mv) {
            const float* w = mode == 0 ? Wt + ((long long)m * Ci + k) * 9 : Wt + ((long long)k * Ci + m) * 9;
            float gg[3][3];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) gg[a][b] = mode == 0 ? w[a * 3 + b] : w[(2 - a) * 3 + (2 - b)];
            // t[b] = (G g)[i][b], then v = sum_b t[b] G[j][b]
            float t[3];
#pragma unroll
            for (int b = 0; b < 3; ++b)
                t[b] = i == 0 ? gg[0][b] : i == 1 ? ((gg[0][b] + gg[1][b]) + gg[2][b]) * 0.5f : i == 2 ? ((gg[0][b] - gg[1][b]) + gg[2][b]) * 0.5f : gg[2][b];
            v = j == 0 ? t[0] : j == 1 ? ((t[0] + t[1]) + t[2]) * 0.5f : j == 2 ? ((t[0] - t[1]) + t[2]) * 0.5f : t[2];
        }
        dst[idx] = v;
    }
}

extern "C" int dp_pack_weight_wino2d(const float* W, int Co, int Ci, int mode, float* dst, int ld, void* stream) {
    const long long total = 16ll * (mode == 0 ? Ci : Co) * ld;
    if (total <= 0) return 0;
    long long nb = (total + 255) / 256;
    if (nb > 4096) nb = 4096;
    DP_LAUNCH(pack_weight_wino2d_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, W, Co, Ci, mode, dst, ld);
    return DP_LAUNCH_CHECK();
}
