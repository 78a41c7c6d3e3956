// Fused attention forward for the no-grad forwards (DDPM / DDIM sampling, the CFG sampling of the LDM importance pass):
//   O[c][i] = sum_j V[c][j] * softmax_j(scale * sum_c' Q[c'][i] K[c'][j])
// replaces QK^T launch -> softmax launch -> P.V launch (attention_processor.py:415-470 get_attention_scores + bmm,
// ldm/modules/attention.py:168-193) without materialising the [T, T] scores.  Exact fp32 on the matrix cores
// (v_mfma_f32_32x32x2_f32), online softmax (running max / sum per query).  The forwards that save for a backward keep the
// three launches: the hand-written backward reads the materialised probabilities (a fused fp32 backward
// executes 7 T x T x d products against 4).
//
// Layout: activations are channel-major ([N, C, T], tokens contiguous), so BOTH products take their operands straight from
// global memory in MFMA operand order, no transposes and no LDS staging:
//   S^T tile [32 keys x 32 queries] = K^T Q :  a = K[c0 + 2s + half][j0 + lane&31],  b = Q[c0 + 2s + half][i0 + lane&31]
//     -> accumulator register r of a lane = S^T[key (r&3) + 8 (r>>2) + 4 half][query lane&31]: a lane holds 16 of the 32 keys
//        of ITS query, the other 16 sit in lane ^ 32 -> the per-query max / sum are in-lane reductions + one cross-half swap.
//   O^T tile [32 channels x 32 queries] += V P^T : MFMA step r takes the key pair (kA, kB) = (8 (r>>2) + (r&3), + 4), which is
//     exactly what accumulator register r of the two lane halves holds: b = p[r] (no shuffle, P never leaves the registers),
//     a = V[c0 + lane&31][j0 + 8 (r>>2) + 4 half + (r&3)] = component r&3 of one 16-byte load.
// Work split: one workgroup (4 wavefronts) per (image, head, 32 queries).  The head width here is 256 ... 576 channels, far too
// wide for one wavefront's registers, so the CHANNELS are split over the wavefronts in 32-channel tiles (wave w owns tiles
// w, w + 4, ...): each wave sums its channels' share of S^T, the four partial tiles are exchanged through LDS (one barrier per
// key block, double-buffered) and added in the fixed order 0..3 -- every wave then holds the same scores bit for bit, runs the
// same softmax update, and multiplies P into ITS channel tiles of O.  Deterministic: fixed order everywhere.
#include <type_traits>
#include "dp_common.h"

// no mul + add contraction in this file: the three schedules of the kernel below must round alike
#pragma clang fp contract(off)

#define AT_RSRC_FLAGS 0x00020000
__device__ __forceinline__ __amdgpu_buffer_rsrc_t at_rsrc(const float* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const float* b = (const float*)(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc((void*)b, 0, (int)__builtin_amdgcn_readfirstlane(bytes), AT_RSRC_FLAGS);
}
// raw buffer loads: an offset past the slab (a channel >= d of a ragged last tile) reads 0.0f
__device__ __forceinline__ float at_load(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0));
}
__device__ __forceinline__ f32x4 at_load4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0));
}

// DEEP: every K tile of key block j + 1 is in flight while block j's softmax and P.V run, every V tile of block j while its
// S^T runs (16 * NT + 16 * NT operand registers); else a rolling pair of tile buffers, one tile (16 MFMAs) ahead.
template <int NT, bool DEEP>
__global__ __launch_bounds__(256) void attn_fwd_fused_kernel(const dp_attention_params p) {
    __shared__ f32x4 xch[2][4][4][64];                   // [buffer][wave][register quad][lane]: 32 KB
    const int lane = threadIdx.x & 63;
    const int li = lane & 31, half = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int T = p.T, d = p.d, dv = p.dv;
    const int nqb = T >> 5;
    // XCD-aware order: workgroup b runs on XCD b % 8; the query blocks of one (image, head) re-read the same K and V slabs,
    // so they go to ONE XCD (one L2) -- XCD x takes the logical blocks [x * n / 8, (x + 1) * n / 8)
    unsigned L = blockIdx.x;
    if ((gridDim.x & 7u) == 0) L = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int z = (int)(L / (unsigned)nqb);
    const int i0 = ((int)L - z * nqb) * 32;
    const int n = z / p.heads, h = z - n * p.heads;
    const __amdgpu_buffer_rsrc_t qr = at_rsrc(p.q + n * p.q_bs + (long long)h * d * T, (unsigned)d * T * 4u);
    const __amdgpu_buffer_rsrc_t kr = at_rsrc(p.k + n * p.k_bs + (long long)h * d * T, (unsigned)d * T * 4u);
    const __amdgpu_buffer_rsrc_t vr = at_rsrc(p.v + n * p.v_bs + (long long)h * dv * T, (unsigned)dv * T * 4u);
    float* __restrict__ ob = p.o + n * p.o_bs + (long long)h * dv * T;

    // this wave's share of Q (constant over the key blocks): channel tiles w, w + 4, ...
    float qreg[NT][16];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int s = 0; s < 16; ++s)
            qreg[t][s] = at_load(qr, (unsigned)((32 * (w + 4 * t) + 2 * s + half) * T + i0 + li) * 4u);

    f32x16 oacc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
    float mrun = -INFINITY, lrun = 0.f;

    constexpr int NB = DEEP ? NT : 2;
    float kb[NB][16];
    f32x4 vb[NB][4];
    // byte offset = per-lane part (one VGPR, advanced per key block) + wave-uniform part (scalar registers): the empty asm keeps
    // hipcc from hoisting one offset VGPR per load out of the key loop (16 * NT + 4 * NT registers, an occupancy step)
    const unsigned klane = (unsigned)(half * T + li) * 4u;
    const unsigned vlane = (unsigned)(li * T + 4 * half) * 4u;
    auto load_k = [&](float (&dst)[16], int t, unsigned kj) {
#pragma unroll
        for (int s = 0; s < 16; ++s)
            dst[s] = at_load(kr, kj + (unsigned)((32 * (w + 4 * t) + 2 * s) * T) * 4u);
    };
    auto load_v = [&](f32x4 (&dst)[4], int t, unsigned vj) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
            dst[g] = at_load4(vr, vj + (unsigned)(32 * (w + 4 * t) * T + 8 * g) * 4u);
    };
    const float c2 = p.scale * 1.44269504088896340736f;              // softmax in base 2: exp(x) = exp2(x * log2 e)

    if (DEEP) {
#pragma unroll
        for (int t = 0; t < NT; ++t) load_k(kb[t], t, klane);
    } else {
        load_k(kb[0], 0, klane);
    }
    for (int jb = 0; jb < nqb; ++jb) {
        unsigned kj = klane + (unsigned)jb * 128u, vj = vlane + (unsigned)jb * 128u;
        asm volatile("" : "+v"(kj), "+v"(vj));
        // ---- partial S^T over this wave's channels ----------------------------------------------------------------
        f32x16 sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
        if (DEEP) {
#pragma unroll
            for (int t = 0; t < NT; ++t) load_v(vb[t], t, vj);       // this block's V under its S^T, exchange and softmax
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (!DEEP) {
                if (t + 1 < NT) load_k(kb[(t + 1) & 1], t + 1, kj);  // next tile's K under this tile's MFMAs
                else load_v(vb[0], 0, vj);                           // first V tile under the exchange + softmax
            }
            if (32 * (w + 4 * t) < d) {
#pragma unroll
                for (int s = 0; s < 16; ++s)
                    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kb[DEEP ? t : (t & 1)][s], qreg[t][s], sacc, 0, 0, 0);
            }
        }
        // ---- exchange: every wave ends up with the same full tile ---------------------------------------------------
        const int xb = jb & 1;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 q4;
            q4.x = sacc[4 * g]; q4.y = sacc[4 * g + 1]; q4.z = sacc[4 * g + 2]; q4.w = sacc[4 * g + 3];
            xch[xb][w][g][lane] = q4;
        }
        __syncthreads();
        if (jb + 1 < nqb) {                                          // next key block's K (its registers are free now)
            if (DEEP) {
#pragma unroll
                for (int t = 0; t < NT; ++t) load_k(kb[t], t, kj + 128u);
            } else {
                load_k(kb[0], 0, kj + 128u);
            }
        }
        float sc[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 a0 = xch[xb][0][g][lane], a1 = xch[xb][1][g][lane], a2 = xch[xb][2][g][lane], a3 = xch[xb][3][g][lane];
            sc[4 * g] = (((a0.x + a1.x) + a2.x) + a3.x) * c2;
            sc[4 * g + 1] = (((a0.y + a1.y) + a2.y) + a3.y) * c2;
            sc[4 * g + 2] = (((a0.z + a1.z) + a2.z) + a3.z) * c2;
            sc[4 * g + 3] = (((a0.w + a1.w) + a2.w) + a3.w) * c2;
        }
        // ---- online softmax for query lane&31 (the two halves of the wave hold 16 keys each) ---------------------------
        float bm = sc[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) bm = fmaxf(bm, sc[r]);
        bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
        const float mnew = fmaxf(mrun, bm);
        const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);     // first block: exp2(-inf) = 0
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sc[r] = __builtin_amdgcn_exp2f(sc[r] - mnew);            // arguments <= 0: v_exp_f32 is exact to ~1 ulp there
            ps += sc[r];
        }
        lrun = lrun * alpha + ps;                                    // this half's keys; the halves are added once, at the end
        mrun = mnew;
        // ---- O^T += V P^T on this wave's channel tiles ----------------------------------------------------------------
        // the running max rarely moves after the first blocks: alpha == 1 in every lane -> skip the multiplies (same bits)
        const bool rescale = __any(alpha != 1.0f);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            __builtin_amdgcn_sched_barrier(0);       // one tile at a time: the accumulators of the other tiles stay where they are
            if (!DEEP && t + 1 < NT) load_v(vb[(t + 1) & 1], t + 1, vj);
            if (32 * (w + 4 * t) < dv) {
                if (rescale) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[t][r] *= alpha;
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 vv = vb[DEEP ? t : (t & 1)][g];
                    oacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv.x, sc[4 * g], oacc[t], 0, 0, 0);
                    oacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv.y, sc[4 * g + 1], oacc[t], 0, 0, 0);
                    oacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv.z, sc[4 * g + 2], oacc[t], 0, 0, 0);
                    oacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv.w, sc[4 * g + 3], oacc[t], 0, 0, 0);
                }
            }
        }
    }
    const float inv = 1.0f / (lrun + __shfl_xor(lrun, 32, 64));
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int c0 = 32 * (w + 4 * t);
        if (c0 < dv) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = c0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (c < dv) ob[(long long)c * T + i0 + li] = oacc[t][r] * inv;
            }
        }
    }
}

// Software-pipelined form.  In the kernel above a wavefront alternates between matrix work (S^T, P.V) and ~230 vector
// instructions of exchange + softmax during which ITS share of the matrix pipe idles.  Here the S^T MFMAs of key block j + 1 are
// issued BETWEEN the softmax stages of block j (ten stages, pinned with sched_barrier): an MFMA runs 64 cycles in the matrix
// pipe after it issues, the vector instructions of the stage issue behind it meanwhile.  Operand registers as in DEEP: K of block
// j + 2 is requested when block j + 1's S^T has issued, V of block j + 1 when block j's P.V has issued.  Same arithmetic in the
// same order as the plain form.
template <int NT>
__global__ __launch_bounds__(256) void attn_fwd_pipe_kernel(const dp_attention_params p) {
    __shared__ f32x4 xch[2][4][4][64];
    const int lane = threadIdx.x & 63;
    const int li = lane & 31, half = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int T = p.T, d = p.d, dv = p.dv;
    const int nqb = T >> 5;
    unsigned L = blockIdx.x;
    if ((gridDim.x & 7u) == 0) L = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int z = (int)(L / (unsigned)nqb);
    const int i0 = ((int)L - z * nqb) * 32;
    const int n = z / p.heads, h = z - n * p.heads;
    const __amdgpu_buffer_rsrc_t qr = at_rsrc(p.q + n * p.q_bs + (long long)h * d * T, (unsigned)d * T * 4u);
    const __amdgpu_buffer_rsrc_t kr = at_rsrc(p.k + n * p.k_bs + (long long)h * d * T, (unsigned)d * T * 4u);
    const __amdgpu_buffer_rsrc_t vr = at_rsrc(p.v + n * p.v_bs + (long long)h * dv * T, (unsigned)dv * T * 4u);
    float* __restrict__ ob = p.o + n * p.o_bs + (long long)h * dv * T;

    float qreg[NT][16];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int s = 0; s < 16; ++s)
            qreg[t][s] = at_load(qr, (unsigned)((32 * (w + 4 * t) + 2 * s + half) * T + i0 + li) * 4u);
    f32x16 oacc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
    float mrun = -INFINITY, lrun = 0.f;

    float kb[NT][16];
    f32x4 vb[NT][4];
    const unsigned klane = (unsigned)(half * T + li) * 4u;
    const unsigned vlane = (unsigned)(li * T + 4 * half) * 4u;
    auto load_k = [&](float (&dst)[16], int t, unsigned kj) {
#pragma unroll
        for (int s = 0; s < 16; ++s)
            dst[s] = at_load(kr, kj + (unsigned)((32 * (w + 4 * t) + 2 * s) * T) * 4u);
    };
    auto load_v = [&](f32x4 (&dst)[4], int t, unsigned vj) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
            dst[g] = at_load4(vr, vj + (unsigned)(32 * (w + 4 * t) * T + 8 * g) * 4u);
    };
    auto publish = [&](const f32x16& sacc, int xb) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 q4;
            q4.x = sacc[4 * g]; q4.y = sacc[4 * g + 1]; q4.z = sacc[4 * g + 2]; q4.w = sacc[4 * g + 3];
            xch[xb][w][g][lane] = q4;
        }
    };
    const float c2 = p.scale * 1.44269504088896340736f;

    // prologue: S^T of key block 0 (channel tiles past d read zeros: their MFMAs add nothing)
    {
#pragma unroll
        for (int t = 0; t < NT; ++t) load_k(kb[t], t, klane);
        f32x16 s0;
#pragma unroll
        for (int r = 0; r < 16; ++r) s0[r] = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int s = 0; s < 16; ++s) s0 = __builtin_amdgcn_mfma_f32_32x32x2f32(kb[t][s], qreg[t][s], s0, 0, 0, 0);
        const unsigned k1 = klane + (nqb > 1 ? 128u : 0u);
#pragma unroll
        for (int t = 0; t < NT; ++t) load_k(kb[t], t, k1);
#pragma unroll
        for (int t = 0; t < NT; ++t) load_v(vb[t], t, vlane);
        publish(s0, 0);
        __syncthreads();
    }

    auto block = [&](int jb, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        const int xb = jb & 1;
        unsigned kj2 = klane + (unsigned)(jb + 2) * 128u, vj1 = vlane + (unsigned)(jb + 1) * 128u;
        asm volatile("" : "+v"(kj2), "+v"(vj1));
        // ---- softmax of block jb, with the S^T MFMAs of block jb + 1 issued between its stages -------------------------
        f32x16 snext;
#pragma unroll
        for (int r = 0; r < 16; ++r) snext[r] = 0.f;
        f32x4 part[2][4];
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) part[0][ww] = xch[xb][ww][0][lane];
        float sc[16];
        float bm = 0.f, mnew = 0.f, alpha = 0.f, ps = 0.f;
        constexpr int M = LAST ? 0 : NT * 16;
        constexpr int NS = 10;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
#pragma unroll
            for (int i = k * M / NS; i < (k + 1) * M / NS; ++i)
                snext = __builtin_amdgcn_mfma_f32_32x32x2f32(kb[i / 16][i % 16], qreg[i / 16][i % 16], snext, 0, 0, 0);
            if (k < 4) {
                if (k < 3) {
#pragma unroll
                    for (int ww = 0; ww < 4; ++ww) part[(k + 1) & 1][ww] = xch[xb][ww][k + 1][lane];
                }
                const f32x4 a0 = part[k & 1][0], a1 = part[k & 1][1], a2 = part[k & 1][2], a3 = part[k & 1][3];
                sc[4 * k] = (((a0.x + a1.x) + a2.x) + a3.x) * c2;
                sc[4 * k + 1] = (((a0.y + a1.y) + a2.y) + a3.y) * c2;
                sc[4 * k + 2] = (((a0.z + a1.z) + a2.z) + a3.z) * c2;
                sc[4 * k + 3] = (((a0.w + a1.w) + a2.w) + a3.w) * c2;
                // the empty asm statements pin each stage's results HERE: hipcc otherwise sinks the whole softmax below the
                // last MFMA (its results are first used by the P.V products)
                asm volatile("" : "+v"(sc[4 * k]), "+v"(sc[4 * k + 1]), "+v"(sc[4 * k + 2]), "+v"(sc[4 * k + 3]));
            } else if (k == 4) {
                bm = sc[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) bm = fmaxf(bm, sc[r]);
                asm volatile("" : "+v"(bm));
            } else if (k == 5) {
                bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
                mnew = fmaxf(mrun, bm);
                alpha = __builtin_amdgcn_exp2f(mrun - mnew);
                asm volatile("" : "+v"(mnew), "+v"(alpha));
            } else {
#pragma unroll
                for (int r = 4 * (k - 6); r < 4 * (k - 6) + 4; ++r) {
                    sc[r] = __builtin_amdgcn_exp2f(sc[r] - mnew);
                    ps += sc[r];
                }
                asm volatile("" : "+v"(sc[4 * (k - 6)]), "+v"(sc[4 * (k - 6) + 1]), "+v"(sc[4 * (k - 6) + 2]),
                             "+v"(sc[4 * (k - 6) + 3]), "+v"(ps));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        lrun = lrun * alpha + ps;
        mrun = mnew;
        if (!LAST && jb + 2 < nqb) {                                 // K registers are free: block jb + 2
#pragma unroll
            for (int t = 0; t < NT; ++t) load_k(kb[t], t, kj2);
        }
        // ---- O^T += V P^T ---------------------------------------------------------------------------------------------
        const bool rescale = __any(alpha != 1.0f);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            __builtin_amdgcn_sched_barrier(0);
            if (32 * (w + 4 * t) < dv) {
                if (rescale) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[t][r] *= alpha;
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 vv = vb[t][g];
                    oacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv.x, sc[4 * g], oacc[t], 0, 0, 0);
                    oacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv.y, sc[4 * g + 1], oacc[t], 0, 0, 0);
                    oacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv.z, sc[4 * g + 2], oacc[t], 0, 0, 0);
                    oacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vv.w, sc[4 * g + 3], oacc[t], 0, 0, 0);
                }
            }
        }
        if (!LAST) {                                                 // V registers are free: block jb + 1; publish its S^T
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NT; ++t) load_v(vb[t], t, vj1);
            publish(snext, xb ^ 1);
            __syncthreads();
        }
    };
    for (int jb = 0; jb + 1 < nqb; ++jb) block(jb, std::false_type{});
    block(nqb - 1, std::true_type{});

    const float inv = 1.0f / (lrun + __shfl_xor(lrun, 32, 64));
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int c0 = 32 * (w + 4 * t);
        if (c0 < dv) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = c0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (c < dv) ob[(long long)c * T + i0 + li] = oacc[t][r] * inv;
            }
        }
    }
}

// 1 = the shapes the fused kernel takes: tokens in whole 32-blocks, head widths up to 640 channels (5 tiles per wavefront)
extern "C" int dp_attention_fwd_supported(int T, int d, int dv) {
    const int tiles = ((d > dv ? d : dv) + 31) / 32;
    const int nt = (tiles + 3) / 4;
    return T >= 32 && (T & 31) == 0 && d >= 1 && dv >= 1 && nt <= 5 && (long long)(128 * nt + 32) * T * 4 < (1ll << 31);
}

extern "C" int dp_attention_fwd(const dp_attention_params* p, void* stream) {
    if (p->N <= 0) return 0;
    if (!dp_attention_fwd_supported(p->T, p->d, p->dv) || p->heads < 1) return (int)hipErrorInvalidValue;
    const int tiles = ((p->d > p->dv ? p->d : p->dv) + 31) / 32;
    const int nt = (tiles + 3) / 4;
    const dim3 grid((unsigned)((long long)p->N * p->heads * (p->T / 32)));
    hipStream_t st = (hipStream_t)stream;
    const int deep = p->variant == 2;
#define AT_GO(NT_) do { if (deep) DP_LAUNCH((attn_fwd_fused_kernel<NT_, true>), grid, dim3(256), 0, st, *p); \
                        else DP_LAUNCH((attn_fwd_fused_kernel<NT_, false>), grid, dim3(256), 0, st, *p); } while (0)
    // variant: 1 = rolling tile buffers, 2 = whole-block prefetch, 3 = software-pipelined, 0 = the library's choice per width
    // [measured, tools/bench_attention.py]: the pipelined form wins for heads of <= 256 channels (2 tiles per wavefront, two
    // workgroups per CU), the rolling buffers above that (the other two need > 256 registers there: one workgroup per CU)
    if (p->variant == 3 || (p->variant == 0 && nt <= 2)) {
        switch (nt) {
            case 1: DP_LAUNCH(attn_fwd_pipe_kernel<1>, grid, dim3(256), 0, st, *p); break;
            case 2: DP_LAUNCH(attn_fwd_pipe_kernel<2>, grid, dim3(256), 0, st, *p); break;
            case 3: DP_LAUNCH(attn_fwd_pipe_kernel<3>, grid, dim3(256), 0, st, *p); break;
            case 4: DP_LAUNCH(attn_fwd_pipe_kernel<4>, grid, dim3(256), 0, st, *p); break;
            default: DP_LAUNCH(attn_fwd_pipe_kernel<5>, grid, dim3(256), 0, st, *p); break;
        }
        return DP_LAUNCH_CHECK();
    }
    switch (nt) {
        case 1: AT_GO(1); break;
        case 2: AT_GO(2); break;
        case 3: AT_GO(3); break;
        case 4: AT_GO(4); break;
        default: AT_GO(5); break;
    }
#undef AT_GO
    return DP_LAUNCH_CHECK();
}
