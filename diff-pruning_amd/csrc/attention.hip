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

// What both schedules share: workgroup -> (image, head, query block), the three operand slabs, this wave's lane offsets.
struct AtCtx {
    int lane, li, half, w, T, d, dv, nqb, i0;
    __amdgpu_buffer_rsrc_t qr, kr, vr;
    float* __restrict__ ob;
    unsigned klane, vlane;       // per-lane byte offsets of the K (= Q) and V operand loads inside a 32-key block
    float c2;                    // scale * log2 e: softmax in base 2 (v_exp_f32)
};

__device__ __forceinline__ AtCtx at_setup(const dp_attention_params& p) {
    AtCtx c;
    c.lane = threadIdx.x & 63;
    c.li = c.lane & 31;
    c.half = c.lane >> 5;
    c.w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    c.T = p.T;
    c.d = p.d;
    c.dv = p.dv;
    c.nqb = p.T >> 5;
    // XCD-aware order: workgroup b runs on XCD b % 8; the query blocks of one (image, head) re-read the same K and V slabs,
    // so they go to ONE XCD (one L2) -- XCD x takes the logical blocks [x * n / 8, (x + 1) * n / 8)
    unsigned L = blockIdx.x;
    if ((gridDim.x & 7u) == 0) L = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int z = (int)(L / (unsigned)c.nqb);
    c.i0 = ((int)L - z * c.nqb) * 32;
    const int n = z / p.heads, h = z - n * p.heads;
    c.qr = at_rsrc(p.q + n * p.q_bs + (long long)h * c.d * c.T, (unsigned)c.d * c.T * 4u);
    c.kr = at_rsrc(p.k + n * p.k_bs + (long long)h * c.d * c.T, (unsigned)c.d * c.T * 4u);
    c.vr = at_rsrc(p.v + n * p.v_bs + (long long)h * c.dv * c.T, (unsigned)c.dv * c.T * 4u);
    c.ob = p.o + n * p.o_bs + (long long)h * c.dv * c.T;
    // byte offset of a load = per-lane part (one VGPR, advanced per key block) + wave-uniform part (scalar registers); the
    // callers make the per-block value opaque (empty asm) so that hipcc does not hoist one offset VGPR per load out of the
    // key loop (16 * NT + 4 * NT registers, an occupancy step)
    c.klane = (unsigned)(c.half * c.T + c.li) * 4u;
    c.vlane = (unsigned)(c.li * c.T + 4 * c.half) * 4u;
    c.c2 = p.scale * 1.44269504088896340736f;
    return c;
}

// this wave's share of Q (constant over the key blocks): channel tiles w, w + 4, ...
template <int NT>
__device__ __forceinline__ void at_load_q(const AtCtx& c, float (&qreg)[NT][16]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int s = 0; s < 16; ++s)
            qreg[t][s] = at_load(c.qr, (unsigned)((32 * (c.w + 4 * t) + 2 * s + c.half) * c.T + c.i0 + c.li) * 4u);
}
// K operand of channel tile t for the 32 keys at lane offset kj; V operand likewise (four 16-byte loads)
__device__ __forceinline__ void at_load_k(const AtCtx& c, float (&dst)[16], int t, unsigned kj) {
#pragma unroll
    for (int s = 0; s < 16; ++s) dst[s] = at_load(c.kr, kj + (unsigned)((32 * (c.w + 4 * t) + 2 * s) * c.T) * 4u);
}
__device__ __forceinline__ void at_load_v(const AtCtx& c, f32x4 (&dst)[4], int t, unsigned vj) {
#pragma unroll
    for (int g = 0; g < 4; ++g) dst[g] = at_load4(c.vr, vj + (unsigned)(32 * (c.w + 4 * t) * c.T + 8 * g) * 4u);
}
// O^T tile += V tile * P^T: MFMA step r = 4 g + e multiplies component e of the g-th 16-byte V load with probability register r
__device__ __forceinline__ void at_pv_tile(f32x16& oacc, const f32x4 (&vt)[4], const float (&sc)[16]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 vv = vt[g];
        oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(vv.x, sc[4 * g], oacc, 0, 0, 0);
        oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(vv.y, sc[4 * g + 1], oacc, 0, 0, 0);
        oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(vv.z, sc[4 * g + 2], oacc, 0, 0, 0);
        oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(vv.w, sc[4 * g + 3], oacc, 0, 0, 0);
    }
}
// partial S^T tile of this wave -> its slot of the exchange buffer
__device__ __forceinline__ void at_publish(f32x4 (&slot)[4][64], const f32x16& sacc, int lane) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 q4;
        q4.x = sacc[4 * g]; q4.y = sacc[4 * g + 1]; q4.z = sacc[4 * g + 2]; q4.w = sacc[4 * g + 3];
        slot[g][lane] = q4;
    }
}
// O = accumulator / (sum of both lane halves' exponentials)
template <int NT>
__device__ __forceinline__ void at_store(const AtCtx& c, const f32x16 (&oacc)[NT], float lrun) {
    const float inv = 1.0f / (lrun + __shfl_xor(lrun, 32, 64));
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int c0 = 32 * (c.w + 4 * t);
        if (c0 < c.dv) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ch = c0 + (r & 3) + 8 * (r >> 2) + 4 * c.half;
                if (ch < c.dv) c.ob[(long long)ch * c.T + c.i0 + c.li] = oacc[t][r] * inv;
            }
        }
    }
}

// Plain schedule: S^T -> exchange -> softmax -> P.V per key block, a rolling pair of operand buffers one channel tile (16 MFMAs)
// ahead.  Taken for heads wider than 256 channels, where the pipelined form below needs more than 256 registers.
template <int NT>
__global__ __launch_bounds__(256) void attn_fwd_fused_kernel(const dp_attention_params p) {
    __shared__ f32x4 xch[2][4][4][64];                   // [buffer][wave][register quad][lane]: 32 KB
    const AtCtx c = at_setup(p);
    float qreg[NT][16];
    at_load_q<NT>(c, qreg);
    f32x16 oacc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
    float mrun = -INFINITY, lrun = 0.f;
    float kb[2][16];
    f32x4 vb[2][4];

    at_load_k(c, kb[0], 0, c.klane);
    for (int jb = 0; jb < c.nqb; ++jb) {
        unsigned kj = c.klane + (unsigned)jb * 128u, vj = c.vlane + (unsigned)jb * 128u;
        asm volatile("" : "+v"(kj), "+v"(vj));
        // ---- partial S^T over this wave's channels ----------------------------------------------------------------
        f32x16 sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t + 1 < NT) at_load_k(c, kb[(t + 1) & 1], t + 1, kj);      // next tile's K under this tile's MFMAs
            else at_load_v(c, vb[0], 0, vj);                               // first V tile under the exchange + softmax
            if (32 * (c.w + 4 * t) < c.d) {
#pragma unroll
                for (int s = 0; s < 16; ++s)
                    sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kb[t & 1][s], qreg[t][s], sacc, 0, 0, 0);
            }
        }
        // ---- exchange: every wave ends up with the same full tile ---------------------------------------------------
        const int xb = jb & 1;
        at_publish(xch[xb][c.w], sacc, c.lane);
        __syncthreads();
        if (jb + 1 < c.nqb) at_load_k(c, kb[0], 0, kj + 128u);             // next key block's first K tile (kb[0] is free)
        float sc[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 a0 = xch[xb][0][g][c.lane], a1 = xch[xb][1][g][c.lane], a2 = xch[xb][2][g][c.lane],
                        a3 = xch[xb][3][g][c.lane];
            sc[4 * g] = (((a0.x + a1.x) + a2.x) + a3.x) * c.c2;
            sc[4 * g + 1] = (((a0.y + a1.y) + a2.y) + a3.y) * c.c2;
            sc[4 * g + 2] = (((a0.z + a1.z) + a2.z) + a3.z) * c.c2;
            sc[4 * g + 3] = (((a0.w + a1.w) + a2.w) + a3.w) * c.c2;
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
            sc[r] = __builtin_amdgcn_exp2f(sc[r] - mnew);            // arguments <= 0: v_exp_f32 is good to ~1 ulp there
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
            if (t + 1 < NT) at_load_v(c, vb[(t + 1) & 1], t + 1, vj);
            if (32 * (c.w + 4 * t) < c.dv) {
                if (rescale) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[t][r] *= alpha;
                }
                at_pv_tile(oacc[t], vb[t & 1], sc);
            }
        }
    }
    at_store<NT>(c, oacc, lrun);
}

// Software-pipelined schedule.  In the plain one a wavefront alternates between matrix work (S^T, P.V) and ~230 vector
// instructions of exchange + softmax during which ITS share of the matrix pipe idles.  Here the S^T MFMAs of key block j + 1 are
// issued BETWEEN the softmax stages of block j (ten stages, pinned with sched_barrier): an MFMA runs 64 cycles in the matrix
// pipe after it issues, the vector instructions of the stage issue behind it meanwhile.  Whole-block operand registers: K of
// block j + 2 is requested when block j + 1's S^T has issued, V of block j + 1 when block j's P.V has issued.  Same arithmetic
// in the same order as the plain schedule (channel tiles past d read zeros here instead of being skipped).
template <int NT>
__global__ __launch_bounds__(256) void attn_fwd_pipe_kernel(const dp_attention_params p) {
    __shared__ f32x4 xch[2][4][4][64];
    const AtCtx c = at_setup(p);
    float qreg[NT][16];
    at_load_q<NT>(c, qreg);
    f32x16 oacc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
    float mrun = -INFINITY, lrun = 0.f;
    float kb[NT][16];
    f32x4 vb[NT][4];

    // prologue: S^T of key block 0
    {
#pragma unroll
        for (int t = 0; t < NT; ++t) at_load_k(c, kb[t], t, c.klane);
        f32x16 s0;
#pragma unroll
        for (int r = 0; r < 16; ++r) s0[r] = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int s = 0; s < 16; ++s) s0 = __builtin_amdgcn_mfma_f32_32x32x2f32(kb[t][s], qreg[t][s], s0, 0, 0, 0);
        const unsigned k1 = c.klane + (c.nqb > 1 ? 128u : 0u);
#pragma unroll
        for (int t = 0; t < NT; ++t) at_load_k(c, kb[t], t, k1);
#pragma unroll
        for (int t = 0; t < NT; ++t) at_load_v(c, vb[t], t, c.vlane);
        at_publish(xch[0][c.w], s0, c.lane);
        __syncthreads();
    }

    auto block = [&](int jb, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        const int xb = jb & 1;
        unsigned kj2 = c.klane + (unsigned)(jb + 2) * 128u, vj1 = c.vlane + (unsigned)(jb + 1) * 128u;
        asm volatile("" : "+v"(kj2), "+v"(vj1));
        // ---- softmax of block jb, with the S^T MFMAs of block jb + 1 issued between its stages -------------------------
        f32x16 snext;
#pragma unroll
        for (int r = 0; r < 16; ++r) snext[r] = 0.f;
        f32x4 part[2][4];
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) part[0][ww] = xch[xb][ww][0][c.lane];
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
                    for (int ww = 0; ww < 4; ++ww) part[(k + 1) & 1][ww] = xch[xb][ww][k + 1][c.lane];
                }
                const f32x4 a0 = part[k & 1][0], a1 = part[k & 1][1], a2 = part[k & 1][2], a3 = part[k & 1][3];
                sc[4 * k] = (((a0.x + a1.x) + a2.x) + a3.x) * c.c2;
                sc[4 * k + 1] = (((a0.y + a1.y) + a2.y) + a3.y) * c.c2;
                sc[4 * k + 2] = (((a0.z + a1.z) + a2.z) + a3.z) * c.c2;
                sc[4 * k + 3] = (((a0.w + a1.w) + a2.w) + a3.w) * c.c2;
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
        if (!LAST && jb + 2 < c.nqb) {                               // K registers are free: block jb + 2
#pragma unroll
            for (int t = 0; t < NT; ++t) at_load_k(c, kb[t], t, kj2);
        }
        // ---- O^T += V P^T ---------------------------------------------------------------------------------------------
        const bool rescale = __any(alpha != 1.0f);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            __builtin_amdgcn_sched_barrier(0);
            if (32 * (c.w + 4 * t) < c.dv) {
                if (rescale) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[t][r] *= alpha;
                }
                at_pv_tile(oacc[t], vb[t], sc);
            }
        }
        if (!LAST) {                                                 // V registers are free: block jb + 1; publish its S^T
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NT; ++t) at_load_v(c, vb[t], t, vj1);
            at_publish(xch[xb ^ 1][c.w], snext, c.lane);
            __syncthreads();
        }
    };
    for (int jb = 0; jb + 1 < c.nqb; ++jb) block(jb, std::false_type{});
    block(c.nqb - 1, std::true_type{});
    at_store<NT>(c, oacc, lrun);
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
    // variant: 1 = plain schedule (rolling operand buffers), 2 = software-pipelined, 0 = the library's choice per width
    // [measured, tools/bench_attention.py]: the pipelined form wins for heads of <= 256 channels (2 tiles per wavefront, two
    // workgroups per CU); above that it needs > 256 registers (one workgroup per CU) and the plain one wins
    const bool pipe = p->variant == 2 || (p->variant == 0 && nt <= 2);
#define AT_GO(NT_) do { if (pipe) DP_LAUNCH(attn_fwd_pipe_kernel<NT_>, grid, dim3(256), 0, st, *p); \
                        else DP_LAUNCH(attn_fwd_fused_kernel<NT_>, grid, dim3(256), 0, st, *p); } while (0)
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
