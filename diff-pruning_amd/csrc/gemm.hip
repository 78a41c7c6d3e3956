// Implicit-GEMM contractions on the CDNA4 matrix cores in exact fp32
// (v_mfma_f32_32x32x2_f32: one rounding per product, bitwise an fmaf chain; 157 TFLOP/s dense peak).
//
//   conv_gemm : D[m][pix] = sum_{tap,c} A(tap,c,m) * X(pix,tap,c)        (forward / dgrad / linear / QK^T / dP)
//   nt_gemm   : D[m][n]   = sum_{pix}   A[m][pix]  * X(pix, n=(c,tap))   (wgrad / P.V / dQ), split over pix
//
// 256-thread workgroups (4 wavefronts, 2x2), 32x32x2 MFMA tiles, LDS double buffering with one
// barrier per K-tile: global loads for tile i+1 are issued before the MFMAs of tile i and committed
// to the other LDS buffer afterwards.  LDS images are laid out so that every ds_read_b32 /
// ds_write_b32 lane group touches 32 distinct banks ([k][m] for m-contiguous operands, [m][BK+1]
// for k-contiguous ones).
#include <cstdlib>
#include <type_traits>
#include "dp_common.h"

// compile-time unrolled loop: f(std::integral_constant<int, J>) for J in [0, N) -- the LDS-DMA builtins need constant
// instruction offsets
template <int J, int N, class F>
__device__ __forceinline__ void dp_static_for(F&& f) {
    if constexpr (J < N) {
        f(std::integral_constant<int, J>{});
        dp_static_for<J + 1, N>(f);
    }
}

// Global -> LDS DMA for the lane-linear tiles of conv_gemm (A/B: same speed on the large shapes, +4..8 % on the 8x8 / 4x4
// resolution layers, 16 fewer VGPRs -> 5 wavefronts/SIMD); compile with -UDP_LDSDMA ... to get the register-staged path.
#ifndef DP_NO_LDSDMA
#define DP_LDSDMA 1
#endif

// Raw buffer loads: the hardware range check (offset >= num_records -> 0.0f) replaces every bounds / padding /
// tail select, so the loaders are straight-line code and hipcc is free to hoist all global loads of tile i+1 above the
// MFMAs of tile i.  Invalid elements set bit 31 of the byte offset (all real offsets are < 2 GiB, enforced by ops.py).
#define DP_RSRC_FLAGS 0x00020000
#define DP_OOB 0x80000000u
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dp_rsrc(const float* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, DP_RSRC_FLAGS);
}
// Descriptor for a base that is wave-uniform by construction but not provably so to the compiler: pinning the words
// with readfirstlane keeps hipcc from wrapping every load in a waterfall loop (cdna_hip_programming.md T20).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dp_rsrc_uniform(const float* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const float* b = (const float*)(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc((void*)b, 0, (int)__builtin_amdgcn_readfirstlane(bytes), DP_RSRC_FLAGS);
}
// The empty asm makes the (selected) offset opaque: without it hipcc turns load(select(valid, off, OOB)) back into
// two predicated loads behind exec branches with a full vmcnt(0) wait after each.
__device__ __forceinline__ float dp_bload(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    asm volatile("" : "+v"(byte_off));
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0));   // b32 = raw bits
}
// LDS-DMA forms (buffer_load_dword[x4] ... lds): the wavefront writes base + lane*size directly into LDS, no VGPR
// staging and no ds_write; out-of-range lanes write 0 (probed on gfx950: tools/probe/lds_probe.hip).
typedef __attribute__((address_space(3))) void dp_lds_void;
__device__ __forceinline__ void dp_bload_lds(__amdgpu_buffer_rsrc_t r, float* lds_wave_base, unsigned byte_off) {
    asm volatile("" : "+v"(byte_off));
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (dp_lds_void*)lds_wave_base, 4, (int)byte_off, 0, 0, 0);
}
__device__ __forceinline__ void dp_bload4_lds(__amdgpu_buffer_rsrc_t r, float* lds_wave_base, unsigned byte_off) {
    asm volatile("" : "+v"(byte_off));
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (dp_lds_void*)lds_wave_base, 16, (int)byte_off, 0, 0, 0);
}
__device__ __forceinline__ float4 dp_bload4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    asm volatile("" : "+v"(byte_off));
    f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0));
    return make_float4(v.x, v.y, v.z, v.w);
}

// ------------------------------------------------------------------------------------------------
// MFMA core: acc[TM][TN] += A_tile * B_tile for one LDS-resident K-tile.
// A_KC: As[m*(BK+1) + k]  else As[k*BM + m];   B_KC: Bs[n*(BK+1) + k]  else Bs[k*BN + n].
// mfma_f32_32x32x2f32 operand map: a = A[i = lane&31][k = lane>>5], b = B[k = lane>>5][j = lane&31].
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, int BK, int TM, int TN, bool A_KC, bool B_KC>
__device__ __forceinline__ void mfma_tile(const float* __restrict__ As, const float* __restrict__ Bs,
                                          int wm0, int wn0, int lane, f32x16 (&acc)[TM][TN]) {
    const int li = lane & 31;
    const int lk = lane >> 5;
    // fragment reads are software-pipelined one k-step ahead of the MFMAs that consume them
    float a[2][TM], b[2][TN];
    auto frag = [&](int ks, float (&fa)[TM], float (&fb)[TN]) {
        const int kk = ks * 2 + lk;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int m = wm0 + tm * 32 + li;
            fa[tm] = A_KC ? As[m * (BK + 1) + kk] : As[kk * BM + m];
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int n = wn0 + tn * 32 + li;
            fb[tn] = B_KC ? Bs[n * (BK + 1) + kk] : Bs[kk * BN + n];
        }
    };
    frag(0, a[0], b[0]);
#ifdef DP_SETPRIO
    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int ks = 0; ks < BK / 2; ++ks) {
        const int cur = ks & 1;
        if (ks + 1 < BK / 2) frag(ks + 1, a[cur ^ 1], b[cur ^ 1]);
        __builtin_amdgcn_sched_barrier(0);      // keep the next k-step's LDS reads ahead of this k-step's MFMAs
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][tm], b[cur][tn], acc[tm][tn], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
#ifdef DP_SETPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
}

// Experiment knob: extra (unused) dynamic LDS per workgroup, to cap the workgroups resident per CU (DP_LDS_PAD bytes).
// ------------------------------------------------------------------------------------------------
// Epilogue shared by the conv_gemm kernels.  C/D map of a 32x32 MFMA tile: col j = lane&31,
// row i = (r&3) + 8*(r>>2) + 4*(lane>>5).  Sub-tile (tm, tn) of this wave starts at (row0 + tm*TMS, col0 + tn*TNS).
// ------------------------------------------------------------------------------------------------
// One 32x32 sub-tile (16 values per lane), 8 values at a time.  FULL: all 32 rows are < M.  The optional operands are
// tested once per half sub-tile and the 8 loads of an operand are issued before the first use: a null test plus a
// load + wait per element made the epilogue a chain of ~64 dependent memory latencies per lane.  Rows >= M (partial
// tiles) load from a clamped row and skip the store.
// [round 5, measured] four rows x ALL operands per batch (one wait per quarter sub-tile instead of one per operand and half): the
// 128 x 128 kernels fell from 107.9 to 103-105 TFLOP/s -- most launches carry 0-2 operands, for which that form waits four times
// per sub-tile instead of two; eight rows x all operands spills at the 128-VGPR cap.  Kept as it was.
template <bool FULL>
__device__ __forceinline__ void conv_epilogue_tile(const dp_conv_gemm_params& p, const f32x16& acc, int mrow0, int lane,
                                                   float* __restrict__ optr, const float* __restrict__ rptr,
                                                   const float* __restrict__ tptr, int HoWo) {
    const int mb = mrow0 + 4 * (lane >> 5);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float v[8];
        int mc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = h * 8 + q;
            const int m = mb + (r & 3) + 8 * (r >> 2);
            mc[q] = FULL ? m : min(m, p.M - 1);
            v[q] = p.alpha * acc[r];
        }
        if (p.bias) {
            float t[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) t[q] = p.bias[mc[q]];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] += t[q];
        }
        if (tptr) {
            float t[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) t[q] = tptr[mc[q]];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] += t[q];
        }
        if (rptr) {
            float t[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) t[q] = rptr[(long long)mc[q] * HoWo];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] += t[q];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] *= p.post_scale;
        if (p.act == 1) {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], 0.f);
        }
        if (p.accumulate) {
            float t[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) t[q] = optr[(long long)mc[q] * HoWo];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] += t[q];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = h * 8 + q;
            if (FULL || mb + (r & 3) + 8 * (r >> 2) < p.M) optr[(long long)mc[q] * HoWo] = v[q];
        }
    }
}

template <int TM, int TN, int TMS, int TNS>
__device__ __forceinline__ void conv_epilogue(const dp_conv_gemm_params& p, const f32x16 (&acc)[TM][TN], int row0, int col0,
                                              int lane, int z, bool ksplit) {
    const int HoWo = p.g.Ho * p.g.Wo;
#ifdef DP_EXP_NOEPI      // experiment: price the epilogue (keeps the accumulators alive, stores nothing in practice)
    {
        float s = 0.f;
        for (int tm = 0; tm < TM; ++tm) for (int tn = 0; tn < TN; ++tn) for (int r = 0; r < 16; ++r) s += acc[tm][tn][r];
        if (s == 1.2345e30f) p.out[0] = s;
        return;
    }
#endif
    if (ksplit) {
        float* __restrict__ wsb = p.ws + (long long)z * p.M * p.NPIX;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int pix = col0 + tn * TNS + (lane & 31);
            if (pix >= p.NPIX) continue;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = row0 + tm * TMS + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (m < p.M) wsb[(long long)m * p.NPIX + pix] = acc[tm][tn][r];
                }
        }
        return;
    }
    float* __restrict__ outb = p.out + (long long)z * p.o_bs;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int pix = col0 + tn * TNS + (lane & 31);
        if (pix >= p.NPIX) continue;
        const int img = pix / HoWo;
        const int r_in = pix - img * HoWo;
        float* optr = outb + (long long)img * p.o_img_stride + r_in;
        const float* rptr = p.res ? p.res + (long long)img * p.r_img_stride + r_in : nullptr;
        const float* tptr = p.tadd ? p.tadd + (long long)img * p.tadd_stride : nullptr;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int mrow0 = row0 + tm * TMS;
            if (mrow0 + 32 <= p.M) conv_epilogue_tile<true>(p, acc[tm][tn], mrow0, lane, optr, rptr, tptr, HoWo);
            else if (mrow0 < p.M)  conv_epilogue_tile<false>(p, acc[tm][tn], mrow0, lane, optr, rptr, tptr, HoWo);
        }
    }
}

#ifdef DP_CLOCK_PROBE
// Experiment build only: shader-clock / 100 MHz wall-clock ticks spent by workgroup (0,0,0) of the last conv_gemm launch.
__device__ unsigned long long dp_clk[2];
extern "C" int dp_debug_read_clock(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dp_clk), 16);
}
#endif

__device__ __forceinline__ float dp_splitk_sum(const float* __restrict__ ws, long long stride, int splits);

// Split-K without a second launch (round 4: MI355X_MICROARCH.md "publish-large" / cdna_hip_programming.md "in-launch split-K
// reduction", the write-through form).  Every workgroup of a tile stores its raw accumulators as a SLAB in the workspace --
// lane-linear 16-byte rows in accumulator-register order (float4 (sub-tile, q) of thread tid at ((sub*4 + q)*256 + tid)*16 bytes:
// each wave instruction writes 1 KB contiguous), with `sc1` WRITE-THROUGH buffer stores, so no L2 write-back fence is needed --
// drains its own stores (s_waitcnt vmcnt(0)), meets at the workgroup barrier, and ONE lane takes a ticket with a relaxed
// agent-scope atomic.  The workgroup that draws the last ticket reads all ksplit slabs of the tile back with `sc1` loads (L2 /
// fabric-served, never from its own L1) INTO the accumulator registers, summing in ascending split order with the arithmetic of
// dp_splitk_sum -- the order of conv_splitk_epilogue_kernel, so the bits do not depend on who arrives last -- and then runs the
// ordinary epilogue on them.  No per-thread fence, no per-element index arithmetic.
// The "I am last" flag lives in the kernel's ONE LDS array (free after the K loop's final barrier): a second __shared__ object
// made hipcc put an s_waitcnt vmcnt(0) behind the B-tile DMA of EVERY K tile of every conv_gemm_fast instantiation (round 3's
// fold did exactly that -- measured on the ISA, /tmp/isa in DESIGN section 4 item 26).
typedef unsigned dp_u32x4 __attribute__((ext_vector_type(4)));
#define DP_AUX_SC1 16
#ifndef DP_FOLD_ST_AUX
#define DP_FOLD_ST_AUX DP_AUX_SC1
#endif
#ifndef DP_FOLD_LD_AUX
#define DP_FOLD_LD_AUX DP_AUX_SC1
#endif

template <int TM, int TN, int TMS, int TNS>
__device__ __forceinline__ void conv_splitk_fold(const dp_conv_gemm_params& p, f32x16 (&acc)[TM][TN], int row0, int col0,
                                                 unsigned tile_id, int z, float* lds) {
    constexpr unsigned SLAB_BYTES = TM * TN * 16u * 256u * 4u;        // = BM * BN * 4
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const float* tile_ws = p.ws + (size_t)tile_id * (size_t)p.ksplit * (SLAB_BYTES / 4);
    const __amdgpu_buffer_rsrc_t rs = dp_rsrc_uniform(tile_ws, (unsigned)p.ksplit * SLAB_BYTES);
    // The store descriptor points at THIS slice's slab and the stores carry NO scalar offset: hipcc pads the "VALU overwrites the
    // data registers of a 128-bit buffer store" hazard only when soffset is not a register (GCNHazardRecognizer's rule), and with
    // soffset in an SGPR the next v_or really did clobber v[2:3] of the float4 before the store had read them on gfx950 --
    // sporadically, for 16 lanes at a time: [measured, round 4] slab element = the address temporary, bit for bit.
    const __amdgpu_buffer_rsrc_t rw = dp_rsrc_uniform(tile_ws + (size_t)z * (SLAB_BYTES / 4), SLAB_BYTES);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = {acc[tm][tn][4 * q], acc[tm][tn][4 * q + 1], acc[tm][tn][4 * q + 2], acc[tm][tn][4 * q + 3]};
                const unsigned off = (unsigned)((((tm * TN + tn) * 4 + q) * 256 + tid) * 16);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(dp_u32x4, v), rw, (int)off, 0, DP_FOLD_ST_AUX);
            }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this wave's slab rows have left for memory
    __syncthreads();                                                 // ... and so have the other three waves'
#ifdef DP_FOLD_RELEASE
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#endif
    unsigned* flag = reinterpret_cast<unsigned*>(lds);
    if (tid == 0) {
        unsigned* c = p.tile_counters + tile_id;
        const unsigned t = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned last = (t == (unsigned)p.ksplit - 1u) ? 1u : 0u;
        if (last) __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // zero again for the next launch
        *flag = last;
    }
    __syncthreads();
    if (*flag == 0u) return;
#ifdef DP_FOLD_ACQUIRE
    if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
#endif
    // reducer: sum_z slab[z] in ascending z, 8 slabs' loads in flight per accumulator row (dp_splitk_sum's arithmetic: the
    // accumulator starts at 0.0f and missing tail entries add 0.0f -- what a load with bit 31 of its vector offset set returns)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned off = (unsigned)((((tm * TN + tn) * 4 + q) * 256 + tid) * 16);
                f32x4 a = {0.f, 0.f, 0.f, 0.f};
                for (int z0 = 0; z0 < p.ksplit; z0 += 8) {
                    f32x4 v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {      // the range check covers the VECTOR offset only, not the scalar one
                        unsigned o = (z0 + j < p.ksplit) ? off : DP_OOB;
                        asm volatile("" : "+v"(o));
                        v[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                            rs, (int)o, (int)((unsigned)(z0 + j) * SLAB_BYTES), DP_FOLD_LD_AUX));
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) a += v[j];
                }
                acc[tm][tn][4 * q] = a.x; acc[tm][tn][4 * q + 1] = a.y; acc[tm][tn][4 * q + 2] = a.z; acc[tm][tn][4 * q + 3] = a.w;
            }
    conv_epilogue<TM, TN, TMS, TNS>(p, acc, row0, col0, lane, 0, false);
}

static unsigned dp_lds_pad() {
    static const unsigned pad = [] { const char* e = getenv("DP_LDS_PAD"); return e ? (unsigned)atoi(e) : 0u; }();
    return pad;
}

// ------------------------------------------------------------------------------------------------
// conv_gemm
// ------------------------------------------------------------------------------------------------
// 64 x 128 tiles: under the 128-VGPR cap of four workgroups per CU the compiler spilled 104-107 registers of this variant to
// scratch (204-220 bytes per lane, `hipcc -S` metadata, round 6) -- memory the HIP runtime has to find at dispatch; with two
// workgroups per CU asked for it keeps everything in registers.  The other variants fit (0-2 spills).
template <int BM, int BN, bool A_KC, bool STRADDLE>
__global__ __launch_bounds__(256, (BM == 64 && BN == 128) ? 2 : 4) void conv_gemm_kernel(const dp_conv_gemm_params p) {
    constexpr int BK = 16;
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int A_SZ = A_KC ? BM * (BK + 1) : BK * BM;
    constexpr int B_SZ = BK * BN;
    constexpr int STAGE = A_SZ + B_SZ;
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm0 = (wave >> 1) * WM;
    const int wn0 = (wave & 1) * WN;
    const int m0 = blockIdx.y * BM;
#ifdef DP_XCD
    // workgroup b is dispatched to XCD b % 8: remap so that each XCD owns a contiguous run of pixel tiles (neighbouring
    // tiles share input halo rows and all tiles share the weights -> private-L2 reuse).  Bijective for any grid size.
    int bx = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bx & 7, k = bx >> 3;
        bx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int n0 = bx * BN;
#else
    const int n0 = blockIdx.x * BN;
#endif
    const int z = blockIdx.z;

    const ConvGeom& g = p.g;
    const int HoWo = g.Ho * g.Wo;
    const int HsWs = g.Hs * g.Ws;
    const int C = p.C;
    const int nch = (C + BK - 1) / BK;
    const int nIterAll = p.ntaps * nch;
    // split-K (ksplit > 1, non-batched): workgroup z reduces K-tiles [it0, it0 + nIter) and stores its raw partial
    // tile to ws[z][m][pix]; dp_conv_splitk_epilogue sums the partials in ascending z and applies the epilogue.
    const bool ksplit = p.ksplit > 1;
    const int per = ksplit ? (nIterAll + p.ksplit - 1) / p.ksplit : nIterAll;
    const int it0 = ksplit ? z * per : 0;
    const int nIter = ksplit ? max(0, min(per, nIterAll - it0)) : nIterAll;
    const int zb = ksplit ? 0 : z;            // batch index (0 in split-K mode)

    const float* __restrict__ Ab = p.A + (long long)zb * p.a_bs;
    const float* __restrict__ X1 = p.X1 + (long long)zb * p.x_bs;
    const float* __restrict__ X2 = p.X2 ? p.X2 + (long long)zb * p.x_bs : nullptr;

    // ---- B loader: this thread owns pixel column bn of the tile for the whole K loop
    constexpr int NB = BK * BN / 256;
    constexpr int BROWS = 256 / BN;            // k rows covered per pass
    const int bn = tid % BN;
    const int bk0 = tid / BN;
    const int bpix = n0 + bn;
    const bool bpv = bpix < p.NPIX;
    int b_ho = 0, b_wo = 0;
    long long b_img1 = 0, b_img2 = 0;
    {
        const int pp = bpv ? bpix : 0;
        const int img = pp / HoWo;
        const int r = pp - img * HoWo;
        b_ho = r / g.Wo;
        b_wo = r - b_ho * g.Wo;
        b_img1 = (long long)img * g.x1_img_stride;
        b_img2 = (long long)img * g.x2_img_stride;
    }

    // ---- A loader
    constexpr int NA4 = A_KC ? 1 : (BK * BM / 4 / 256);      // float4 per thread (m-contiguous)
    constexpr int NAS = A_KC ? (BM * BK / 256) : 1;          // scalars per thread (k-contiguous)
    float4 ra4[NA4];
    float ras[NAS];
    float rb[NB];

    const __amdgpu_buffer_rsrc_t rA = dp_rsrc(Ab, p.a_bytes);
    const __amdgpu_buffer_rsrc_t r1 = dp_rsrc(X1, p.x1_bytes);
    const float* __restrict__ X2s = X2 ? X2 : X1;
    const unsigned x2b = X2 ? p.x2_bytes : p.x1_bytes;
    const __amdgpu_buffer_rsrc_t r2 = dp_rsrc(X2s, x2b);
    const unsigned pb1 = (unsigned)b_img1 * 4u;        // byte offset of this thread's image in X1 / X2
    const unsigned pb2 = (unsigned)b_img2 * 4u;
    const int csplit = g.c_split;

    // K order: channel chunk outer, kernel tap inner -- the 16-channel input slab (tile + halo, ~10 KB) is re-read by
    // the 9 taps back-to-back and stays in L1/L2 instead of being evicted between taps.
    auto load_tile = [&](int itr, bool live) {
        const int it = it0 + itr;
        const int ch = it / p.ntaps;
        const int tap = it - ch * p.ntaps;
        const int c0 = ch * BK;
        // A
        if constexpr (!A_KC) {
#pragma unroll
            for (int j = 0; j < NA4; ++j) {
                const int e = tid + 256 * j;
                const int k = e / (BM / 4);
                const int m = m0 + 4 * (e % (BM / 4));
                const bool v = live && (c0 + k < C) && (m < p.lda);
                const unsigned o = (unsigned)(((tap * C + c0 + k) * p.lda + m) * 4);
                ra4[j] = dp_bload4(rA, v ? o : DP_OOB);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NAS; ++j) {
                const int k = tid & (BK - 1);
                const int m = m0 + (tid / BK) + (256 / BK) * j;
                const bool v = live && (m < p.M) && (c0 + k < C);
                const unsigned o = (unsigned)((m * p.lda + c0 + k) * 4);
                ras[j] = dp_bload(rA, v ? o : DP_OOB);
            }
        }
        // B
        const int ky = tap / g.kw;
        const int kx = tap - ky * g.kw;
        int off;
        const bool tv = dp_gather(g, b_ho, b_wo, ky, kx, off) && bpv && live;
        const int cb = c0 + bk0;
        if constexpr (!STRADDLE) {                          // every K-chunk lives in one source (host-checked)
            const bool first = c0 < csplit;                  // chunk start decides (c_split % 16 == 0 or single source)
            const __amdgpu_buffer_rsrc_t rs = dp_rsrc_uniform(first ? X1 : X2s, first ? p.x1_bytes : x2b);
            const unsigned o0 = (first ? pb1 : pb2) + (unsigned)(((first ? cb : cb - csplit) * HsWs + off) * 4);
            const unsigned step = (unsigned)(BROWS * HsWs * 4);
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const bool v = tv && (cb + BROWS * j < C);
                rb[j] = dp_bload(rs, v ? (o0 + j * step) : DP_OOB);
            }
        } else {                                            // chunk straddles the concat boundary (pruned widths)
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int c = cb + BROWS * j;
                const bool v = tv && (c < C);
                const bool f1 = c < csplit;
                const unsigned o1 = pb1 + (unsigned)((c * HsWs + off) * 4);
                const unsigned o2 = pb2 + (unsigned)(((c - csplit) * HsWs + off) * 4);
                rb[j] = dp_bload(r1, (v && f1) ? o1 : DP_OOB) + dp_bload(r2, (v && !f1) ? o2 : DP_OOB);
            }
        }
    };

#ifdef DP_LDSDMA
    // global -> LDS directly (only for the m-contiguous, non-straddling variant: its LDS images are lane-linear)
    auto dma_tile = [&](int itr, bool live, int buf) {
        float* As = smem + buf * STAGE;
        float* Bs = As + A_SZ;
        const int it = it0 + itr;
        const int ch = it / p.ntaps;
        const int tap = it - ch * p.ntaps;
        const int c0 = ch * BK;
#pragma unroll
        for (int j = 0; j < NA4; ++j) {
            const int e = tid + 256 * j;
            const int k = e / (BM / 4);
            const int m = m0 + 4 * (e % (BM / 4));
            const bool v = live && (c0 + k < C) && (m < p.lda);
            const unsigned o = (unsigned)(((tap * C + c0 + k) * p.lda + m) * 4);
            dp_bload4_lds(rA, As + 4 * (e - lane), v ? o : DP_OOB);
        }
        const int ky = tap / g.kw;
        const int kx = tap - ky * g.kw;
        int off;
        const bool tv = dp_gather(g, b_ho, b_wo, ky, kx, off) && bpv && live;
        const int cb = c0 + bk0;
        const bool first = c0 < csplit;
        const __amdgpu_buffer_rsrc_t rs = dp_rsrc_uniform(first ? X1 : X2s, first ? p.x1_bytes : x2b);
        const unsigned o0 = (first ? pb1 : pb2) + (unsigned)(((first ? cb : cb - csplit) * HsWs + off) * 4);
        const unsigned step = (unsigned)(BROWS * HsWs * 4);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const bool v = tv && (cb + BROWS * j < C);
            dp_bload_lds(rs, Bs + (bk0 + BROWS * j) * BN + (bn - lane), v ? (o0 + j * step) : DP_OOB);
        }
    };
#endif

    auto store_tile = [&](int buf) {
        float* As = smem + buf * STAGE;
        float* Bs = As + A_SZ;
        if constexpr (!A_KC) {
#pragma unroll
            for (int j = 0; j < NA4; ++j) {
                const int e = tid + 256 * j;
                const int k = e / (BM / 4);
                const int m = 4 * (e % (BM / 4));
                *reinterpret_cast<float4*>(As + k * BM + m) = ra4[j];
            }
        } else {
#pragma unroll
            for (int j = 0; j < NAS; ++j) {
                const int k = tid & (BK - 1);
                const int m = (tid / BK) + (256 / BK) * j;
                As[m * (BK + 1) + k] = ras[j];
            }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) Bs[(bk0 + BROWS * j) * BN + bn] = rb[j];
    };

#ifdef DP_CLOCK_PROBE
    const unsigned long long clk0 = clock64(), wclk0 = wall_clock64();
#endif
    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    // The prefetch of tile it+1 is unconditional: past the last tile every offset is out of range (the buffer
    // unit returns zeros without touching memory), so there is no branch and no register merge that would force
    // hipcc to drain the loads before the MFMA block.
#ifdef DP_LDSDMA
    if constexpr (!A_KC && !STRADDLE) {
        dma_tile(0, nIter > 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int it = 0; it < nIter; ++it) {
            const int buf = it & 1;
            dma_tile(it + 1, it + 1 < nIter, buf ^ 1);       // lands in the other buffer while this one is consumed
            const float* As = smem + buf * STAGE;
            mfma_tile<BM, BN, BK, TM, TN, A_KC, false>(As, As + A_SZ, wm0, wn0, lane, acc);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // DMA complete before anyone reads buf^1
            __syncthreads();
        }
    } else
#endif
    {
        load_tile(0, nIter > 0);
        store_tile(0);
        __syncthreads();
        for (int it = 0; it < nIter; ++it) {
            const int buf = it & 1;
            load_tile(it + 1, it + 1 < nIter);
            const float* As = smem + buf * STAGE;
            mfma_tile<BM, BN, BK, TM, TN, A_KC, false>(As, As + A_SZ, wm0, wn0, lane, acc);
            store_tile(buf ^ 1);
            __syncthreads();
        }
    }

    // ---- epilogue.  C/D map of the 32x32 tile: col j = lane&31, row i = (r&3) + 8*(r>>2) + 4*(lane>>5).
#ifdef DP_CLOCK_PROBE
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0) {
        dp_clk[0] = clock64() - clk0;
        dp_clk[1] = wall_clock64() - wclk0;
    }
#endif
    if (ksplit && p.tile_counters)
        conv_splitk_fold<TM, TN, 32, 32>(p, acc, m0 + wm0, n0 + wn0, blockIdx.y * gridDim.x + (unsigned)(n0 / BN), z, smem);
    else
        conv_epilogue<TM, TN, 32, 32>(p, acc, m0 + wm0, n0 + wn0, lane, z, ksplit);
}


// ------------------------------------------------------------------------------------------------
// conv_gemm_fast: the stride-1 / no-upsample case, any channel count and concat split (every resnet 3x3, shortcut, attention
// projection and their dgrads; > 95 % of the conv_gemm time), 128x128 tile, same math and K order as conv_gemm_kernel.
// Plain VALU instructions are not free next to the matrix pipe (tools/probe/mfma_valu.hip: ~1.7 cycles each, the
// general loader spends ~100 per K tile = -9 %), so everything per-lane is hoisted out of the K loop:
//   * A: two constant per-lane byte offsets; the K-tile row offset goes in the buffer instruction's scalar offset.
//   * B: eight constant per-lane byte offsets (8 channels of one pixel); channel chunk and kernel tap move the SCALAR
//        offset ((c0*HW + ky*Ws + kx)*4, the descriptor base is shifted back by the padding so it is never negative);
//        zero padding = one precomputed tap-validity bit mask per lane -> 1 and + 1 cmp + 8 cndmask per K tile.
//   * LDS destinations (M0) are scalar; fragment reads use the interleaved wave tile (sub-tiles 64 apart), whose
//     k-step / sub-tile offsets all fit ds_read2st64_b32 immediates -> one address VGPR per operand.
// ------------------------------------------------------------------------------------------------
#ifndef DP_FAST_MINBLOCKS
#define DP_FAST_MINBLOCKS 4
#endif
// TAILS = false: every chunk is 16 channels wide (C % 16 == 0, split % 16 == 0): the K-tile bookkeeping is one scalar compare.
// X4 (output rows a multiple of 4 pixels wide, at most one padding column per side): the B tile is fetched with TWO
//   16-byte LDS-DMA loads per lane instead of eight 4-byte ones -- a lane owns 4 consecutive pixels of one channel row,
//   32 lanes one 128-pixel row, a wave-instruction two rows (1 KB).  The loads are only 4-byte aligned for the shifted
//   taps (legal for buffer loads; probed with tools/probe/lds_probe_x4.hip, range check per dword).  Vertical padding is
//   still the per-lane out-of-range offset; a horizontal shift makes the one element that falls on the left / right
//   image border read its neighbour row instead of zero: the owning lane overwrites it with 0.0f after its own
//   vmcnt(0) and before the barrier (6 of 9 taps, two ds_write_b32 for 1/8 .. 1/2 of the lanes).
template <int BM, int BN, bool TAILS, bool X4>
__global__ __launch_bounds__(256, DP_FAST_MINBLOCKS) void conv_gemm_fast_kernel(const dp_conv_gemm_params p) {
    // BM = 128: waves 2x2, each 64x64 as interleaved 32x32 sub-tiles (64 apart).  BM = 96 (pruned widths such as 90 or
    // 180 channels lose 30 % of a 128-row tile): waves 1x4, each all 96 rows x 32 columns.
    // BN = 64 (X4 only, BM = 128): half-width tiles for launches that would otherwise be ONE round of workgroups (256-channel
    // layers at 16x16: 1024 tiles = 4 per CU, all ramping up and storing their 64 KB outputs at the same time): twice the
    // workgroups in two rounds overlap the prologue / epilogue of one round with the K loop of the other.
    static_assert((BM == 128 || BM == 96) && (BN == 128 || (BN == 64 && BM == 128 && X4)), "fast path tiles: 128x128, 96x128, 128x64");
    constexpr int BK = 16;
    constexpr int WAVES_M = (BM == 128) ? 2 : 1;
    constexpr int TM = BM / 32 / WAVES_M, TN = BN / 32 / (4 / WAVES_M);
    constexpr int TMS = (BM == 128) ? 64 : 32, TNS = 64;        // sub-tile strides inside the workgroup tile
    constexpr int A_F4 = BK * BM / 4;                           // float4 elements of the A tile (512 or 384)
    constexpr int A_SZ = BK * BM;
    constexpr int B_SZ = BK * BN;
    constexpr int STAGE = A_SZ + B_SZ;
    __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wrow = (BM == 128) ? (wave >> 1) * 32 : 0;        // first row / column of this wave's sub-tile 0
    const int wcol = (BM == 128) ? (wave & 1) * 32 : wave * 32;
    const int m0 = blockIdx.y * BM;
    const int n0 = blockIdx.x * BN;
    const int z = blockIdx.z;

    const ConvGeom& g = p.g;
    const int HoWo = g.Ho * g.Wo;
    const int HsWs = g.Hs * g.Ws;
    const int C = p.C;
    const int ntaps = p.ntaps;
    // channel chunks of 16 per concat source; the last chunk of a source may be narrower (pruned widths)
    const int C1 = p.X2 ? g.c_split : C;
    const int nch1 = (C1 + BK - 1) / BK;
    const int nch = nch1 + (C - C1 + BK - 1) / BK;
    const int nIterAll = ntaps * nch;
    const bool ksplit = p.ksplit > 1;
    const int per = ksplit ? (nIterAll + p.ksplit - 1) / p.ksplit : nIterAll;
    const int it0 = ksplit ? z * per : 0;
    const int nIter = ksplit ? max(0, min(per, nIterAll - it0)) : nIterAll;
    const int zb = ksplit ? 0 : z;

    const float* __restrict__ Ab = p.A + (long long)zb * p.a_bs;
    const float* __restrict__ X1 = p.X1 + (long long)zb * p.x_bs;
    const float* __restrict__ X2 = p.X2 ? p.X2 + (long long)zb * p.x_bs : X1;
    // descriptor bases moved back by the padding (never-negative scalar offsets) and by IMM_MAX bytes: the 8 B-tile
    // loads of a lane land 1 KB apart in LDS, so 4 of them share one M0 and differ in the 12-bit instruction offset --
    // which the hardware adds to the memory address too; the per-lane offsets carry the compensation (IMM_MAX - imm).
    constexpr unsigned IMM_MAX = 3 * 1024;
    const int shift = g.pad_t * g.Ws + g.pad_l + (int)(IMM_MAX / 4);
    const __amdgpu_buffer_rsrc_t rA = dp_rsrc(Ab, p.a_bytes);
    const __amdgpu_buffer_rsrc_t r1 = dp_rsrc(X1 - shift, p.x1_bytes + 4u * (unsigned)shift);
    const __amdgpu_buffer_rsrc_t r2 = dp_rsrc(X2 - shift, (p.X2 ? p.x2_bytes : p.x1_bytes) + 4u * (unsigned)shift);
    auto chunk_width = [&](int c) {                              // valid channels of chunk c
        const int left = (c < nch1) ? C1 - c * BK : (C - C1) - (c - nch1) * BK;
        return left < BK ? left : BK;
    };

    // ---- per-lane constants
    // A: element e = tid + 256*j of the [16][BM] tile, 4 floats each: row k = e/(BM/4), column 4*(e%(BM/4))
    unsigned a_voff[2], a_base[2];
    int a_k[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int e = tid + 256 * j;
        a_k[j] = e / (BM / 4);
        const int m = m0 + 4 * (e - a_k[j] * (BM / 4));
        a_base[j] = (e < A_F4 && m < p.lda) ? (unsigned)((a_k[j] * p.lda + m) * 4) : DP_OOB;
    }
    const unsigned a_row_bytes = (unsigned)p.lda * 4u;
    // B: pixel column bn, rows bk0 + 2*j
    const int bn = tid & (BN - 1);
    const int bk0 = tid / BN;
    const int bpix = n0 + bn;
    const bool bpv = bpix < p.NPIX;
    unsigned b_pix1, b_pix2, vmask = 0;
    {
        const int pp = bpv ? bpix : 0;
        const int img = pp / HoWo;
        const int r = pp - img * HoWo;
        const int ho = r / g.Wo;
        const int wo = r - ho * g.Wo;
        const unsigned lin = (unsigned)(ho * g.Ws + wo + bk0 * HsWs);
        b_pix1 = (unsigned)((long long)img * g.x1_img_stride) + lin;
        b_pix2 = (unsigned)((long long)img * g.x2_img_stride) + lin;
        if (bpv)
            for (int t = 0; t < ntaps; ++t) {
                const int ky = t / g.kw;
                int off;
                if (dp_gather(g, ho, wo, ky, t - ky * g.kw, off)) vmask |= 1u << t;
            }
    }
    // X4: lane = (row of the wave's group, pixel group): pixels n0 + 4*g4 .. +3 of channel rows 4*RPW*j + RPW*wave + rsub
    constexpr int RPW = 256 / BN;                        // rows one wave-instruction covers (2 for BN = 128, 4 for BN = 64)
    constexpr int NJ = BK / (4 * RPW);                   // 16-byte loads per lane and K tile (2 / 1)
    const int g4 = lane & (BN / 4 - 1), rsub = lane / (BN / 4);
    unsigned x_pix1 = 0, x_pix2 = 0, vrow = 0;
    bool fix_l = false, fix_r = false;
    if constexpr (X4) {
        const int gp = n0 + 4 * g4;
        const bool gv = gp < p.NPIX;
        const int pp = gv ? gp : 0;
        const int img = pp / HoWo;
        const int r = pp - img * HoWo;
        const int ho = r / g.Wo;
        const int wo = r - ho * g.Wo;
        const unsigned lin = (unsigned)(ho * g.Ws + wo + (RPW * wave + rsub) * HsWs);
        x_pix1 = (unsigned)((long long)img * g.x1_img_stride) + lin;
        x_pix2 = (unsigned)((long long)img * g.x2_img_stride) + lin;
        if (gv) {
            const int kh = ntaps / g.kw;
            for (int ky = 0; ky < kh; ++ky)
                if ((unsigned)(ho + ky - g.pad_t) < (unsigned)g.Hs) vrow |= 1u << ky;
            fix_l = g.pad_l == 1 && wo == 0;                                  // tap column 0 reads column -1
            fix_r = wo + 3 + (g.kw - 1 - g.pad_l) >= g.Ws;                    // the last tap column reads column Ws
        }
    }
    unsigned b_voff[X4 ? (BK * BN / 1024) : 8];
    // (re)build the per-lane offsets for a chunk of `cw` valid channels from source `first`: rows beyond cw are
    // permanently out of range (zeros), so the K loop itself carries no channel test.  Runs only when the source or the
    // width changes (at most 4 times per workgroup).
    auto set_chunk = [&](bool first, int cw) {
        if constexpr (X4) {
            const unsigned b = first ? x_pix1 : x_pix2;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                b_voff[j] = (4 * RPW * j + RPW * wave + rsub < cw) ? (b + (unsigned)(4 * RPW * j * HsWs)) * 4u + IMM_MAX : DP_OOB;
        } else {
            const unsigned b = first ? b_pix1 : b_pix2;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                b_voff[j] = (bk0 + 2 * j < cw) ? (b + (unsigned)(2 * j * HsWs)) * 4u + (IMM_MAX - 1024u * (j & 3)) : DP_OOB;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) a_voff[j] = (a_k[j] < cw) ? a_base[j] : DP_OOB;
    };

    // ---- scalar K-tile state: chunk ch (16 channels), tap, and the two scalar byte offsets, advanced incrementally
    //      (every scalar instruction here sits in front of this wave's DMA issue: ~45 of them cost 4 % of the kernel)
    // (uniform integer divisions are expanded on the VALU: pin the results back to SGPRs, or every buffer instruction
    //  that takes them as its scalar offset gets wrapped in a waterfall loop)
    int ch = __builtin_amdgcn_readfirstlane(it0 / ntaps);
    int tap = it0 - ch * ntaps;
    int kx, kyc;
    bool first = ch < nch1;
    int cw = (ch < nch) ? chunk_width(ch) : BK;
    set_chunk(first, cw);
    const unsigned a_tap_step = (unsigned)C * a_row_bytes;
    const unsigned b_row_step = (unsigned)(g.Ws - g.kw) * 4u;
    unsigned a_soff, b_soff;
    auto chunk_offsets = [&]() {                                 // offsets of (ch, tap): full recompute (rare)
        const int ky = __builtin_amdgcn_readfirstlane(tap / g.kw);
        kyc = ky;
        kx = tap - ky * g.kw;
        const int cbase = first ? ch * BK : C1 + (ch - nch1) * BK;       // first channel of the chunk in the concat order
        a_soff = (unsigned)(tap * C + cbase) * a_row_bytes;
        b_soff = (unsigned)(((first ? ch : ch - nch1) * BK) * HsWs + ky * g.Ws + kx) * 4u;
    };
    chunk_offsets();

    float* const ldsA = smem + 4 * (wave * 64);                 // + buf*STAGE + 1024*j   (float4 per lane)
    float* const ldsB = smem + A_SZ + bk0 * 0 + (wave & 1) * 64 + (wave >> 1) * BN;   // row bk0 = wave>>1, cols (wave&1)*64..

    auto dma_A = [&](int buf) {
        float* As = ldsA + buf * STAGE;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (256 * j + 64 * wave >= A_F4) continue;          // BM = 96: the second pass belongs to waves 0 and 1 only
            unsigned o = a_voff[j];
            asm volatile("" : "+v"(o));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (dp_lds_void*)(As + 1024 * j), 16, (int)o, (int)a_soff, 0, 0);
        }
    };
    auto dma_B = [&](int buf) {
        float* Bs = ldsB + buf * STAGE;
        const __amdgpu_buffer_rsrc_t rs = first ? r1 : r2;
        if constexpr (X4) {
            const bool tv = (vrow >> kyc) & 1u;
            float* B4 = smem + A_SZ + buf * STAGE + RPW * wave * BN;         // this wave's row group of pass 0
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                unsigned o = tv ? b_voff[j] : DP_OOB;
                asm volatile("" : "+v"(o));
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (dp_lds_void*)(B4 + 4 * RPW * j * BN), 16, (int)o, (int)b_soff, 0, 0);
            }
        } else {
            const bool tv = (vmask >> tap) & 1u;
            dp_static_for<0, 8>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                unsigned o = tv ? b_voff[j] : DP_OOB;
                asm volatile("" : "+v"(o));
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (dp_lds_void*)(Bs + 8 * (j >> 2) * BN), 4, (int)o, (int)b_soff,
                                                         1024 * (j & 3), 0);
            });
        }
    };
    auto dma_tile = [&](int buf) { dma_A(buf); dma_B(buf); };
    // X4: zero the border element of the tile that has just landed in `buf` (tap column state = the tile's)
    const int kx_last = g.kw - 1, r_over = g.kw - 1 - g.pad_l;
    auto fix_border = [&](int buf) {
        if constexpr (X4) {
            float* B4 = smem + A_SZ + buf * STAGE + RPW * wave * BN + 4 * lane;
            if (kx == 0 && g.pad_l == 1) {
                if (fix_l) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) B4[4 * RPW * j * BN] = 0.f;
                }
            }
            if (kx == kx_last && r_over == 1) {
                if (fix_r) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) B4[4 * RPW * j * BN + 3] = 0.f;
                }
            }
        }
    };
    auto advance = [&]() {                                       // next K tile (chunk outer, tap inner)
        ++tap; ++kx;
        a_soff += a_tap_step;
        b_soff += 4u;
        if (kx == g.kw) { kx = 0; ++kyc; b_soff += b_row_step; }
        if (tap == ntaps) {
            tap = 0; ++ch;
            if constexpr (!TAILS) {
                if (first && ch == nch1 && ch < nch) { first = false; set_chunk(false, BK); }
            } else if (ch < nch) {
                const bool f = ch < nch1;
                const int w = chunk_width(ch);
                if (f != first || w != cw) { first = f; cw = w; set_chunk(f, w); }
            }
            chunk_offsets();
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    const int li = lane & 31, lk = lane >> 5;
    const float* fragA = smem + lk * BM + wrow + li;
    const float* fragB = smem + A_SZ + lk * BN + wcol + li;

    if (nIter > 0) {
        dma_tile(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        fix_border(0);
        __syncthreads();
        for (int it = 0; it < nIter; ++it) {
            const int buf = it & 1;
            // prefetch of tile it+1 (on the last iteration the current tile again: harmless, no predicate needed)
#ifndef DP_EXP_NOADV
            if (it + 1 < nIter) advance();
#endif
            // The prefetch DMA of tile it+1 is issued INSIDE the MFMA sequence (A after k-step DP_DMA_A_KS, B after k-step
            // DP_DMA_B_KS) instead of in front of it: the co-resident waves of a SIMD run in near lockstep (they share the
            // matrix pipe round-robin), so a block of DMA issue slots in front of the first MFMA left the pipe idle on all of
            // them at once.  [measured, round 2: 256->256 @16x16 119.8 -> 121.1, 128->128 @32x32 130.8 -> 134.6 TFLOP/s]
#ifndef DP_DMA_A_KS
#define DP_DMA_A_KS 1
#define DP_DMA_B_KS 4
#endif
#if !defined(DP_EXP_NODMA)
            if (DP_DMA_A_KS < 0) dma_A(buf ^ 1);
            if (DP_DMA_B_KS < 0) dma_B(buf ^ 1);
#endif
            const float* Af = fragA + buf * STAGE;
            const float* Bf = fragB + buf * STAGE;
            float a[2][TM], b[2][TN];
#pragma unroll
            for (int t = 0; t < TM; ++t) a[0][t] = Af[TMS * t];
#pragma unroll
            for (int t = 0; t < TN; ++t) b[0][t] = Bf[TNS * t];
#pragma unroll
            for (int ks = 0; ks < BK / 2; ++ks) {
                const int cur = ks & 1;
                if (ks + 1 < BK / 2) {
#pragma unroll
                    for (int t = 0; t < TM; ++t) a[cur ^ 1][t] = Af[(ks + 1) * 2 * BM + TMS * t];
#pragma unroll
                    for (int t = 0; t < TN; ++t) b[cur ^ 1][t] = Bf[(ks + 1) * 2 * BN + TNS * t];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][tm], b[cur][tn], acc[tm][tn], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#if !defined(DP_EXP_NODMA)
                if (ks == DP_DMA_A_KS) { dma_A(buf ^ 1); __builtin_amdgcn_sched_barrier(0); }
                if (ks == DP_DMA_B_KS) { dma_B(buf ^ 1); __builtin_amdgcn_sched_barrier(0); }
#endif
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            fix_border(buf ^ 1);
            __syncthreads();
        }
    }
    if (ksplit && p.tile_counters)
        conv_splitk_fold<TM, TN, TMS, TNS>(p, acc, m0 + wrow, n0 + wcol, blockIdx.y * gridDim.x + blockIdx.x, z, smem);
    else
        conv_epilogue<TM, TN, TMS, TNS>(p, acc, m0 + wrow, n0 + wcol, lane, z, ksplit);
}

static bool conv_fast_ok(const dp_conv_gemm_params& p) {
    static const bool no_fast = getenv("DP_NO_FAST") != nullptr;
    const dp_conv_geom& g = p.g;
    return !no_fast && !p.a_kc && g.stride == 1 && g.sden == 1 && g.ups == 0 && p.ntaps <= 32 && g.Hs == g.Hv &&
           g.Ws == g.Wv;
}

static bool conv_fast_tails(const dp_conv_gemm_params& p) {
    return (p.C % 16) != 0 || (p.X2 && (p.g.c_split % 16) != 0);
}

// 16-byte B-tile loads: 4 consecutive output pixels = 4 consecutive input columns of one row, <= 1 padding column per side
static bool conv_fast_x4(const dp_conv_gemm_params& p) {
    const dp_conv_geom& g = p.g;
    if ((!p.x_guard && g.pad_l > 0) || getenv("DP_NO_X4") || g.kw < 1 || p.ntaps % g.kw) return false;
    const int kh = p.ntaps / g.kw;
    if (kh > 8) return false;
    if (p.ntaps == 1 && g.pad_l == 0 && g.pad_t == 0 && g.Ws == g.Wo && g.Hs == g.Ho) return (g.Ho * g.Wo) % 4 == 0;
    return g.Wo % 4 == 0 && g.Ws >= 4 && g.pad_l >= 0 && g.pad_l <= 1 && g.kw - 1 - g.pad_l >= 0 && g.kw - 1 - g.pad_l <= 1;
}

// 128x64 tiles: same kernel, half-width B tile (see the BN = 64 note in the kernel)
static bool launch_conv_fast_n64(const dp_conv_gemm_params& p, hipStream_t st) {
    if (!conv_fast_ok(p) || conv_fast_tails(p) || !conv_fast_x4(p)) return false;
    dim3 grid((p.NPIX + 63) / 64, (p.M + 127) / 128, p.ksplit > 1 ? p.ksplit : (p.batches > 0 ? p.batches : 1));
    DP_LAUNCH((conv_gemm_fast_kernel<128, 64, false, true>), grid, dim3(256), dp_lds_pad(), st, p);
    return true;
}

template <int BM, bool TAILS>
static void launch_conv_fast(const dp_conv_gemm_params& p, dim3 grid, hipStream_t st) {
    if (conv_fast_x4(p)) DP_LAUNCH((conv_gemm_fast_kernel<BM, 128, TAILS, true>), grid, dim3(256), dp_lds_pad(), st, p);
    else                 DP_LAUNCH((conv_gemm_fast_kernel<BM, 128, TAILS, false>), grid, dim3(256), dp_lds_pad(), st, p);
}

template <int BM, int BN>
static int launch_conv_gemm(const dp_conv_gemm_params& p, hipStream_t st) {
    dim3 grid((p.NPIX + BN - 1) / BN, (p.M + BM - 1) / BM, p.ksplit > 1 ? p.ksplit : (p.batches > 0 ? p.batches : 1));
    if constexpr (BM == 128 && BN == 128) {
        if (conv_fast_ok(p)) {
            if (conv_fast_tails(p)) launch_conv_fast<128, true>(p, grid, st);
            else                    launch_conv_fast<128, false>(p, grid, st);
            return DP_LAUNCH_CHECK();
        }
    }
    // a K-chunk of 16 channels can straddle the concat boundary only when c_split is not a multiple of 16
    const bool straddle = p.X2 != nullptr && (p.g.c_split % 16) != 0;
    if (p.a_kc) {
        if (straddle) DP_LAUNCH((conv_gemm_kernel<BM, BN, true, true>), grid, dim3(256), dp_lds_pad(), st, p);
        else          DP_LAUNCH((conv_gemm_kernel<BM, BN, true, false>), grid, dim3(256), dp_lds_pad(), st, p);
    } else {
        if (straddle) DP_LAUNCH((conv_gemm_kernel<BM, BN, false, true>), grid, dim3(256), dp_lds_pad(), st, p);
        else          DP_LAUNCH((conv_gemm_kernel<BM, BN, false, false>), grid, dim3(256), dp_lds_pad(), st, p);
    }
    return DP_LAUNCH_CHECK();
}

// sum_z ws[z*stride] in ascending z (the fixed order that makes split-K deterministic).  The loads of 8 partials are issued
// together and only the ADDS are ordered: one dependent memory latency per 8 splits instead of one per split (these kernels
// were latency-, not bandwidth-bound: 13-60 us for a few MB).  Adding 0.0f for the missing tail entries changes nothing.
__device__ __forceinline__ float dp_splitk_sum(const float* __restrict__ ws, long long stride, int splits) {
    float a = 0.f;
    for (int z = 0; z < splits; z += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (z + j < splits) ? ws[(long long)(z + j) * stride] : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) a += v[j];
    }
    return a;
}

// out = epilogue(sum_z ws[z][m][pix])  -- fixed summation order; same epilogue arithmetic as the fused path
__global__ __launch_bounds__(256) void conv_splitk_epilogue_kernel(const dp_conv_gemm_params p) {
    const long long total = (long long)p.M * p.NPIX;
    const int HoWo = p.g.Ho * p.g.Wo;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int m = (int)(i / p.NPIX);
        const int pix = (int)(i - (long long)m * p.NPIX);
        const float a = dp_splitk_sum(p.ws + i, total, p.ksplit);
        const int img = pix / HoWo;
        const int r_in = pix - img * HoWo;
        float v = p.alpha * a;
        if (p.bias) v += p.bias[m];
        if (p.tadd) v += p.tadd[(long long)img * p.tadd_stride + m];
        if (p.res) v += p.res[(long long)img * p.r_img_stride + (long long)m * HoWo + r_in];
        v *= p.post_scale;
        if (p.act == 1) v = fmaxf(v, 0.f);
        float* o = p.out + (long long)img * p.o_img_stride + (long long)m * HoWo + r_in;
        if (p.accumulate) v += *o;
        *o = v;
    }
}

// ------------------------------------------------------------------------------------------------
// conv_few_out: 3x3 stride-1 'same' convolution with <= 4 output channels (conv_out of the UNets: C -> 3).  On the matrix
// path such a layer fills 3 of the 64 rows of the smallest tile (0.39 ms at 4.6 TFLOP/s for 0.9 GFLOP at B = 256); here it
// is what it is -- an HBM-bound stencil: one workgroup per 16x16 pixel tile of one image, 16 input channels at a time
// through LDS (18x18 halo tile), one thread per output pixel with <= 4 fp32 accumulators, the 4 weights of a (tap, channel)
// pair fetched as ONE uniform 16-byte load from the packed operand ([(tap*C + c)][4]).  fmaf chain over (channel, tap).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_few_out_kernel(const dp_conv_gemm_params p) {
    constexpr int TS = 16, CH = 16, PW = TS + 2;
    __shared__ float sx[CH][PW][PW + 1];
    const ConvGeom& g = p.g;
    const int H = g.Hs, W = g.Ws, C = p.C;
    const int tiles_x = (W + TS - 1) / TS;
    const int img = blockIdx.y;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int h0 = ty * TS, w0 = tx * TS;
    const int lh = threadIdx.x / TS, lw = threadIdx.x - lh * TS;
    const float* __restrict__ xb = p.X1 + (long long)img * g.x1_img_stride;
    const float4* __restrict__ A4 = reinterpret_cast<const float4*>(p.A);          // lda == 4
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    for (int c0 = 0; c0 < C; c0 += CH) {
        const int cw = min(CH, C - c0);
        __syncthreads();
        for (int e = threadIdx.x; e < cw * PW * PW; e += 256) {
            const int c = e / (PW * PW), r = e - c * (PW * PW);
            const int y = r / PW, xq = r - y * PW;
            const int h = h0 + y - 1, w = w0 + xq - 1;
            sx[c][y][xq] = ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) ? xb[((long long)(c0 + c) * H + h) * W + w] : 0.f;
        }
        __syncthreads();
        for (int c = 0; c < cw; ++c) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float4 w4 = A4[(long long)t * C + c0 + c];                  // uniform address: scalar load
                const float v = sx[c][lh + t / 3][lw + t % 3];
                acc0 = fmaf(w4.x, v, acc0); acc1 = fmaf(w4.y, v, acc1); acc2 = fmaf(w4.z, v, acc2); acc3 = fmaf(w4.w, v, acc3);
            }
        }
    }
    const int h = h0 + lh, w = w0 + lw;
    if (h >= H || w >= W) return;
    const float accs[4] = {acc0, acc1, acc2, acc3};
    float* ob = p.out + (long long)img * p.o_img_stride + (long long)h * W + w;
    for (int m = 0; m < p.M; ++m) {
        float v = p.alpha * accs[m];
        if (p.bias) v += p.bias[m];
        v *= p.post_scale;
        if (p.act == 1) v = fmaxf(v, 0.f);
        ob[(long long)m * H * W] = v;
    }
}

static bool conv_few_out_ok(const dp_conv_gemm_params& p) {
    const dp_conv_geom& g = p.g;
    return !getenv("DP_NO_FEW_OUT") && p.M <= 4 && p.lda == 4 && !p.a_kc && p.ntaps == 9 && g.kw == 3 && g.stride == 1 &&
           g.sden == 1 && g.ups == 0 && g.pad_t == 1 && g.pad_l == 1 && g.Ho == g.Hs && g.Wo == g.Ws && g.Hs == g.Hv &&
           g.Ws == g.Wv && !p.X2 && !p.tadd && !p.res && !p.accumulate && p.ksplit <= 1 && p.batches <= 1 &&
           p.NPIX % (g.Ho * g.Wo) == 0;
}

// the reduction launch of a split-K convolution (also used by dp_conv_wino): out = epilogue(sum_z ws[z][m][pix])
// The same reduction for 4 consecutive pixels per thread (round 5): ksplit 16-byte loads per thread, eight of them in flight,
// all branch-free (split indices past the end re-read the last slice and are not added) -- the scalar form above keeps eight
// 4-byte loads in flight behind a branch each (8.6 us per launch x 3 981 launches of an LDM importance step).  Needs Ho*Wo % 4 == 0
// (so the four pixels share an image and a row of the [M][NPIX] partials) and 16-byte aligned rows everywhere.  Same additions
// in the same order per element -> the same bits.
__global__ __launch_bounds__(256) void conv_splitk_epilogue4_kernel(const dp_conv_gemm_params p) {
    const long long total4 = (long long)p.M * p.NPIX / 4;
    const long long total = (long long)p.M * p.NPIX;
    const int HoWo = p.g.Ho * p.g.Wo;
    const int NPIX4 = p.NPIX / 4;
    for (long long i4 = (long long)blockIdx.x * 256 + threadIdx.x; i4 < total4; i4 += (long long)gridDim.x * 256) {
        const int m = (int)(i4 / NPIX4);
        const int pix = (int)(i4 - (long long)m * NPIX4) * 4;
        const float4* w4 = reinterpret_cast<const float4*>(p.ws) + i4;
        const int img = pix / HoWo;
        const int r_in = pix - img * HoWo;
        // the epilogue operands go out with the first partials (a load per `if` after the reduction is a round trip each)
        float4* o = reinterpret_cast<float4*>(p.out + (long long)img * p.o_img_stride + (long long)m * HoWo + r_in);
        float b = 0.f, t = 0.f;
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f), prev = r;
        if (p.bias) b = p.bias[m];
        if (p.tadd) t = p.tadd[(long long)img * p.tadd_stride + m];
        if (p.res) r = *reinterpret_cast<const float4*>(p.res + (long long)img * p.r_img_stride + (long long)m * HoWo + r_in);
        if (p.accumulate) prev = *o;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int z = 0; z < p.ksplit; z += 8) {
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int zc = z + j < p.ksplit ? z + j : p.ksplit - 1;
                v[j] = w4[(long long)zc * (total / 4)];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (z + j < p.ksplit) { a.x += v[j].x; a.y += v[j].y; a.z += v[j].z; a.w += v[j].w; }
        }
        float4 v = make_float4(p.alpha * a.x, p.alpha * a.y, p.alpha * a.z, p.alpha * a.w);
        if (p.bias) { v.x += b; v.y += b; v.z += b; v.w += b; }
        if (p.tadd) { v.x += t; v.y += t; v.z += t; v.w += t; }
        if (p.res) { v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
        v.x *= p.post_scale; v.y *= p.post_scale; v.z *= p.post_scale; v.w *= p.post_scale;
        if (p.act == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (p.accumulate) { v.x += prev.x; v.y += prev.y; v.z += prev.z; v.w += prev.w; }
        *o = v;
    }
}

extern "C" int dp_conv_splitk_epilogue(const dp_conv_gemm_params* pp, void* stream) {
    const dp_conv_gemm_params& p = *pp;
    const int HoWo = p.g.Ho * p.g.Wo;
    const bool v4 = HoWo % 4 == 0 && p.NPIX % 4 == 0 && ((uintptr_t)p.ws | (uintptr_t)p.out | (uintptr_t)p.res) % 16 == 0 &&
                    p.o_img_stride % 4 == 0 && (!p.res || p.r_img_stride % 4 == 0) && !getenv("DP_NO_EPI4");
    if (v4) {
        long long nb = ((long long)p.M * p.NPIX / 4 + 255) / 256;
        if (nb > 8192) nb = 8192;
        DP_LAUNCH(conv_splitk_epilogue4_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, p);
        return DP_LAUNCH_CHECK();
    }
    long long nb = ((long long)p.M * p.NPIX + 255) / 256;
    if (nb > 8192) nb = 8192;
    DP_LAUNCH(conv_splitk_epilogue_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, p);
    return DP_LAUNCH_CHECK();
}

extern "C" int dp_conv_gemm(const dp_conv_gemm_params* pp, void* stream) {
    const dp_conv_gemm_params& p = *pp;
    hipStream_t st = (hipStream_t)stream;
    if (p.M <= 0 || p.NPIX <= 0) return 0;
    if (p.ksplit > 1 && (p.batches > 1 || !p.ws)) return (int)hipErrorInvalidValue;
    if (!p.a_kc && (p.lda & 3)) return (int)hipErrorInvalidValue;
    if (p.a_kc && p.ntaps != 1) return (int)hipErrorInvalidValue;
    if (conv_few_out_ok(p)) {
        const dp_conv_geom& g = p.g;
        dim3 grid(((g.Wo + 15) / 16) * ((g.Ho + 15) / 16), p.NPIX / (g.Ho * g.Wo));
        DP_LAUNCH(conv_few_out_kernel, grid, dim3(256), 0, st, p);
        return DP_LAUNCH_CHECK();
    }
    int e;
    switch (p.tile) {
        case 3:                                                  // 96x128: fast kernel only, else the 128x128 path
            if (conv_fast_ok(p)) {
                dim3 grid((p.NPIX + 127) / 128, (p.M + 95) / 96, p.ksplit > 1 ? p.ksplit : (p.batches > 0 ? p.batches : 1));
                launch_conv_fast<96, true>(p, grid, st);
                e = DP_LAUNCH_CHECK();
                break;
            }
            [[fallthrough]];
        case 4:                                                  // 128x64: fast x4 kernel only, else the 128x128 path
            if (launch_conv_fast_n64(p, st)) { e = DP_LAUNCH_CHECK(); break; }
            [[fallthrough]];
        case 0: e = launch_conv_gemm<128, 128>(p, st); break;
        case 1: e = launch_conv_gemm<64, 128>(p, st); break;
        case 2: e = launch_conv_gemm<64, 64>(p, st); break;
        default: return (int)hipErrorInvalidValue;
    }
    if (e || p.ksplit <= 1 || p.tile_counters) return e;
    return dp_conv_splitk_epilogue(pp, stream);
}

// ------------------------------------------------------------------------------------------------
// nt_gemm: D[m][c] (per tap) = alpha * sum_pix A[m][pix] * X(pix, c, tap)
//   blockIdx.z = batch (batched mode) or split*ntaps + tap: one kernel tap per workgroup, so the gather geometry is
//   wave-uniform and the per-element address is one add (same loader cost as conv_gemm).
//   Output element (m, c, tap) at out[zo*o_bs + m*ldo + c*ntaps + tap]  (torch [Cout][Cin][kh][kw] layout).
// ------------------------------------------------------------------------------------------------
#ifndef NT_BK
#define NT_BK 16
#endif
// MERGE (few input channels, e.g. conv_in / conv_out weight gradients): the kernel taps are folded into the column
//   index, col = c*ntaps + tap, so one tile holds all C*ntaps columns instead of ntaps workgroups with C of BN
//   columns each; merge bit 1 mirrors the taps (the gathered tensor is dy and the plain rows are x: the roles of the
//   two operands swapped, for few OUTPUT channels).  Output element (row, c, tap) at row*ldo + c*ocs + tap.
template <int BM, int BN, bool STRADDLE, bool MERGE = false>
__global__ __launch_bounds__(256, 4) void nt_gemm_kernel(const dp_nt_gemm_params p) {
    constexpr int BK = NT_BK;
    constexpr int LD = BK + 1;
    constexpr int RSTEP = 256 / BK;      // rows covered per pass of the 256 threads
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int A_SZ = BM * LD;
    constexpr int B_SZ = BN * LD;
    constexpr int STAGE = A_SZ + B_SZ;
    __shared__ float smem[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm0 = (wave >> 1) * WM;
    const int wn0 = (wave & 1) * WN;
    const int m0 = blockIdx.y * BM;
    const int n0 = blockIdx.x * BN;
    const int z = blockIdx.z;

    const ConvGeom& g = p.g;
    const int HoWo = g.Ho * g.Wo;
    const int HsWs = g.Hs * g.Ws;
    const int batch = p.batched ? z : 0;
    const int split = p.batched ? 0 : (MERGE ? z : z / p.ntaps);
    const int tap = (p.batched || MERGE) ? 0 : z - split * p.ntaps;
    const int ky = tap / g.kw;
    const int kx = tap - ky * g.kw;
    const int p_begin = split * p.p_per_split;
    int p_end = p.batched ? p.P : (p_begin + p.p_per_split);
    if (p_end > p.P) p_end = p.P;
    const int nIter = (p_end > p_begin) ? (p_end - p_begin + BK - 1) / BK : 0;

    const float* __restrict__ Ab = p.A + (long long)batch * p.a_bs;
    const float* __restrict__ X1 = p.X1 + (long long)batch * p.x_bs;
    const float* __restrict__ X2 = p.X2 ? p.X2 + (long long)batch * p.x_bs : X1;
    const int csplit = g.c_split;

    constexpr int NA = BM / RSTEP;
    constexpr int NB = BN / RSTEP;
    const int lk = tid & (BK - 1);  // this thread's pixel within the K-tile
    const int r0 = tid / BK;        // first row; rows r0 + RSTEP*j
    float ra[NA], rb[NB];
    // the N-tile [n0, n0+BN) is either entirely inside one source or straddles the concat boundary (block-uniform)
    const bool first_src = n0 < csplit;               // tile start decides in the non-straddling variant

    const __amdgpu_buffer_rsrc_t rA = dp_rsrc(Ab, p.a_bytes);
    const __amdgpu_buffer_rsrc_t r1 = dp_rsrc(X1, p.x1_bytes);
    const __amdgpu_buffer_rsrc_t r2 = dp_rsrc(X2, p.X2 ? p.x2_bytes : p.x1_bytes);
    const __amdgpu_buffer_rsrc_t rs_one = dp_rsrc_uniform(first_src ? X1 : X2, (first_src || !p.X2) ? p.x1_bytes : p.x2_bytes);

    auto load_tile = [&](int it, bool live) {
        const int pp = p_begin + it * BK + lk;
        const bool pv = live && (pp < p_end);
        const int pq = pv ? pp : p_begin;
        const int img = pq / HoWo;
        const int r = pq - img * HoWo;
        const int ho = r / g.Wo;
        const int wo = r - ho * g.Wo;
        const unsigned ab = (unsigned)(((long long)img * p.a_img_stride + r) * 4);
        const unsigned astep = (unsigned)(HoWo * 4);
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int m = m0 + r0 + RSTEP * j;
            const bool v = pv && (m < p.M);
            ra[j] = dp_bload(rA, v ? (ab + (unsigned)m * astep) : DP_OOB);
        }
        const unsigned pb1 = (unsigned)((long long)img * g.x1_img_stride * 4);
        const unsigned pb2 = (unsigned)((long long)img * g.x2_img_stride * 4);
        const int cb = n0 + r0;
        if constexpr (MERGE) {
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int col = cb + RSTEP * j;
                const int c = col / p.ntaps;
                int t = col - c * p.ntaps;
                if (p.merge & 2) t = p.ntaps - 1 - t;
                const int tky = t / g.kw;
                int o;
                const bool v = dp_gather(g, ho, wo, tky, t - tky * g.kw, o) && pv && (col < p.NCOLS);
                rb[j] = dp_bload(r1, v ? (pb1 + (unsigned)((c * HsWs + o) * 4)) : DP_OOB);
            }
            return;
        }
        int off;
        const bool tv = dp_gather(g, ho, wo, ky, kx, off) && pv;
        if constexpr (!STRADDLE) {
            const __amdgpu_buffer_rsrc_t rs = rs_one;
            const unsigned o0 = (first_src ? pb1 : pb2) + (unsigned)(((first_src ? cb : cb - csplit) * HsWs + off) * 4);
            const unsigned step = (unsigned)(RSTEP * HsWs * 4);
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const bool v = tv && (cb + RSTEP * j < p.NCOLS);
                rb[j] = dp_bload(rs, v ? (o0 + j * step) : DP_OOB);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int c = cb + RSTEP * j;
                const bool v = tv && (c < p.NCOLS);
                const bool f1 = c < csplit;
                const unsigned o1 = pb1 + (unsigned)((c * HsWs + off) * 4);
                const unsigned o2 = pb2 + (unsigned)(((c - csplit) * HsWs + off) * 4);
                rb[j] = dp_bload(r1, (v && f1) ? o1 : DP_OOB) + dp_bload(r2, (v && !f1) ? o2 : DP_OOB);
            }
        }
    };
    auto store_tile = [&](int buf) {
        float* As = smem + buf * STAGE;
        float* Bs = As + A_SZ;
#pragma unroll
        for (int j = 0; j < NA; ++j) As[(r0 + RSTEP * j) * LD + lk] = ra[j];
#pragma unroll
        for (int j = 0; j < NB; ++j) Bs[(r0 + RSTEP * j) * LD + lk] = rb[j];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    load_tile(0, nIter > 0);
    store_tile(0);
    __syncthreads();
    for (int it = 0; it < nIter; ++it) {
        const int buf = it & 1;
        load_tile(it + 1, it + 1 < nIter);
        const float* As = smem + buf * STAGE;
        mfma_tile<BM, BN, BK, TM, TN, true, true>(As, As + A_SZ, wm0, wn0, lane, acc);
        store_tile(buf ^ 1);
        __syncthreads();
    }

    const int zo = p.batched ? z : split;
    const int o_cs = p.o_col_stride ? p.o_col_stride : p.ntaps;          // defaults: torch [Cout][Cin][kh][kw] layout
    float* __restrict__ outb = p.out + (long long)zo * p.o_bs + (long long)tap * (p.o_tap_stride ? p.o_tap_stride : 1);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int col = n0 + wn0 + tn * 32 + (lane & 31);
        if (col >= p.NCOLS) continue;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m >= p.M) continue;
                float* o;
                if constexpr (MERGE) {
                    const int c = col / p.ntaps;
                    o = outb + (long long)m * p.ldo + (long long)c * p.ocs + (col - c * p.ntaps);
                } else {
                    o = outb + (long long)m * p.ldo + (long long)col * o_cs;
                }
                float v = p.alpha * acc[tm][tn][r];
                if (p.col_bias && split == 0) v += p.col_bias[col];
                if (p.accumulate) v += *o;
                *o = v;
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// nt_gemm_fast: weight gradients of the stride-1 convolutions whose pixel rows are powers of two (every CIFAR / LSUN /
// LDM resnet and attention layer), 128x128 tile, same math, split-K ranges and summation order as nt_gemm_kernel.
// A K tile = 16 consecutive pixels of one image (HoWo % 16 == 0), so image / row / column of the tile are scalars and a
// lane's pixel is (ho0 + lk / Wo, wo0 + lk % Wo) with lane-constant second terms:
//   * 8 + 8 constant per-lane byte offsets (4 rows x 16 pixels per DMA instruction), tile position in the scalar offset;
//   * zero padding: 2 adds + 2 unsigned compares per K tile, then 8 cndmask on the gathered operand only;
//   * global -> LDS DMA into a [row/4][4 rows x 16 pixels + 1 pad dword] image: each instruction writes 64 consecutive
//     dwords, the pad dword per 4-row group makes the 32-lane fragment reads hit 32 distinct banks, and the k-step
//     offsets are ds_read immediates (no address arithmetic, no register staging, no ds_write).
// ------------------------------------------------------------------------------------------------
// NW = 4: 128x128 tile, waves 2x2 of 64x64.  NW = 3: 96x96 tile, 192 threads, wave w = rows [32w, 32w+32) x 96 columns
// (pruned widths: 90 or 180 channels fill 49 % of 128x128 tiles, 88 % of 96x96 ones).
// Two concat sources with ANY split: the tile columns are *virtual* channels v -- source 1 padded to a multiple of 4
// (C1p), then source 2 -- so every 4-row DMA group reads one source (scalar descriptor choice), and the epilogue maps
// v back to the real channel.
template <int NW, bool TWO>
__global__ __launch_bounds__(NW * 64, 4) void nt_gemm_fast_kernel(const dp_nt_gemm_params p) {
    constexpr int BM = 32 * NW, BN = 32 * NW, BK = 16;
    constexpr int TM = (NW == 4) ? 2 : 1, TN = (NW == 4) ? 2 : 3;
    constexpr int GRP = 4 * BK + 1;                 // dwords per 4-row group
    constexpr int OP_SZ = (BM / 4) * GRP;           // 2080 dwords per operand tile
    constexpr int STAGE = 2 * OP_SZ;
    __shared__ float smem[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (NW == 4) ? (wave >> 1) * 64 : wave * 32;
    const int wn0 = (NW == 4) ? (wave & 1) * 64 : 0;
    // Workgroup -> (tile, tap, split).  Workgroup b is dispatched to XCD b % 8 and every XCD has its own L2: with the plain
    // (x, y, z) order the gx*gy*ntaps workgroups that read ONE pixel range (one split) are spread over all 8 XCDs and each L2
    // fetches its own copy of dy / x (FETCH_SIZE 2.5x the operands, profiles/round1_pmc_bench_traffic.json).  p.xcd: XCD
    // x owns a contiguous run of the split-major order, so the workgroups of a split share one L2.  Bijective for any grid.
    int bx = blockIdx.x, by = blockIdx.y, split, tap;
    if (p.xcd) {
        const int gxy = gridDim.x * gridDim.y;
        const int per_split = gxy * p.ntaps;
        const int nwg = per_split * p.splits;
        const int b = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const int q = nwg >> 3, r = nwg & 7, xcd = b & 7, k = b >> 3;
        const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
        split = __builtin_amdgcn_readfirstlane(v / per_split);
        const int rem = v - split * per_split;
        tap = __builtin_amdgcn_readfirstlane(rem / gxy);
        const int rem2 = rem - tap * gxy;
        by = __builtin_amdgcn_readfirstlane(rem2 / (int)gridDim.x);
        bx = rem2 - by * (int)gridDim.x;
    } else {
        split = blockIdx.z / p.ntaps;
        tap = blockIdx.z - split * p.ntaps;
    }
    const int m0 = by * BM;
    const int n0 = bx * BN;

    const ConvGeom& g = p.g;
    const int HoWo = g.Ho * g.Wo;
    const int HsWs = g.Hs * g.Ws;
    const int ky = tap / g.kw;
    const int kx = tap - ky * g.kw;
    const int p_begin = split * p.p_per_split;
    const int p_end = min(p_begin + p.p_per_split, p.P);
    const int nIter = (p_end > p_begin) ? (p_end - p_begin) / BK : 0;

    const int C1 = p.X2 ? g.c_split : p.NCOLS;                   // channels of source 1
    const int C1p = p.X2 ? ((C1 + 3) & ~3) : p.NCOLS;            // virtual index of source 2's first channel
    // The 8 DMA groups of a lane are GRP dwords apart in LDS: one M0 per operand, the 12-bit instruction offset picks
    // the group.  The hardware adds that offset to the memory address too: bases are moved back by IMM_MAX bytes and the
    // per-lane offsets carry (IMM_MAX - imm).
    constexpr unsigned IMM_MAX = 7 * GRP * 4;
    const int shift = g.pad_t * g.Ws + g.pad_l + (int)(IMM_MAX / 4);
    const __amdgpu_buffer_rsrc_t rA = dp_rsrc(p.A - IMM_MAX / 4, p.a_bytes + IMM_MAX);
    const __amdgpu_buffer_rsrc_t rB1 = dp_rsrc(p.X1 - shift, p.x1_bytes + 4u * (unsigned)shift);
    const __amdgpu_buffer_rsrc_t rB2 = dp_rsrc((p.X2 ? p.X2 : p.X1) - shift, (p.X2 ? p.x2_bytes : p.x1_bytes) + 4u * (unsigned)shift);

    // ---- per-lane constants: pixel lk of the K tile, row sub of each 4-row group; this wave owns groups wave*8 .. +7
    const int lk = lane & 15, sub = lane >> 4;
    const int dho = lk / g.Wo, wol = lk - dho * g.Wo;
    const int hc = dho + ky - g.pad_t;
    const int wc = wol + kx - g.pad_l;
    unsigned a_voff[8], b_voff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row = 4 * (wave * 8 + j) + sub;
        const int m = m0 + row;
        a_voff[j] = (m < p.M) ? (unsigned)((m * HoWo + lk) * 4) + (IMM_MAX - (unsigned)(j * GRP * 4)) : DP_OOB;
        const int v = n0 + row;                                  // virtual channel -> (source, channel in source)
        const bool s1 = v < C1p;
        const int cs = s1 ? v : v - C1p;
        const bool cv = s1 ? (v < C1) : (C1 + cs < p.NCOLS);
        b_voff[j] = cv ? (unsigned)((cs * HsWs + (dho + ky) * g.Ws + wol + kx) * 4) + (IMM_MAX - (unsigned)(j * GRP * 4)) : DP_OOB;
    }
    float* const ldsW = smem + wave * 8 * GRP;      // this wave's first group, operand A of stage 0

    // The 16 DMA instructions of a K tile are issued in NT_DMA_PER_KS-sized groups BETWEEN the MFMA groups of the current tile
    // (see the note in conv_gemm_fast_kernel): tile_setup() computes the scalar state, dma_group<q>() issues group q.
    unsigned t_a_soff = 0, t_b_soff1 = 0, t_b_soff2 = 0;
    bool t_v = false;
    auto tile_setup = [&](int it) {
        const int pp0 = p_begin + it * BK;
        const int img = pp0 / HoWo;
        const int r0 = pp0 - img * HoWo;
        const int ho0 = r0 / g.Wo;
        const int wo0 = r0 - ho0 * g.Wo;
        t_a_soff = (unsigned)((long long)img * p.a_img_stride + r0) * 4u;
        t_b_soff1 = (unsigned)((long long)img * g.x1_img_stride + ho0 * g.Ws + wo0) * 4u;
        t_b_soff2 = (unsigned)((long long)img * g.x2_img_stride + ho0 * g.Ws + wo0) * 4u;
        t_v = ((unsigned)(ho0 + hc) < (unsigned)g.Hs) && ((unsigned)(wo0 + wc) < (unsigned)g.Ws);
    };
    auto dma_one = [&](auto jc, int buf) {                           // instruction jc in 0..15: 0..7 operand A, 8..15 operand B
        constexpr int jj = decltype(jc)::value;
        float* As = ldsW + buf * STAGE;
        if constexpr (jj < 8) {
            constexpr int j = jj;
            unsigned o = a_voff[j];
            asm volatile("" : "+v"(o));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (dp_lds_void*)As, 4, (int)o, (int)t_a_soff, j * GRP * 4, 0);
        } else {
            constexpr int j = jj - 8;
            float* Bs = As + OP_SZ;
            unsigned o = t_v ? b_voff[j] : DP_OOB;
            asm volatile("" : "+v"(o));
            const bool s1 = !TWO || n0 + 4 * (wave * 8 + j) < C1p;      // scalar: the whole 4-row group is in one source
            __builtin_amdgcn_raw_ptr_buffer_load_lds(s1 ? rB1 : rB2, (dp_lds_void*)Bs, 4, (int)o,
                                                     (int)(s1 ? t_b_soff1 : t_b_soff2), j * GRP * 4, 0);
        }
    };
    auto dma_tile = [&](int it, int buf) {
        tile_setup(it);
        dp_static_for<0, 16>([&](auto jc) { dma_one(jc, buf); });
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    const int li = lane & 31, lkk = lane >> 5;
    auto rowoff = [](int row) { return (row >> 2) * GRP + (row & 3) * BK; };
    const float* fragA[TM];
    const float* fragB[TN];
#pragma unroll
    for (int t = 0; t < TM; ++t) fragA[t] = smem + rowoff(wm0 + 32 * t + li) + lkk;
#pragma unroll
    for (int t = 0; t < TN; ++t) fragB[t] = smem + OP_SZ + rowoff(wn0 + 32 * t + li) + lkk;

    if (nIter > 0) {
        dma_tile(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#ifndef NT_DMA_SPREAD
#define NT_DMA_SPREAD 1
#endif
        for (int it = 0; it < nIter; ++it) {
            const int buf = it & 1;
#if NT_DMA_SPREAD
            tile_setup(it + 1 < nIter ? it + 1 : it);              // last iteration: reloads the current tile (harmless)
#else
            dma_tile(it + 1 < nIter ? it + 1 : it, buf ^ 1);
#endif
            const int bo = buf * STAGE;
            float a[2][TM], b[2][TN];
#pragma unroll
            for (int t = 0; t < TM; ++t) a[0][t] = fragA[t][bo];
#pragma unroll
            for (int t = 0; t < TN; ++t) b[0][t] = fragB[t][bo];
#pragma unroll
            for (int ks = 0; ks < BK / 2; ++ks) {
                const int cur = ks & 1;
                if (ks + 1 < BK / 2) {
                    const int o = bo + 2 * (ks + 1);
#pragma unroll
                    for (int t = 0; t < TM; ++t) a[cur ^ 1][t] = fragA[t][o];
#pragma unroll
                    for (int t = 0; t < TN; ++t) b[cur ^ 1][t] = fragB[t][o];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][tm], b[cur][tn], acc[tm][tn], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#if NT_DMA_SPREAD
                dp_static_for<0, 2>([&](auto qc) {                   // two of the 16 prefetch loads after every MFMA group
                    constexpr int q = decltype(qc)::value;
                    dp_static_for<0, 8>([&](auto kc) {
                        constexpr int kk = decltype(kc)::value;
                        if (ks == kk) dma_one(std::integral_constant<int, 2 * kk + q>{}, buf ^ 1);
                    });
                });
                __builtin_amdgcn_sched_barrier(0);
#endif
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }

    const int o_cs = p.o_col_stride ? p.o_col_stride : p.ntaps;
    float* __restrict__ outb = p.out + (long long)split * p.o_bs + (long long)tap * (p.o_tap_stride ? p.o_tap_stride : 1);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int vcol = n0 + wn0 + tn * 32 + (lane & 31);
        const int col = (vcol < C1p) ? vcol : C1 + (vcol - C1p);
        if ((vcol < C1p && vcol >= C1) || col >= p.NCOLS) continue;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m >= p.M) continue;
                float* o = outb + (long long)m * p.ldo + (long long)col * o_cs;
                float v = p.alpha * acc[tm][tn][r];
                if (p.accumulate) v += *o;
                *o = v;
            }
        }
    }
}

static bool nt_fast_ok(const dp_nt_gemm_params& p) {
    static const bool no_fast = getenv("DP_NO_FAST") != nullptr;
    const dp_conv_geom& g = p.g;
    const int HoWo = g.Ho * g.Wo;
    return !no_fast && !p.batched && !p.merge && !p.col_bias && g.stride == 1 && g.sden == 1 && g.ups == 0 && g.Wo > 0 &&
           ((g.Wo % 16) == 0 || (16 % g.Wo) == 0) && (HoWo % 16) == 0 && (p.P % 16) == 0 && (p.p_per_split % 16) == 0 &&
           g.Hs == g.Hv && g.Ws == g.Wv;
}

template <int BM, int BN>
static int launch_nt_gemm(const dp_nt_gemm_params& p, hipStream_t st) {
    const int gz = p.batched ? p.batches : p.splits * p.ntaps;
    dim3 grid((p.NCOLS + BN - 1) / BN, (p.M + BM - 1) / BM, gz > 0 ? gz : 1);
    const bool straddle = p.X2 != nullptr && (p.g.c_split % BN) != 0;     // an N-tile may span both concat sources
    if constexpr (BM == 128 && BN == 128) {
        if (nt_fast_ok(p)) {
            const int ncv = p.X2 ? ((p.g.c_split + 3) & ~3) + (p.NCOLS - p.g.c_split) : p.NCOLS;   // virtual columns
            if (p.tile == 3) {
                dim3 g96((ncv + 95) / 96, (p.M + 95) / 96, gz > 0 ? gz : 1);
                if (p.X2) DP_LAUNCH((nt_gemm_fast_kernel<3, true>), g96, dim3(192), dp_lds_pad(), st, p);
                else      DP_LAUNCH((nt_gemm_fast_kernel<3, false>), g96, dim3(192), dp_lds_pad(), st, p);
            } else {
                dim3 g128((ncv + 127) / 128, (p.M + 127) / 128, gz > 0 ? gz : 1);
                if (p.X2) DP_LAUNCH((nt_gemm_fast_kernel<4, true>), g128, dim3(256), dp_lds_pad(), st, p);
                else      DP_LAUNCH((nt_gemm_fast_kernel<4, false>), g128, dim3(256), dp_lds_pad(), st, p);
            }
            return DP_LAUNCH_CHECK();
        }
    }
    if (straddle) DP_LAUNCH((nt_gemm_kernel<BM, BN, true>), grid, dim3(256), dp_lds_pad(), st, p);
    else          DP_LAUNCH((nt_gemm_kernel<BM, BN, false>), grid, dim3(256), dp_lds_pad(), st, p);
    return DP_LAUNCH_CHECK();
}

extern "C" int dp_nt_gemm(const dp_nt_gemm_params* pp, void* stream) {
    const dp_nt_gemm_params& p = *pp;
    hipStream_t st = (hipStream_t)stream;
    if (p.M <= 0 || p.NCOLS <= 0) return 0;
    if (!p.batched && (p.p_per_split <= 0 || (p.p_per_split & (NT_BK - 1)))) return (int)hipErrorInvalidValue;
    if (p.merge) {               // NCOLS = C*ntaps merged columns, one source, 64x64 tiles, blockIdx.z = split
        if (p.batched || p.X2 || p.splits <= 0) return (int)hipErrorInvalidValue;
        dim3 grid((p.NCOLS + 63) / 64, (p.M + 63) / 64, p.splits);
        DP_LAUNCH((nt_gemm_kernel<64, 64, false, true>), grid, dim3(256), dp_lds_pad(), st, p);
        return DP_LAUNCH_CHECK();
    }
    switch (p.tile) {
        case 3:                                                  // 96x96: fast kernel only, else as tile 0
        case 0: return launch_nt_gemm<128, 128>(p, st);
        case 1: return launch_nt_gemm<64, 128>(p, st);
        case 2: return launch_nt_gemm<64, 64>(p, st);
        default: return (int)hipErrorInvalidValue;
    }
}

// ------------------------------------------------------------------------------------------------
// split-K epilogue: out[i] (+)= sum_s ws[s*stride + i]  in ascending s (deterministic)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, long long stride, int splits,
                                                            float* __restrict__ out, long long n, int accumulate) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float s = dp_splitk_sum(ws + i, stride, splits);
        out[i] = accumulate ? out[i] + s : s;
    }
}

// tap-major partials ws[s][tap][m*C + c] (written coalesced by the weight-gradient kernels) -> out[(m*C + c)*ntaps + tap]
__global__ __launch_bounds__(256) void splitk_reduce_taps_kernel(const float* __restrict__ ws, long long stride, int splits,
                                                                 float* __restrict__ out, long long mc, int ntaps,
                                                                 int accumulate) {
    const long long n = mc * ntaps;
    for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < n; j += (long long)gridDim.x * 256) {
        const float s = dp_splitk_sum(ws + j, stride, splits);
        const long long tap = j / mc;
        const long long o = (j - tap * mc) * ntaps + tap;
        out[o] = accumulate ? out[o] + s : s;
    }
}

// One thread per (m, c), all NT taps (round 5): the tap planes ws[s][tap][mc] are read coalesced (consecutive threads = consecutive
// mc), 4 splits x NT taps = up to 36 independent loads in flight per thread (the scalar form above: eight behind a branch each and
// 4-byte stores 4 * ntaps bytes apart), and the NT sums of a thread leave as one contiguous run of out[(m*C + c)*NT ..].  Per
// element the same ascending-split additions as dp_splitk_sum -> the same bits.  [21.8 us x 104 launches per finetune step]
template <int NT>
__global__ __launch_bounds__(256) void splitk_reduce_taps_mc_kernel(const float* __restrict__ ws, long long stride, int splits,
                                                                    float* __restrict__ out, long long mc, int accumulate) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= mc) return;
    float acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = 0.f;
    const float* w = ws + i;
    for (int z = 0; z < splits; z += 4) {
        float v[4][NT];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int zc = z + j < splits ? z + j : splits - 1;
#pragma unroll
            for (int t = 0; t < NT; ++t) v[j][t] = w[(long long)zc * stride + (long long)t * mc];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (z + j < splits) {
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] += v[j][t];
            }
    }
    float* o = out + i * NT;
    if (accumulate) {
        float prev[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) prev[t] = o[t];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = prev[t] + acc[t];
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) o[t] = acc[t];
}

extern "C" int dp_splitk_reduce_taps(const float* ws, long long stride, int splits, float* out, long long mc, int ntaps,
                                     int accumulate, void* stream) {
    if (mc <= 0 || ntaps <= 0) return 0;
    const bool no_mc = getenv("DP_NO_REDUCE_MC") != nullptr;        // read per call: the parity test flips it
    if (!no_mc && (ntaps == 9 || ntaps == 4)) {
        const unsigned nbm = (unsigned)((mc + 255) / 256);
        if (ntaps == 9) DP_LAUNCH((splitk_reduce_taps_mc_kernel<9>), dim3(nbm), dim3(256), 0, (hipStream_t)stream, ws, stride, splits, out, mc, accumulate);
        else            DP_LAUNCH((splitk_reduce_taps_mc_kernel<4>), dim3(nbm), dim3(256), 0, (hipStream_t)stream, ws, stride, splits, out, mc, accumulate);
        return DP_LAUNCH_CHECK();
    }
    long long nb = (mc * ntaps + 255) / 256;
    if (nb > 4096) nb = 4096;
    DP_LAUNCH(splitk_reduce_taps_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, ws, stride, splits,
                       out, mc, ntaps, accumulate);
    return DP_LAUNCH_CHECK();
}

extern "C" int dp_splitk_reduce(const float* ws, long long stride, int splits, float* out, long long n, int accumulate,
                                void* stream) {
    if (n <= 0) return 0;
    long long nb = (n + 255) / 256;
    if (nb > 4096) nb = 4096;
    DP_LAUNCH(splitk_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, ws, stride, splits, out,
                       n, accumulate);
    return DP_LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------
// weight packing (A operand of conv_gemm)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ W, int Co, int Ci, int taps, int mode,
                                                          float* __restrict__ dst, int ld) {
    // one thread per destination element; destination is [taps][K][ld]
    const int K = mode == 0 ? Ci : Co;     // reduction rows per tap
    const int Mv = mode == 0 ? Co : Ci;    // valid columns
    const long long total = (long long)taps * K * ld;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int m = (int)(i % ld);
        const long long rk = i / ld;
        const int k = (int)(rk % K);
        const int tap = (int)(rk / K);
        float v = 0.f;
        if (m < Mv) {
            if (mode == 0) v = W[((long long)m * Ci + k) * taps + tap];
            else           v = W[((long long)k * Ci + m) * taps + (taps - 1 - tap)];
        }
        dst[i] = v;
    }
}

extern "C" int dp_pack_weight(const float* W, int Co, int Ci, int taps, int mode, float* dst, int ld, void* stream) {
    const int K = mode == 0 ? Ci : Co;
    const long long total = (long long)taps * K * ld;
    if (total <= 0) return 0;
    long long nb = (total + 255) / 256;
    if (nb > 8192) nb = 8192;
    DP_LAUNCH(pack_weight_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, W, Co, Ci, taps, mode, dst,
                       ld);
    return DP_LAUNCH_CHECK();
}

extern "C" int dp_version(void) { return 100; }
