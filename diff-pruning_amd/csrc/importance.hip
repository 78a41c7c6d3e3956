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

// ------------------------------------------------------------------------------------------------
// Batched forms for the prune tail (one group of coupled layers at a time; ddpm_prune.py:108-110 walks ~50 groups, each
// scored from the weights and gradients the previous group's pruning left behind).  The per-member launches above cost
// ~20 us of host time each and a group has 4 ... 22 members: dp_group_score computes every member's per-channel vector in ONE
// launch (same per-member code, hence the same bits) and folds them into the group score in a second one, in member order
// (score = ((0 + m_1) + m_2) + ..., what the chain of dp_axpby / dp_gather_add calls computes); dp_slice_batch gathers the
// kept channels of every weight, bias and gradient tensor of the group in one launch.
// ------------------------------------------------------------------------------------------------
#define DP_SCORE_BATCH 24
struct ScoreBatch {
    int n;
    dp_score_member m[DP_SCORE_BATCH];
};

__global__ __launch_bounds__(256) void group_score_part_kernel(const ScoreBatch b, float* __restrict__ scratch) {
    __shared__ float red[4];
    __shared__ float part[4][64];
    int i = 0;
    while (i + 1 < b.n && (int)blockIdx.x >= b.m[i + 1].blk0) ++i;              // block-uniform scan
    const dp_score_member& m = b.m[i];
    const int blk = (int)blockIdx.x - m.blk0;
    const int mode = m.mode;
    if (mode == 3) {                                                            // GroupNorm member: |w g| per channel
        const int c = blk * 256 + threadIdx.x;
        if (c < m.R) scratch[m.full_off + c] = fabsf(m.w[c] * m.g[c]);
        return;
    }
    if (m.dim == 0) {                                                           // wg_rows_kernel
        const long long inner = (long long)m.C * m.T;
        const float* wr = m.w + (long long)blk * inner;
        const float* gr = m.g + (long long)blk * inner;
        float s = 0.f;
        for (long long e = threadIdx.x; e < inner; e += 256) s += wg_f(wr[e], gr[e], mode);
        s = dp_block_sum_256(s, red);
        if (threadIdx.x == 0) scratch[m.full_off + blk] = (mode == 2) ? fabsf(s) : s;
        return;
    }
    const long long CT = (long long)m.C * m.T;                                  // wg_cols_ct_kernel
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const long long col = (long long)blk * 64 + lane;
    float s = 0.f;
    if (col < CT) {
        for (int r = wave; r < m.R; r += 4) s += wg_f(m.w[(long long)r * CT + col], m.g[(long long)r * CT + col], mode);
    }
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && col < CT) scratch[m.col_off + col] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

__global__ void group_score_combine_kernel(const ScoreBatch b, const float* __restrict__ scratch, const int64_t* __restrict__ idx,
                                           int n0, float* __restrict__ score, int accumulate) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n0) return;
    float acc = accumulate ? score[j] : 0.f;
    for (int i = 0; i < b.n; ++i) {
        const dp_score_member& m = b.m[i];
        const long long c = m.idx_off >= 0 ? (long long)idx[m.idx_off + j] : (long long)j;
        float v;
        if (m.mode != 3 && m.dim == 1) {                                        // wg_fold_taps_kernel
            float s = 0.f;
            for (int t = 0; t < m.T; ++t) s += scratch[m.col_off + c * m.T + t];
            v = (m.mode == 2) ? fabsf(s) : s;
        } else {
            v = scratch[m.full_off + c];
        }
        acc = acc + v;
    }
    score[j] = acc;
}

extern "C" int dp_group_score(const dp_score_member* members, int n, int n0, const int64_t* idx, float* scratch, float* score,
                              void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (n <= 0 || n0 <= 0) return 0;
    for (int lo = 0; lo < n; lo += DP_SCORE_BATCH) {
        ScoreBatch b;
        b.n = (n - lo < DP_SCORE_BATCH) ? n - lo : DP_SCORE_BATCH;
        int blocks = 0;
        for (int i = 0; i < b.n; ++i) {
            b.m[i] = members[lo + i];
            dp_score_member& m = b.m[i];
            if (m.R <= 0 || m.C <= 0 || m.T <= 0) return (int)hipErrorInvalidValue;
            m.blk0 = blocks;
            if (m.mode == 3)      blocks += (m.R + 255) / 256;
            else if (m.dim == 0)  blocks += m.R;
            else                  blocks += (int)(((long long)m.C * m.T + 63) / 64);
        }
        DP_LAUNCH(group_score_part_kernel, dim3(blocks), dim3(256), 0, st, b, scratch);
        DP_LAUNCH(group_score_combine_kernel, dim3((n0 + 255) / 256), dim3(256), 0, st, b, scratch, idx, n0, score, lo > 0 ? 1 : 0);
    }
    return DP_LAUNCH_CHECK();
}

// dst = src with only the kept channels along `dim` of the [R][C][T] view (function.py:85-146,168-207,274-302: weight, bias
// and their accumulated gradients of every member of a group, one launch)
#define DP_SLICE_BATCH 48
struct SliceBatch {
    int n;
    dp_slice_item it[DP_SLICE_BATCH];
};

__global__ __launch_bounds__(256) void slice_batch_kernel(const SliceBatch b, const int64_t* __restrict__ keep) {
    int i = 0;
    while (i + 1 < b.n && (int)blockIdx.x >= b.it[i + 1].blk0) ++i;
    const dp_slice_item& it = b.it[i];
    const long long CT = (long long)it.C * it.T;
    const long long total = it.dim == 0 ? (long long)it.n_keep * CT : (long long)it.R * it.n_keep * it.T;
    const long long step = (long long)it.nblk * 256;
    const int64_t* kp = keep + it.keep_off;
    for (long long e = (long long)((int)blockIdx.x - it.blk0) * 256 + threadIdx.x; e < total; e += step) {
        long long s;
        if (it.dim == 0) {
            const long long r = e / CT;
            s = (long long)kp[r] * CT + (e - r * CT);
        } else {
            const long long nkT = (long long)it.n_keep * it.T;
            const long long r = e / nkT;
            const long long rem = e - r * nkT;
            const long long c = rem / it.T;
            s = r * CT + (long long)kp[c] * it.T + (rem - c * it.T);
        }
        it.dst[e] = it.src[s];
    }
}

extern "C" int dp_slice_batch(const dp_slice_item* items, int n, const int64_t* keep, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    for (int lo = 0; lo < n; lo += DP_SLICE_BATCH) {
        SliceBatch b;
        b.n = (n - lo < DP_SLICE_BATCH) ? n - lo : DP_SLICE_BATCH;
        int blocks = 0;
        for (int i = 0; i < b.n; ++i) {
            b.it[i] = items[lo + i];
            dp_slice_item& it = b.it[i];
            if (it.R <= 0 || it.C <= 0 || it.T <= 0 || it.n_keep <= 0 || (it.dim != 0 && it.dim != 1)) return (int)hipErrorInvalidValue;
            const long long total = it.dim == 0 ? (long long)it.n_keep * it.C * it.T : (long long)it.R * it.n_keep * it.T;
            long long nb = (total + 1023) / 1024;
            if (nb > 256) nb = 256;
            if (nb < 1) nb = 1;
            it.blk0 = blocks;
            it.nblk = (int)nb;
            blocks += (int)nb;
        }
        DP_LAUNCH(slice_batch_kernel, dim3(blocks), dim3(256), 0, st, b, keep);
    }
    return DP_LAUNCH_CHECK();
}
