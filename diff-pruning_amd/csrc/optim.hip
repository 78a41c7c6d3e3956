// Fused finetune update over flat parameter / gradient / moment / EMA buffers (HBM-bound, one pass):
//   global-norm clip (ddpm_train.py:462) -> Adam (ddpm_train.py:331-337,463; torch.optim.Adam, wd = 0)
//   -> EMA with constant decay (diffusers/training_utils.py:201,215-216).
#include "dp_common.h"

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, long long n, float* __restrict__ partial) {
    __shared__ float red[4];
    const long long per = (n + gridDim.x - 1) / gridDim.x;
    const long long lo = (long long)blockIdx.x * per;
    long long hi = lo + per;
    if (hi > n) hi = n;
    float s = 0.f;
    for (long long i = lo + threadIdx.x; i < hi; i += 256) s += x[i] * x[i];
    s = dp_block_sum_256(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
extern "C" int dp_sumsq_partials(const float* x, long long n, float* partial, int nblocks, void* stream) {
    if (nblocks <= 0) return (int)hipErrorInvalidValue;
    DP_LAUNCH(sumsq_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, x, n, partial);
    return DP_LAUNCH_CHECK();
}

__global__ __launch_bounds__(256) void clip_coef_kernel(const float* __restrict__ partial, int n, float max_norm,
                                                        float* __restrict__ norm_out, float* __restrict__ coef_out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
    s = dp_block_sum_256(s, red);
    if (threadIdx.x == 0) {
        const float nrm = sqrtf(s);
        norm_out[0] = nrm;
        const float c = max_norm / (nrm + 1e-6f);
        coef_out[0] = c < 1.0f ? c : 1.0f;
    }
}
extern "C" int dp_clip_coef(const float* partial, int n, float max_norm, float* norm_out, float* coef_out, void* stream) {
    DP_LAUNCH(clip_coef_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, n, max_norm, norm_out, coef_out);
    return DP_LAUNCH_CHECK();
}

__global__ __launch_bounds__(256) void adam_ema_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, float* __restrict__ ema, long long n,
                                                       const float* __restrict__ clip_coef, float lr, float b1, float b2,
                                                       float eps, float bc1, float bc2, float ema_decay) {
    const float coef = clip_coef ? clip_coef[0] : 1.0f;
    const float step_size = lr / bc1;
    const float inv_sqrt_bc2 = 1.0f / sqrtf(bc2);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float gi = g[i] * coef;
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
        const float pi = p[i] - step_size * (mi / denom);
        p[i] = pi;
        if (ema) ema[i] = (1.0f - ema_decay) * pi + ema_decay * ema[i];
    }
}
extern "C" int dp_adam_ema(float* p, const float* g, float* m, float* v, float* ema, long long n, const float* clip_coef,
                           float lr, float b1, float b2, float eps, float bc1, float bc2, float ema_decay, void* stream) {
    if (n <= 0) return 0;
    long long nb = (n + 255) / 256;
    if (nb > 8192) nb = 8192;
    DP_LAUNCH(adam_ema_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, p, g, m, v, ema, n, clip_coef,
                       lr, b1, b2, eps, bc1, bc2, ema_decay);
    return DP_LAUNCH_CHECK();
}

// ---- the same update with the per-step scalars on the device (a captured finetune step replays with unchanged kernel arguments)
__global__ void set_step_scalars_kernel(float* __restrict__ hyper, float lr, float bc1, float bc2, unsigned step) {
    hyper[0] = lr;
    hyper[1] = bc1;
    hyper[2] = bc2;
    reinterpret_cast<unsigned*>(hyper)[3] = step;
}
extern "C" int dp_set_step_scalars(float* hyper, float lr, float bc1, float bc2, unsigned step, void* stream) {
    DP_LAUNCH(set_step_scalars_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, hyper, lr, bc1, bc2, step);
    return DP_LAUNCH_CHECK();
}

__global__ __launch_bounds__(256) void adam_ema_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                           float* __restrict__ v, float* __restrict__ ema, long long n,
                                                           const float* __restrict__ clip_coef, const float* __restrict__ hyper,
                                                           float b1, float b2, float eps, float ema_decay) {
    const float coef = clip_coef ? clip_coef[0] : 1.0f;
    const float lr = hyper[0], bc1 = hyper[1], bc2 = hyper[2];
    const float step_size = lr / bc1;
    const float inv_sqrt_bc2 = 1.0f / sqrtf(bc2);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float gi = g[i] * coef;
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
        const float pi = p[i] - step_size * (mi / denom);
        p[i] = pi;
        if (ema) ema[i] = (1.0f - ema_decay) * pi + ema_decay * ema[i];
    }
}
extern "C" int dp_adam_ema_dev(float* p, const float* g, float* m, float* v, float* ema, long long n, const float* clip_coef,
                               const float* hyper, float b1, float b2, float eps, float ema_decay, void* stream) {
    if (n <= 0) return 0;
    if (!hyper) return (int)hipErrorInvalidValue;
    long long nb = (n + 255) / 256;
    if (nb > 8192) nb = 8192;
    DP_LAUNCH(adam_ema_dev_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, p, g, m, v, ema, n, clip_coef, hyper,
                       b1, b2, eps, ema_decay);
    return DP_LAUNCH_CHECK();
}

