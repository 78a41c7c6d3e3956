// Transformer-block glue of the LDM (CompVis) UNet on channel-major tokens x[n][c][t] (t contiguous):
// LayerNorm over channels (forward / backward), GEGLU (forward / backward), per-(image, channel) broadcast add.
// All HBM-bound; one thread per token so that every access is coalesced across the wavefront.
// Reference: ldm_exp/ldm/modules/attention.py:37-46 (GEGLU), :196-212 (BasicTransformerBlock LayerNorms).
#include "dp_common.h"

// stats[(n*T + t)*2 + {0,1}] = {mean, rstd} over the C channels of token t.
// Workgroup = 64 consecutive tokens x 4 channel groups (wavefront w walks channels w, w+4, ...): every access is a
// contiguous 256-byte run of tokens, the channel loop is 4x shorter and carries two independent load streams per lane
// (one thread per token with 3 serial passes over up to 960 channels was latency bound: 31 % of an LDM step).
// Variance in one pass around the token's first channel as the shift (stable: the shifted mean is small).
// TOK tokens x (256 / TOK) channel groups per workgroup: 64 x 4 for large token counts; 16 x 16 when 64-token workgroups would
// leave the chip mostly idle (LDM importance pass: 6 latents x 1024 tokens = 96 workgroups on 256 CUs; 161 us for 28 MB).
template <int TOK>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, long long x_img_stride,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta, int N,
                                                     int C, int T, float eps, float* __restrict__ y, long long y_img_stride,
                                                     float* __restrict__ stats) {
    constexpr int CG = 256 / TOK;
    __shared__ float r1[CG][TOK], r2[CG][TOK];
    const int tl = threadIdx.x % TOK, cg = threadIdx.x / TOK;
    const long long tok = (long long)blockIdx.x * TOK + tl;
    const bool valid = tok < (long long)N * T;
    const long long tk = valid ? tok : 0;
    const int n = (int)(tk / T);
    const int t = (int)(tk - (long long)n * T);
    const float* xp = x + (long long)n * x_img_stride + t;
    const float k = xp[0];
    float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
    int c = cg;
    for (; c + CG < C; c += 2 * CG) {
        const float v0 = xp[(long long)c * T] - k, v1 = xp[(long long)(c + CG) * T] - k;
        s0 += v0; q0 += v0 * v0;
        s1 += v1; q1 += v1 * v1;
    }
    if (c < C) { const float v0 = xp[(long long)c * T] - k; s0 += v0; q0 += v0 * v0; }
    r1[cg][tl] = s0 + s1;
    r2[cg][tl] = q0 + q1;
    __syncthreads();
    float rs = 0.f, rq = 0.f;
#pragma unroll
    for (int j = 0; j < CG; ++j) { rs += r1[j][tl]; rq += r2[j][tl]; }            // fixed order
    const float ms = rs / (float)C;
    const float var = fmaxf(rq / (float)C - ms * ms, 0.f);
    const float mean = k + ms;
    const float rstd = 1.0f / sqrtf(var + eps);
    if (!valid) return;
    if (cg == 0) {
        stats[tok * 2 + 0] = mean;
        stats[tok * 2 + 1] = rstd;
    }
    float* yp = y + (long long)n * y_img_stride + t;
    for (c = cg; c < C; c += CG) yp[(long long)c * T] = (xp[(long long)c * T] - mean) * rstd * gamma[c] + beta[c];
}

extern "C" int dp_layernorm_fwd(const float* x, long long x_img_stride, const float* gamma, const float* beta, int N, int C,
                                int T, float eps, float* y, long long y_img_stride, float* stats, void* stream) {
    const long long ntok = (long long)N * T;
    if (ntok <= 0) return 0;
    if (ntok >= 64 * 1024)
        DP_LAUNCH(ln_fwd_kernel<64>, dim3((unsigned)((ntok + 63) / 64)), dim3(256), 0, (hipStream_t)stream, x,
                  x_img_stride, gamma, beta, N, C, T, eps, y, y_img_stride, stats);
    else
        DP_LAUNCH(ln_fwd_kernel<16>, dim3((unsigned)((ntok + 15) / 16)), dim3(256), 0, (hipStream_t)stream, x,
                  x_img_stride, gamma, beta, N, C, T, eps, y, y_img_stride, stats);
    return DP_LAUNCH_CHECK();
}

// dx = rstd * (gamma*dy - mean_c(gamma*dy) - xhat * mean_c(gamma*dy*xhat))  (+ add)      (same 64 x 4 decomposition)
template <int TOK>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ x, long long x_img_stride,
                                                     const float* __restrict__ gamma, const float* __restrict__ stats,
                                                     const float* __restrict__ dy, long long dy_img_stride, int N, int C,
                                                     int T, float* __restrict__ dx, long long dx_img_stride,
                                                     const float* __restrict__ add, long long add_img_stride) {
    constexpr int CG = 256 / TOK;
    __shared__ float r1[CG][TOK], r2[CG][TOK];
    const int tl = threadIdx.x % TOK, cg = threadIdx.x / TOK;
    const long long tok = (long long)blockIdx.x * TOK + tl;
    const bool valid = tok < (long long)N * T;
    const long long tk = valid ? tok : 0;
    const int n = (int)(tk / T);
    const int t = (int)(tk - (long long)n * T);
    const float mean = stats[tk * 2 + 0], rstd = stats[tk * 2 + 1];
    const float* xp = x + (long long)n * x_img_stride + t;
    const float* dp = dy + (long long)n * dy_img_stride + t;
    float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
    int c = cg;
    for (; c + CG < C; c += 2 * CG) {
        const float g0 = gamma[c] * dp[(long long)c * T], g1 = gamma[c + CG] * dp[(long long)(c + CG) * T];
        const float h0 = (xp[(long long)c * T] - mean) * rstd, h1 = (xp[(long long)(c + CG) * T] - mean) * rstd;
        a0 += g0; b0 += g0 * h0;
        a1 += g1; b1 += g1 * h1;
    }
    if (c < C) {
        const float g0 = gamma[c] * dp[(long long)c * T];
        a0 += g0; b0 += g0 * ((xp[(long long)c * T] - mean) * rstd);
    }
    r1[cg][tl] = a0 + a1;
    r2[cg][tl] = b0 + b1;
    __syncthreads();
    float ra = 0.f, rb = 0.f;
#pragma unroll
    for (int j = 0; j < CG; ++j) { ra += r1[j][tl]; rb += r2[j][tl]; }            // fixed order
    const float a = ra / (float)C;
    const float b = rb / (float)C;
    if (!valid) return;
    float* op = dx + (long long)n * dx_img_stride + t;
    const float* ap = add ? add + (long long)n * add_img_stride + t : nullptr;
    for (c = cg; c < C; c += CG) {
        const float xh = (xp[(long long)c * T] - mean) * rstd;
        float v = rstd * (gamma[c] * dp[(long long)c * T] - a - xh * b);
        if (ap) v += ap[(long long)c * T];
        op[(long long)c * T] = v;
    }
}

// pws[(n*C + c)*2 + {0,1}] = {sum_t dy, sum_t dy*xhat}: one wavefront per (n, c) row (reduce over n with dp_colsum_accum)
__global__ __launch_bounds__(256) void ln_param_kernel(const float* __restrict__ x, long long x_img_stride,
                                                       const float* __restrict__ stats, const float* __restrict__ dy,
                                                       long long dy_img_stride, int N, int C, int T, float* __restrict__ pws) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long long)N * C) return;
    const int n = (int)(row / C);
    const int c = (int)(row - (long long)n * C);
    const float* xp = x + (long long)n * x_img_stride + (long long)c * T;
    const float* dp = dy + (long long)n * dy_img_stride + (long long)c * T;
    const float* st = stats + (long long)n * T * 2;
    float s1 = 0.f, s2 = 0.f;
    for (int t = threadIdx.x & 63; t < T; t += 64) {
        const float d = dp[t];
        s1 += d;
        s2 += d * ((xp[t] - st[t * 2]) * st[t * 2 + 1]);
    }
    s1 = dp_wave_sum(s1);
    s2 = dp_wave_sum(s2);
    if ((threadIdx.x & 63) == 0) {
        pws[row * 2 + 0] = s1;
        pws[row * 2 + 1] = s2;
    }
}

extern "C" int dp_layernorm_bwd(const float* x, long long x_img_stride, const float* gamma, const float* stats,
                                const float* dy, long long dy_img_stride, int N, int C, int T, float* dx,
                                long long dx_img_stride, const float* add, long long add_img_stride, float* pws, void* stream) {
    const long long ntok = (long long)N * T;
    if (ntok <= 0) return 0;
    if (ntok >= 64 * 1024)
        DP_LAUNCH(ln_bwd_kernel<64>, dim3((unsigned)((ntok + 63) / 64)), dim3(256), 0, (hipStream_t)stream, x,
                  x_img_stride, gamma, stats, dy, dy_img_stride, N, C, T, dx, dx_img_stride, add, add_img_stride);
    else
        DP_LAUNCH(ln_bwd_kernel<16>, dim3((unsigned)((ntok + 15) / 16)), dim3(256), 0, (hipStream_t)stream, x,
                  x_img_stride, gamma, stats, dy, dy_img_stride, N, C, T, dx, dx_img_stride, add, add_img_stride);
    int e = DP_LAUNCH_CHECK();
    if (e) return e;
    const long long nrows = (long long)N * C;
    DP_LAUNCH(ln_param_kernel, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, x_img_stride,
                       stats, dy, dy_img_stride, N, C, T, pws);
    return DP_LAUNCH_CHECK();
}

// ---- GEGLU: out[n][c][t] = in[n][c][t] * gelu(in[n][c + D][t]),  c < D  (exact erf GELU, F.gelu default)
__device__ __forceinline__ float dp_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float dp_gelu_grad(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
}

__global__ void geglu_fwd_kernel(const float* __restrict__ in, int N, long long half_plane, float* __restrict__ out) {
    const long long total = (long long)N * half_plane;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / half_plane;
        const long long r = i - n * half_plane;
        const float* p = in + n * 2 * half_plane;
        out[i] = p[r] * dp_gelu(p[half_plane + r]);
    }
}
__global__ void geglu_bwd_kernel(const float* __restrict__ in, const float* __restrict__ dout, int N, long long half_plane,
                                 float* __restrict__ din) {
    const long long total = (long long)N * half_plane;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / half_plane;
        const long long r = i - n * half_plane;
        const float* p = in + n * 2 * half_plane;
        float* d = din + n * 2 * half_plane;
        const float a = p[r], g = p[half_plane + r], dy = dout[i];
        d[r] = dy * dp_gelu(g);
        d[half_plane + r] = dy * a * dp_gelu_grad(g);
    }
}
static inline unsigned tf_grid(long long n) {
    long long nb = (n + 255) / 256;
    if (nb > 8192) nb = 8192;
    return (unsigned)(nb < 1 ? 1 : nb);
}
extern "C" int dp_geglu_fwd(const float* in, int N, long long half_plane, float* out, void* stream) {
    if ((long long)N * half_plane <= 0) return 0;
    DP_LAUNCH(geglu_fwd_kernel, dim3(tf_grid((long long)N * half_plane)), dim3(256), 0, (hipStream_t)stream, in, N,
                       half_plane, out);
    return DP_LAUNCH_CHECK();
}
extern "C" int dp_geglu_bwd(const float* in, const float* dout, int N, long long half_plane, float* din, void* stream) {
    if ((long long)N * half_plane <= 0) return 0;
    DP_LAUNCH(geglu_bwd_kernel, dim3(tf_grid((long long)N * half_plane)), dim3(256), 0, (hipStream_t)stream, in, dout,
                       N, half_plane, din);
    return DP_LAUNCH_CHECK();
}

// out[n][c][t] = x[n][c][t] + v[n*C + c]   (cross-attention with a single context token broadcasts one vector per image)
__global__ void add_rowvec_kernel(const float* __restrict__ x, long long x_img_stride, const float* __restrict__ v, int N,
                                  int C, int T, float* __restrict__ out, long long o_img_stride) {
    const long long per = (long long)C * T;
    const long long total = (long long)N * per;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / per;
        const long long r = i - n * per;
        out[n * o_img_stride + r] = x[n * x_img_stride + r] + v[n * C + r / T];
    }
}
extern "C" int dp_add_rowvec(const float* x, long long x_img_stride, const float* v, int N, int C, int T, float* out,
                             long long o_img_stride, void* stream) {
    if ((long long)N * C * T <= 0) return 0;
    DP_LAUNCH(add_rowvec_kernel, dim3(tf_grid((long long)N * C * T)), dim3(256), 0, (hipStream_t)stream, x,
                       x_img_stride, v, N, C, T, out, o_img_stride);
    return DP_LAUNCH_CHECK();
}
