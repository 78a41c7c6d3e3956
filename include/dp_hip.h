/* dp_hip.h -- C-ABI of the MI355X-native Diff-Pruning hot path (libdp_hip.so).
 *
 * The reference (VainF/Diff-Pruning) is pure Python on top of ATen; it has no FFI of its own.  The
 * boundary a maintainer would bind is therefore the set of ATen calls its hot path makes; each entry
 * point below names the reference call site it replaces (paths relative to the reference root).
 * All pointers are DEVICE pointers owned by the caller (PyTorch owns every allocation); `stream` is a
 * hipStream_t passed as void*; every function only enqueues work on `stream` and returns a
 * hipError_t-style int (0 = success).  Nothing here throws, allocates or synchronises.
 *
 * Activation layout everywhere: channel-major planes, element (img, c, h, w) at
 *     base + img*img_stride + c*H*W + h*W + w          (NCHW with an explicit image stride)
 * Linear layers on [B, C] vectors use the same convention with H = W = 1.
 */
#ifndef DP_HIP_H
#define DP_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Gather geometry shared by the implicit-GEMM kernels (see csrc/dp_common.h ConvGeom). */
typedef struct dp_conv_geom {
    int Ho, Wo, Hs, Ws, Hv, Wv, kw, stride, sden, pad_t, pad_l, ups, c_split, _pad;
    long long x1_img_stride, x2_img_stride;
} dp_conv_geom;

/* D[m][pix] = sum_{tap,c} A(tap,c,m) * X(pix,tap,c)  -- conv3x3 / conv1x1 / linear forward and dgrad,
 * attention QK^T and dP (batched).  Replaces F.conv2d / F.linear / torch.baddbmm forward and the
 * input-gradient half of ConvolutionBackward / AddmmBackward:
 *   diffusers/models/resnet.py:606,630,633 (conv1, conv2, conv_shortcut), :612 (time_emb_proj),
 *   diffusers/models/resnet.py:166,218 (Upsample2D/Downsample2D conv), unet_2d.py:273,304 (conv_in/out),
 *   diffusers/models/attention_processor.py:438-445 (to_q/k/v), :339-345 (baddbmm), :455 (to_out).
 * a_kc = 0: A is a packed weight  A[(tap*C + c)*lda + m]  (m contiguous, lda % 4 == 0, zero padded)
 * a_kc = 1: A[m*lda + c]  (c contiguous; taps must be 1)
 * Epilogue: v = alpha*acc (+bias[m]) (+tadd[img*tadd_stride + m]) (+res[img*r_img_stride + m*HoWo + r]);
 *           v *= post_scale; (act: v = max(v, 0)); out = accumulate ? out + v : v.
 * ksplit > 1 (small pixel counts): the K loop is split over workgroups, raw partial tiles go to ws and a second
 * kernel sums them in a fixed order and applies the same epilogue (deterministic). */
typedef struct dp_conv_gemm_params {
    const float* A; long long a_bs; int lda; int a_kc;
    const float* X1; const float* X2; long long x_bs;
    unsigned a_bytes, x1_bytes, x2_bytes, _pad0;   /* readable extent of A / X1 / X2 (per batch), each < 2 GiB */
    dp_conv_geom g;
    int M, C, NPIX, ntaps, batches, tile;      /* tile: 0 = 128x128, 1 = 64x128, 2 = 64x64, 3 = 96x128 (stride-1 fast kernel; else as 0) */
    float* out; long long o_img_stride; long long o_bs;
    float alpha; float post_scale;
    const float* bias; const float* tadd; long long tadd_stride;
    const float* res; long long r_img_stride;
    int accumulate; int ksplit;            /* ksplit > 1: split the K loop over blockIdx.z, partials in ws (non-batched only) */
    float* ws;                             /* >= ksplit*M*NPIX floats when ksplit > 1 */
    int x_guard; int act;                  /* act = 1: ReLU after post_scale (BasicConv2d of the FID Inception network).
                                              x_guard = 1: the 4 bytes in front of X1 (and X2) are readable memory.  The 16-byte B-tile
                                              loads of the stride-1 fast kernel read one element to the left of an image
                                              row for the shifted taps (overwritten with 0 in LDS); for the first row of the
                                              tensor that element lies in front of it.  0 = use the 4-byte loads. */
    unsigned* tile_counters;               /* ksplit > 1, optional: one zero-initialised counter per output tile (grid x * grid y).
                                              The LAST workgroup to deliver its partial tile sums the ksplit partials of that tile in
                                              ascending split order (the same fixed order as the separate reduction kernel: same bits,
                                              whoever arrives last), applies the epilogue and re-zeroes the counter; NULL = a second
                                              launch does that. */
} dp_conv_gemm_params;
int dp_conv_gemm(const dp_conv_gemm_params* p, void* stream);

/* The same convolution for the 3x3, stride 1, pad 1 case as a one-dimensional Winograd F(2, 3) implicit GEMM (csrc/winograd.hip):
 * 2/3 of the multiplies of dp_conv_gemm (4 per 2 outputs x 3 taps along W), input transform at fragment-read time, output transform
 * in the epilogue.  Replaces the same call sites as dp_conv_gemm's 3x3 forward / input-gradient launches
 * (diffusers/models/resnet.py:606,630 conv1 / conv2 and their ConvolutionBackward input gradients;
 * ldm/modules/diffusionmodules/openaimodel.py:214-232 in_layers / out_layers).  Same parameter block and epilogue; A is the
 * operand of dp_pack_weight_wino, U[(ky*4 + pos)*C + c][lda]; batches / a_kc / tile_counters must be unset; ksplit > 1 splits the
 * (channel chunk, kernel row) loop over blockIdx.z with partials in ws and the reduction launch of dp_conv_gemm.
 * dp_conv_wino_supported returns the K-chunk width the kernel would use (16 / 8) or 0 when the shape is not taken
 * (W a power of two in 4..256, channel counts per concat source multiples of 8, 3x3 / stride 1 / pad 1). */
int dp_conv_wino(const dp_conv_gemm_params* p, void* stream);
int dp_conv_wino_supported(const dp_conv_gemm_params* p);
/* the reduction launch of a split-K convolution (ksplit > 1, no tile_counters): out = epilogue(sum_z ws[z][m][pix]), ascending z */
int dp_conv_splitk_epilogue(const dp_conv_gemm_params* p, void* stream);
/* mode 0: forward operand (K = Ci, columns = co); mode 1: input-gradient operand (K = Co, columns = ci, taps flipped).
 * dst holds 12 * K * ld floats. */
int dp_pack_weight_wino(const float* W, int Co, int Ci, int mode, float* dst, int ld, void* stream);

/* The same convolution as a TWO-DIMENSIONAL Winograd F(2x2, 3x3) implicit GEMM (csrc/winograd2d.hip, round 6): 16 multiplies per
 * 2x2 output tile and channel instead of 36 -- 4/9 of dp_conv_gemm, 2/3 of dp_conv_wino.  One workgroup = 64 output channels x 32
 * tiles (128 output pixels = whole image rows) x 16 positions, the four ROWS of the position matrix on the four wavefronts; input
 * transform at fragment-read time, output transform in registers (columns) and through LDS (rows).  Replaces the same call sites as
 * dp_conv_wino (diffusers/models/resnet.py:606,630 and their input gradients; ldm/modules/diffusionmodules/openaimodel.py:214-232).
 * Same parameter block, epilogue operands and split-K contract; A is dp_pack_weight_wino2d's operand U[(pos*K + k)][lda],
 * pos = 4 i + j of U = G g G^T (dst holds 16 * K * ld floats; mode 0 forward, mode 1 input gradient with both tap axes flipped).
 * Takes W a power of two in 4..256 (beyond 64 pixels a block is a 2-row x 64-column segment with a halo tile), H even, channel counts
 * per concat source in multiples of 8, even image strides. */
int dp_conv_wino2d(const dp_conv_gemm_params* p, void* stream);
int dp_conv_wino2d_supported(const dp_conv_gemm_params* p);
int dp_pack_weight_wino2d(const float* W, int Co, int Ci, int mode, float* dst, int ld, void* stream);

/* The same convolution as Winograd F(4, 3) along W (csrc/winograd43.hip): HALF the multiplies of dp_conv_gemm (6 per 4 outputs x 3
 * taps), for the forwards that keep nothing for a backward -- the sampling loops (diffusers/pipelines/ddim/pipeline_ddim.py:101-116,
 * pipelines/ddpm/pipeline_ddpm.py:87-96) and the CFG sampler of the LDM importance pass (ldm_exp/prune_ldm.py:111-118,
 * ldm/models/diffusion/ddim.py:165-203).  fp32 error vs fp64 ~1e-6 (F(2, 3): ~3e-7): not used for scored gradients.  Same parameter
 * block, epilogue and split-K contract as dp_conv_wino; A is dp_pack_weight_wino43's operand U[(ky*6 + pos)*C + c][lda]
 * (forward flavour only; dst holds 18 * Ci * ld floats).  Takes W a power of two in 4..256, channel counts per concat source in
 * multiples of 8, pixel count / image strides in multiples of 4, 16-byte aligned tensors. */
int dp_conv_wino43(const dp_conv_gemm_params* p, void* stream);
int dp_conv_wino43_supported(const dp_conv_gemm_params* p);
int dp_pack_weight_wino43(const float* W, int Co, int Ci, float* dst, int ld, void* stream);

/* D[m][c][tap] = alpha * sum_pix A[m][pix] * X(pix, c, tap)  -- weight gradients (split over pixels, one kernel tap
 * per workgroup) and the k-contiguous batched products of attention (P.V, dQ).  Replaces the weight-gradient half of
 * ConvolutionBackward / AddmmBackward reached from loss.backward() (ddpm_prune.py:102) and torch.bmm
 * (attention_processor.py:446).
 * A element (m, pix=(img,r)) at A + z*a_bs + img*a_img_stride + m*HoWo + r;  NCOLS = number of channels c.
 * batched = 1: blockIdx.z = batch, output out[z*o_bs + m*ldo + c]  (ntaps must be 1);
 * batched = 0: blockIdx.z = split*ntaps + tap, partial sums to out[split*o_bs + m*ldo + c*ntaps + tap]
 *              (reduce with dp_splitk_reduce), or direct (+accumulate) when splits == 1.
 * merge != 0 (few channels on one side, conv_in / conv_out): one tile holds all C*ntaps columns; with bit 1 the plain
 *              rows are the layer INPUT and the gathered operand is dy with mirrored taps (stride-1 'same' convs only).
 * nn.Linear forward y[n][o] = sum_i x[n][i] W[o][i] + b[o] (embeddings.py:192-212, resnet.py:611) is the batched form
 * with one batch: A = x, X1 = W (both row-major, reduction index contiguous), col_bias = b. */
typedef struct dp_nt_gemm_params {
    const float* A; long long a_bs; long long a_img_stride;
    const float* X1; const float* X2; long long x_bs;
    unsigned a_bytes, x1_bytes, x2_bytes, _pad0;   /* readable extent of A / X1 / X2 (per batch), each < 2 GiB */
    dp_conv_geom g;
    int M, C, NCOLS, ntaps, P, batches, splits, p_per_split, tile, batched;   /* tile: 0 = 128x128, 1 = 64x128, 2 = 64x64 */
    float* out; long long o_bs; int ldo; int accumulate;
    float alpha; int merge;                /* merge: bit 0 = taps folded into the columns (NCOLS = C*ntaps), bit 1 = mirrored taps */
    long long ocs;                         /* merge: output element (row, c, tap) at row*ldo + c*ocs + tap */
    long long o_tap_stride; int o_col_stride; int xcd;       /* 0 = defaults (1, ntaps); (M*NCOLS, 1) with ldo = NCOLS gives
                                              tap-major partials [tap][m][c] whose stores are contiguous */
    /* xcd != 0 (fast weight-gradient kernel only): XCD-aware workgroup order, the workgroups of one split share an L2 */
    const float* col_bias;                 /* optional [NCOLS]: added per output column by split 0 (nn.Linear bias), or NULL */
} dp_nt_gemm_params;
int dp_nt_gemm(const dp_nt_gemm_params* p, void* stream);

/* Weight gradient of the same convolutions by the transposed algorithm (3 taps from output pairs: 4 multiplies instead of 6 per
 * pair, kernel row and channel pair), csrc/winograd.hip wgrad_wino_kernel.  Parameter block as for dp_nt_gemm's weight-gradient
 * launches (A = dy, X1 / X2 = the layer input, ntaps = 9, tap-major split-K partials); the K range is counted in 32-pixel tiles
 * over P/32 + 1 tiles, so splits * p_per_split must cover P + 32.  Replaces the ConvolutionBackward weight gradient of
 * diffusers/models/resnet.py:606,630 (conv1 / conv2).  Shapes: 3x3 / stride 1 / pad 1, W a power of two in 8..256, H*W a power of
 * two >= 64, P % 32 == 0, c_split % 64 == 0 with two sources. */
int dp_wgrad_wino(const dp_nt_gemm_params* p, void* stream);
int dp_wgrad_wino_supported(const dp_nt_gemm_params* p);

/* The same weight gradient by the TWO-dimensional transposed algorithm F(3x3, 2x2) (csrc/wgrad2d.hip, round 6): 16 multiplies per
 * (2x2 block of output gradients, m, c) instead of 36 -- 2/3 of dp_wgrad_wino's.  One workgroup = a 64 x 32 (m, c) tile of all 16
 * position matrices over a range of 64-pixel K tiles; the nine taps are folded out of the positions in the epilogue and written as
 * the same tap-major split-K partials (dp_splitk_reduce_taps).  Parameter block as for dp_wgrad_wino; splits * p_per_split must cover
 * P, counted in 64-pixel tiles.  Shapes: W in {8, 16, 32}, H even, H*W a power of two >= 64, P % 64 == 0, c_split % 32 == 0 with two
 * sources.  Replaces the same ConvolutionBackward weight gradients (diffusers/models/resnet.py:606,630). */
int dp_wgrad_wino2d(const dp_nt_gemm_params* p, void* stream);
int dp_wgrad_wino2d_supported(const dp_nt_gemm_params* p);

/* out[i] (+)= sum_s ws[s*stride + i], fixed summation order (deterministic split-K epilogue). */
int dp_splitk_reduce(const float* ws, long long stride, int splits, float* out, long long n, int accumulate, void* stream);
/* same for tap-major partials ws[s][tap][mc] -> out[mc][ntaps] (the torch weight layout) */
int dp_splitk_reduce_taps(const float* ws, long long stride, int splits, float* out, long long mc, int ntaps,
                          int accumulate, void* stream);

/* Weight packing for dp_conv_gemm's A operand.  W is a torch Conv2d/Linear weight [Co][Ci][taps].
 * mode 0 (forward): dst[(tap*Ci + ci)*ld + co] = W[co][ci][tap],            ld = roundup4(Co)
 * mode 1 (dgrad)  : dst[(tap*Co + co)*ld + ci] = W[co][ci][taps-1-tap],     ld = roundup4(Ci)
 * dst must hold taps*K*ld floats; padding columns are written as zeros. */
int dp_pack_weight(const float* W, int Co, int Ci, int taps, int mode, float* dst, int ld, void* stream);

/* Dropout of the finetune step (nn.Dropout in ResnetBlock2D, resnet.py:628, and Attention.to_out[1],
 * attention_processor.py:457; probability set by utils.set_dropout, utils.py:26-29, ddpm_train.py:380-382).
 * The mask is a pure function of (seed, site, step, logical element index) through Philox4x32-10 -- see csrc/dp_common.h --
 * so the backward kernels regenerate it instead of reading a stored mask.  thr24 = ceil(p * 2^24) (0 disables),
 * scale = 1/(1-p), site = a stable id of the layer, step = optimizer step, n_off = global index of the shard's first
 * image (masks do not depend on how the batch is sharded over ranks). */
typedef struct dp_dropout {
    unsigned thr24; float scale; unsigned long long seed; unsigned site; unsigned step; long long n_off;
    const unsigned* step_dev;   /* optional DEVICE counter read instead of `step`: a captured / replayed finetune step must not bake
                                 * the optimizer step into its kernel arguments (dp_set_step_scalars writes it) */
} dp_dropout;

/* Standalone forms: y[i] = x[i] * m(i) (in place allowed; forward and backward are the same map), and the bare mask
 * multipliers m(i) in {0, scale} for logical elements [idx0, idx0 + n) (parity tests export masks through it).
 * x is [N][per_img] with image stride x_img_stride (logical index = (n_off + n) * per_img + r). */
int dp_dropout_apply(const float* x, long long x_img_stride, float* y, long long y_img_stride, int N, long long per_img,
                     const dp_dropout* drop, void* stream);
int dp_dropout_mask(float* m, long long idx0, long long n, const dp_dropout* drop, void* stream);
/* out[i] = standard-normal draw of logical element idx0 + i: Philox4x32-10 on counter (idx >> 2 lo, hi, stream_id, step), key
 * seed, Box-Muller on the word pairs (see csrc/elementwise.hip).  Replaces the device-RNG draws of the LDM importance pass
 * (ldm/models/diffusion/ddim.py:122 `torch.randn(shape, device=device)`, ddpm.py:1023 `torch.randn_like(x_start)`) with a
 * stream that does not depend on how the latents are sharded over ranks. */
int dp_randn_philox(float* out, long long idx0, long long n, unsigned long long seed, unsigned stream_id, unsigned step,
                    void* stream);

/* GroupNorm (+ optional SiLU) (+ optional dropout of the result, drop may be NULL) forward over a (virtually
 * concatenated) NCHW tensor.
 * Replaces F.group_norm + F.silu: resnet.py:596-598,622-628, unet_2d.py:302-303, attention_processor.py:433.
 * Channel c < c_split is read from x1 (image stride x1_img_stride), else from x2.  y is contiguous
 * [N][C][HW] with image stride y_img_stride.  stats[(n*G+g)*2 + {0,1}] = {mean, rstd}. */
int dp_groupnorm_silu_fwd(const float* x1, const float* x2, int c_split, long long x1_img_stride, long long x2_img_stride,
                          const float* gamma, const float* beta, int N, int C, int HW, int G, float eps, int silu,
                          float* y, long long y_img_stride, float* stats, const dp_dropout* drop, void* stream);

/* Backward of the above (drop: the same descriptor as in the forward; dz is masked on the fly).  dz = gradient w.r.t. the (SiLU'd) output, image stride dz_img_stride.
 * dx (image stride dx_img_stride) = gn_backward(dz) (+ add1) (+ add2); add tensors carry their own image
 * strides.  pws[(n*C + c)*2 + {0,1}] = per-image sums {sum dy, sum dy*xhat} (reduce over n with
 * dp_colsum_accum to obtain dbeta / dgamma).  Replaces NativeGroupNormBackward + SiluBackward. */
int dp_groupnorm_silu_bwd(const float* x1, const float* x2, int c_split, long long x1_img_stride, long long x2_img_stride,
                          const float* gamma, const float* beta, const float* stats, const float* dz, long long dz_img_stride,
                          int N, int C, int HW, int G, int silu,
                          float* dx, long long dx_img_stride,
                          const float* add1, long long add1_img_stride, const float* add2, long long add2_img_stride,
                          float* pws, const dp_dropout* drop, float* rows, void* stream);
/* rows (optional, [N][C]): rows[n*C + c] = sum_hw dx[n][c][hw] -- the bias / time-embedding-projection gradient rows of the
 * layer that produced x fall out of the same pass (otherwise a separate dp_rowsum_nc re-reads dx). */

/* The same two operations for FEW, LARGE groups (e.g. 256x256 images at batch 4: N*G = 128 groups of 1 MB): the work unit
 * is one of `slices` equal slices of one channel plane, partial statistics go through ws and are combined in a fixed
 * order (Chan's parallel variance in the forward).  Requires HW % (4*slices) == 0 and 16-byte aligned planes.
 * ws: forward >= N*C*slices*2 floats, backward >= N*C*slices*2 + N*G*2 floats. */
int dp_groupnorm_silu_fwd_split(const float* x1, const float* x2, int c_split, long long x1_img_stride, long long x2_img_stride,
                                const float* gamma, const float* beta, int N, int C, int HW, int G, float eps, int silu,
                                float* y, long long y_img_stride, float* stats, int slices, float* ws,
                                const dp_dropout* drop, void* stream);
int dp_groupnorm_silu_bwd_split(const float* x1, const float* x2, int c_split, long long x1_img_stride, long long x2_img_stride,
                                const float* gamma, const float* beta, const float* stats, const float* dz,
                                long long dz_img_stride, int N, int C, int HW, int G, int silu, float* dx,
                                long long dx_img_stride, const float* add1, long long add1_img_stride, const float* add2,
                                long long add2_img_stride, float* pws, int slices, float* ws, const dp_dropout* drop,
                                void* stream);

/* out[c*ostride] (+)= sum_n ws[(n*C + c)*wstride + woff]   (deterministic, n ascending) */
int dp_colsum_accum(const float* ws, int N, int C, int wstride, int woff, float* out, int accumulate, void* stream);

/* The same for n independent items in one launch (per 80 items): dst[c] (+)= sum_n src[(n*C + c)*wstride + woff].  The host
 * queues the bias / GroupNorm-parameter gradient sums of a whole backward pass and flushes them together.
 * With ld != 0 element (n, c) sits at src[(n*ld + c)*wstride + woff] (a column slice of a wider matrix). */
typedef struct dp_colsum_item {
    const float* src; float* dst; int N, C, wstride, woff, accumulate, ld;   /* ld: row pitch in elements, 0 = C */
} dp_colsum_item;
int dp_colsum_accum_batch(const dp_colsum_item* items, int n, void* stream);

/* rows[n*C + c] = sum_hw x[n*img_stride + c*HW + hw]  (bias / time-embedding-projection gradients) */
int dp_rowsum_nc(const float* x, long long img_stride, int N, int C, int HW, float* rows, void* stream);

/* y = silu(x) ; dx = dy * silu'(x)   (resnet.py:611 nonlinearity(temb), embeddings.py:206 act) */
int dp_silu_fwd(const float* x, float* y, long long n, void* stream);
int dp_silu_bwd(const float* x, const float* dy, float* dx, long long n, int accumulate, void* stream);

/* y = a*x + b*y  (elementwise) */
int dp_axpby(const float* x, float a, float* y, float b, long long n, void* stream);
/* strided plane copy / add: dst[img*d_stride + i] (+)= src[img*s_stride + i], i < per_img */
int dp_copy_strided(const float* src, long long s_stride, float* dst, long long d_stride, int N, long long per_img, int accumulate, void* stream);

/* Row softmax over the last dim (in place allowed).  attention_processor.py:350-353. */
int dp_softmax_fwd(const float* s, float* p, long long rows, int cols, void* stream);
/* Fused attention forward (no score tensor): for every image n < N and head h < heads
 *   o[n][h*dv + c][i] = sum_j v[n][h*dv + c][j] * softmax_j(scale * sum_c' q[n][h*d + c'][i] * k[n][h*d + c'][j]),  i, j < T.
 * Replaces get_attention_scores (baddbmm + softmax) + bmm of attention_processor.py:415-470 / the einsum-softmax-einsum of
 * ldm/modules/attention.py:168-193 for forwards that keep nothing for a backward (sampling).  Tensors are channel-major
 * ([N, C, T], tokens contiguous; head h = channel rows [h*d, (h+1)*d) -- head_to_batch_dim as a view); *_bs = elements between
 * consecutive images (q / k / v may be channel slices of one fused QKV tensor).  d = query / key width per head, dv = value
 * width per head (they differ after pruning), any d, dv >= 1 with dp_attention_fwd_supported(T, d, dv) != 0 (T a multiple of 32,
 * max(d, dv) <= 640); other shapes return hipErrorInvalidValue -- the caller keeps the three-launch path for those. */
typedef struct dp_attention_params {
    const float* q; const float* k; const float* v; float* o;
    long long q_bs, k_bs, v_bs, o_bs;
    int N, heads, d, dv, T;
    float scale;
    int variant;                           /* 0 = library's choice; 1 / 2 force the plain / the software-pipelined schedule
                                              (same arithmetic, different instruction order) */
    int _pad;
} dp_attention_params;
int dp_attention_fwd(const dp_attention_params* p, void* stream);
int dp_attention_fwd_supported(int T, int d, int dv);
/* ds = scale * p * (dp - sum_j p_j dp_j)   (SoftmaxBackward followed by the baddbmm alpha) */
int dp_softmax_bwd(const float* p, const float* dp, float* ds, long long rows, int cols, float scale, void* stream);

/* Sinusoidal timestep embedding, out[b][dim] (embeddings.py:22-62).  t given as float (timesteps.float()). */
int dp_timestep_embedding(const float* t, int B, int dim, int flip_sin_to_cos, float freq_shift, float max_period, float* out, void* stream);

/* noisy = sqrt(acp[t_b]) * x0 + sqrt(1 - acp[t_b]) * noise   (scheduling_ddpm.py:408-429) */
int dp_add_noise(const float* x0, const float* noise, const float* acp, const int64_t* t, int B, long long per_img, float* out, void* stream);

/* loss partial sums + dOut for the eps-prediction loss.
 * dout = gscale * (out - noise);  partial[block] = sum (out-noise)^2 over the block's slice.
 * F.mse_loss (ddpm_prune.py:101): gscale = 2/numel;  sum-CHW/mean-B loss (ddpm_train.py:459): gscale = 2/B. */
int dp_mse_fwd_bwd(const float* out, const float* noise, long long n, float gscale, float* dout, float* partial, int nblocks,
                   const float* stop_state, void* stream);
/* Diff-Pruning early exit kept on the device (ddpm_prune.py:104-106: `if loss > loss_max: loss_max = loss; if loss <
 * loss_max * thr: break`, fp32 as the reference's 0-d tensors).  state = [loss_max, stopped, executed steps] (zero-initialised),
 * losses[k] = loss of executed step k.  Once stopped, dp_mse_fwd_bwd(stop_state = state) produces dOut = 0, so timesteps the
 * host enqueued past the stop are exact no-ops on the accumulated gradients; the host polls `stopped` every few steps.
 * dp_zero_if_stopped cancels an already computed dOut (ddpm_exp flavour: threshold test before the backward pass). */
int dp_early_exit_update(const float* loss, float thr, float* state, float* losses, int max_steps, void* stream);
int dp_zero_if_stopped(float* x, long long n, const float* state, void* stream);
/* The same state machine in the LDM script's form (ldm_exp/prune_ldm.py:104,124-129: `max_loss = -1` -- the caller initialises
 * state[0] = -1 -- `if loss > max_loss: max_loss = loss; if loss / max_loss < thres: break`, the quotient rounded to fp32).
 * thr < 0 never stops (plain Taylor pass: losses are only recorded). */
int dp_early_exit_update_ratio(const float* loss, float thr, float* state, float* losses, int max_steps, void* stream);
/* dst[0] = scale * sum_i partial[i]  (single block, fixed order) */
int dp_sum_partials(const float* partial, int n, float scale, float* dst, void* stream);

/* dx[n][c][h][w] = sum of the 2x2 block of dy (backward of nearest x2 upsampling, resnet.py:155) */
int dp_downsum2x2(const float* dy, long long dy_img_stride, int N, int C, int H, int W, float* dx, long long dx_img_stride, void* stream);

/* y[n][c][2h+i][2w+j] = x[n][c][h][w]: nearest x2 upsampling, materialised (F.interpolate(scale_factor=2.0,
 * mode="nearest") in Upsample2D, diffusers/models/resnet.py:155; openaimodel.py Upsample).  The conv that follows is then a
 * plain stride-1 conv (dp_conv_gemm with ups = 0) instead of the gather form (ups = 1). */
int dp_upsample2x(const float* x, long long x_img_stride, int N, int C, int H, int W, float* y, long long y_img_stride, void* stream);

/* dx[n][c][2i+ph][2j+pw] = q[(2ph+pw)*q_class_stride + n*q_img_stride + (c*Ho+i)*Wo + j] (+ add[n][c][2i+ph][2j+pw]).
 * Assembles the input gradient of a stride-2 3x3 convolution (Downsample2D, diffusers/models/resnet.py:218; openaimodel.py
 * Downsample) from its four parity classes, each of which is a small stride-1 convolution over dy with the 4 / 2 / 2 / 1 taps
 * that can reach that parity (dp_conv_gemm): 9/4 taps per input pixel instead of the 9 of the zero-inserted form.
 * `add` (optional) is the skip-connection gradient the caller adds to dx (ConvolutionBackward + AddBackward in autograd). */
int dp_interleave2x2(const float* q, long long q_class_stride, long long q_img_stride, int N, int C, int Ho, int Wo,
                     const float* add, long long add_img_stride, float* dx, long long dx_img_stride, void* stream);

/* q[(2ph+pw)*q_class_stride + n*q_img_stride + (c*Ho+i)*Wo + j] = y[n][c][2i+ph][2j+pw]: inverse of dp_interleave2x2. */
int dp_deinterleave2x2(const float* y, long long y_img_stride, int N, int C, int Ho, int Wo, float* q,
                       long long q_class_stride, long long q_img_stride, void* stream);

/* Upsample2D = F.interpolate(nearest, x2) + Conv2d(3x3, pad 1) (diffusers/models/resnet.py:131-166; openaimodel.py Upsample)
 * as four 2x2 convolutions on the low-resolution input, one per parity class (ph, pw) of the output position, top / left
 * padding (1 - ph, 1 - pw); mathematically the same function with 16 instead of 36 multiply-adds per low-resolution pixel.
 * dp_ups_weff: weff[4][M][2][2] from w[M][3][3] (M = Cout*Cin), sums of the taps that read the same source pixel;
 * dp_ups_wfold: gw[M][3][3] (+)= the transposed map applied to the class weight gradients gweff[4][M][2][2]. */
int dp_ups_weff(const float* w, long long M, float* weff, void* stream);
int dp_ups_wfold(const float* gweff, long long M, float* gw, int accumulate, void* stream);

/* Taylor-importance reductions  (ddpm_exp/torch_pruning/importance.py:375-434).
 * Weight viewed as [R][C][T]; dim = 0: out[r] = sum_{c,t} f(w*g); dim = 1: out[c] = sum_{r,t} f(w*g);
 * mode 0: f = (w g)^2 (vendored), mode 1: f = |w g| (sum_abs), mode 2: signed sum then |.| (abs_sum),
 * mode 3: out[i] = |w_i g_i| (GroupNorm member; R = channels, C = T = 1),
 * mode 4: f = g^2 (FisherImportance, importance.py:715-781), mode 5: signed sum, no |.| (FullTaylorImportance,
 * importance.py:482-548: the absolute value is taken after the members are summed).
 * out[i] = (accumulate ? out[i] : 0) + value.  `scratch` (>= C*T floats) is required for dim == 1. */
int dp_wg_reduce(const float* w, const float* g, int R, int C, int T, int dim, int mode, float* out, int accumulate,
                 float* scratch, void* stream);

/* dst[i] += src[idx[i]], i < n : adds one member's channel sums into the group score (importance.py:427-428). */
int dp_gather_add(const float* src, const int64_t* idx, int n, float* dst, void* stream);

/* Batched forms for the prune tail (one launch pair per GROUP of coupled layers instead of two or three per member):
 * dp_group_score = the dp_wg_reduce of every member + the dp_axpby / dp_gather_add chain that folds them into the group
 * score (importance.py:375-434), same per-member arithmetic and member order, hence the same bits.  Member i views its
 * weight / gradient as [R][C][T]; dim / mode as in dp_wg_reduce (mode 3 = GroupNorm member, R channels); its per-channel
 * vector lives at scratch[full_off ...] (R or C floats), dim-1 members also need C*T floats at scratch[col_off ...];
 * idx_off >= 0: score[j] += vector[idx[idx_off + j]], idx_off < 0: identity.  blk0 is filled by the launcher.
 * dp_slice_batch = function.py:85-146,168-207,274-302 for every tensor of a group: dst = the kept channels (keep[keep_off ..
 * keep_off + n_keep), ascending) of src along `dim` of its [R][C][T] view.  blk0 / nblk are filled by the launcher. */
typedef struct dp_score_member {
    const float* w; const float* g;
    int R, C, T, dim, mode, blk0;
    long long full_off, col_off, idx_off;
} dp_score_member;
int dp_group_score(const dp_score_member* members, int n, int n0, const int64_t* idx, float* scratch, float* score, void* stream);
typedef struct dp_slice_item {
    const float* src; float* dst;
    int R, C, T, dim, n_keep, blk0, nblk, _pad;
    long long keep_off;
} dp_slice_item;
int dp_slice_batch(const dp_slice_item* items, int n, const int64_t* keep, void* stream);

/* Fused finetune update over flat buffers (ddpm_train.py:462-469, training_utils.py:201-216):
 *   g *= clip_coef (clip_coef read from device: min(1, max_norm/(norm+1e-6)));  Adam;  EMA with constant decay. */
int dp_sumsq_partials(const float* x, long long n, float* partial, int nblocks, void* stream);
/* The per-step scalars of a REPLAYED finetune step live on the device: hyper[0..2] = {lr, bc1 = 1 - b1^step, bc2 = 1 - b2^step}
 * (floats, the values the host computes for dp_adam_ema), hyper[3] = the optimizer step as an unsigned (the dropout masks'
 * `step`, read through dp_dropout.step_dev).  dp_set_step_scalars is the ONE launch per step that carries them by value;
 * dp_adam_ema_dev is dp_adam_ema reading {lr, bc1, bc2} from hyper.  (ddpm_train.py:462-469; LR schedule: optimization.py:282) */
int dp_set_step_scalars(float* hyper, float lr, float bc1, float bc2, unsigned step, void* stream);
int dp_adam_ema_dev(float* p, const float* g, float* m, float* v, float* ema, long long n, const float* clip_coef,
                    const float* hyper, float b1, float b2, float eps, float ema_decay, void* stream);
int dp_clip_coef(const float* partial, int n, float max_norm, float* norm_out, float* coef_out, void* stream);
int dp_adam_ema(float* p, const float* g, float* m, float* v, float* ema, long long n, const float* clip_coef,
                float lr, float b1, float b2, float eps, float bc1, float bc2, float ema_decay, void* stream);

/* DDIM update (scheduling_ddim.py:324-370, eta = 0 or with supplied noise):
 *   x0 = clamp((x - sqrt(1-a_t) eps)/sqrt(a_t), +-clip_range);  prev = sqrt(a_prev) x0 + sqrt(1-a_prev-std^2) eps (+ std*noise) */
int dp_ddim_step(const float* x, const float* eps, const float* vnoise, float a_t, float a_prev, float std, int clip,
                 float clip_range, float* out, long long n, void* stream);

/* DDPM ancestral update (scheduling_ddpm.py:312-406, epsilon prediction):
 *   x0 = clamp((x - sqrt_b_t eps)/sqrt_a_t, +-clip_range);  prev = c_x0 * x0 + c_xt * x (+ sigma * noise)
 * The 0-d coefficient arithmetic stays on the host in the reference's own fp32 operation order. */
int dp_ddpm_step(const float* x, const float* eps, const float* vnoise, float sqrt_a_t, float sqrt_b_t, float c_x0,
                 float c_xt, float sigma, int clip, float clip_range, float* out, long long n, void* stream);

/* ---- LDM (CompVis) transformer-block glue on channel-major tokens x[n][c][t]  (ldm_exp/ldm/modules/attention.py) ---- */
/* LayerNorm over the C channels of every token (attention.py:200-212 norm1/2/3); stats[(n*T+t)*2+{0,1}] = {mean, rstd}. */
int dp_layernorm_fwd(const float* x, long long x_img_stride, const float* gamma, const float* beta, int N, int C, int T,
                     float eps, float* y, long long y_img_stride, float* stats, void* stream);
/* dx = layernorm_backward(dy) (+ add);  pws[(n*C+c)*2+{0,1}] = {sum_t dy, sum_t dy*xhat} (reduce over n: dp_colsum_accum). */
int dp_layernorm_bwd(const float* x, long long x_img_stride, const float* gamma, const float* stats, const float* dy,
                     long long dy_img_stride, int N, int C, int T, float* dx, long long dx_img_stride, const float* add,
                     long long add_img_stride, float* pws, void* stream);
/* GEGLU (attention.py:37-46): in [N][2D][T] -> out [N][D][T] = in[:, :D] * gelu(in[:, D:]); half_plane = D*T. */
int dp_geglu_fwd(const float* in, int N, long long half_plane, float* out, void* stream);
int dp_geglu_bwd(const float* in, const float* dout, int N, long long half_plane, float* din, void* stream);
/* out[n][c][t] = x[n][c][t] + v[n*C + c]  (cross-attention over a single context token, attention.py:168-193) */
int dp_add_rowvec(const float* x, long long x_img_stride, const float* v, int N, int C, int T, float* out,
                  long long o_img_stride, void* stream);

/* q_sample with precomputed fp32 sqrt tables (ldm_exp/ldm/models/diffusion/ddpm.py q_sample / extract_into_tensor) */
int dp_q_sample(const float* x0, const float* noise, const float* sqrt_acp, const float* sqrt_1m_acp, const int64_t* t,
                int B, long long per_img, float* out, void* stream);
/* classifier-free guidance: out = e_uncond + scale * (e_cond - e_uncond)  (ldm_exp/ldm/models/diffusion/ddim.py:178-183) */
int dp_cfg_combine(const float* e_uncond, const float* e_cond, float scale, float* out, long long n, void* stream);

/* version / build info (smoke-tested by the CPU suite: library loads, symbols resolve) */
/* ---- FID / SSIM evaluation: fid_score.py:100-322, inception.py:16-340, compute_ssim.py:14-53 ---- */
/* k x k pooling, symmetric padding; mode 0: F.max_pool2d, mode 1: F.avg_pool2d(count_include_pad=False) (inception.py:238,266,300,333) */
int dp_pool2d(const float* x, long long x_img_stride, int N, int C, int H, int W, int k, int stride, int pad, int mode,
              float* y, long long y_img_stride, void* stream);
/* y = a * bilinear_resize(x -> Ho x Wo, align_corners = False) + b   (inception.py:147-154: 299 x 299, then 2x - 1) */
int dp_resize_bilinear(const float* x, long long x_img_stride, int N, int C, int H, int W, int Ho, int Wo, float a, float b,
                       float* y, void* stream);
/* out[n] = SSIM(x_n, y_n) as pytorch_msssim.ssim(x, y, data_range, size_average=False) (compute_ssim.py:43): 11-tap Gaussian,
 * sigma 1.5, valid filtering per channel, mean over the map, mean over channels.  x, y contiguous [N][C][H][W];
 * part: workspace of dp_ssim_workspace(N, C, H, W) floats.  dp_mse_per_image: compute_ssim.py:45. */
int dp_ssim(const float* x, const float* y, int N, int C, int H, int W, float data_range, float* part, float* out, void* stream);
long long dp_ssim_workspace(int N, int C, int H, int W);
int dp_mse_per_image(const float* a, const float* b, int N, long long per, float* out, void* stream);

/* Input pipeline (utils.py:8-58 get_dataset transforms; ddpm_exp/datasets/__init__.py:176-192 data_transform): decoded
 * uint8 images -> fp32 NCHW batch: x/255 (ToTensor), horizontal flip of image n with probability flip_thr24 / 2^24
 * (RandomHorizontalFlip; Philox decision on counter (n_off + n, 0, rng.site, rng.step), key rng.seed), mode 1:
 * (v - 0.5)/0.5 (Normalize(0.5, 0.5)), mode 2: 2v - 1 (rescaled), mode 0: none; dequant != 0: v/256*255 + u/256 first.
 * hwc != 0: src is [N][H][W][C], else [N][C][H][W].  rng: only seed / site / step / n_off are read. */
int dp_u8_to_float(const unsigned char* src, int hwc, int N, int C, int H, int W, float* out, long long out_img_stride,
                   int mode, unsigned flip_thr24, int dequant, const dp_dropout* rng, void* stream);

/* Native replay list (csrc/replay.hip): re-issue the kernels / memsets of a stream-captured step from a C loop.  The
 * reference's loop re-launches ~700 ATen kernels per timestep from Python (ddpm_prune.py:94-106); here one timestep is captured
 * once into a hipGraph (never instantiated: hipGraphLaunch of these graphs costs more host time than eager launches on this
 * stack), its nodes are read back, list-scheduled onto two streams along the captured dependency edges, and
 * dp_replay_launch re-issues them with hipLaunchKernel.  `graph`: hipGraph_t; it must outlive the list.
 * info[8] = nodes, kernels, memsets, memcpys, empty nodes, cross-stream waits, nodes on the side stream, events. */
int dp_replay_build(void* graph, void** out_handle);
int dp_replay_launch(void* handle, void* main_stream, void* side_stream);
int dp_replay_info(void* handle, int* info8);
int dp_replay_free(void* handle);

/* dp_pack_weight for many layers in one launch (blk0 / nblk are filled by the launcher).  mode 0 / 1 as dp_pack_weight;
 * mode 2 / 3: the dp_pack_weight_wino operand (mode - 2) of a 3x3 weight (taps = 9, dst holds 12 * K * ld floats);
 * mode 4 / 5: the dp_pack_weight_wino2d operand (mode - 4) of a 3x3 weight (taps = 9, dst holds 16 * K * ld floats). */
typedef struct dp_pack_item {
    const float* W; float* dst;
    int Co, Ci, taps, mode, ld, blk0, nblk, _pad;
} dp_pack_item;
int dp_pack_weight_batch(const dp_pack_item* items, int n, void* stream);

int dp_version(void);
/* Number of kernel launches this library has issued since it was loaded (host counter; bench.py reports launches per step).
 * Returned in place of an error code. */
long long dp_launch_count(void);

#ifdef __cplusplus
}
#endif
#endif
