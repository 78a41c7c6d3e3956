"""ctypes binding of libdp_hip.so (the C-ABI declared in include/dp_hip.h).

There is NO fallback: if the shared library is missing or a symbol does not resolve, importing the
compute path raises.  Device pointers are taken from torch tensors (`data_ptr()`), the stream from
`torch.cuda.current_stream()`; torch is plumbing (memory + streams), every FLOP runs in the HIP kernels.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('DP_HIP_LIB') or os.path.join(_HERE, 'libdp_hip.so')   # DP_HIP_LIB: kernel A/B experiments

c_float_p = C.c_void_p
LL = C.c_longlong


class ConvGeom(C.Structure):
    _fields_ = [(n, C.c_int) for n in ('Ho', 'Wo', 'Hs', 'Ws', 'Hv', 'Wv', 'kw', 'stride', 'sden', 'pad_t', 'pad_l',
                                       'ups', 'c_split', '_pad')] + [('x1_img_stride', LL), ('x2_img_stride', LL)]


class ConvGemmParams(C.Structure):
    _fields_ = [('A', C.c_void_p), ('a_bs', LL), ('lda', C.c_int), ('a_kc', C.c_int),
                ('X1', C.c_void_p), ('X2', C.c_void_p), ('x_bs', LL),
                ('a_bytes', C.c_uint), ('x1_bytes', C.c_uint), ('x2_bytes', C.c_uint), ('_pad0', C.c_uint),
                ('g', ConvGeom),
                ('M', C.c_int), ('C', C.c_int), ('NPIX', C.c_int), ('ntaps', C.c_int), ('batches', C.c_int),
                ('tile', C.c_int),
                ('out', C.c_void_p), ('o_img_stride', LL), ('o_bs', LL),
                ('alpha', C.c_float), ('post_scale', C.c_float),
                ('bias', C.c_void_p), ('tadd', C.c_void_p), ('tadd_stride', LL),
                ('res', C.c_void_p), ('r_img_stride', LL),
                ('accumulate', C.c_int), ('ksplit', C.c_int), ('ws', C.c_void_p), ('x_guard', C.c_int), ('act', C.c_int),
                ('tile_counters', C.c_void_p)]


class NtGemmParams(C.Structure):
    _fields_ = [('A', C.c_void_p), ('a_bs', LL), ('a_img_stride', LL),
                ('X1', C.c_void_p), ('X2', C.c_void_p), ('x_bs', LL),
                ('a_bytes', C.c_uint), ('x1_bytes', C.c_uint), ('x2_bytes', C.c_uint), ('_pad0', C.c_uint),
                ('g', ConvGeom),
                ('M', C.c_int), ('C', C.c_int), ('NCOLS', C.c_int), ('ntaps', C.c_int), ('P', C.c_int),
                ('batches', C.c_int), ('splits', C.c_int), ('p_per_split', C.c_int), ('tile', C.c_int),
                ('batched', C.c_int),
                ('out', C.c_void_p), ('o_bs', LL), ('ldo', C.c_int), ('accumulate', C.c_int),
                ('alpha', C.c_float), ('merge', C.c_int), ('ocs', LL),
                ('o_tap_stride', LL), ('o_col_stride', C.c_int), ('xcd', C.c_int), ('col_bias', C.c_void_p)]


class AttentionParams(C.Structure):
    _fields_ = [('q', C.c_void_p), ('k', C.c_void_p), ('v', C.c_void_p), ('o', C.c_void_p),
                ('q_bs', LL), ('k_bs', LL), ('v_bs', LL), ('o_bs', LL),
                ('N', C.c_int), ('heads', C.c_int), ('d', C.c_int), ('dv', C.c_int), ('T', C.c_int), ('scale', C.c_float),
                ('variant', C.c_int), ('_pad', C.c_int)]


class Dropout(C.Structure):
    _fields_ = [('thr24', C.c_uint), ('scale', C.c_float), ('seed', C.c_ulonglong), ('site', C.c_uint), ('step', C.c_uint),
                ('n_off', LL), ('step_dev', C.c_void_p)]


class ScoreMember(C.Structure):
    _fields_ = [('w', C.c_void_p), ('g', C.c_void_p), ('R', C.c_int), ('C', C.c_int), ('T', C.c_int), ('dim', C.c_int),
                ('mode', C.c_int), ('blk0', C.c_int), ('full_off', LL), ('col_off', LL), ('idx_off', LL)]


class SliceItem(C.Structure):
    _fields_ = [('src', C.c_void_p), ('dst', C.c_void_p), ('R', C.c_int), ('C', C.c_int), ('T', C.c_int), ('dim', C.c_int),
                ('n_keep', C.c_int), ('blk0', C.c_int), ('nblk', C.c_int), ('_pad', C.c_int), ('keep_off', LL)]


class PackItem(C.Structure):
    _fields_ = [('W', C.c_void_p), ('dst', C.c_void_p), ('Co', C.c_int), ('Ci', C.c_int), ('taps', C.c_int), ('mode', C.c_int),
                ('ld', C.c_int), ('blk0', C.c_int), ('nblk', C.c_int), ('_pad', C.c_int)]


class ColsumItem(C.Structure):
    _fields_ = [('src', C.c_void_p), ('dst', C.c_void_p), ('N', C.c_int), ('C', C.c_int), ('wstride', C.c_int),
                ('woff', C.c_int), ('accumulate', C.c_int), ('ld', C.c_int)]


_vp, _i, _f, _ll = C.c_void_p, C.c_int, C.c_float, LL
_dr = C.POINTER(Dropout)

# name -> argtypes (restype is always int); this table is also what the CPU test-suite checks against
# include/dp_hip.h (every declared symbol must resolve).
SIGNATURES = {
    'dp_conv_gemm': [C.POINTER(ConvGemmParams), _vp],
    'dp_conv_wino': [C.POINTER(ConvGemmParams), _vp],
    'dp_conv_wino_supported': [C.POINTER(ConvGemmParams)],
    'dp_conv_splitk_epilogue': [C.POINTER(ConvGemmParams), _vp],
    'dp_pack_weight_wino': [_vp, _i, _i, _i, _vp, _i, _vp],
    'dp_conv_wino2d': [C.POINTER(ConvGemmParams), _vp],
    'dp_conv_wino2d_supported': [C.POINTER(ConvGemmParams)],
    'dp_pack_weight_wino2d': [_vp, _i, _i, _i, _vp, _i, _vp],
    'dp_conv_wino43': [C.POINTER(ConvGemmParams), _vp],
    'dp_conv_wino43_supported': [C.POINTER(ConvGemmParams)],
    'dp_pack_weight_wino43': [_vp, _i, _i, _vp, _i, _vp],
    'dp_wgrad_wino': [C.POINTER(NtGemmParams), _vp],
    'dp_wgrad_wino2d': [C.POINTER(NtGemmParams), _vp],
    'dp_wgrad_wino2d_supported': [C.POINTER(NtGemmParams)],
    'dp_wgrad_wino_supported': [C.POINTER(NtGemmParams)],
    'dp_nt_gemm': [C.POINTER(NtGemmParams), _vp],
    'dp_splitk_reduce': [_vp, _ll, _i, _vp, _ll, _i, _vp],
    'dp_splitk_reduce_taps': [_vp, _ll, _i, _vp, _ll, _i, _i, _vp],
    'dp_pack_weight': [_vp, _i, _i, _i, _i, _vp, _i, _vp],
    'dp_pack_weight_batch': [C.POINTER(PackItem), _i, _vp],
    'dp_groupnorm_silu_fwd': [_vp, _vp, _i, _ll, _ll, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp, _ll, _vp, _dr, _vp],
    'dp_groupnorm_silu_bwd': [_vp, _vp, _i, _ll, _ll, _vp, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _vp, _ll, _vp, _ll,
                              _vp, _ll, _vp, _dr, _vp, _vp],
    'dp_groupnorm_silu_fwd_split': [_vp, _vp, _i, _ll, _ll, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp, _ll, _vp, _i, _vp, _dr,
                                    _vp],
    'dp_groupnorm_silu_bwd_split': [_vp, _vp, _i, _ll, _ll, _vp, _vp, _vp, _vp, _ll, _i, _i, _i, _i, _i, _vp, _ll, _vp, _ll,
                                    _vp, _ll, _vp, _i, _vp, _dr, _vp],
    'dp_colsum_accum': [_vp, _i, _i, _i, _i, _vp, _i, _vp],
    'dp_colsum_accum_batch': [C.POINTER(ColsumItem), _i, _vp],
    'dp_rowsum_nc': [_vp, _ll, _i, _i, _i, _vp, _vp],
    'dp_silu_fwd': [_vp, _vp, _ll, _vp],
    'dp_silu_bwd': [_vp, _vp, _vp, _ll, _i, _vp],
    'dp_axpby': [_vp, _f, _vp, _f, _ll, _vp],
    'dp_copy_strided': [_vp, _ll, _vp, _ll, _i, _ll, _i, _vp],
    'dp_attention_fwd': [C.POINTER(AttentionParams), _vp],
    'dp_attention_fwd_supported': [_i, _i, _i],
    'dp_softmax_fwd': [_vp, _vp, _ll, _i, _vp],
    'dp_softmax_bwd': [_vp, _vp, _vp, _ll, _i, _f, _vp],
    'dp_timestep_embedding': [_vp, _i, _i, _i, _f, _f, _vp, _vp],
    'dp_add_noise': [_vp, _vp, _vp, _vp, _i, _ll, _vp, _vp],
    'dp_mse_fwd_bwd': [_vp, _vp, _ll, _f, _vp, _vp, _i, _vp, _vp],
    'dp_early_exit_update': [_vp, _f, _vp, _vp, _i, _vp],
    'dp_zero_if_stopped': [_vp, _ll, _vp, _vp],
    'dp_early_exit_update_ratio': [_vp, _f, _vp, _vp, _i, _vp],
    'dp_randn_philox': [_vp, _ll, _ll, C.c_ulonglong, C.c_uint, C.c_uint, _vp],
    'dp_sum_partials': [_vp, _i, _f, _vp, _vp],
    'dp_downsum2x2': [_vp, _ll, _i, _i, _i, _i, _vp, _ll, _vp],
    'dp_upsample2x': [_vp, _ll, _i, _i, _i, _i, _vp, _ll, _vp],
    'dp_interleave2x2': [_vp, _ll, _ll, _i, _i, _i, _i, _vp, _ll, _vp, _ll, _vp],
    'dp_deinterleave2x2': [_vp, _ll, _i, _i, _i, _i, _vp, _ll, _ll, _vp],
    'dp_ups_weff': [_vp, _ll, _vp, _vp],
    'dp_ups_wfold': [_vp, _ll, _vp, _i, _vp],
    'dp_wg_reduce': [_vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp],
    'dp_gather_add': [_vp, _vp, _i, _vp, _vp],
    'dp_group_score': [C.POINTER(ScoreMember), _i, _i, _vp, _vp, _vp, _vp],
    'dp_slice_batch': [C.POINTER(SliceItem), _i, _vp, _vp],
    'dp_sumsq_partials': [_vp, _ll, _vp, _i, _vp],
    'dp_clip_coef': [_vp, _i, _f, _vp, _vp, _vp],
    'dp_adam_ema': [_vp, _vp, _vp, _vp, _vp, _ll, _vp, _f, _f, _f, _f, _f, _f, _f, _vp],
    'dp_set_step_scalars': [_vp, _f, _f, _f, C.c_uint, _vp],
    'dp_adam_ema_dev': [_vp, _vp, _vp, _vp, _vp, _ll, _vp, _vp, _f, _f, _f, _f, _vp],
    'dp_ddim_step': [_vp, _vp, _vp, _f, _f, _f, _i, _f, _vp, _ll, _vp],
    'dp_ddpm_step': [_vp, _vp, _vp, _f, _f, _f, _f, _f, _i, _f, _vp, _ll, _vp],
    'dp_dropout_apply': [_vp, _ll, _vp, _ll, _i, _ll, _dr, _vp],
    'dp_dropout_mask': [_vp, _ll, _ll, _dr, _vp],
    'dp_layernorm_fwd': [_vp, _ll, _vp, _vp, _i, _i, _i, _f, _vp, _ll, _vp, _vp],
    'dp_layernorm_bwd': [_vp, _ll, _vp, _vp, _vp, _ll, _i, _i, _i, _vp, _ll, _vp, _ll, _vp, _vp],
    'dp_geglu_fwd': [_vp, _i, _ll, _vp, _vp],
    'dp_geglu_bwd': [_vp, _vp, _i, _ll, _vp, _vp],
    'dp_add_rowvec': [_vp, _ll, _vp, _i, _i, _i, _vp, _ll, _vp],
    'dp_q_sample': [_vp, _vp, _vp, _vp, _vp, _i, _ll, _vp, _vp],
    'dp_cfg_combine': [_vp, _vp, _f, _vp, _ll, _vp],
    'dp_u8_to_float': [_vp, _i, _i, _i, _i, _i, _vp, _ll, _i, C.c_uint, _i, _dr, _vp],
    'dp_pool2d': [_vp, _ll, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _ll, _vp],
    'dp_resize_bilinear': [_vp, _ll, _i, _i, _i, _i, _i, _i, _f, _f, _vp, _vp],
    'dp_ssim': [_vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp],
    'dp_ssim_workspace': [_i, _i, _i, _i],
    'dp_mse_per_image': [_vp, _vp, _i, _ll, _vp, _vp],
    'dp_replay_build': [_vp, C.POINTER(C.c_void_p)],
    'dp_replay_launch': [_vp, _vp, _vp],
    'dp_replay_info': [_vp, C.POINTER(C.c_int)],
    'dp_replay_free': [_vp],
    'dp_version': [],
    'dp_launch_count': [],
}

_lib = None


class DpHipError(RuntimeError):
    code = None              # the hipError_t a library call returned (None: the library itself is missing)


def load():
    """Load libdp_hip.so and bind every symbol.  Raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DpHipError('libdp_hip.so not found at %s -- run ./build.sh (or __graft_entry__.build()); '
                         'there is no CPU / PyTorch fallback for the hot path' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = C.c_longlong if name in ('dp_launch_count', 'dp_ssim_workspace') else C.c_int
    _lib = lib
    return lib


def check(err, what):
    if err != 0:
        e = DpHipError('%s failed with hipError %d' % (what, err))
        e.code = int(err)
        raise e
