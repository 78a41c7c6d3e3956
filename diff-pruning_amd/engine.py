"""Forward + hand-written backward of UNet2DModel on the HIP kernels (no autograd, no ATen compute).

The engine is *functional* over a flat {diffusers_state_dict_key: device tensor} dictionary -- channel
counts are read off the tensors, so the same code runs un-pruned and pruned networks -- and mirrors
`UNet2DModel.forward` of the reference (diffusers/models/unet_2d.py:219-316, resnet.py:589-639,
attention_processor.py:415-470).  Gradients are *accumulated* (+=) straight into caller-provided
buffers (normally `param.grad`), which is what the Taylor sweep needs (ddpm_prune.py:102: grads are
never zeroed between timesteps), fusing the reference's separate `add_` pass into the wgrad epilogue.

Layout: activations are channel-major planes [N, C, H, W] with a free image stride.  `torch.cat`
(unet_2d_blocks.py:2035) is never materialised: GroupNorm and the 1x1 shortcut read both halves in
place (two-source gather), and the concat gradient is handed to its two producers as channel-slice views.
"""
import os

import torch

from . import ops

_SPEC3 = ops.ConvSpec(3, 1, 1, 0)
_SPEC1 = ops.ConvSpec(1, 1, 0, 0)
_SPEC_UP = ops.ConvSpec(3, 1, 1, 1)
UPS_COPY = not os.environ.get('DP_NO_UPS_COPY')
UPS_SUBPIXEL = not os.environ.get('DP_NO_UPS_SUBPIXEL')      # upsample convolutions as four 2x2 convolutions at low resolution
_UPS_SPECS = ops.UPS_CLASS_SPECS
S2_PARITY = not os.environ.get('DP_NO_S2_PARITY')
FUSE_QKV_WGRAD = not os.environ.get('DP_NO_FUSED_QKV_WGRAD')      # to_q / to_k / to_v weight gradients as one M = 3C contraction
OVERLAP_MIN_WORK = 4_000_000     # images x pixels x base width from which the weight-gradient side stream pays (UNetEngine.__init__)

# parameter-name suffixes of a residual block: Diffusers ResnetBlock2D / CompVis ResBlock (openaimodel.py:163-275)
RES_DIFFUSERS = dict(norm1='.norm1', conv1='.conv1', temb='.time_emb_proj', norm2='.norm2', conv2='.conv2',
                     shortcut='.conv_shortcut', dropout='.dropout')
RES_LDM = dict(norm1='.in_layers.0', conv1='.in_layers.2', temb='.emb_layers.1', norm2='.out_layers.0',
               conv2='.out_layers.3', shortcut='.skip_connection')


# most bytes of side-stream operands kept alive between two joins of the weight-gradient stream (UNetEngine._side_stream)
SIDE_RECORD_STREAM = os.environ.get('DP_SIDE_RECORD_STREAM') == '1'
SIDE_KEEP_MAX_BYTES = int(float(os.environ.get('DP_SIDE_KEEP_GB', '4')) * (1 << 30))


_stream_cache = {}


def shared_stream(device, role, slot=0, make=None):
    """ONE stream per (device, role, slot) for the life of the process.  Every engine used to create its own streams: the caching
    allocator keeps a separate pool of cached blocks per stream, so each new sweep / engine object left ~11 GB (CIFAR UNet,
    batch 256) of blocks behind that no later stream could reuse (round 6: reserved memory grew by that much per sweep of a
    process), and the raw HIP streams of the weight-gradient side were never destroyed.  Objects alive at the same time that
    share a stream are merely ordered on it; each forks and joins with wait_stream on both sides as before.
    role: 'wgrad' (lowest priority), 'pipeline' (a second timestep / half-batch pipeline), 'replay' (native replay's side)."""
    device = torch.device(device)
    key = (device.index if device.index is not None else torch.cuda.current_device(), role, int(slot))
    st = _stream_cache.get(key)
    if st is None:
        st = _stream_cache[key] = (make or (lambda d: torch.cuda.Stream(device=d)))(device)
    return st


def _low_priority_stream(device):
    """Side stream for the weight-gradient work at the LOWEST HIP priority: the backward chain on the main stream gets
    the CUs first, the side stream fills what is left.  torch only exposes priorities <= 0, so the stream is created
    through the HIP runtime and wrapped; any failure falls back to a plain torch stream."""
    if os.environ.get('DP_SIDE_PRIORITY', '1') == '0':
        return torch.cuda.Stream(device=device)
    try:
        import ctypes as C
        hip = C.CDLL('libamdhip64.so')
        least, greatest = C.c_int(0), C.c_int(0)
        if hip.hipDeviceGetStreamPriorityRange(C.byref(least), C.byref(greatest)) != 0:
            raise OSError('hipDeviceGetStreamPriorityRange')
        h = C.c_void_p()
        with torch.cuda.device(device):
            if hip.hipStreamCreateWithPriority(C.byref(h), C.c_uint(1), least) != 0:      # 1 = hipStreamNonBlocking
                raise OSError('hipStreamCreateWithPriority')
        return torch.cuda.ExternalStream(h.value, device=device)
    except (OSError, AttributeError, RuntimeError):
        return torch.cuda.Stream(device=device)


class _Packs:
    """Cache of packed (m-contiguous) weight operands, keyed by the identity + version of the source tensor.

    Neither key detects an in-place write through `.data` (`param.data.copy_()` -- how the vendored EMAModel.copy_to /
    restore swap weights, training_utils.py:231,286 -- leaves `_version` untouched), so the cache is only trusted while
    it is *pinned*: `model.engine()` drops it on every re-bind unless a caller that knows the weights are frozen (the
    sweep, a sampling loop) holds `model.pin_weights()`."""

    def __init__(self):
        self._c = {}
        self.pin_depth = 0
        self.lazy = {}                # (name, ('up', class, 0 | 1)) / (prefix, 'qkv') asked for so far: prepare_packs() batches them

    def get(self, name, w, mode):
        key = (name, mode)
        tag = (w.data_ptr(), w._version, tuple(w.shape))
        hit = self._c.get(key)
        if hit is not None and hit[0] == tag:
            return hit[1], hit[2]
        if isinstance(mode, tuple) and mode[0] == 'wino43':  # ('wino43', 0): Winograd F(4, 3) operand of the no-grad forwards
            buf, ld = ops.pack_weight_wino43(w)
        elif isinstance(mode, tuple) and mode[0] == 'wino':    # ('wino', 0 | 1): Winograd F(2, 3) operand (ops.pack_weight_wino)
            buf, ld = ops.pack_weight_wino(w, mode[1])
        elif isinstance(mode, tuple) and mode[0] == 'wino2d':  # ('wino2d', 0 | 1): Winograd F(2x2, 3x3) operand (ops.pack_weight_wino2d)
            buf, ld = ops.pack_weight_wino2d(w, mode[1])
        elif isinstance(mode, tuple) and mode[0] == 'up':      # ('up', class, 0 | 1): class kernel of an upsample convolution
            self.lazy[(name, mode)] = True
            weff = self.get_weff(name, w)
            buf, ld = ops.pack_weight(weff[mode[1]], mode[2])
        elif isinstance(mode, tuple):             # ('s2', ph, pw, pad): one parity class of a stride-2 dgrad (ops.conv_dgrad_s2)
            buf, ld = ops.pack_weight_s2(w, mode[1], mode[2], mode[3])
        else:
            buf, ld = ops.pack_weight(w, mode)
        self._c[key] = (tag, buf, ld)
        return buf, ld

    def has(self, name, w, mode):
        hit = self._c.get((name, mode))
        return hit is not None and hit[0] == (w.data_ptr(), w._version, tuple(w.shape))

    def put(self, name, w, mode, buf, ld):
        self._c[(name, mode)] = ((w.data_ptr(), w._version, tuple(w.shape)), buf, ld)

    def get_weff(self, name, w):
        """[4, Cout, Cin, 2, 2] class kernels of the upsample convolution `name` (ops.ups_weff), cached like the packs."""
        key = (name, 'weff')
        tag = (w.data_ptr(), w._version, tuple(w.shape))
        hit = self._c.get(key)
        if hit is not None and hit[0] == tag:
            return hit[1]
        weff = ops.ups_weff(w)
        self._c[key] = (tag, weff, 0)
        return weff

    def clear(self):
        self._c.clear()

    def rebind(self):
        """Called by `model.engine()`: weights may have been rewritten in place since the last call."""
        if self.pin_depth == 0:
            self._c.clear()


class _PinnedWeights:
    """`with model.pin_weights():` -- the parameters are not written inside the block, packed operands are kept
    across `model(...)` calls (one pack per layer instead of one per forward)."""

    def __init__(self, model):
        self.model = model

    def __enter__(self):
        eng = self.model.engine()                 # re-bind (and re-pack lazily) once, against the current weights
        eng.packs.pin_depth += 1
        self.eng = eng
        return self.model

    def __exit__(self, *exc):
        self.eng.packs.pin_depth -= 1
        return False


class UNetEngine:
    def __init__(self, cfg):
        self.cfg = dict(cfg)
        self.packs = _Packs()
        self.P = None      # name -> parameter tensor (device)
        self.G = None      # name -> gradient accumulation buffer (device), same shapes
        self.ctx = None
        # Weight / bias gradients do not feed the backward chain: they run on a second HIP stream, so their MFMA work
        # fills the CUs while the main stream is in the HBM-bound GroupNorm / reduction kernels and in the launch
        # ramp / tail of its contraction kernels.  Same kernels, same accumulation order -> same bits.
        # [measured, round 3] every fork / join of the two streams costs tens of microseconds of cross-queue synchronisation on
        # this stack (~80 per timestep): CIFAR UNet batch 4: 15.4 ms per timestep with the side stream, 11.2 without; batch 16:
        # 16.1 / 12.0; batch 64: 24.4 / 25.4; LDM UNet, 6 latents: 51.1 / 55.8.  overlap_wgrad = None (default) decides per
        # backward pass from the work of the step (images x pixels x base width >= OVERLAP_MIN_WORK); True / False force it.
        self._nograd = False
        self.overlap_wgrad = False if os.environ.get('DP_NO_OVERLAP') else (True if os.environ.get('DP_OVERLAP') else None)
        self._overlap_now = True
        self._side, self._side_dev = None, None
        self.stream_slot = 0                 # engines that run concurrently (timestep pipelines) use different slots (shared_stream)
        self._side_keep = []                 # tensors the side stream reads, alive until the next join (_side_stream)
        self._side_keep_bytes = 0
        # Dropout (training mode only; utils.set_dropout, ddpm_train.py:380-382): {module name: p} of the nn.Dropout
        # holders with p > 0, or None.  Masks are Philox functions of (seed, crc32(module name), step, element index):
        # the backward regenerates them, nothing is stored (csrc/dp_common.h).
        self.dropout = None
        self.drop_seed, self.drop_step, self.drop_n_off = 0, 0, 0
        # bias / GroupNorm-parameter gradient sums are queued during the backward pass and reduced together at its end
        self.defer_colsum = not os.environ.get('DP_NO_COLSUM_BATCH')
        self._cq = None
        # the GroupNorm backward kernels also emit the per-(image, channel) sums of their output: the bias and time-embedding
        # projection gradients of the layer below need exactly those, and a separate row-sum pass would re-read the tensor
        self.fuse_rows = not os.environ.get('DP_NO_FUSED_ROWS')
        self._rows_src = None
        self._temb_rows = None
        self.segment_hook = None
        # Time-embedding projections batched over the network (Diffusers UNet only): the 22 `time_emb_proj` Linear layers all read
        # the same silu(emb), and their input gradient is only needed once the whole backward pass is over -- so forward is ONE
        # GEMM against the concatenated weights ([sum C, 512]; every ResnetBlock2D takes its column slice as the conv epilogue's
        # per-(image, channel) addend), and backward is two GEMMs at the end (d silu(emb) = R_all W_all, dW_all = R_all^T silu(emb))
        # over the concatenated row sums instead of 22 x (dgrad + wgrad + split-K reduce) latency-bound launches.
        self.temb_batch = not os.environ.get('DP_NO_TEMB_BATCH')
        self._temb = None
        # Attention q / k / v projections as ONE 1x1 convolution over the concatenated weights (single-head blocks): the
        # normalised input is read once instead of three times, and the three input gradients that used to accumulate into one
        # buffer (3 x read + 2 x read-modify-write) are one K = 3C contraction.  q, k, v are channel slices of the result.
        self.fuse_qkv = not os.environ.get('DP_NO_FUSED_QKV')

    # ------------------------------------------------------------------------------------------
    def bind(self, params, grads=None):
        self.P = params
        self.G = grads

    def set_dropout(self, table, seed=0, step=0, n_off=0, step_dev=None):
        """step_dev: data_ptr of a device uint32 the kernels read instead of `step` (train.FinetuneEngine's replayed step)."""
        self.dropout = table if table else None
        self.drop_seed, self.drop_step, self.drop_n_off = seed, step, n_off
        self.drop_step_dev = step_dev

    def _drop(self, site):
        if not self.dropout:
            return None
        sd = getattr(self, 'drop_step_dev', None)
        if sd is not None:
            return ops.dropout_desc(self.dropout.get(site, 0.0), self.drop_seed, site, self.drop_step, self.drop_n_off, step_dev=sd)
        return ops.dropout_desc(self.dropout.get(site, 0.0), self.drop_seed, site, self.drop_step, self.drop_n_off)

    def _colsum(self, ws, N, C, wstride, woff, out):
        """out[c] += sum_n ws[(n*C + c)*wstride + woff] -- now, or with everything else at the end of backward().
        ws may be a column slice [N, C] of a wider row-major matrix (wstride == 1)."""
        ld = 0
        if ws.dim() == 2 and wstride == 1 and ws.stride(0) != C:
            if self._cq is None:
                ws = ws.contiguous()
            else:
                ld = ws.stride(0)
        if self._cq is not None:
            self._cq.add(ws, N, C, wstride, woff, out, True, ld)
        else:
            ops.colsum_accum(ws, N, C, wstride, woff, out, True)

    def _rows_of(self, t):
        """Per-(image, channel) sums of gradient tensor `t` if the GroupNorm backward that produced it (or the wider tensor `t`
        is the leading channel slice of) also emitted them; else one dp_rowsum_nc pass."""
        src = self._rows_src
        if src is not None:
            full, rows = src
            if t.data_ptr() == full.data_ptr() and t.shape[0] == full.shape[0] and t.shape[2:] == full.shape[2:] \
                    and t.stride() == full.stride() and t.shape[1] <= full.shape[1]:
                return rows if t.shape[1] == full.shape[1] else rows[:, :t.shape[1]]
        return ops.rowsum_nc(t)

    def _gn_bwd(self, x, x2, gamma, beta, stats, dz, G, silu, **kw):
        """groupnorm_bwd that remembers the row sums of its output for the next layer's bias / temb gradients."""
        if self.fuse_rows and hasattr(ops, 'ColsumQueue'):
            dx, pws, rows = ops.groupnorm_bwd(x, x2, gamma, beta, stats, dz, G, silu, want_rows=True, **kw)
            self._rows_src = (dx, rows) if rows is not None else None
        else:
            dx, pws = ops.groupnorm_bwd(x, x2, gamma, beta, stats, dz, G, silu, **kw)
            self._rows_src = None
        return dx, pws

    def decide_overlap(self, x):
        """Called by forward(save=True) with the network input: whether this step's weight gradients go to the side stream."""
        if self.overlap_wgrad is None:
            width = self.cfg.get('block_out_channels', [self.cfg.get('model_channels', 128)])[0]
            self._overlap_now = x.shape[0] * x.shape[2] * x.shape[3] * width >= OVERLAP_MIN_WORK
        else:
            self._overlap_now = bool(self.overlap_wgrad)

    def _begin_backward(self):
        self._cq = ops.ColsumQueue() if (self.defer_colsum and hasattr(ops, 'ColsumQueue')) else None

    def segment_done(self, segment):
        """backward() reached the end of a segment ('up': output head + up blocks, 'mid', 'down'): all of that segment's
        parameter gradients are final once the weight-gradient stream is joined and the queued sums are flushed.  The
        data-parallel finetune step hangs its bucketed gradient all-reduce here (train.FinetuneEngine)."""
        if self.segment_hook is None:
            return
        self._join_side()
        if self._cq is not None:
            self._cq.flush()
        self.segment_hook(segment)

    def _end_backward(self):
        """Join the weight-gradient stream, then flush the queued sums (their sources were produced on either stream)."""
        self._join_side()
        if self._cq is not None:
            self._cq.flush()
            self._cq = None

    _temb_suffix = '.time_emb_proj'

    def _resnet_prefixes(self):
        cfg = self.cfg
        Lr, nb = cfg['layers_per_block'], len(cfg['block_out_channels'])
        out = ['down_blocks.%d.resnets.%d' % (i, j) for i in range(nb) for j in range(Lr)]
        out += ['mid_block.resnets.0', 'mid_block.resnets.1']
        out += ['up_blocks.%d.resnets.%d' % (i, j) for i in range(nb) for j in range(Lr + 1)]
        return out

    def _temb_pack(self):
        """(prefixes, {prefix: (offset, C)}, W_all [sum C, tdim], b_all [sum C]) -- cached with the packed conv operands."""
        hit = self.packs._c.get(('__temb_all__', 0))
        if hit is not None:
            return hit
        names = self._resnet_prefixes()
        sfx = self._temb_suffix
        ws = [self.P[n + sfx + '.weight'] for n in names]
        offs, o = {}, 0
        for n, w in zip(names, ws):
            offs[n] = (o, w.shape[0])
            o += w.shape[0]
        val = (names, offs, torch.cat(ws, 0).contiguous(), torch.cat([self.P[n + sfx + '.bias'] for n in names], 0).contiguous())
        self.packs._c[('__temb_all__', 0)] = val
        return val

    def _qkv_pack(self, pre):
        """Packed operands of cat(to_q, to_k, to_v): (fwd pack, ld, bias, dgrad pack, ld, (cq, ck, cv)); cached with the packs."""
        key = (pre + '.__qkv__', 0)
        hit = self.packs._c.get(key)
        if hit is None:
            self.packs.lazy[(pre, 'qkv')] = True
            w, b, sizes = self._qkv_cat(pre)
            wp, ld = ops.pack_weight(w, 0)
            wd, ldd = ops.pack_weight(w, 1)
            hit = (wp, ld, b, wd, ldd, sizes)
            self.packs._c[key] = hit
        return hit

    def _qkv_cat(self, pre):
        ws = [self.P[pre + n + '.weight'] for n in ('.to_q', '.to_k', '.to_v')]
        w = torch.cat(ws, 0).contiguous()
        b = (torch.cat([self.P[pre + n + '.bias'] for n in ('.to_q', '.to_k', '.to_v')], 0).contiguous()
             if (pre + '.to_q.bias') in self.P else None)            # the LDM transformer's projections are bias-free
        return w, b, tuple(x.shape[0] for x in ws)

    def prepare_packs(self):
        """Pack every conv / linear weight in both operand layouts now (needed before hipGraph capture: packing
        must not be recorded into the replayed graph; and once per finetune step, whose optimizer update invalidates every
        pack): all of them in a few batched launches (ops.pack_weight_batch)."""
        todo = []
        # Winograd operands the LAST pass (or the one in progress) asked for -- decided by activation shape at launch time.  Layers
        # that stopped qualifying (a prune left 90 channels, the batch or resolution changed) age out instead of being re-packed
        # for ever  [advisor, round 4]
        gen = getattr(self, '_wino_gen', 0)
        seen = getattr(self, '_wino_seen', None) or {}
        for k in [k for k, g in seen.items() if g < gen - 1]:
            del seen[k]
        for name, w in self.P.items():
            if name.endswith('.weight') and w.dim() >= 2:
                for mode in (0, 1, ('wino', 0), ('wino', 1), ('wino2d', 0), ('wino2d', 1)):
                    if isinstance(mode, tuple) and (name[:-7], mode) not in seen:
                        continue
                    if not self.packs.has(name[:-7], w, mode):
                        if isinstance(mode, tuple) and mode[0] == 'wino2d' and not getattr(ops, 'PACK_BATCH_WINO2D', False):
                            self.packs.get(name[:-7], w, mode)                     # (a packer without modes 4 / 5: one launch each)
                        else:
                            todo.append((name[:-7], w, mode))
        # operands that are packed from a DERIVED tensor and were asked for in an earlier pass: the four class kernels of an upsample
        # convolution (both layouts) and the concatenated q | k | v projections of an attention block
        derived = []                                  # (finish(buf, ld), source tensor, mode 0 | 1)
        if hasattr(ops, 'pack_weight_batch') and getattr(ops, 'PACK_BATCH_DERIVED', True):
            qkv_parts = {}
            for (name, what) in list(self.packs.lazy):
                if what == 'qkv':
                    key = (name + '.__qkv__', 0)
                    if key in self.packs._c or (name + '.to_q.weight') not in self.P:
                        continue
                    w, b, sizes = self._qkv_cat(name)
                    qkv_parts[key] = [None, None, b, None, None, sizes]
                    for m in (0, 1):
                        derived.append((lambda buf, ld, key=key, m=m: qkv_parts[key].__setitem__(slice(3 * m, 3 * m + 2), [buf, ld]), w, m))
                else:
                    w = self.P.get(name + '.weight')
                    if w is None or self.packs.has(name, w, what):
                        continue
                    weff = self.packs.get_weff(name, w)
                    derived.append((lambda buf, ld, name=name, w=w, what=what: self.packs.put(name, w, what, buf, ld), weff[what[1]], what[2]))
        if hasattr(ops, 'pack_weight_batch') and (todo or derived) and all(w.is_contiguous() for _, w, _ in todo):
            packed = ops.pack_weight_batch([(w, mode) for _, w, mode in todo] + [(w, m) for _, w, m in derived])
            for (name, w, mode), (buf, ld) in zip(todo, packed):
                self.packs.put(name, w, mode, buf, ld)
            for (finish, _, _), (buf, ld) in zip(derived, packed[len(todo):]):
                finish(buf, ld)
            if derived:
                for key, parts in qkv_parts.items():
                    self.packs._c[key] = tuple(parts)
        else:
            for name, w, mode in todo:
                self.packs.get(name, w, mode)

    def attn_scale(self, channels):
        hd = self.cfg.get('attention_head_dim')
        return float(hd if hd is not None else channels) ** -0.5

    # ---- primitive layers ---------------------------------------------------------------------
    def _wino_pack(self, name, w, mode, kind='wino'):
        if not hasattr(self, '_wino_seen'):
            self._wino_seen = {}
        self._wino_seen[(name, (kind, mode))] = getattr(self, '_wino_gen', 0)     # prepare_packs() batches it from the next pass on
        return self.packs.get(name, w, (kind, mode))

    def _wino_operand(self, name, w, mode, M, C_sources, N, H, W, spec):
        """The Winograd operand of a 3x3 / stride 1 / pad 1 layer for ops.conv_forward / conv_dgrad (`wino=`), or None: the
        two-dimensional F(2x2, 3x3) form (csrc/winograd2d.hip: 4/9 of the direct multiplies) where the shape and the grid suit it,
        else the one-dimensional F(2, 3) form (csrc/winograd.hip: 2/3), else the direct kernel."""
        if not hasattr(ops, 'wino_wanted'):
            return None
        if hasattr(ops, 'wino2d_wanted') and ops.wino2d_wanted(M, C_sources, N, H, W, spec):
            return ('2d',) + tuple(self._wino_pack(name, w, mode, 'wino2d'))
        if ops.wino_wanted(M, C_sources, N, H, W, spec):
            return self._wino_pack(name, w, mode)
        return None

    def _conv(self, name, x, x2, spec, **kw):
        w = self.P[name + '.weight']
        wp, ld = self.packs.get(name, w, 0)
        # 3x3 / stride 1 / pad 1 layers with a grid worth it: Winograd F(2, 3) along W, 2/3 of the multiplies (csrc/winograd.hip)
        if w.dim() == 4 and w.shape[2] == 3 and hasattr(ops, 'wino_wanted'):
            cs = (x.shape[1],) + ((x2.shape[1],) if x2 is not None else ())
            wo = self._wino_operand(name, w, 0, w.shape[0], cs, x.shape[0], x.shape[2], x.shape[3], spec)
            if wo is not None:
                kw['wino'] = wo
            # a forward that keeps nothing for a backward (sampling loops, the LDM importance pass's CFG sampler): F(4, 3), half the
            # multiplies at ~1e-6 instead of ~3e-7 fp32 error -- never for a scored forward (csrc/winograd43.hip)
            if self._nograd and hasattr(ops, 'wino43_wanted') and ops.wino43_wanted(w.shape[0], cs, x.shape[0], x.shape[2], x.shape[3], spec):
                kw['wino43'] = self.packs.get(name, w, ('wino43', 0))
        return ops.conv_forward(x, x2, wp, ld, w.shape[0], spec, bias=self.P.get(name + '.bias'), **kw)

    def _linear(self, name, x2d):
        return ops.linear_forward(x2d, self.P[name + '.weight'], self.P.get(name + '.bias'))

    def _linear_bwd(self, name, dy2d, x2d, *, need_dx=True, dx_out=None, dx_accumulate=False):
        """Accumulate weight / bias gradients of nn.Linear `name`; return (or accumulate) the input gradient."""
        w = self.P[name + '.weight']
        ops.linear_wgrad(dy2d, x2d, self.G[name + '.weight'], accumulate=True)
        if (name + '.bias') in self.P:
            self._colsum(dy2d, dy2d.shape[0], dy2d.shape[1], 1, 0, self.G[name + '.bias'])
        if not need_dx:
            return None
        return ops.linear_dgrad(dy2d, w, out=dx_out, accumulate=dx_accumulate)

    def _side_stream(self, *tensors):
        """Fork: returns the side stream (ordered after everything enqueued so far on the current stream) or None.
        `tensors` are read by the side-stream work: the allocator must not recycle them before that work is done.  They are
        kept ALIVE (a reference each) until the next join instead of being handed to `Tensor.record_stream`: a recorded block is
        only reusable once an event on the side stream has been seen complete by a later allocation, so with the host a few
        timesteps ahead of the device every block of those timesteps is still pending and the allocator reserves new memory
        instead -- [measured, round 6, CIFAR UNet batch 256, tools/mem_probe.py] 124 GB reserved after 300 timesteps for a 10 GB
        working set, the whole 288 GB within a 1000-timestep sweep, after which the HIP runtime's own allocations (scratch,
        kernel arguments, "svm hidden buffer") start to fail: the abort of the round-5 driver run.  A block released AFTER the join
        is reused by main-stream work that is ordered behind the join, i.e. behind its last side-stream reader."""
        if not self._overlap_now or not hasattr(torch.cuda, 'current_stream') or tensors[0].device.type != 'cuda':
            return None
        if self._side is None or self._side_dev != tensors[0].device:        # the model may have moved to another GPU
            self._join_side()
            self._side = shared_stream(tensors[0].device, 'wgrad', self.stream_slot, _low_priority_stream)
            self._side_dev = tensors[0].device
        if self._side_keep_bytes > SIDE_KEEP_MAX_BYTES:
            self._join_side()                    # bounds what a backward pass keeps alive beyond its own needs; same kernels and chains
        self._side.wait_stream(torch.cuda.current_stream())
        for t in tensors:
            if t is None:
                continue
            if SIDE_RECORD_STREAM:               # the round-5 form, kept ONLY so that tools/abort_repro.py can show the failure it caused
                t.record_stream(self._side)
                continue
            self._side_keep.append(t)
            self._side_keep_bytes += t.numel() * t.element_size()
        return self._side

    def replay_side_stream(self, device):
        """Second stream for the native replay of a captured step (ops.ReplayList): the forked chains of the capture alternate
        between the two replay streams, so it has the same priority as the main one."""
        if getattr(self, '_replay_side', None) is None or self._replay_side.device != torch.device(device):
            self._replay_side = shared_stream(device, 'replay', self.stream_slot)
        return self._replay_side

    def _join_side(self):
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)
        self._side_keep.clear()                  # (after the join: see _side_stream)
        self._side_keep_bytes = 0

    def _conv_bwd(self, name, dy, x, x2, spec, in_hw, *, need_dx=True, rows=None, dx_out=None, dx_accumulate=False,
                  alpha=1.0, dx_add=None):
        """Accumulate weight / bias gradients of conv `name`; return gradient w.r.t. its (virtual) input (+ dx_add)."""
        w = self.P[name + '.weight']

        if rows is None and (name + '.bias') in self.P and self._rows_src is not None and self._rows_src[0].data_ptr() == dy.data_ptr():
            rows = self._rows_of(dy)               # dy came out of a GroupNorm backward that already summed it

        def param_grads(rows):
            ops.conv_wgrad(dy, x, x2, self.G[name + '.weight'], spec, alpha=alpha, accumulate=True)
            if (name + '.bias') in self.P:
                if rows is None:
                    rows = ops.rowsum_nc(dy)
                if alpha != 1.0:
                    rows = rows.contiguous()
                    rows = ops.axpby(rows, alpha, torch.empty_like(rows), 0.0)
                self._colsum(rows, rows.shape[0], rows.shape[1], 1, 0, self.G[name + '.bias'])

        side = self._side_stream(dy, x, x2, rows)
        if side is None:
            param_grads(rows)
        else:
            with torch.cuda.stream(side):
                param_grads(rows)
        if not need_dx:
            return None
        if (S2_PARITY and spec.stride == 2 and spec.k == 3 and not spec.ups and spec.pad in (0, 1) and alpha == 1.0
                and dx_out is None and in_hw[0] == 2 * dy.shape[2] and in_hw[1] == 2 * dy.shape[3]):
            packs = [self.packs.get(name, w, ('s2', ph, pw, spec.pad)) for ph in (0, 1) for pw in (0, 1)]
            return ops.conv_dgrad_s2(dy, packs, w.shape[1], spec, in_hw, add=dx_add)
        wd, ldd = self.packs.get(name, w, 1)
        wino = None
        if w.dim() == 4 and w.shape[2] == 3 and hasattr(ops, 'wino_wanted') and tuple(in_hw) == tuple(dy.shape[2:]):
            wino = self._wino_operand(name, w, 1, w.shape[1], (w.shape[0],), dy.shape[0], dy.shape[2], dy.shape[3], spec)
        dx = ops.conv_dgrad(dy, wd, ldd, w.shape[1], spec, in_hw, alpha=alpha, out=dx_out, accumulate=dx_accumulate,
                            **({'wino': wino} if wino is not None else {}))
        if dx_add is not None:
            ops.copy_strided(dx_add, dx, accumulate=True)
        return dx

    # ---- Upsample2D: nearest x2 + conv3x3 (resnet.py:131-166) as four 2x2 convolutions on the low-resolution input --------
    # One per parity class of the output position, with class kernels that are sums of the 3x3 taps reading the same source
    # pixel (csrc/elementwise.hip, dp_ups_weff): the same function with 16 instead of 36 multiply-adds per low-resolution
    # pixel and channel pair, in the forward pass, the input gradient and the weight gradient alike.
    def _ups_conv_fwd(self, name, x):
        w = self.P[name + '.weight']
        N, _, H, W = x.shape
        q = ops.empty_act((4, N, w.shape[0], H, W), x.device)
        for c, spec in enumerate(_UPS_SPECS):
            wp, ld = self.packs.get(name, w, ('up', c, 0))
            ops.conv_forward(x, None, wp, ld, w.shape[0], spec, bias=self.P.get(name + '.bias'), out=q[c])
        return ops.interleave2x2(q)

    def _ups_conv_bwd(self, name, dy, x):
        """Accumulate the weight / bias gradients of the upsample convolution `name`; return the gradient w.r.t. its
        low-resolution input x (what Upsample2D's autograd returns after the 2x2 sum of UpsampleNearest2DBackward)."""
        w = self.P[name + '.weight']
        dyq = ops.deinterleave2x2(dy)
        rows = None
        if (name + '.bias') in self.P and self._rows_src is not None and self._rows_src[0].data_ptr() == dy.data_ptr():
            rows = self._rows_of(dy)

        def param_grads(rows):
            gweff = torch.empty((4,) + tuple(w.shape[:2]) + (2, 2), dtype=torch.float32, device=dy.device)
            for c, spec in enumerate(_UPS_SPECS):
                ops.conv_wgrad(dyq[c], x, None, gweff[c], spec, accumulate=False)
            ops.ups_wfold(gweff, self.G[name + '.weight'], accumulate=True)
            if (name + '.bias') in self.P:
                if rows is None:
                    rows = ops.rowsum_nc(dy)
                self._colsum(rows, rows.shape[0], rows.shape[1], 1, 0, self.G[name + '.bias'])

        side = self._side_stream(dyq, dy, x, rows)
        if side is None:
            param_grads(rows)
        else:
            with torch.cuda.stream(side):
                param_grads(rows)
        dx = None
        for c, spec in enumerate(_UPS_SPECS):
            wd, ldd = self.packs.get(name, w, ('up', c, 1))
            dx = ops.conv_dgrad(dyq[c], wd, ldd, w.shape[1], spec, tuple(x.shape[2:]), out=dx, accumulate=c > 0)
        return dx

    def _gn_param_grads(self, name, pws):
        N, C = pws.shape[0], pws.shape[1]
        self._colsum(pws, N, C, 2, 1, self.G[name + '.weight'])
        self._colsum(pws, N, C, 2, 0, self.G[name + '.bias'])

    # ---- ResnetBlock2D (resnet.py:589-639) -----------------------------------------------------
    def resnet_fwd(self, pre, xa, xb, semb, out_scale, save, names=RES_DIFFUSERS, G=None, eps=None):
        P, cfg = self.P, self.cfg
        G = cfg['norm_num_groups'] if G is None else G
        eps = cfg['norm_eps'] if eps is None else eps
        nm = names
        n1, st1 = ops.groupnorm_fwd(xa, xb, P[pre + nm['norm1'] + '.weight'], P[pre + nm['norm1'] + '.bias'], G, eps, True)
        if self._temb is not None:
            o, c = self._temb[1][pre]
            tproj = self._temb[2][:xa.shape[0], o:o + c]   # column slice of the batched projection (row stride = sum C); the
                                                           # rows of this block's images (LdmEngine: shared CFG stem)
        else:
            tproj = self._linear(pre + nm['temb'], semb)
        h = self._conv(pre + nm['conv1'], n1, None, _SPEC3, tadd=tproj)
        # resnet.py:622-630: norm2 -> SiLU -> dropout -> conv2; the dropout is fused into the GroupNorm kernel's store
        drop = self._drop(pre + nm['dropout']) if 'dropout' in nm else None
        n2, st2 = ops.groupnorm_fwd(h, None, P[pre + nm['norm2'] + '.weight'], P[pre + nm['norm2'] + '.bias'], G, eps, True,
                                    drop=drop)
        has_sc = (pre + nm['shortcut'] + '.weight') in P
        if has_sc:
            res = self._conv(pre + nm['shortcut'], xa, xb, _SPEC1)
        else:
            if xb is not None:       # identity shortcut over a concat: materialise it (not reached by UNet2DModel configs)
                res = torch.cat([xa, xb], 1)
            else:
                res = xa
        out = self._conv(pre + nm['conv2'], n2, None, _SPEC3, res=res, post_scale=1.0 / out_scale)
        if save is not None:
            save[pre] = (xa, xb, st1, n1, h, st2, n2, has_sc, out_scale, nm, G, drop)
        return out

    def resnet_bwd(self, pre, dout, semb, d_semb, extra=None):
        """Returns d(input) over the (virtually concatenated) input channels."""
        xa, xb, st1, n1, h, st2, n2, has_sc, out_scale, nm, G, drop = self.ctx.pop(pre)
        P = self.P
        hw = tuple(h.shape[2:])
        d = dout
        if out_scale != 1.0:
            d = ops.axpby(dout.contiguous(), 1.0 / out_scale, torch.empty_like(dout, memory_format=torch.contiguous_format), 0.0)
        rows_d = self._rows_of(d)
        dn2 = self._conv_bwd(pre + nm['conv2'], d, n2, None, _SPEC3, hw, rows=rows_d)
        dh, pws2 = self._gn_bwd(h, None, P[pre + nm['norm2'] + '.weight'], P[pre + nm['norm2'] + '.bias'], st2, dn2, G, True,
                                drop=drop)
        self._gn_param_grads(pre + nm['norm2'], pws2)
        del dn2
        # time-embedding projection: d tproj[n, c] = sum_hw dh  (also conv1's bias-gradient rows)
        rows_h = self._rows_of(dh)
        if self._temb_rows is not None:
            self._temb_rows[pre] = rows_h              # batched at the end of backward()
        else:
            self._linear_bwd(pre + nm['temb'], rows_h, semb, dx_out=d_semb, dx_accumulate=True)
        dn1 = self._conv_bwd(pre + nm['conv1'], dh, n1, None, _SPEC3, hw, rows=rows_h)
        del dh
        if has_sc:
            add1 = self._conv_bwd(pre + nm['shortcut'], d, xa, xb, _SPEC1, hw, rows=rows_d)
        else:
            add1 = d
        dx, pws1 = self._gn_bwd(xa, xb, P[pre + nm['norm1'] + '.weight'], P[pre + nm['norm1'] + '.bias'], st1, dn1, G,
                                True, add1=add1, add2=extra)
        self._gn_param_grads(pre + nm['norm1'], pws1)
        return dx

    # ---- Attention (attention_processor.py:415-470, heads == 1) ---------------------------------
    def attn_heads(self, channels):
        """Heads of an attention block over `channels` (un-pruned width; fixed at construction, unet_2d_blocks.py:722-723)."""
        hd = self.cfg.get('attention_head_dim')
        return channels // hd if hd is not None else 1

    def attn_fwd(self, pre, x, scale, rescale, save, heads=1):
        P, cfg = self.P, self.cfg
        G, eps = cfg['norm_num_groups'], cfg['norm_eps']
        N, C, H, W = x.shape
        T = H * W
        n, st = ops.groupnorm_fwd(x, None, P[pre + '.group_norm.weight'], P[pre + '.group_norm.bias'], G, eps, False)
        fused = self.fuse_qkv and heads == 1 and hasattr(ops, 'empty_act') and (pre + '.to_q.bias') in P
        if fused:
            wp, ld, b_cat, _, _, (cq, ck, cv) = self._qkv_pack(pre)
            qkv = ops.conv_forward(n, None, wp, ld, cq + ck + cv, _SPEC1, bias=b_cat)
            q, k, v = qkv[:, :cq], qkv[:, cq:cq + ck], qkv[:, cq + ck:]
        else:
            q = self._conv(pre + '.to_q', n, None, _SPEC1)
            k = self._conv(pre + '.to_k', n, None, _SPEC1)
            v = self._conv(pre + '.to_v', n, None, _SPEC1)
        inner = q.shape[1]
        # heads: channel-major tokens make head_to_batch_dim (attention_processor.py:283-305) a view: head h owns the
        # contiguous channel rows [h*d, (h+1)*d) of every image -> batch index n*heads + h
        Z, d = N * heads, inner // heads
        vd = v.shape[1] // heads                      # the value width may differ from the query / key width after pruning
        if save is None and getattr(ops, 'FUSED_ATTN', False) and ops.attention_fused_ok(T, d, vd):
            p = None                                  # sampling forward: one kernel, no [T, T] scores (csrc/attention.hip)
            o = ops.attention_fwd(q, k, v, heads, scale)
        else:
            s = ops.bmm_tn(q.view(Z, d, T), k.view(Z, d, T), alpha=scale)
            p = ops.softmax_fwd(s, out=s)
            o = ops.bmm_nt(v.view(Z, vd, T), p)
        drop = self._drop(pre + '.to_out.1')
        if drop is None:
            out = self._conv(pre + '.to_out.0', o.view(N, vd * heads, H, W), None, _SPEC1, res=x, post_scale=1.0 / rescale)
        else:
            # attention_processor.py:455-466: to_out[0] -> to_out[1] (dropout) -> + residual -> / rescale_output_factor
            out = self._conv(pre + '.to_out.0', o.view(N, vd * heads, H, W), None, _SPEC1)
            ops.dropout_apply(out, drop, out=out)
            ops.copy_strided(x, out, accumulate=True)
            if rescale != 1.0:
                ops.axpby(out, 1.0 / rescale, out, 0.0)
        if save is not None:
            save[pre] = (x, st, n, q, k, v, p, o, scale, rescale, heads, drop, fused)
        return out

    def _qkv_param_grads(self, pre, d_qkv, n, widths):
        """Weight / bias gradients of to_q, to_k, to_v in ONE contraction (round 5): the three projections read the same normalised
        input n, and their output gradients are the channel slices of d_qkv -- so dW_cat[3C, C] = d_qkv (x) n is one M = 3C launch
        (a third of the split-K partial traffic per weight, three times the K range per workgroup: the M = 256 1x1 weight gradients
        ran at 62 TFLOP/s, 20 launches per timestep) whose row blocks are then added to the three gradients; one row-sum pass over
        d_qkv gives the three bias gradients.  Same products, the pixel sum re-associated by the different split-K partition."""
        names = ('.to_q', '.to_k', '.to_v')
        has_bias = (pre + '.to_q.bias') in self.P

        def work():
            gcat = torch.empty((sum(widths), n.shape[1]), dtype=torch.float32, device=n.device)
            ops.conv_wgrad(d_qkv, n, None, gcat, _SPEC1, accumulate=False)
            rows = ops.rowsum_nc(d_qkv) if has_bias else None
            o = 0
            for name, c in zip(names, widths):
                gw = self.G[pre + name + '.weight']
                ops.axpby(gcat[o:o + c].reshape(-1), 1.0, gw.view(-1), 1.0)
                if has_bias:
                    self._colsum(rows[:, o:o + c], rows.shape[0], c, 1, 0, self.G[pre + name + '.bias'])
                o += c

        side = self._side_stream(d_qkv, n)
        if side is None:
            work()
        else:
            with torch.cuda.stream(side):
                work()

    def attn_bwd(self, pre, dout, extra=None):
        x, st, n, q, k, v, p, o, scale, rescale, heads, drop, fused = self.ctx.pop(pre)
        P, cfg = self.P, self.cfg
        G = cfg['norm_num_groups']
        N, C, H, W = x.shape
        T = H * W
        inner = q.shape[1]
        hw = (H, W)
        d = dout
        if rescale != 1.0:
            d = ops.axpby(dout.contiguous(), 1.0 / rescale, torch.empty_like(dout, memory_format=torch.contiguous_format), 0.0)
        dproj = d if drop is None else ops.dropout_apply(d.contiguous(), drop)
        do = self._conv_bwd(pre + '.to_out.0', dproj, o.view(N, -1, H, W), None, _SPEC1, hw)
        Z, dh = N * heads, inner // heads
        do3 = do.view(Z, do.shape[1] // heads, T)
        if fused:
            # dq | dk | dv written straight into channel slices of one buffer; one K = 3C input-gradient contraction
            _, _, _, wd, ldd, (cq, ck, cv) = self._qkv_pack(pre)
            d_qkv = ops.empty_act((N, cq + ck + cv, H, W), x.device)
            sl = (d_qkv[:, :cq], d_qkv[:, cq:cq + ck], d_qkv[:, cq + ck:])
            ops.bmm_nn(do3, p, out=sl[2].view(N, cv, T))
            dp = ops.bmm_tn(do3, v.view(N, cv, T))
            ds = ops.softmax_bwd(p, dp, scale, out=dp)
            ops.bmm_nt(k.view(N, ck, T), ds, out=sl[0].view(N, cq, T))
            ops.bmm_nn(q.view(N, cq, T), ds, out=sl[1].view(N, ck, T))
            if FUSE_QKV_WGRAD:
                self._qkv_param_grads(pre, d_qkv, n, (cq, ck, cv))
            else:
                for dproj, name in zip(sl, ('.to_q', '.to_k', '.to_v')):
                    self._conv_bwd(pre + name, dproj, n, None, _SPEC1, hw, need_dx=False)       # weight / bias gradients only
            dn = ops.conv_dgrad(d_qkv, wd, ldd, n.shape[1], _SPEC1, hw)
        else:
            dv = ops.bmm_nn(do3, p)
            dp = ops.bmm_tn(do3, v.view(Z, dh, T))
            ds = ops.softmax_bwd(p, dp, scale, out=dp)
            dq = ops.bmm_nt(k.view(Z, dh, T), ds)
            dk = ops.bmm_nn(q.view(Z, dh, T), ds)
            dn = torch.empty_like(n)
            first = True
            for dproj, name in ((dq, '.to_q'), (dk, '.to_k'), (dv, '.to_v')):
                self._conv_bwd(pre + name, dproj.view(N, inner, H, W), n, None, _SPEC1, hw, dx_out=dn, dx_accumulate=not first)
                first = False
        dx, pws = self._gn_bwd(x, None, P[pre + '.group_norm.weight'], P[pre + '.group_norm.bias'], st, dn, G, False,
                               add1=d, add2=extra)
        self._gn_param_grads(pre + '.group_norm', pws)
        return dx

    # ---- whole network ------------------------------------------------------------------------
    def forward(self, sample, timesteps, save=False):
        """sample [B, Cin, H, W] fp32 device tensor, timesteps [B] (int64 or float) device tensor."""
        P, cfg = self.P, self.cfg
        boc = list(cfg['block_out_channels'])
        Lr = cfg['layers_per_block']
        nb = len(boc)
        ctx = {} if save else None
        self._nograd = not save                                  # F(4, 3) convolutions only where nothing is kept for a backward
        self._wino_gen = getattr(self, '_wino_gen', 0) + 1       # one generation per forward pass (see prepare_packs)
        if save:
            self.decide_overlap(sample)
        if cfg.get('center_input_sample', False):
            sample = 2 * sample - 1.0
        sample = sample.contiguous()
        t_emb = ops.timestep_embedding(timesteps.to(torch.float32), boc[0], cfg['flip_sin_to_cos'], cfg['freq_shift'])
        h1 = self._linear('time_embedding.linear_1', t_emb)
        a1 = ops.silu_fwd(h1)
        emb = self._linear('time_embedding.linear_2', a1)
        semb = ops.silu_fwd(emb)
        self._temb = None
        if self.temb_batch and 'down_blocks.0.resnets.0.time_emb_proj.weight' in P:
            names, offs, W_all, b_all = self._temb_pack()
            self._temb = (names, offs, ops.linear_forward(semb, W_all, b_all), W_all)
        x = self._conv('conv_in', sample, None, _SPEC3)
        skips = [x]
        for i, bt in enumerate(cfg['down_block_types']):
            pre = 'down_blocks.%d' % i
            for j in range(Lr):
                x = self.resnet_fwd('%s.resnets.%d' % (pre, j), x, None, semb, 1.0, ctx)
                if bt == 'AttnDownBlock2D':
                    x = self.attn_fwd('%s.attentions.%d' % (pre, j), x, self.attn_scale(boc[i]), 1.0, ctx, self.attn_heads(boc[i]))
                skips.append(x)
            if i != nb - 1:
                spec = ops.ConvSpec(3, 2, cfg['downsample_padding'], 0)
                xin = x
                x = self._conv(pre + '.downsamplers.0.conv', xin, None, spec)
                if ctx is not None:
                    ctx[pre + '.down'] = (xin, spec)
                skips.append(x)
        msf = float(cfg.get('mid_block_scale_factor', 1))
        x = self.resnet_fwd('mid_block.resnets.0', x, None, semb, msf, ctx)
        if cfg.get('add_attention', True):
            x = self.attn_fwd('mid_block.attentions.0', x, self.attn_scale(boc[-1]), msf, ctx, self.attn_heads(boc[-1]))
        x = self.resnet_fwd('mid_block.resnets.1', x, None, semb, msf, ctx)
        rev = list(reversed(boc))
        n_skips = len(skips)
        for i, bt in enumerate(cfg['up_block_types']):
            pre = 'up_blocks.%d' % i
            for j in range(Lr + 1):
                skip = skips.pop()
                x = self.resnet_fwd('%s.resnets.%d' % (pre, j), x, skip, semb, 1.0, ctx)
                if bt == 'AttnUpBlock2D':
                    x = self.attn_fwd('%s.attentions.%d' % (pre, j), x, self.attn_scale(rev[i]), 1.0, ctx, self.attn_heads(rev[i]))
            if i != nb - 1:
                # the x2-upsampled tensor is materialised (4x a low-resolution activation, HBM-bound) so that the convolution
                # and its weight gradient are plain stride-1 launches on the LDS-DMA kernels (see ops.upsample2x)
                if UPS_SUBPIXEL:
                    xin = x
                    x = self._ups_conv_fwd(pre + '.upsamplers.0.conv', xin)
                else:
                    xin = ops.upsample2x(x) if UPS_COPY else x
                    x = self._conv(pre + '.upsamplers.0.conv', xin, None, _SPEC3 if UPS_COPY else _SPEC_UP)
                if ctx is not None:
                    ctx[pre + '.up'] = xin
        G, eps = cfg['norm_num_groups'], cfg['norm_eps']
        xo = x
        no, sto = ops.groupnorm_fwd(xo, None, P['conv_norm_out.weight'], P['conv_norm_out.bias'], G, eps, True)
        out = self._conv('conv_out', no, None, _SPEC3)
        if ctx is not None:
            ctx['_head'] = (sample, t_emb, h1, a1, emb, semb, xo, no, sto, n_skips)
            ctx['_temb'] = None if self._temb is None else (self._temb[0], self._temb[1], self._temb[3])
            self.ctx = ctx
        self._temb = None
        return out

    def backward(self, dout):
        """Accumulate d(loss)/d(param) into self.G given d(loss)/d(output).  Consumes the saved context."""
        P, cfg, ctx = self.P, self.cfg, self.ctx
        assert ctx is not None, 'forward(save=True) must precede backward()'
        boc = list(cfg['block_out_channels'])
        Lr = cfg['layers_per_block']
        nb = len(boc)
        G = cfg['norm_num_groups']
        sample, t_emb, h1, a1, emb, semb, xo, no, sto, n_skips = ctx.pop('_head')
        temb = ctx.pop('_temb', None)
        self._temb_rows = {} if temb is not None else None
        self._begin_backward()
        d_semb = torch.zeros_like(semb)
        hw = tuple(xo.shape[2:])
        dno = self._conv_bwd('conv_out', dout, no, None, _SPEC3, hw)
        dx, pws = ops.groupnorm_bwd(xo, None, P['conv_norm_out.weight'], P['conv_norm_out.bias'], sto, dno, G, True)
        self._gn_param_grads('conv_norm_out', pws)
        del dno
        skip_grads = []          # filled in pop order: skip_grads[k] is the gradient of skips[n_skips-1-k]
        for i in reversed(range(nb)):
            bt = cfg['up_block_types'][i]
            pre = 'up_blocks.%d' % i
            if i != nb - 1:
                xin = ctx.pop(pre + '.up')
                if UPS_SUBPIXEL:
                    dx = self._ups_conv_bwd(pre + '.upsamplers.0.conv', dx, xin)
                    dxv = None
                elif UPS_COPY:
                    dxv = self._conv_bwd(pre + '.upsamplers.0.conv', dx, xin, None, _SPEC3, (xin.shape[2], xin.shape[3]))
                else:
                    dxv = self._conv_bwd(pre + '.upsamplers.0.conv', dx, xin, None, _SPEC_UP,
                                         (2 * xin.shape[2], 2 * xin.shape[3]))
                if dxv is not None:
                    dx = ops.downsum2x2(dxv)
                del dxv
            local = []
            for j in reversed(range(Lr + 1)):
                if bt == 'AttnUpBlock2D':
                    dx = self.attn_bwd('%s.attentions.%d' % (pre, j), dx)
                rp = '%s.resnets.%d' % (pre, j)
                c1 = self.ctx[rp][0].shape[1]
                dcat = self.resnet_bwd(rp, dx, semb, d_semb)
                dx = dcat[:, :c1]
                local.append(dcat[:, c1:])
            skip_grads.append(local)
        # skip k (forward push order) was popped by up block i at resnet j; rebuild index -> gradient view
        order = []
        for i in range(nb):          # forward pop order: up block 0 first, resnet 0 first
            blk = skip_grads[nb - 1 - i]           # skip_grads was filled for i = nb-1 .. 0
            for j in range(Lr + 1):
                order.append(blk[Lr - j])          # local was filled for j = Lr .. 0
        sg = {n_skips - 1 - k: g for k, g in enumerate(order)}     # skips index -> grad view

        self.segment_done('up')
        msf = float(cfg.get('mid_block_scale_factor', 1))
        dx = self.resnet_bwd('mid_block.resnets.1', dx, semb, d_semb)
        if cfg.get('add_attention', True):
            dx = self.attn_bwd('mid_block.attentions.0', dx)
        # mid resnet 0 consumes skips[-1] (the last down output) together with the up path
        idx = n_skips - 1
        dx = self.resnet_bwd('mid_block.resnets.0', dx, semb, d_semb, extra=sg.pop(idx))
        self.segment_done('mid')
        for i in reversed(range(nb)):
            bt = cfg['down_block_types'][i]
            pre = 'down_blocks.%d' % i
            if i != nb - 1:
                # dx is the full gradient of the downsampler output (skips[idx]); its input is skips[idx-1]
                xin, spec = ctx.pop(pre + '.down')
                idx -= 1
                dx = self._conv_bwd(pre + '.downsamplers.0.conv', dx, xin, None, spec, tuple(xin.shape[2:]), dx_add=sg.pop(idx))
            for j in reversed(range(Lr)):
                # dx = full gradient of skips[idx] (output of resnet j / its attention)
                if bt == 'AttnDownBlock2D':
                    dx = self.attn_bwd('%s.attentions.%d' % (pre, j), dx)
                idx -= 1
                dx = self.resnet_bwd('%s.resnets.%d' % (pre, j), dx, semb, d_semb, extra=sg.pop(idx))
        assert idx == 0 and not sg
        self.segment_done('down')
        self._conv_bwd('conv_in', dx, sample, None, _SPEC3, None, need_dx=False)
        if temb is not None:
            # all time_emb_proj layers at once: rows of every ResnetBlock2D side by side, two GEMMs, gradients scattered back
            names, offs, W_all = temb
            rows = self._temb_rows
            self._temb_rows = None
            R_all = torch.cat([rows[n] for n in names], 1)
            d_semb = ops.linear_dgrad(R_all, W_all)
            dW_all = torch.empty_like(W_all)
            ops.linear_wgrad(R_all, semb, dW_all, accumulate=False)
            for n in names:
                o, c = offs[n]
                gw = self.G[n + '.time_emb_proj.weight']
                ops.axpby(dW_all[o:o + c].view(-1), 1.0, gw.view(-1), 1.0)
                self._colsum(rows[n], rows[n].shape[0], c, 1, 0, self.G[n + '.time_emb_proj.bias'])
        # time embedding MLP (embeddings.py:200-212)
        d_emb = ops.silu_bwd(emb, d_semb)
        d_a1 = self._linear_bwd('time_embedding.linear_2', d_emb, a1)
        d_h1 = ops.silu_bwd(h1, d_a1)
        self._linear_bwd('time_embedding.linear_1', d_h1, t_emb, need_dx=False)
        self._end_backward()
        assert not ctx, 'unconsumed context: %s' % list(ctx)
        self.ctx = None
