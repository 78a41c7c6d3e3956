"""UNet2DModel with the Diffusers interface, executed by the HIP engine.

Mirrors the public surface of the reference's `diffusers.models.UNet2DModel` (unet_2d.py:38-316) that the
hot path touches: constructor arguments / `.config`, `forward(sample, timestep, class_labels=None,
return_dict=True) -> UNet2DOutput`, `.dtype`, `.device`, and -- crucially -- the exact `state_dict`
key names, with every weight held by a real `nn.Conv2d` / `nn.Linear` / `nn.GroupNorm` so that
`torch_pruning`'s pruning functions, `EMAModel`, optimizers and `torch.save(model)` keep working.
Those layer objects are *parameter holders only*: their own `forward` is never called; all arithmetic
runs in the HIP kernels through `UNetEngine`.  There is no PyTorch fallback.
"""
import math
from collections import OrderedDict
from types import SimpleNamespace

import torch
import torch.nn as nn

from .engine import UNetEngine, _PinnedWeights


class UNet2DOutput(OrderedDict):
    """diffusers.utils.BaseOutput behaviour (unet_2d.py:28-35): `.sample`, `["sample"]`, `[0]`, `.to_tuple()`, and -- being a
    dict -- it is flattened by `torch_pruning.utils.flatten_as_list` when the reference's tracer walks the outputs."""

    def __init__(self, sample=None):
        super().__init__()
        self['sample'] = sample

    @property
    def sample(self):
        return OrderedDict.__getitem__(self, 'sample')

    def __getitem__(self, k):
        return OrderedDict.__getitem__(self, k) if isinstance(k, str) else self.to_tuple()[k]

    def to_tuple(self):
        return tuple(self.values())


class FrozenConfig(dict):
    """dict with attribute access (Diffusers' FrozenDict behaviour: `unet.config.sample_size`)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name) from None          # keeps copy / pickle protocol probes working


# ---- parameter-holder modules, named as in Diffusers so that state_dict keys match ----------------
class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift):
        super().__init__()
        self.num_channels, self.flip_sin_to_cos, self.downscale_freq_shift = num_channels, flip_sin_to_cos, downscale_freq_shift


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)


class ResnetBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, temb_channels, groups, eps, output_scale_factor=1.0):
        super().__init__()
        self.in_channels, self.out_channels, self.output_scale_factor = in_channels, out_channels, output_scale_factor
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps, affine=True)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, 1, 1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps, affine=True)
        self.dropout = nn.Dropout(0.0)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, 1, 1)
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1, 1, 0) if in_channels != out_channels else None


class Attention(nn.Module):
    def __init__(self, query_dim, heads, dim_head, groups, eps, rescale_output_factor=1.0):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.scale = dim_head ** -0.5          # fixed at construction, like attention_processor.py:85-86
        self.rescale_output_factor = rescale_output_factor
        self.residual_connection = True
        self.group_norm = nn.GroupNorm(groups, query_dim, eps=eps, affine=True)
        self.to_q = nn.Linear(query_dim, inner, bias=True)
        self.to_k = nn.Linear(query_dim, inner, bias=True)
        self.to_v = nn.Linear(query_dim, inner, bias=True)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim, bias=True), nn.Dropout(0.0)])


class Downsample2D(nn.Module):
    def __init__(self, channels, padding):
        super().__init__()
        self.channels, self.out_channels, self.padding, self.use_conv = channels, channels, padding, True
        self.conv = nn.Conv2d(channels, channels, 3, stride=2, padding=padding)


class Upsample2D(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.channels, self.out_channels, self.use_conv = channels, channels, True
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)


class _Block(nn.Module):
    pass


def _attn(cfg, channels, rescale=1.0):
    hd = cfg['attention_head_dim']
    heads = channels // hd if hd is not None else 1
    dim_head = hd if hd is not None else channels
    # heads > 1 (e.g. CompVis/ldm-celebahq-256, attention_head_dim 32): head h owns the contiguous channel rows
    # [h*d, (h+1)*d) of every image, so head_to_batch_dim (attention_processor.py:283-305) is a view of the channel-major
    # tokens and the batched products run with batch index n*heads + h; pruning selects head-grouped channels (ldm_prune.py:73-79)
    return Attention(channels, heads, dim_head, cfg['norm_num_groups'], cfg['norm_eps'], rescale)


_DEFAULTS = dict(
    sample_size=None, in_channels=3, out_channels=3, center_input_sample=False, time_embedding_type='positional',
    freq_shift=0, flip_sin_to_cos=True,
    down_block_types=('DownBlock2D', 'AttnDownBlock2D', 'AttnDownBlock2D', 'AttnDownBlock2D'),
    up_block_types=('AttnUpBlock2D', 'AttnUpBlock2D', 'AttnUpBlock2D', 'UpBlock2D'),
    block_out_channels=(224, 448, 672, 896), layers_per_block=2, mid_block_scale_factor=1, downsample_padding=1,
    act_fn='silu', attention_head_dim=8, norm_num_groups=32, norm_eps=1e-5, resnet_time_scale_shift='default',
    add_attention=True, class_embed_type=None, num_class_embeds=None)


class UNet2DModel(nn.Module):
    """Construction order follows unet_2d.py:84-217, so `state_dict()` key order matches the reference."""

    def __init__(self, **kwargs):
        super().__init__()
        cfg = dict(_DEFAULTS)
        unknown = set(kwargs) - set(cfg)
        if unknown:
            raise TypeError('unexpected UNet2DModel arguments: %s' % sorted(unknown))
        cfg.update(kwargs)
        cfg['down_block_types'] = tuple(cfg['down_block_types'])
        cfg['up_block_types'] = tuple(cfg['up_block_types'])
        cfg['block_out_channels'] = tuple(cfg['block_out_channels'])
        if cfg['time_embedding_type'] != 'positional' or cfg['act_fn'] not in ('silu', 'swish') \
                or cfg['resnet_time_scale_shift'] != 'default' or cfg['class_embed_type'] is not None \
                or cfg['num_class_embeds'] is not None:
            raise NotImplementedError('only the DDPM configuration family of the hot path is implemented')
        if len(cfg['down_block_types']) != len(cfg['up_block_types']) or \
                len(cfg['block_out_channels']) != len(cfg['down_block_types']):
            raise ValueError('down_block_types, up_block_types and block_out_channels must have equal lengths')
        for bt in cfg['down_block_types']:
            if bt not in ('DownBlock2D', 'AttnDownBlock2D'):
                raise NotImplementedError(bt)
        for bt in cfg['up_block_types']:
            if bt not in ('UpBlock2D', 'AttnUpBlock2D'):
                raise NotImplementedError(bt)
        self.config = FrozenConfig(cfg)
        self.sample_size = cfg['sample_size']
        boc = cfg['block_out_channels']
        G, eps, L = cfg['norm_num_groups'], cfg['norm_eps'], cfg['layers_per_block']
        tdim = boc[0] * 4

        self.conv_in = nn.Conv2d(cfg['in_channels'], boc[0], 3, padding=(1, 1))
        self.time_proj = Timesteps(boc[0], cfg['flip_sin_to_cos'], cfg['freq_shift'])
        self.time_embedding = TimestepEmbedding(boc[0], tdim)
        self.class_embedding = None
        self.down_blocks = nn.ModuleList([])
        self.up_blocks = nn.ModuleList([])

        out_c = boc[0]
        for i, bt in enumerate(cfg['down_block_types']):
            in_c, out_c = out_c, boc[i]
            blk = _Block()
            resnets, attns = [], []
            for j in range(L):
                resnets.append(ResnetBlock2D(in_c if j == 0 else out_c, out_c, tdim, G, eps))
                if bt == 'AttnDownBlock2D':
                    attns.append(_attn(cfg, out_c))
            if bt == 'AttnDownBlock2D':
                blk.attentions = nn.ModuleList(attns)
            blk.resnets = nn.ModuleList(resnets)
            blk.downsamplers = nn.ModuleList([Downsample2D(out_c, cfg['downsample_padding'])]) if i != len(boc) - 1 else None
            self.down_blocks.append(blk)

        msf = cfg['mid_block_scale_factor']
        mid = _Block()
        mid.add_attention = cfg['add_attention']
        r0 = ResnetBlock2D(boc[-1], boc[-1], tdim, G, eps, msf)
        att = [_attn(cfg, boc[-1], msf)] if cfg['add_attention'] else [None]
        r1 = ResnetBlock2D(boc[-1], boc[-1], tdim, G, eps, msf)
        mid.attentions = nn.ModuleList(att)
        mid.resnets = nn.ModuleList([r0, r1])
        self.mid_block = mid

        rev = list(reversed(boc))
        out_c = rev[0]
        for i, bt in enumerate(cfg['up_block_types']):
            prev, out_c = out_c, rev[i]
            in_c = rev[min(i + 1, len(boc) - 1)]
            blk = _Block()
            resnets, attns = [], []
            for j in range(L + 1):
                res_skip = in_c if j == L else out_c
                res_in = prev if j == 0 else out_c
                resnets.append(ResnetBlock2D(res_in + res_skip, out_c, tdim, G, eps))
                if bt == 'AttnUpBlock2D':
                    attns.append(_attn(cfg, out_c))
            if bt == 'AttnUpBlock2D':
                blk.attentions = nn.ModuleList(attns)
            blk.resnets = nn.ModuleList(resnets)
            blk.upsamplers = nn.ModuleList([Upsample2D(out_c)]) if i != len(boc) - 1 else None
            self.up_blocks.append(blk)

        self.conv_norm_out = nn.GroupNorm(G, boc[0], eps=eps)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(boc[0], cfg['out_channels'], 3, padding=1)
        self._engine = None
        self._multi_head = any(getattr(m, 'heads', 1) != 1 for m in self.modules())
        # dropout masks are Philox functions of (dropout_seed, layer, step, element): see engine.UNetEngine.set_dropout
        self.dropout_seed = 0
        self._dropout_step = 0

    def __getstate__(self):
        """Whole-module pickles (`torch.save(model, 'unet_pruned.pth')`, ddpm_prune.py:135) carry parameters and shapes
        only: the HIP engine (packed operands, streams) is rebuilt on first use after loading."""
        state = dict(self.__dict__)
        state['_engine'] = None
        for k in ('_leaf_cache', '_structure_tracing', '_hook_warned'):
            state.pop(k, None)
        return state

    # ------------------------------------------------------------------------------------------
    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    @property
    def device(self):
        return self.conv_in.weight.device

    @classmethod
    def from_config(cls, config):
        return cls(**{k: v for k, v in dict(config).items() if not k.startswith('_')})

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, subfolder=None, **kwargs):
        """config.json + diffusion_pytorch_model.{safetensors,bin} in a local directory (modeling_utils.py layout)."""
        from . import checkpoint
        return checkpoint.load_unet(pretrained_model_name_or_path, subfolder)

    def save_pretrained(self, save_directory, safe_serialization=False):
        from . import checkpoint
        checkpoint.save_unet(self, save_directory, safe_serialization)

    def engine(self):
        """The HIP execution engine bound to the *current* parameter tensors (re-bound after pruning)."""
        if self.conv_in.weight.device.type != 'cuda':
            raise RuntimeError('UNet2DModel runs on the MI355X HIP kernels only: move the model to a cuda device '
                               '(there is no CPU / PyTorch fallback)')
        if self._engine is None:
            self._engine = UNetEngine(self.config)
        params = {n: p.detach() for n, p in self.named_parameters()}
        self._engine.packs.rebind()           # in-place weight writes (EMAModel.copy_to / restore) are invisible to the cache
        self._engine.bind(params, None)
        self._engine.set_dropout(self.dropout_table() if self.training else None, getattr(self, 'dropout_seed', 0),
                                 getattr(self, '_dropout_step', 0))
        return self._engine

    def dropout_table(self):
        """{module name: p} of the nn.Dropout holders with p > 0 (what utils.set_dropout, utils.py:26-29, has set)."""
        return {n: float(m.p) for n, m in self.named_modules() if isinstance(m, nn.Dropout) and m.p > 0}

    def pin_weights(self):
        """Context manager: the weights are frozen inside the block (sweep, sampling loop) -> keep the packed operands."""
        return _PinnedWeights(self)

    def sampling_forward(self, shape, n_calls, replay=None):
        """The no-grad forward of a sampling loop (pipeline_ddim.py:101-116, pipeline_ddpm.py:87-96) as a callable
        `f(sample, t: int) -> eps` plus `f.close()`.  The weights are frozen for the lifetime of `f` (`pin_weights`).
        With `replay` (default: automatic -- a cuda model in eval mode without foreign forward hooks, called at least
        REPLAY_MIN_CALLS times; DP_SAMPLE_REPLAY=0 never, =1 from the first call) the forward is captured ONCE at `shape` and every
        call re-issues its ~180 launches from the library's C loop (ops.CapturedCall, csrc/replay.hip): same kernels, same
        arguments, same order -> the same bits as the eager call, without ~10 ms of Python / ctypes per UNet forward."""
        import os
        if replay is None:
            env = os.environ.get('DP_SAMPLE_REPLAY')
            replay = (env != '0' and n_calls >= (1 if env == '1' else REPLAY_MIN_CALLS) and not self.training
                      and self.conv_in.weight.device.type == 'cuda' and self._leaf_hook_owner() is None
                      and not self._forward_hooks and not self._forward_pre_hooks        # module-level hooks see eager calls only
                      and not self.__dict__.get('_structure_tracing') and hasattr(torch.cuda, 'CUDAGraph'))
        return _CapturedForward(self, shape) if replay else _EagerForward(self)

    def _timesteps(self, sample, timestep):
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([t], dtype=torch.long, device=sample.device)
        elif t.dim() == 0:
            t = t[None].to(sample.device)
        t = t.to(sample.device)
        return t * torch.ones(sample.shape[0], dtype=t.dtype, device=sample.device)

    # ---- forward hooks on the holder leaves (boundary B2, ddpm_prune.py:79-89) --------------------------------------------
    # The HIP engine never calls the Conv2d / Linear / GroupNorm holder modules and is ONE autograd node, so hooks registered on
    # them cannot fire from the numerics path.  What a hooked forward does is decided by WHO owns the hooks -- never silently:
    #   * an autograd tracer (`tp.pruner.MagnitudePruner(model, example_inputs, ...)` builds its DependencyGraph by hooking the
    #     leaves, calling `model(**example_inputs)` and walking the `grad_fn`s of the result, ddpm_exp/torch_pruning/
    #     dependency.py:631-690; this package's trace.py does the same) or an explicit `with model.structure_tracing():`
    #     -> `structure_forward`: the reference's layer sequence (unet_2d.py:219-316, resnet.py:589-639,
    #     attention_processor.py:870-935) over the holder modules on a batch of ZERO images: every leaf is called once, per-layer
    #     `grad_fn`s exist, shapes propagate, there are NO VALUES (a tracer never reads any).  One warning per model.
    #   * any other hook (a MAC counter such as `tp.utils.count_ops_and_params`, ddpm_prune.py:89,118; a profiler; a debugging
    #     hook): the same layer sequence runs at the CALLER'S batch under FakeTensorMode -- hooks fire once per leaf and see
    #     shape-correct FakeTensors (reading a value out of one raises) -- and the returned sample is then computed by the HIP
    #     engine as in an un-hooked call: `-> FloatTensor[B, C, H, W]` always holds.  One warning per model.
    TRACER_HOOK_MODULES = ('torch_pruning.dependency',)      # suffixes of `hook.__module__` that identify an autograd tracer

    def _leaf_hook_dicts(self):
        dicts = self.__dict__.get('_leaf_cache')
        if dicts is None:        # the holder leaves are fixed at construction (pruning slices parameters, never swaps modules)
            leaves = [m for m in self.modules() if isinstance(m, (nn.Conv2d, nn.Linear, nn.GroupNorm))]
            dicts = [m._forward_hooks for m in leaves] + [m._forward_pre_hooks for m in leaves]
            self.__dict__['_leaf_cache'] = dicts
        return dicts

    def _leaf_hook_owner(self):
        """None (no hook on any holder leaf), 'tracer' (every hook belongs to an autograd tracer) or 'other'."""
        dicts = self._leaf_hook_dicts()
        if not any(dicts):
            return None
        own = __name__.rsplit('.', 1)[0] + '.trace'
        for d in dicts:
            for fn in d.values():
                fn = getattr(fn, 'func', fn)                                  # functools.partial
                mod = getattr(getattr(fn, '__func__', fn), '__module__', None) or ''
                if not (mod == own or any(mod == s or mod.endswith('.' + s) for s in self.TRACER_HOOK_MODULES)):
                    return 'other'
        return 'tracer'

    def _warn_once(self, key, msg):
        import warnings
        seen = self.__dict__.setdefault('_hook_warned', set())
        if key not in seen:
            seen.add(key)
            warnings.warn(msg, stacklevel=4)

    def structure_tracing(self):
        """Context manager: every forward inside the block is the zero-image `structure_forward` (for autograd tracers this
        package does not recognise by their hooks' module)."""
        return _StructureTracing(self)

    def structure_forward(self, sample, timestep):
        """Forward through the holder modules on a batch of ZERO images: every leaf runs once with the real ATen autograd
        nodes behind it (the same graph the reference's model produces), shapes propagate, and not one value is computed."""
        dev = self.conv_in.weight.device
        x = torch.empty((0,) + tuple(sample.shape[1:]), dtype=torch.float32, device=dev)
        with torch.enable_grad():
            return UNet2DOutput(sample=self._structure_layers(x))

    def shape_forward(self, sample):
        """The layer sequence at the caller's batch under FakeTensorMode: leaf hooks fire with shape-correct FakeTensors, no
        kernel of any backend runs.  Returns the output shape."""
        from torch._subclasses.fake_tensor import FakeTensorMode
        dev = self.conv_in.weight.device
        with FakeTensorMode(allow_non_fake_inputs=True), torch.no_grad():
            x = torch.empty(tuple(sample.shape), dtype=torch.float32, device=dev)
            return tuple(self._structure_layers(x).shape)

    def _structure_layers(self, x):
        F = torch.nn.functional
        cfg = self.config
        half = cfg['block_out_channels'][0] // 2
        temb = torch.empty((x.shape[0], 2 * half), dtype=torch.float32, device=x.device)    # sinusoidal table: no parameters
        emb = self.time_embedding.linear_2(F.silu(self.time_embedding.linear_1(temb)))

        def resnet(r, h):
            y = r.conv1(F.silu(r.norm1(h)))
            y = y + r.time_emb_proj(F.silu(emb))[:, :, None, None]
            y = r.conv2(r.dropout(F.silu(r.norm2(y))))
            sc = r.conv_shortcut(h) if r.conv_shortcut is not None else h
            return (sc + y) / r.output_scale_factor

        def attention(a, h):                      # AttnProcessor2_0, op for op (the tracer's member order follows the op order)
            N, C, H, W = h.shape
            t = h.view(N, C, H * W).transpose(1, 2)
            t = a.group_norm(t.transpose(1, 2)).transpose(1, 2)
            q, k, v = a.to_q(t), a.to_k(t), a.to_v(t)
            d = q.shape[-1] // a.heads
            q, k, v = [z.view(N, H * W, a.heads, d).transpose(1, 2) for z in (q, k, v)]
            o = F.scaled_dot_product_attention(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False)
            o = o.transpose(1, 2).reshape(N, H * W, a.heads * d)
            o = a.to_out[1](a.to_out[0](o)).transpose(-1, -2).reshape(N, C, H, W)
            return (o + h) / a.rescale_output_factor

        h = self.conv_in(x)
        skips = [h]
        for blk in self.down_blocks:
            for j, r in enumerate(blk.resnets):
                h = resnet(r, h)
                if hasattr(blk, 'attentions'):
                    h = attention(blk.attentions[j], h)
                skips.append(h)
            if blk.downsamplers is not None:
                d = blk.downsamplers[0]
                h = d.conv(F.pad(h, (0, 1, 0, 1)) if d.padding == 0 else h)
                skips.append(h)
        h = resnet(self.mid_block.resnets[0], h)
        if self.mid_block.attentions[0] is not None:
            h = attention(self.mid_block.attentions[0], h)
        h = resnet(self.mid_block.resnets[1], h)
        for blk in self.up_blocks:
            for j, r in enumerate(blk.resnets):
                h = resnet(r, torch.cat([h, skips.pop()], dim=1))
                if hasattr(blk, 'attentions'):
                    h = attention(blk.attentions[j], h)
            if blk.upsamplers is not None:
                h = blk.upsamplers[0].conv(F.interpolate(h, scale_factor=2.0, mode='nearest'))
        return self.conv_out(F.silu(self.conv_norm_out(h)))

    def forward(self, sample, timestep, class_labels=None, return_dict=True):
        if class_labels is not None:
            raise ValueError('class conditioning is not part of this model')
        if self.__dict__.get('_structure_tracing'):
            out = self.structure_forward(sample, timestep)
            return out if return_dict else out.to_tuple()
        owner = self._leaf_hook_owner()
        if owner == 'tracer':
            self._warn_once('tracer', 'UNet2DModel: the forward hooks on its Conv2d / Linear / GroupNorm leaves belong to an '
                            'autograd tracer -> structure-only forward on a batch of ZERO images (per-layer grad_fns and '
                            'shapes, no values); the returned sample has no elements')
            out = self.structure_forward(sample, timestep)
            return out if return_dict else out.to_tuple()
        if owner == 'other':
            self._warn_once('other', 'UNet2DModel: the HIP engine does not call the Conv2d / Linear / GroupNorm holder modules; '
                            'forward hooks on them fire once per layer in a shape-only pass at the caller\'s batch '
                            '(FakeTensors: shapes and dtypes, no values) and the returned sample is computed by the engine. '
                            'Use `with model.structure_tracing():` for an autograd tracer.')
            self.shape_forward(sample)
        t = self._timesteps(sample, timestep)
        if self.training:
            self._dropout_step = getattr(self, '_dropout_step', 0) + 1       # a fresh mask per training-mode forward
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if needs_grad:
            names = [n for n, _ in self.named_parameters()]
            out = _UNetFunction.apply(self, names, sample, t, *list(self.parameters()))
        else:
            out = self.engine().forward(sample.to(torch.float32), t, save=False)
        if not return_dict:
            return (out,)
        return UNet2DOutput(sample=out)


# A capture costs an eager forward + the capture pass + the node read-back (~35 ms for the 195 nodes of the CIFAR UNet) and the packed
# operands are only valid for ONE pinned scope (the next pipeline call captures again), while a GPU-bound step gains ~0.6 ms from
# the replay (batch 256: 14.12 -> 13.54 ms) and a host-bound one several ms: from 32 calls on it pays in both regimes.
REPLAY_MIN_CALLS = 32


class _EagerForward:
    def __init__(self, model):
        self.model = model
        self._pin = model.pin_weights()                    # sampling never writes weights: pack the operands once
        self._pin.__enter__()

    def __call__(self, sample, t):
        return self.model(sample, t).sample

    def close(self):
        if self._pin is not None:
            self._pin.__exit__(None, None, None)
            self._pin = None


class _CapturedForward:
    """One captured no-grad forward at a fixed [B, C, H, W], replayed natively (UNet2DModel.sampling_forward)."""

    def __init__(self, model, shape):
        from . import ops
        self.model = model
        self._pin = model.pin_weights()
        self._pin.__enter__()
        try:
            dev = model.device
            eng = self._pin.eng
            self.x = ops.empty_act(tuple(shape), dev)
            self.x.zero_()
            self.t = torch.zeros(shape[0], dtype=torch.long, device=dev)
            with torch.no_grad():
                eng.forward(self.x, self.t, save=False)              # eager once: code objects, Winograd operands asked for
                eng.prepare_packs()                                  # nothing is packed inside the capture
                self.call = ops.CapturedCall(lambda: eng.forward(self.x, self.t, save=False))
        except BaseException:
            self.close()
            raise

    def __call__(self, sample, t):
        if self._pin is None:
            # the capture holds the ADDRESSES of the pinned packed operands: after close() they may have been freed or recycled
            raise RuntimeError('sampling_forward: called after close() (the captured forward\'s packed operands are no longer pinned)')
        self.x.copy_(sample)
        if torch.is_tensor(t) and t.dim() > 0 and t.numel() > 1:
            self.t.copy_(t)
        else:
            self.t.fill_(int(t))
        return self.call.launch()          # a tensor of the capture's pool: valid until the next call

    def close(self):
        if self._pin is not None:
            if getattr(self, 'call', None) is not None:
                torch.cuda.synchronize(self.model.device)    # no replayed launch may still read the operands about to be unpinned
                self.call = None                             # ... or the capture's pool, which goes with the graph
            self._pin.__exit__(None, None, None)
            self._pin = None


class _StructureTracing:
    def __init__(self, model):
        self.model = model

    def __enter__(self):
        self.prev = self.model.__dict__.get('_structure_tracing', False)
        self.model.__dict__['_structure_tracing'] = True
        return self.model

    def __exit__(self, *exc):
        self.model.__dict__['_structure_tracing'] = self.prev
        return False


class _UNetFunction(torch.autograd.Function):
    """Bridges the engine into autograd as ONE node (DDP hooks, optimizers and `loss.backward()` keep working)."""

    @staticmethod
    def forward(ctx, model, names, sample, t, *params):
        eng = model.engine()
        out = eng.forward(sample.detach().to(torch.float32), t, save=True)
        ctx.model, ctx.names, ctx.saved_ctx = model, names, eng.ctx
        eng.ctx = None
        return out

    @staticmethod
    def backward(ctx, dout):
        model = ctx.model
        eng = model.engine()
        grads = {n: torch.zeros_like(p.data) for n, p in model.named_parameters()}
        eng.bind(eng.P, grads)
        eng.ctx = ctx.saved_ctx
        eng.backward(dout.contiguous())
        return (None, None, None, None) + tuple(grads[n] for n in ctx.names)
