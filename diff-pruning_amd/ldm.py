"""LDM (CompVis latent-diffusion) UNet on the HIP engine: the model of the reference's LDM importance pass
(ldm_exp/prune_ldm.py:86-132, config ldm_exp/configs/latent-diffusion/cin256-v2.yaml).

`UNetModel` mirrors ldm_exp/ldm/modules/diffusionmodules/openaimodel.py:413-742 for the configuration family the
reference prunes (use_spatial_transformer=True, no class labels; any num_heads / num_head_channels / transformer_depth): same constructor
arguments, same `forward(x, timesteps, context)` and the same state-dict keys, with every weight held by a real
nn.Conv2d / nn.Linear / nn.GroupNorm / nn.LayerNorm (parameter holders; all arithmetic is in the HIP kernels).

Engine notes.  Tokens stay channel-major ([N][C][T]), so proj_in / q,k,v / FF projections are the same implicit-GEMM
kernel as the 1x1 convolutions and LayerNorm reduces over the strided channel axis with one thread per token.
Cross-attention (`attn2`) over the ONE class-embedding token (attention.py:168-193 with context [B,1,512]): the softmax
over a single key is identically 1, so its output is to_out(to_v(context)) broadcast over the tokens; norm2, to_q and
to_k receive exactly-zero gradients (as under autograd in the reference) and are not evaluated.  A context of L > 1 tokens
(attention.py:152-193 takes any; no configuration the reference prunes has one) takes the general form: the context as
channel-major tokens [B, D, 1, L], K / V as 1x1 projections of it, and the three launches of the self-attention (round 5).
"""
import torch
import torch.nn as nn

from . import ops
from . import engine as engine_mod
from .engine import UNetEngine, RES_LDM, _SPEC1, _SPEC3, _SPEC_UP, _PinnedWeights

_SPEC_DOWN = ops.ConvSpec(3, 2, 1, 0)


def ldm_blocks(cfg):
    """Block structure of UNetModel(**cfg) (openaimodel.py:517-692)."""
    mc, mult, nres = cfg['model_channels'], list(cfg['channel_mult']), cfg['num_res_blocks']
    att = set(cfg['attention_resolutions'])
    inp = [[('conv_in', cfg['in_channels'], mc)]]
    chans = [mc]
    ch, ds = mc, 1
    for level, m in enumerate(mult):
        for _ in range(nres):
            items = [('res', ch, m * mc)]
            ch = m * mc
            if ds in att:
                items.append(('st', ch))
            inp.append(items)
            chans.append(ch)
        if level != len(mult) - 1:
            inp.append([('down', ch)])
            chans.append(ch)
            ds *= 2
    mid = ch
    out = []
    for level, m in list(enumerate(mult))[::-1]:
        for i in range(nres + 1):
            ich = chans.pop()
            items = [('res', ch + ich, mc * m)]
            ch = mc * m
            if ds in att:
                items.append(('st', ch))
            if level and i == nres:
                items.append(('up', ch))
                ds //= 2
            out.append(items)
    return inp, out, mid


def st_heads(cfg, ch):
    """(heads, dim_head) of the SpatialTransformer of a `ch`-channel level (openaimodel.py:542-549, legacy=True with
    use_spatial_transformer: `num_heads` heads, or ch // num_head_channels of them; dim_head = ch // heads)."""
    nhc = cfg.get('num_head_channels', -1)
    heads = cfg.get('num_heads', 1) if nhc in (-1, None) else ch // nhc
    return heads, ch // heads


# ---- parameter-holder modules (names as in the CompVis code base) -----------------------------------------------
class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)


class FeedForward(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.Sequential(GEGLU(dim, dim * 4), nn.Dropout(0.0), nn.Linear(dim * 4, dim))


class CrossAttention(nn.Module):
    def __init__(self, query_dim, context_dim, heads, dim_head):
        super().__init__()
        inner = heads * dim_head
        self.scale, self.heads = dim_head ** -0.5, heads
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(context_dim, inner, bias=False)
        self.to_v = nn.Linear(context_dim, inner, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, query_dim), nn.Dropout(0.0))


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, n_heads, d_head, context_dim):
        super().__init__()
        self.attn1 = CrossAttention(dim, dim, n_heads, d_head)
        self.ff = FeedForward(dim)
        self.attn2 = CrossAttention(dim, context_dim, n_heads, d_head)
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(dim), nn.LayerNorm(dim), nn.LayerNorm(dim)


class SpatialTransformer(nn.Module):
    def __init__(self, in_channels, n_heads, d_head, context_dim, depth=1):
        super().__init__()
        inner = n_heads * d_head
        self.in_channels = in_channels
        self.norm = nn.GroupNorm(32, in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Conv2d(in_channels, inner, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(inner, n_heads, d_head, context_dim) for _ in range(depth)])
        self.proj_out = nn.Conv2d(inner, in_channels, 1)


class ResBlock(nn.Module):
    def __init__(self, channels, emb_channels, out_channels):
        super().__init__()
        self.channels, self.out_channels = channels, out_channels
        self.in_layers = nn.Sequential(nn.GroupNorm(32, channels), nn.SiLU(), nn.Conv2d(channels, out_channels, 3, padding=1))
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_channels, out_channels))
        self.out_layers = nn.Sequential(nn.GroupNorm(32, out_channels), nn.SiLU(), nn.Dropout(0.0),
                                        nn.Conv2d(out_channels, out_channels, 3, padding=1))
        self.skip_connection = nn.Identity() if channels == out_channels else nn.Conv2d(channels, out_channels, 1)


class Downsample(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.channels = self.out_channels = channels
        self.op = nn.Conv2d(channels, channels, 3, stride=2, padding=1)


class Upsample(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.channels = self.out_channels = channels
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)


class UNetModel(nn.Module):
    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 dropout=0, channel_mult=(1, 2, 4, 8), num_heads=1, use_spatial_transformer=True, transformer_depth=1,
                 context_dim=None, num_head_channels=-1):
        super().__init__()
        if not use_spatial_transformer or context_dim is None or dropout:
            raise NotImplementedError('only the spatial-transformer (cross-attention conditioned, dropout-free) family of the LDM '
                                      'importance pass is implemented')
        if num_head_channels in (-1, None):
            num_head_channels = -1
            if num_heads is None or num_heads < 1:                                  # openaimodel.py:483-487
                raise ValueError('Either num_heads or num_head_channels has to be set')
        if transformer_depth < 1:
            raise ValueError('transformer_depth >= 1')
        self.config = dict(image_size=image_size, in_channels=in_channels, model_channels=model_channels,
                           out_channels=out_channels, num_res_blocks=num_res_blocks,
                           attention_resolutions=list(attention_resolutions), channel_mult=list(channel_mult),
                           num_heads=num_heads, num_head_channels=num_head_channels, use_spatial_transformer=True,
                           transformer_depth=int(transformer_depth), context_dim=context_dim)
        tdim = model_channels * 4
        self.time_embed = nn.Sequential(nn.Linear(model_channels, tdim), nn.SiLU(), nn.Linear(tdim, tdim))
        inp, out, mid = ldm_blocks(self.config)

        def st_module(ch):
            heads, d_head = st_heads(self.config, ch)
            if heads < 1 or heads * d_head != ch:
                raise ValueError('q,k,v channels %d is not divisible by the head width / count' % ch)      # openaimodel.py:299-300
            return SpatialTransformer(ch, heads, d_head, context_dim, depth=int(transformer_depth))

        def make(it):
            if it[0] == 'conv_in':
                return nn.Conv2d(it[1], it[2], 3, padding=1)
            if it[0] == 'res':
                return ResBlock(it[1], tdim, it[2])
            if it[0] == 'st':
                return st_module(it[1])
            if it[0] == 'down':
                return Downsample(it[1])
            return Upsample(it[1])

        self.input_blocks = nn.ModuleList([nn.Sequential(*[make(it) for it in items]) for items in inp])
        self.middle_block = nn.Sequential(ResBlock(mid, tdim, mid), st_module(mid),
                                          ResBlock(mid, tdim, mid))
        self.output_blocks = nn.ModuleList([nn.Sequential(*[make(it) for it in items]) for items in out])
        self.out = nn.Sequential(nn.GroupNorm(32, model_channels), nn.SiLU(), nn.Conv2d(model_channels, out_channels, 3, padding=1))
        self._engine = None

    def __getstate__(self):
        """Whole-module pickles (`torch.save(model, 'unet_pruned.pth')`, ddpm_prune.py:135) carry parameters and shapes
        only: the HIP engine (packed operands, streams) is rebuilt on first use after loading."""
        state = dict(self.__dict__)
        state['_engine'] = None
        return state

    @property
    def device(self):
        return self.out[2].weight.device

    def engine(self):
        if self.device.type != 'cuda':
            raise RuntimeError('UNetModel runs on the MI355X HIP kernels only (no CPU / PyTorch fallback)')
        if self._engine is None:
            self._engine = LdmEngine(self.config)
        self._engine.packs.rebind()           # in-place weight writes are invisible to the pack cache
        self._engine.bind({n: p.detach() for n, p in self.named_parameters()}, None)
        return self._engine

    def pin_weights(self):
        """Context manager: the weights are frozen inside the block (sampling loop, importance pass)."""
        return _PinnedWeights(self)

    def forward(self, x, timesteps=None, context=None, y=None, **kwargs):
        if y is not None:
            raise ValueError('class-label conditioning (num_classes) is not part of this configuration')
        return self.engine().forward(x.to(torch.float32), timesteps, context, save=False)

    def forward_cfg_pair(self, x, timesteps, context2):
        """`self(cat([x, x]), cat([t, t]), context=context2)` of the classifier-free-guidance sampler (ddim.py:178-181) with the
        context-free stem evaluated once (LdmEngine.forward, cfg_pair): [2B, C, H, W] = [uncond half; cond half]."""
        return self.engine().forward(x.to(torch.float32), timesteps, context2, save=False, cfg_pair=True)


class LdmEngine(UNetEngine):
    """Forward + hand-written backward of the LDM UNet (openaimodel.py:710-742) on the HIP kernels."""

    _temb_suffix = '.emb_layers.1'

    def _resnet_prefixes(self):
        inp, out, _ = ldm_blocks(self.cfg)
        names = ['input_blocks.%d.%d' % (bi, li) for bi, items in enumerate(inp) for li, it in enumerate(items) if it[0] == 'res']
        names += ['middle_block.0', 'middle_block.2']
        names += ['output_blocks.%d.%d' % (bi, li) for bi, items in enumerate(out) for li, it in enumerate(items) if it[0] == 'res']
        return names

    def res_fwd(self, pre, xa, xb, semb, save):
        return self.resnet_fwd(pre, xa, xb, semb, 1.0, save, names=RES_LDM, G=32, eps=1e-5)

    # ---- SpatialTransformer (attention.py:215-258) ------------------------------------------------------------
    def st_fwd(self, pre, x, ctx2d, ch, save):
        """`ch`: the UN-pruned channel count of the level -- heads and the softmax scale dim_head ** -0.5 are fixed at construction
        (openaimodel.py:542-549, attention.py:158) and survive pruning.  `transformer_depth` BasicTransformerBlocks."""
        P = self.P
        heads, dim_head = st_heads(self.cfg, ch)
        scale = float(dim_head) ** -0.5
        n0, st0 = ops.groupnorm_fwd(x, None, P[pre + '.norm.weight'], P[pre + '.norm.bias'], 32, 1e-6, False)
        h = self._conv(pre + '.proj_in', n0, None, _SPEC1)
        blocks = []
        for d in range(self.cfg.get('transformer_depth', 1)):
            h, kept = self._tb_fwd('%s.transformer_blocks.%d' % (pre, d), h, ctx2d, scale, heads, save is not None)
            blocks.append(kept)
        out = self._conv(pre + '.proj_out', h, None, _SPEC1, res=x)
        if save is not None:
            save[pre] = (x, st0, n0, blocks, h, scale, heads)
        return out

    def _tb_fwd(self, tb, h, ctx2d, scale, heads, keep):
        """BasicTransformerBlock (attention.py:196-212) on channel-major tokens [N, inner, H, W]: self-attention, cross-attention,
        GEGLU feed-forward, each with its residual.  Heads: 'b n (h d) -> (b h) n d' (attention.py:177) is a VIEW here -- head j owns
        the contiguous channel rows [j d, (j + 1) d) of every image, batch index n * heads + j."""
        P = self.P
        N, inner, H, W = h.shape
        T = H * W
        # attn1: self-attention
        l1, ls1 = ops.layernorm_fwd(h, P[tb + '.norm1.weight'], P[tb + '.norm1.bias'])
        fused = self.fuse_qkv and heads == 1 and hasattr(ops, 'empty_act')
        if fused:                                     # one M = 3 * inner contraction instead of three (see UNetEngine.attn_fwd)
            wp, ld, _, _, _, (cq, ck, cv) = self._qkv_pack(tb + '.attn1')
            qkv = ops.conv_forward(l1, None, wp, ld, cq + ck + cv, _SPEC1)
            q, k, v = qkv[:, :cq], qkv[:, cq:cq + ck], qkv[:, cq + ck:]
        else:
            q = self._conv(tb + '.attn1.to_q', l1, None, _SPEC1)
            k = self._conv(tb + '.attn1.to_k', l1, None, _SPEC1)
            v = self._conv(tb + '.attn1.to_v', l1, None, _SPEC1)
        ai, av = q.shape[1], v.shape[1]               # the value width may differ from the query / key width after pruning
        Z = N * heads
        if not keep and getattr(ops, 'FUSED_ATTN', False) and ops.attention_fused_ok(T, ai // heads, av // heads):
            p = None                                  # sampling forward: one kernel, no [T, T] scores (csrc/attention.hip)
            o = ops.attention_fwd(q, k, v, heads, scale)
        else:
            s = ops.bmm_tn(q.view(Z, ai // heads, T), k.view(Z, ai // heads, T), alpha=scale)
            p = ops.softmax_fwd(s, out=s)
            o = ops.bmm_nt(v.view(Z, av // heads, T), p)
        h1 = self._conv(tb + '.attn1.to_out.0', o.view(N, av, H, W), None, _SPEC1, res=h)
        hit = self._ctx_cache.get(tb) if self._ctx_cache is not None else None
        x2 = None
        if ctx2d is not None:
            # attn2: cross-attention over a single context token == broadcast of to_out(to_v(context)) (the softmax over one key
            # is 1 for every head)
            if hit is not None:
                v2, o2 = hit                          # sampling loop: same context and weights at every DDIM step
            else:
                v2 = self._linear(tb + '.attn2.to_v', ctx2d)
                o2 = self._linear(tb + '.attn2.to_out.0', v2).contiguous()
                if self._ctx_cache is not None:
                    self._ctx_cache[tb] = (v2, o2)
            h2 = ops.add_rowvec(h1, o2)
        else:
            # attn2 over L > 1 context tokens (ldm/modules/attention.py:152-193, general form): the context as channel-major
            # "tokens" [N, D, 1, L] makes to_k / to_v 1x1 convolutions whose outputs K, V [N, inner, L] are the operands of the
            # same three launches as the self-attention: S = Q^T K [T, L], softmax over the L keys, O = V P^T.
            cx = self._ctx_cm                         # [N, D, 1, L]
            L_ = cx.shape[3]
            l2, ls2 = ops.layernorm_fwd(h1, P[tb + '.norm2.weight'], P[tb + '.norm2.bias'])
            q2 = self._conv(tb + '.attn2.to_q', l2, None, _SPEC1)
            if hit is not None:
                k2, v2 = hit
            else:
                k2 = self._conv(tb + '.attn2.to_k', cx, None, _SPEC1)
                v2 = self._conv(tb + '.attn2.to_v', cx, None, _SPEC1)
                if self._ctx_cache is not None:
                    self._ctx_cache[tb] = (k2, v2)
            a2, b2 = q2.shape[1], v2.shape[1]
            s2 = ops.bmm_tn(q2.view(Z, a2 // heads, T), k2.view(Z, a2 // heads, L_), alpha=scale)
            p2 = ops.softmax_fwd(s2, out=s2)
            o2 = ops.bmm_nt(v2.view(Z, b2 // heads, L_), p2)
            h2 = self._conv(tb + '.attn2.to_out.0', o2.view(N, b2, H, W), None, _SPEC1, res=h1)
            x2 = (l2, ls2, q2, k2, p2, o2, cx)
        # feed-forward (GEGLU)
        l3, ls3 = ops.layernorm_fwd(h2, P[tb + '.norm3.weight'], P[tb + '.norm3.bias'])
        pr = self._conv(tb + '.ff.net.0.proj', l3, None, _SPEC1)
        gg = ops.geglu_fwd(pr)
        h3 = self._conv(tb + '.ff.net.2', gg, None, _SPEC1, res=h2)
        kept = (h, l1, ls1, q, k, v, p, o, h1, v2, ctx2d, h2, l3, ls3, pr, gg, fused, x2) if keep else None
        return h3, kept

    # The cross-attention branch of every transformer block depends on the context token and the weights only
    # (module docstring): inside a sampling loop -- frozen weights, one context for all DDIM steps -- it is computed once.
    _ctx_cache = None

    def context_cache(self, context):
        """Context manager: `context` (the [B, 1, D] tensor handed to every forward inside the block) and the weights do not
        change inside -> the 2 x 16 cross-attention projections run once instead of once per forward."""
        eng = self

        class _Scope:
            def __enter__(self):
                eng._ctx_cache, eng._ctx_key = {}, (context.data_ptr(), tuple(context.shape))
                return eng

            def __exit__(self, *exc):
                eng._ctx_cache = None
                return False
        return _Scope()

    def _ln_param_grads(self, name, pws):
        N, C = pws.shape[0], pws.shape[1]
        self._colsum(pws, N, C, 2, 1, self.G[name + '.weight'])
        self._colsum(pws, N, C, 2, 0, self.G[name + '.bias'])

    def st_bwd(self, pre, dout, extra=None):
        x, st0, n0, blocks, h_last, scale, heads = self.ctx.pop(pre)
        P = self.P
        hw = (x.shape[2], x.shape[3])
        dh = self._conv_bwd(pre + '.proj_out', dout, h_last, None, _SPEC1, hw)
        for d in range(len(blocks) - 1, -1, -1):
            dh = self._tb_bwd('%s.transformer_blocks.%d' % (pre, d), blocks[d], dh, scale, heads)
            blocks[d] = None
        dn0 = self._conv_bwd(pre + '.proj_in', dh, n0, None, _SPEC1, hw)
        dx, pws = ops.groupnorm_bwd(x, None, P[pre + '.norm.weight'], P[pre + '.norm.bias'], st0, dn0, 32, False,
                                    add1=dout, add2=extra)
        self._gn_param_grads(pre + '.norm', pws)
        return dx

    def _tb_bwd(self, tb, kept, dh3, scale, heads):
        """Backward of _tb_fwd: dh3 = gradient of the block's output; returns the gradient of its input."""
        (h, l1, ls1, q, k, v, p, o, h1, v2, ctx2d, h2, l3, ls3, pr, gg, fused, x2) = kept
        P = self.P
        N, inner, H, W = h.shape
        T = H * W
        hw = (H, W)
        ai, av = q.shape[1], v.shape[1]
        Z = N * heads
        # feed-forward
        dgg = self._conv_bwd(tb + '.ff.net.2', dh3, gg, None, _SPEC1, hw)
        dpr = ops.geglu_bwd(pr, dgg)
        dl3 = self._conv_bwd(tb + '.ff.net.0.proj', dpr, l3, None, _SPEC1, hw)
        dh2, pws = ops.layernorm_bwd(h2, P[tb + '.norm3.weight'], ls3, dl3, add=dh3)
        self._ln_param_grads(tb + '.norm3', pws)
        if x2 is None:
            # attn2 (context token): d o2[n, c] = sum_t dh2
            rows = ops.rowsum_nc(dh2)
            dv2 = self._linear_bwd(tb + '.attn2.to_out.0', rows, v2)
            self._linear_bwd(tb + '.attn2.to_v', dv2, ctx2d, need_dx=False)
            dh1 = dh2                                  # h2 = h1 + (a row vector that does not depend on h1)
        else:
            # attn2 over L > 1 context tokens: the self-attention backward with K, V projected from the context; norm2 / to_q get
            # gradients now (over a single key the softmax is constant and they are exactly zero), the context itself gets none
            # (the importance pass differentiates the UNet's parameters only)
            l2, ls2, q2, k2, p2, o2, cx = x2
            a2, b2, L_ = q2.shape[1], v2.shape[1], cx.shape[3]
            do2 = self._conv_bwd(tb + '.attn2.to_out.0', dh2, o2.view(N, b2, H, W), None, _SPEC1, hw)
            do23 = do2.view(Z, b2 // heads, T)
            dv2 = ops.bmm_nn(do23, p2)                                         # [Z, dv, L]
            dp2 = ops.bmm_tn(do23, v2.view(Z, b2 // heads, L_))                # [Z, T, L]
            ds2 = ops.softmax_bwd(p2, dp2, scale, out=dp2)
            dq2 = ops.bmm_nt(k2.view(Z, a2 // heads, L_), ds2)                 # [Z, d, T]
            dk2 = ops.bmm_nn(q2.view(Z, a2 // heads, T), ds2)                  # [Z, d, L]
            self._conv_bwd(tb + '.attn2.to_k', dk2.view(N, a2, 1, L_), cx, None, _SPEC1, (1, L_), need_dx=False)
            self._conv_bwd(tb + '.attn2.to_v', dv2.view(N, b2, 1, L_), cx, None, _SPEC1, (1, L_), need_dx=False)
            dl2 = self._conv_bwd(tb + '.attn2.to_q', dq2.view(N, a2, H, W), l2, None, _SPEC1, hw)
            dh1, pws2 = ops.layernorm_bwd(h1, P[tb + '.norm2.weight'], ls2, dl2, add=dh2)
            self._ln_param_grads(tb + '.norm2', pws2)
        dh2 = dh1
        # attn1
        do = self._conv_bwd(tb + '.attn1.to_out.0', dh2, o.view(N, av, H, W), None, _SPEC1, hw)
        do3 = do.view(Z, av // heads, T)
        if fused:
            # dq | dk | dv written straight into channel slices of one buffer; one K = 3 * inner input-gradient contraction
            _, _, _, wd, ldd, (cq, ck, cv) = self._qkv_pack(tb + '.attn1')
            d_qkv = ops.empty_act((N, cq + ck + cv, H, W), h.device)
            sl = (d_qkv[:, :cq], d_qkv[:, cq:cq + ck], d_qkv[:, cq + ck:])
            ops.bmm_nn(do3, p, out=sl[2].view(N, cv, T))
            dp = ops.bmm_tn(do3, v.view(N, cv, T))
            ds = ops.softmax_bwd(p, dp, scale, out=dp)
            ops.bmm_nt(k.view(N, ck, T), ds, out=sl[0].view(N, cq, T))
            ops.bmm_nn(q.view(N, cq, T), ds, out=sl[1].view(N, ck, T))
            from .engine import FUSE_QKV_WGRAD
            if FUSE_QKV_WGRAD and hasattr(self, '_qkv_param_grads'):
                self._qkv_param_grads(tb + '.attn1', d_qkv, l1, (cq, ck, cv))              # one M = 3 x inner weight-gradient launch
            else:
                for dproj, name in zip(sl, ('.attn1.to_q', '.attn1.to_k', '.attn1.to_v')):
                    self._conv_bwd(tb + name, dproj, l1, None, _SPEC1, hw, need_dx=False)       # weight gradients only
            dl1 = ops.conv_dgrad(d_qkv, wd, ldd, l1.shape[1], _SPEC1, hw)
        else:
            dv = ops.bmm_nn(do3, p)
            dp = ops.bmm_tn(do3, v.view(Z, av // heads, T))
            ds = ops.softmax_bwd(p, dp, scale, out=dp)
            dq = ops.bmm_nt(k.view(Z, ai // heads, T), ds)
            dk = ops.bmm_nn(q.view(Z, ai // heads, T), ds)
            dl1 = torch.empty_like(l1)
            first = True
            for dproj, name, c in ((dq, '.attn1.to_q', ai), (dk, '.attn1.to_k', ai), (dv, '.attn1.to_v', av)):
                self._conv_bwd(tb + name, dproj.view(N, c, H, W), l1, None, _SPEC1, hw, dx_out=dl1, dx_accumulate=not first)
                first = False
        dh, pws = ops.layernorm_bwd(h, P[tb + '.norm1.weight'], ls1, dl1, add=dh2)
        self._ln_param_grads(tb + '.norm1', pws)
        return dh

    # ---- whole network ----------------------------------------------------------------------------------------
    def forward(self, x, timesteps, context, save=False, cfg_pair=False):
        """cfg_pair (no-grad only): classifier-free guidance evaluates the network on [x; x] with contexts [uncond; cond]
        (ddim.py:178-183).  Until the first SpatialTransformer the two halves are the SAME computation -- conv_in, the two
        ResBlocks and the Downsample of the full-resolution level see identical inputs and the same timestep embedding -- so
        with cfg_pair=True `x` / `timesteps` hold ONE copy (B images), `context` the 2B tokens, the context-free stem runs at
        batch B and its output and skip tensors are duplicated where the first context-dependent block starts (5.8 of the
        104.2 GMAC per latent forward, for half of the batch).  Returns the 2B outputs [uncond; cond]."""
        P, cfg = self.P, self.cfg
        if context is None or context.dim() != 3:
            raise NotImplementedError('context must be [B, L, context_dim] (cin256-v2: the one class-embedding token, L = 1)')
        if cfg_pair and (save or context.shape[0] != 2 * x.shape[0]):
            raise ValueError('cfg_pair: a no-grad forward of B images against 2B context tokens')
        if context.shape[1] == 1:                      # every configuration the reference prunes: the closed form of st_fwd
            ctx2d = context.reshape(context.shape[0], context.shape[2]).contiguous().float()
        else:                                          # L > 1 tokens: [B, L, D] -> channel-major [B, D, 1, L] for the 1x1 projections
            ctx2d = None
            self._ctx_cm = context.float().transpose(1, 2).contiguous().view(context.shape[0], context.shape[2], 1, context.shape[1])
        inp, out, mid = ldm_blocks(cfg)
        ctx = {} if save else None
        self._nograd = not save                        # F(4, 3) convolutions in the sampler's forwards only (UNetEngine._conv)
        if save:
            self.decide_overlap(x)
        x = x.contiguous()
        if cfg_pair:
            timesteps = torch.cat([timesteps, timesteps])            # 2B embedding rows; the stem reads the first B
        t_emb = ops.timestep_embedding(timesteps.to(torch.float32), cfg['model_channels'], True, 0.0)
        h1 = self._linear('time_embed.0', t_emb)
        a1 = ops.silu_fwd(h1)
        emb = self._linear('time_embed.2', a1)
        semb = ops.silu_fwd(emb)
        # no-grad forwards (the sampling loop: 20 per importance step): the 22 emb_layers projections of silu(emb) as ONE GEMM
        # against the concatenated weights, each ResBlock taking its column slice (UNetEngine.__init__, temb_batch)
        self._temb = None
        if ctx is None and self.temb_batch and hasattr(ops, 'empty_act'):
            names, offs, W_all, b_all = self._temb_pack()
            self._temb = (names, offs, ops.linear_forward(semb, W_all, b_all), W_all)
        if self._ctx_cache is not None and (ctx is not None or self._ctx_key != (context.data_ptr(), tuple(context.shape))):
            raise RuntimeError('context_cache(): a different context (or a gradient step) inside the cached block')
        hs = []
        h = x
        shared = cfg_pair                                   # still inside the context-free stem
        semb_all = semb
        if shared:
            semb = semb_all[:x.shape[0]]
        for bi, items in enumerate(inp):
            if shared and any(it[0] == 'st' for it in items):            # first context-dependent block: both halves from here
                shared = False
                semb = semb_all
                h = torch.cat([h, h])
                hs = [torch.cat([s_, s_]) for s_ in hs]
            for li, it in enumerate(items):
                pre = 'input_blocks.%d.%d' % (bi, li)
                if it[0] == 'conv_in':
                    h = self._conv(pre, h, None, _SPEC3)
                elif it[0] == 'res':
                    h = self.res_fwd(pre, h, None, semb, ctx)
                elif it[0] == 'st':
                    h = self.st_fwd(pre, h, ctx2d, it[1], ctx)
                else:
                    hin = h
                    h = self._conv(pre + '.op', hin, None, _SPEC_DOWN)
                    if ctx is not None:
                        ctx[pre] = hin
            hs.append(h)
        if shared:                                           # a configuration without attention in the encoder
            semb = semb_all
            h = torch.cat([h, h])
            hs = [torch.cat([s_, s_]) for s_ in hs]
        h = self.res_fwd('middle_block.0', h, None, semb, ctx)
        h = self.st_fwd('middle_block.1', h, ctx2d, mid, ctx)
        h = self.res_fwd('middle_block.2', h, None, semb, ctx)
        for bi, items in enumerate(out):
            skip = hs.pop()
            for li, it in enumerate(items):
                pre = 'output_blocks.%d.%d' % (bi, li)
                if it[0] == 'res':
                    h = self.res_fwd(pre, h, skip, semb, ctx)
                elif it[0] == 'st':
                    h = self.st_fwd(pre, h, ctx2d, it[1], ctx)
                else:
                    if engine_mod.UPS_SUBPIXEL:                                    # see UNetEngine._ups_conv_fwd
                        hin = h
                        h = self._ups_conv_fwd(pre + '.conv', hin)
                    else:
                        hin = ops.upsample2x(h) if engine_mod.UPS_COPY else h
                        h = self._conv(pre + '.conv', hin, None, _SPEC3 if engine_mod.UPS_COPY else _SPEC_UP)
                    if ctx is not None:
                        ctx[pre] = hin
        ho = h
        no, sto = ops.groupnorm_fwd(ho, None, P['out.0.weight'], P['out.0.bias'], 32, 1e-5, True)
        y = self._conv('out.2', no, None, _SPEC3)
        self._temb = None
        if ctx is not None:
            ctx['_head'] = (x, t_emb, h1, a1, emb, semb, ho, no, sto)
            self.ctx = ctx
        return y

    def backward(self, dout):
        P, cfg, ctx = self.P, self.cfg, self.ctx
        assert ctx is not None
        inp, out, mid = ldm_blocks(cfg)
        x, t_emb, h1, a1, emb, semb, ho, no, sto = ctx.pop('_head')
        self._begin_backward()
        d_semb = torch.zeros_like(semb)
        dno = self._conv_bwd('out.2', dout, no, None, _SPEC3, tuple(ho.shape[2:]))
        dx, pws = ops.groupnorm_bwd(ho, None, P['out.0.weight'], P['out.0.bias'], sto, dno, 32, True)
        self._gn_param_grads('out.0', pws)
        n_in = len(inp)
        sg = {}                                   # index into hs -> gradient view from the output path
        for bi in reversed(range(len(out))):
            items = out[bi]
            for li in reversed(range(len(items))):
                it = items[li]
                pre = 'output_blocks.%d.%d' % (bi, li)
                if it[0] == 'up':
                    hin = ctx.pop(pre)
                    if engine_mod.UPS_SUBPIXEL:
                        dx = self._ups_conv_bwd(pre + '.conv', dx, hin)
                    else:
                        if engine_mod.UPS_COPY:
                            dxv = self._conv_bwd(pre + '.conv', dx, hin, None, _SPEC3, (hin.shape[2], hin.shape[3]))
                        else:
                            dxv = self._conv_bwd(pre + '.conv', dx, hin, None, _SPEC_UP, (2 * hin.shape[2], 2 * hin.shape[3]))
                        dx = ops.downsum2x2(dxv)
                elif it[0] == 'st':
                    dx = self.st_bwd(pre, dx)
                else:
                    c1 = self.ctx[pre][0].shape[1]
                    dcat = self.resnet_bwd(pre, dx, semb, d_semb)
                    dx = dcat[:, :c1]
                    sg[n_in - 1 - bi] = dcat[:, c1:]          # output block bi popped hs[n_in-1-bi]
        dx = self.resnet_bwd('middle_block.2', dx, semb, d_semb)
        dx = self.st_bwd('middle_block.1', dx)
        dx = self.resnet_bwd('middle_block.0', dx, semb, d_semb, extra=sg.pop(n_in - 1))
        for bi in reversed(range(n_in)):
            items = inp[bi]
            for li in reversed(range(len(items))):
                it = items[li]
                pre = 'input_blocks.%d.%d' % (bi, li)
                extra = sg.pop(bi - 1) if (li == 0 and bi > 0) else None      # this item consumed hs[bi-1]
                if it[0] == 'st':
                    dx = self.st_bwd(pre, dx, extra=extra)
                elif it[0] == 'res':
                    dx = self.resnet_bwd(pre, dx, semb, d_semb, extra=extra)
                elif it[0] == 'down':
                    hin = ctx.pop(pre)
                    dx = self._conv_bwd(pre + '.op', dx, hin, None, _SPEC_DOWN, tuple(hin.shape[2:]), dx_add=extra)
                else:
                    self._conv_bwd(pre, dx, x, None, _SPEC3, None, need_dx=False)
        assert not sg
        d_emb = ops.silu_bwd(emb, d_semb)
        d_a1 = self._linear_bwd('time_embed.2', d_emb, a1)
        d_h1 = ops.silu_bwd(h1, d_a1)
        self._linear_bwd('time_embed.0', d_h1, t_emb, need_dx=False)
        self._end_backward()
        assert not ctx, 'unconsumed context: %s' % list(ctx)
        self.ctx = None
