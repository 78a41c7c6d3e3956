"""ORACLE (test infrastructure only) -- CPU restatement of the LDM (CompVis) UNet and of the LDM importance pass.

Parity: the UNet forward/backward is PINNED against the reference's own `UNetModel` (importable with a 3-line
omegaconf stub; fixtures tests/golden/ldm_unet.npz from tests/golden/make_golden_ldm.py).
The noise schedule and the CFG DDIM sampler are PINNED against the reference's own `DDIMSampler` + `make_beta_schedule`
(both import cleanly) driven over the reference UNetModel: tests/golden/ldm_sampler.npz (make_golden_ldm.py sampler).
get_loss_at_t / p_losses / q_sample / get_learned_conditioning and ClassEmbedder are PINNED against the reference's own
`LatentDiffusion` methods (tests/golden/ldm_loss_at_t.npz, make_golden_ldm.py loss: `ddpm.py` imports with empty stand-in
modules for the absent pytorch_lightning / torchvision / taming / clip / kornia, none of whose code is on this path; the
object's DDPM.__init__ really runs over the reference DiffusionWrapper + UNetModel).  The for-loop of the prune_ldm.py
SCRIPT (lines 103-131: class draw, CFG sampling, loss at t, max-loss bookkeeping, threshold test before backward) is module-
level code, not an importable function; it is PINNED by tests/golden/ldm_driver.json, for which make_golden_ldm.py `driver`
EXECUTES those source lines (read from the reference file at generation time) over that LatentDiffusion object and the
reference DDIMSampler with replayable draws -- a 1000-iteration run that never reaches the threshold and a second run
constructed to break at t = 2.

Reference lines followed (relative to /root/reference/ldm_exp):
  ldm/modules/diffusionmodules/openaimodel.py:710-742      UNetModel.forward
  ldm/modules/diffusionmodules/openaimodel.py:236-275      ResBlock._forward
  ldm/modules/diffusionmodules/openaimodel.py:95-160       Upsample / Downsample
  ldm/modules/attention.py:37-66,152-258                   GEGLU, FeedForward, CrossAttention, BasicTransformerBlock, SpatialTransformer
  ldm/modules/diffusionmodules/util.py:21-25,151-171       beta schedule ("linear" = linspace(sqrt)**2, fp64), timestep_embedding
  ldm/models/diffusion/ddpm.py:881-889,1024-1056           get_loss_at_t / p_losses (eps-parameterisation, l2, logvar = 0)
  ldm/models/diffusion/ddim.py:165-203                     p_sample_ddim with classifier-free guidance
  prune_ldm.py:101-132                                     the importance-pass driver
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def timestep_embedding(timesteps, dim, max_period=10000):
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def _conv(P, n, x, stride=1, padding=1):
    return F.conv2d(x, P[n + '.weight'], P.get(n + '.bias'), stride=stride, padding=padding)


def _lin(P, n, x):
    return F.linear(x, P[n + '.weight'], P.get(n + '.bias'))


def res_block(P, pre, x, emb):
    h = _conv(P, pre + '.in_layers.2', F.silu(F.group_norm(x, 32, P[pre + '.in_layers.0.weight'], P[pre + '.in_layers.0.bias'], 1e-5)))
    e = _lin(P, pre + '.emb_layers.1', F.silu(emb))
    h = h + e[:, :, None, None]
    h = F.silu(F.group_norm(h, 32, P[pre + '.out_layers.0.weight'], P[pre + '.out_layers.0.bias'], 1e-5))
    h = _conv(P, pre + '.out_layers.3', h)
    if (pre + '.skip_connection.weight') in P:
        x = _conv(P, pre + '.skip_connection', x, padding=0)
    return x + h


def cross_attention(P, pre, x, context, scale, heads=1):
    """attention.py:168-193: 'b n (h d) -> (b h) n d', scaled dot-product per head, heads concatenated again."""
    q = _lin(P, pre + '.to_q', x)
    ctx = x if context is None else context
    k = _lin(P, pre + '.to_k', ctx)
    v = _lin(P, pre + '.to_v', ctx)

    def split(t):                                    # rearrange(t, 'b n (h d) -> (b h) n d', h=heads)
        b, n, hd = t.shape
        return t.reshape(b, n, heads, hd // heads).permute(0, 2, 1, 3).reshape(b * heads, n, hd // heads)
    q, k, v = split(q), split(k), split(v)
    sim = torch.einsum('bid,bjd->bij', q, k) * scale
    attn = sim.softmax(dim=-1)
    out = torch.einsum('bij,bjd->bid', attn, v)
    bh, n, d = out.shape                             # rearrange(out, '(b h) n d -> b n (h d)', h=heads)
    out = out.reshape(bh // heads, heads, n, d).permute(0, 2, 1, 3).reshape(bh // heads, n, heads * d)
    return _lin(P, pre + '.to_out.0', out)


def st_heads(cfg, ch):
    """(heads, dim_head) of the SpatialTransformer of a `ch`-channel level (openaimodel.py:542-549, legacy=True with
    use_spatial_transformer: num_heads fixed, or ch // num_head_channels heads; dim_head = ch // heads)."""
    nhc = cfg.get('num_head_channels', -1)
    heads = cfg.get('num_heads', 1) if nhc in (-1, None) else ch // nhc
    return heads, ch // heads


def spatial_transformer(P, pre, x, context, dim_head, heads=1, depth=1):
    """attention.py:218-258.  `dim_head` = the UN-pruned head width of the block (CrossAttention.scale is fixed at
    construction: attention.py:158); `depth` BasicTransformerBlocks (attention.py:196-212) between proj_in and proj_out."""
    b, c, h, w = x.shape
    x_in = x
    x = F.group_norm(x, 32, P[pre + '.norm.weight'], P[pre + '.norm.bias'], 1e-6)
    x = _conv(P, pre + '.proj_in', x, padding=0)
    inner = x.shape[1]
    x = x.reshape(b, inner, h * w).transpose(1, 2)
    scale = float(dim_head) ** -0.5
    for d in range(depth):
        tb = pre + '.transformer_blocks.%d' % d
        x = cross_attention(P, tb + '.attn1', F.layer_norm(x, (inner,), P[tb + '.norm1.weight'], P[tb + '.norm1.bias']), None, scale, heads) + x
        x = cross_attention(P, tb + '.attn2', F.layer_norm(x, (inner,), P[tb + '.norm2.weight'], P[tb + '.norm2.bias']), context, scale, heads) + x
        y = F.layer_norm(x, (inner,), P[tb + '.norm3.weight'], P[tb + '.norm3.bias'])
        a, gate = _lin(P, tb + '.ff.net.0.proj', y).chunk(2, dim=-1)
        x = _lin(P, tb + '.ff.net.2', a * F.gelu(gate)) + x
    x = x.transpose(1, 2).reshape(b, inner, h, w)
    return _conv(P, pre + '.proj_out', x, padding=0) + x_in


def ldm_blocks(cfg):
    """Block structure of UNetModel(**cfg) (openaimodel.py:517-692): list of input blocks, each a list of
    ('res'|'st'|'down'|'conv_in', ...) items; middle; output blocks."""
    mc, mult, nres = cfg['model_channels'], list(cfg['channel_mult']), cfg['num_res_blocks']
    att = set(cfg['attention_resolutions'])
    inp = [[('conv_in',)]]
    ds = 1
    for level, m in enumerate(mult):
        for _ in range(nres):
            items = [('res',)]
            if ds in att:
                items.append(('st', m * mc))
            inp.append(items)
        if level != len(mult) - 1:
            inp.append([('down',)])
            ds *= 2
    mid_ch = mult[-1] * mc
    out = []
    for level, m in list(enumerate(mult))[::-1]:
        for i in range(nres + 1):
            items = [('res',)]
            if ds in att:
                items.append(('st', m * mc))
            if level and i == nres:
                items.append(('up',))
                ds //= 2
            out.append(items)
    return inp, out, mid_ch


def ldm_unet_forward(P, cfg, x, timesteps, context):
    """openaimodel.py:710-742 (num_classes None, use_spatial_transformer True)."""
    inp, out, mid_ch = ldm_blocks(cfg)
    depth = cfg.get('transformer_depth', 1)
    emb = _lin(P, 'time_embed.2', F.silu(_lin(P, 'time_embed.0', timestep_embedding(timesteps, cfg['model_channels']))))
    hs = []
    h = x
    for bi, items in enumerate(inp):
        for li, it in enumerate(items):
            pre = 'input_blocks.%d.%d' % (bi, li)
            if it[0] == 'conv_in':
                h = _conv(P, pre, h)
            elif it[0] == 'res':
                h = res_block(P, pre, h, emb)
            elif it[0] == 'st':
                h = spatial_transformer(P, pre, h, context, st_heads(cfg, it[1])[1], st_heads(cfg, it[1])[0], depth)
            elif it[0] == 'down':
                h = _conv(P, pre + '.op', h, stride=2, padding=1)
        hs.append(h)
    h = res_block(P, 'middle_block.0', h, emb)
    h = spatial_transformer(P, 'middle_block.1', h, context, st_heads(cfg, mid_ch)[1], st_heads(cfg, mid_ch)[0], depth)
    h = res_block(P, 'middle_block.2', h, emb)
    for bi, items in enumerate(out):
        h = torch.cat([h, hs.pop()], dim=1)
        for li, it in enumerate(items):
            pre = 'output_blocks.%d.%d' % (bi, li)
            if it[0] == 'res':
                h = res_block(P, pre, h, emb)
            elif it[0] == 'st':
                h = spatial_transformer(P, pre, h, context, st_heads(cfg, it[1])[1], st_heads(cfg, it[1])[0], depth)
            elif it[0] == 'up':
                h = _conv(P, pre + '.conv', F.interpolate(h, scale_factor=2, mode='nearest'))
    h = F.silu(F.group_norm(h, 32, P['out.0.weight'], P['out.0.bias'], 1e-5))
    return _conv(P, 'out.2', h)


def ldm_param_shapes(cfg):
    """Parameter shapes of an un-pruned UNetModel(**cfg) keyed like its state_dict."""
    mc, mult, nres = cfg['model_channels'], list(cfg['channel_mult']), cfg['num_res_blocks']
    cdim, tdim = cfg['context_dim'], cfg['model_channels'] * 4
    S = {}

    def conv(n, ci, co, k):
        S[n + '.weight'] = (co, ci, k, k)
        S[n + '.bias'] = (co,)

    def lin(n, ci, co, bias=True):
        S[n + '.weight'] = (co, ci)
        if bias:
            S[n + '.bias'] = (co,)

    def norm(n, c):
        S[n + '.weight'] = (c,)
        S[n + '.bias'] = (c,)

    def res(n, ci, co):
        norm(n + '.in_layers.0', ci)
        conv(n + '.in_layers.2', ci, co, 3)
        lin(n + '.emb_layers.1', tdim, co)
        norm(n + '.out_layers.0', co)
        conv(n + '.out_layers.3', co, co, 3)
        if ci != co:
            conv(n + '.skip_connection', ci, co, 1)

    def st(n, c):
        norm(n + '.norm', c)
        conv(n + '.proj_in', c, c, 1)
        for d in range(cfg.get('transformer_depth', 1)):
            tb = n + '.transformer_blocks.%d' % d
            for a, kd in (('attn1', c), ('attn2', cdim)):
                lin(tb + '.%s.to_q' % a, c, c, False)
                lin(tb + '.%s.to_k' % a, kd, c, False)
                lin(tb + '.%s.to_v' % a, kd, c, False)
                lin(tb + '.%s.to_out.0' % a, c, c)
            lin(tb + '.ff.net.0.proj', c, 8 * c)
            lin(tb + '.ff.net.2', 4 * c, c)
            for k in ('norm1', 'norm2', 'norm3'):
                norm(tb + '.' + k, c)
        conv(n + '.proj_out', c, c, 1)

    lin('time_embed.0', mc, tdim)
    lin('time_embed.2', tdim, tdim)
    inp, out, _ = ldm_blocks(cfg)
    chans = []
    ch = mc
    level = 0
    for bi, items in enumerate(inp):
        for li, it in enumerate(items):
            pre = 'input_blocks.%d.%d' % (bi, li)
            if it[0] == 'conv_in':
                conv(pre, cfg['in_channels'], mc, 3)
            elif it[0] == 'res':
                co = mult[level] * mc
                res(pre, ch, co)
                ch = co
            elif it[0] == 'st':
                st(pre, ch)
            elif it[0] == 'down':
                conv(pre + '.op', ch, ch, 3)
                level += 1
        chans.append(ch)
    res('middle_block.0', ch, ch)
    st('middle_block.1', ch)
    res('middle_block.2', ch, ch)
    level = len(mult) - 1
    for bi, items in enumerate(out):
        ich = chans.pop()
        for li, it in enumerate(items):
            pre = 'output_blocks.%d.%d' % (bi, li)
            if it[0] == 'res':
                co = mc * mult[level]
                res(pre, ch + ich, co)
                ch = co
            elif it[0] == 'st':
                st(pre, ch)
            elif it[0] == 'up':
                conv(pre + '.conv', ch, ch, 3)
                level -= 1
    norm('out.0', ch)
    conv('out.2', mc, cfg['out_channels'], 3)
    return S


# ------------------------------------------------------------------------------------------------------
# LatentDiffusion pieces (schedule + sampler pinned by ldm_sampler.npz; q_sample / loss-at-t pinned by ldm_loss_at_t.npz)
# ------------------------------------------------------------------------------------------------------
def ldm_alphas_cumprod(n_timestep=1000, linear_start=0.0015, linear_end=0.0195):
    """util.py:21-25 ('linear') + ddpm.py register_schedule: fp64 tables, cast to fp32."""
    betas = (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=torch.float64) ** 2).numpy()
    return np.cumprod(1.0 - betas, axis=0)


def q_sample(acp64, x0, t, noise):
    sa = torch.from_numpy(np.sqrt(acp64)).float()[t][:, None, None, None]
    sb = torch.from_numpy(np.sqrt(1.0 - acp64)).float()[t][:, None, None, None]
    return sa * x0 + sb * noise


def ldm_loss_at_t(P, cfg, acp64, x0, t, context, noise):
    """ddpm.py:1024-1056 with parameterization eps, l2, learn_logvar False (logvar = 0), l_simple_weight 1,
    original_elbo_weight 0: loss = mean_B(mean_CHW((eps - eps_hat)^2)) (prune_ldm.py:122-123 takes loss[0])."""
    out = ldm_unet_forward(P, cfg, q_sample(acp64, x0, t, noise), t, context)
    return (out - noise).square().mean(dim=(1, 2, 3)).mean()


def ddim_schedule(acp64, S=20, T=1000, eta=0.0):
    """ddim.py make_schedule with ddim_discretize='uniform': steps = arange(0, T, T//S) + 1."""
    c = T // S
    steps = np.asarray(list(range(0, T, c))) + 1
    a = acp64[steps]
    a_prev = np.asarray([acp64[0]] + acp64[steps[:-1]].tolist())
    sig = eta * np.sqrt((1 - a_prev) / (1 - a) * (1 - a / a_prev))
    return steps, a, a_prev, sig


@torch.no_grad()
def ddim_sample_cfg(P, cfg, acp64, x_T, cond, uncond, S=20, scale=3.0):
    """ddim.py:165-203 (eta = 0): batch doubled [uncond; cond], e = e_u + scale (e_c - e_u)."""
    steps, a, a_prev, sig = ddim_schedule(acp64, S)
    x = x_T
    B = x.shape[0]
    for i in reversed(range(len(steps))):
        t = torch.full((B,), int(steps[i]), dtype=torch.long)
        e = ldm_unet_forward(P, cfg, torch.cat([x, x]), torch.cat([t, t]), torch.cat([uncond, cond]))
        e_u, e_c = e.chunk(2)
        e_t = e_u + scale * (e_c - e_u)
        a_t, ap = float(a[i]), float(a_prev[i])
        pred_x0 = (x - math.sqrt(1 - a_t) * e_t) / math.sqrt(a_t)
        x = math.sqrt(ap) * pred_x0 + math.sqrt(1 - ap - float(sig[i]) ** 2) * e_t
    return x
