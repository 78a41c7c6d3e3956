"""ORACLE (test infrastructure only) -- CPU restatement of the reference UNet2DModel forward.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product path (diff-pruning_amd/) never does.  Parity pinned: checked against outputs of the
reference itself (tests/golden/*.npz|json produced by tests/golden/make_golden.py, which imports
/root/reference in the build container).

The model is restated *functionally* over a flat {diffusers_state_dict_key: tensor} dictionary; all
channel counts are read off the tensors, so the same code evaluates un-pruned and pruned networks.
Plain PyTorch fp32 ops on whatever device the tensors live on (CPU in practice); autograd supplies
the backward pass, exactly as in the reference.

Reference lines followed (relative to /root/reference/diffusers/models):
  unet_2d.py:219-316              overall forward (time -> conv_in -> down -> mid -> up -> out)
  embeddings.py:22-62,200-212     sinusoidal embedding, TimestepEmbedding MLP
  resnet.py:589-639               ResnetBlock2D.forward
  resnet.py:131-166,206-220       Upsample2D / Downsample2D forward
  attention_processor.py:415-470  AttnProcessor (legacy baddbmm/softmax/bmm path, heads == 1)
  unet_2d_blocks.py:465-472,749-762,973-994,1817-1831,2030-2060  block wrappers
"""
import math
import torch
import torch.nn.functional as F


def timestep_embedding(timesteps, dim, flip_sin_to_cos, freq_shift, max_period=10000):
    """embeddings.py:22-62."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half, dtype=torch.float32, device=timesteps.device)
    exponent = exponent / (half - freq_shift)
    arg = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(arg), torch.cos(arg)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    if dim % 2 == 1:
        emb = F.pad(emb, (0, 1, 0, 0))
    return emb


def _gn(P, pre, x, groups, eps):
    return F.group_norm(x, groups, P[pre + '.weight'], P[pre + '.bias'], eps)


def _conv(P, pre, x, stride=1, padding=1):
    return F.conv2d(x, P[pre + '.weight'], P.get(pre + '.bias'), stride=stride, padding=padding)


def _lin(P, pre, x):
    return F.linear(x, P[pre + '.weight'], P.get(pre + '.bias'))


def resnet_block(P, pre, x, temb, groups, eps, out_scale, drop=None):
    """resnet.py:589-639 with time_embedding_norm == 'default', no up/down.  drop: philox_ref.DropSpec (training mode with
    nn.Dropout p > 0, resnet.py:628) or None (eval mode / p = 0)."""
    h = F.silu(_gn(P, pre + '.norm1', x, groups, eps))
    h = _conv(P, pre + '.conv1', h)
    t = _lin(P, pre + '.time_emb_proj', F.silu(temb))[:, :, None, None]
    h = h + t
    h = F.silu(_gn(P, pre + '.norm2', h, groups, eps))
    if drop is not None:
        h = drop.apply(pre + '.dropout', h)
    h = _conv(P, pre + '.conv2', h)
    if (pre + '.conv_shortcut.weight') in P:
        x = _conv(P, pre + '.conv_shortcut', x, padding=0)
    return (x + h) / out_scale


def attention_block(P, pre, x, groups, eps, scale, rescale, heads=1, drop=None):
    """attention_processor.py:415-470 (AttnProcessor; head_to_batch_dim / batch_to_head_dim :283-305), residual_connection=True.

    `scale` and `heads` are module attributes fixed at construction (dim_head ** -0.5 with dim_head = attention_head_dim
    or the *un-pruned* channel count, heads = channels // attention_head_dim; attention_processor.py:85-86,
    unet_2d_blocks.py:722-723); the inner width comes from to_q, so after pruning a head has inner // heads channels.
    heads > 1 is pinned by tests/golden/tiny_heads.npz."""
    B, C, H, W = x.shape
    res = x
    h = x.view(B, C, H * W).transpose(1, 2)
    h = F.group_norm(h.transpose(1, 2), groups, P[pre + '.group_norm.weight'], P[pre + '.group_norm.bias'],
                     eps).transpose(1, 2)
    q = _lin(P, pre + '.to_q', h)
    k = _lin(P, pre + '.to_k', h)
    v = _lin(P, pre + '.to_v', h)
    T = q.shape[1]
    if heads > 1:
        def to_batch(t):
            return t.reshape(B, T, heads, t.shape[-1] // heads).permute(0, 2, 1, 3).reshape(B * heads, T, t.shape[-1] // heads)
        q, k, v = to_batch(q), to_batch(k), to_batch(v)
    s = torch.baddbmm(torch.empty(q.shape[0], q.shape[1], k.shape[1], dtype=q.dtype, device=q.device), q,
                      k.transpose(-1, -2), beta=0, alpha=scale)
    p = s.float().softmax(dim=-1).to(q.dtype)
    h = torch.bmm(p, v)
    if heads > 1:
        h = h.reshape(B, heads, T, h.shape[-1]).permute(0, 2, 1, 3).reshape(B, T, heads * h.shape[-1])
    h = _lin(P, pre + '.to_out.0', h)
    h = h.transpose(-1, -2).reshape(B, C, H, W)
    if drop is not None:          # to_out[1] (attention_processor.py:457); elementwise, so it commutes with the reshape
        h = drop.apply(pre + '.to_out.1', h)
    return (h + res) / rescale


def downsample(P, pre, x, padding):
    """resnet.py:206-220: asymmetric zero pad when padding == 0, then 3x3 stride-2 conv."""
    if padding == 0:
        x = F.pad(x, (0, 1, 0, 1), mode='constant', value=0)
    return _conv(P, pre + '.conv', x, stride=2, padding=padding)


def upsample(P, pre, x):
    """resnet.py:131-166: nearest x2 then 3x3 conv."""
    x = F.interpolate(x, scale_factor=2.0, mode='nearest')
    return _conv(P, pre + '.conv', x)


def attn_scale_for(cfg, channels):
    hd = cfg.get('attention_head_dim')
    dim_head = hd if hd is not None else channels
    return float(dim_head) ** -0.5


def attn_heads_for(cfg, channels):
    hd = cfg.get('attention_head_dim')
    return channels // hd if hd is not None else 1


def unet_forward(P, cfg, sample, timesteps, drop=None):
    """unet_2d.py:219-316.  P: flat parameter dict, cfg: Diffusers UNet2DModel config dict.  drop: see resnet_block."""
    boc = list(cfg['block_out_channels'])
    groups, eps = cfg['norm_num_groups'], cfg['norm_eps']
    L = cfg['layers_per_block']
    if cfg.get('center_input_sample', False):
        sample = 2 * sample - 1.0
    if not torch.is_tensor(timesteps):
        timesteps = torch.tensor([timesteps], dtype=torch.long, device=sample.device)
    elif timesteps.dim() == 0:
        timesteps = timesteps[None].to(sample.device)
    timesteps = timesteps * torch.ones(sample.shape[0], dtype=timesteps.dtype, device=timesteps.device)
    t_emb = timestep_embedding(timesteps, boc[0], cfg['flip_sin_to_cos'], cfg['freq_shift'])
    emb = _lin(P, 'time_embedding.linear_2', F.silu(_lin(P, 'time_embedding.linear_1', t_emb)))

    x = _conv(P, 'conv_in', sample)
    skips = [x]
    nb = len(boc)
    for i, bt in enumerate(cfg['down_block_types']):
        pre = 'down_blocks.%d' % i
        for j in range(L):
            x = resnet_block(P, '%s.resnets.%d' % (pre, j), x, emb, groups, eps, 1.0, drop)
            if bt == 'AttnDownBlock2D':
                x = attention_block(P, '%s.attentions.%d' % (pre, j), x, groups, eps,
                                    attn_scale_for(cfg, boc[i]), 1.0, attn_heads_for(cfg, boc[i]), drop)
            skips.append(x)
        if i != nb - 1:
            x = downsample(P, pre + '.downsamplers.0', x, cfg['downsample_padding'])
            skips.append(x)

    msf = float(cfg.get('mid_block_scale_factor', 1))
    x = resnet_block(P, 'mid_block.resnets.0', x, emb, groups, eps, msf, drop)
    if cfg.get('add_attention', True):
        x = attention_block(P, 'mid_block.attentions.0', x, groups, eps, attn_scale_for(cfg, boc[-1]), msf,
                            attn_heads_for(cfg, boc[-1]), drop)
    x = resnet_block(P, 'mid_block.resnets.1', x, emb, groups, eps, msf, drop)

    rev = list(reversed(boc))
    for i, bt in enumerate(cfg['up_block_types']):
        pre = 'up_blocks.%d' % i
        for j in range(L + 1):
            x = torch.cat([x, skips.pop()], dim=1)
            x = resnet_block(P, '%s.resnets.%d' % (pre, j), x, emb, groups, eps, 1.0, drop)
            if bt == 'AttnUpBlock2D':
                x = attention_block(P, '%s.attentions.%d' % (pre, j), x, groups, eps,
                                    attn_scale_for(cfg, rev[i]), 1.0, attn_heads_for(cfg, rev[i]), drop)
        if i != nb - 1:
            x = upsample(P, pre + '.upsamplers.0', x)

    x = F.silu(_gn(P, 'conv_norm_out', x, groups, eps))
    return _conv(P, 'conv_out', x)


def param_shapes(cfg):
    """Shapes of every parameter of an un-pruned UNet2DModel(**cfg), in Diffusers state_dict order
    (unet_2d.py:84-217 construction order)."""
    boc = list(cfg['block_out_channels'])
    L = cfg['layers_per_block']
    tdim = boc[0] * 4
    S = {}

    def conv(n, ci, co, k):
        S[n + '.weight'] = (co, ci, k, k)
        S[n + '.bias'] = (co,)

    def lin(n, ci, co):
        S[n + '.weight'] = (co, ci)
        S[n + '.bias'] = (co,)

    def gn(n, c):
        S[n + '.weight'] = (c,)
        S[n + '.bias'] = (c,)

    def resnet(n, ci, co):
        gn(n + '.norm1', ci)
        conv(n + '.conv1', ci, co, 3)
        lin(n + '.time_emb_proj', tdim, co)
        gn(n + '.norm2', co)
        conv(n + '.conv2', co, co, 3)
        if ci != co:
            conv(n + '.conv_shortcut', ci, co, 1)

    def attn(n, c):
        hd = cfg.get('attention_head_dim')
        inner = c if hd is None else (c // hd) * hd
        gn(n + '.group_norm', c)
        lin(n + '.to_q', c, inner)
        lin(n + '.to_k', c, inner)
        lin(n + '.to_v', c, inner)
        lin(n + '.to_out.0', inner, c)

    conv('conv_in', cfg['in_channels'], boc[0], 3)
    lin('time_embedding.linear_1', boc[0], tdim)
    lin('time_embedding.linear_2', tdim, tdim)
    out_c = boc[0]
    for i, bt in enumerate(cfg['down_block_types']):
        in_c, out_c = out_c, boc[i]
        pre = 'down_blocks.%d' % i
        # Diffusers registers `attentions` before `resnets` in Attn blocks, and resnets first otherwise;
        # ordering inside this dict is irrelevant to the oracle (lookups are by key).
        for j in range(L):
            resnet('%s.resnets.%d' % (pre, j), in_c if j == 0 else out_c, out_c)
            if bt == 'AttnDownBlock2D':
                attn('%s.attentions.%d' % (pre, j), out_c)
        if i != len(boc) - 1:
            conv(pre + '.downsamplers.0.conv', out_c, out_c, 3)
    resnet('mid_block.resnets.0', boc[-1], boc[-1])
    if cfg.get('add_attention', True):
        attn('mid_block.attentions.0', boc[-1])
    resnet('mid_block.resnets.1', boc[-1], boc[-1])
    rev = list(reversed(boc))
    out_c = rev[0]
    for i, bt in enumerate(cfg['up_block_types']):
        prev, out_c = out_c, rev[i]
        in_c = rev[min(i + 1, len(boc) - 1)]
        pre = 'up_blocks.%d' % i
        for j in range(L + 1):
            res_skip = in_c if j == L else out_c
            res_in = prev if j == 0 else out_c
            resnet('%s.resnets.%d' % (pre, j), res_in + res_skip, out_c)
            if bt == 'AttnUpBlock2D':
                attn('%s.attentions.%d' % (pre, j), out_c)
        if i != len(boc) - 1:
            conv(pre + '.upsamplers.0.conv', out_c, out_c, 3)
    gn('conv_norm_out', boc[0])
    conv('conv_out', boc[0], cfg['out_channels'], 3)
    return S
