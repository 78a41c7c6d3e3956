#!/usr/bin/env python3
"""Golden vectors for the LDM (CompVis) UNet -- build container only; imports the reference's own UNetModel
(ldm_exp/ldm/modules/diffusionmodules/openaimodel.py) through the omegaconf stub of SURVEY.md App. E.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_ldm.py

Writes tests/golden/ldm_unet.npz (reduced-width config: forward output, loss, selected full gradients) and
ldm_unet_stats.json (per-parameter gradient statistics, parameter count of the full cin256-v2 config)."""
import json, os, sys, types
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import numpy as np
import torch
import golden_common as gc

stub = types.ModuleType('omegaconf.listconfig'); stub.ListConfig = type('ListConfig', (list,), {})
oc = types.ModuleType('omegaconf'); oc.listconfig = stub
sys.modules.setdefault('omegaconf', oc); sys.modules.setdefault('omegaconf.listconfig', stub)
sys.path.insert(0, '/root/reference/ldm_exp')
from ldm.modules.diffusionmodules.openaimodel import UNetModel      # noqa: E402 (reference)

torch.set_num_threads(8)


def run(cfg, seed, B, tag):
    m = UNetModel(**cfg).eval()
    gc.det_init_(m, seed)
    H = cfg['image_size']
    x = torch.from_numpy(gc.det_noise((B, cfg['in_channels'], H, H), 31))
    ctx = torch.from_numpy(gc.det_noise((B, 1, cfg['context_dim']), 32))
    noise = torch.from_numpy(gc.det_noise((B, cfg['in_channels'], H, H), 33))
    t = torch.tensor([7, 640][:B])
    y = m(x, t, context=ctx)
    loss = (y - noise).square().mean(dim=(1, 2, 3)).mean()
    loss.backward()
    P = dict(m.named_parameters())
    stats = {n: [float(p.grad.double().sum()), float(p.grad.double().abs().sum())] for n, p in P.items()}
    full = {}
    for n in ['input_blocks.0.0.weight', 'input_blocks.4.1.transformer_blocks.0.attn1.to_q.weight',
              'input_blocks.4.1.transformer_blocks.0.attn2.to_v.weight', 'input_blocks.4.1.transformer_blocks.0.ff.net.0.proj.weight',
              'input_blocks.4.1.transformer_blocks.0.norm2.weight', 'input_blocks.4.1.proj_out.weight', 'input_blocks.3.0.op.weight',
              'middle_block.1.transformer_blocks.0.ff.net.2.weight', 'output_blocks.2.1.conv.weight',
              'output_blocks.5.0.skip_connection.weight', 'time_embed.0.weight', 'out.2.bias']:
        if n in P:
            full['grad::' + n] = P[n].grad.numpy().copy()
    np.savez(os.path.join(HERE, 'ldm_unet%s.npz' % tag), fwd_out=y.detach().numpy(), loss=np.array(float(loss)), **full)
    return stats, {n: list(p.shape) for n, p in P.items()}


if __name__ == '__main__':
    stats, shapes = run(gc.LDM_TINY_CFG, 9, 2, '')
    full = UNetModel(**gc.LDM_CIN256_CFG)
    json.dump(dict(grad_stats=stats, shapes=shapes, cin256_params=sum(p.numel() for p in full.parameters()),
                   cin256_shapes={n: list(p.shape) for n, p in full.named_parameters()}),
              open(os.path.join(HERE, 'ldm_unet_stats.json'), 'w'))
    print('ok', len(stats))
