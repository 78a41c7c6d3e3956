#!/usr/bin/env python3
"""Golden vectors for the original-DDPM checkpoint layout (run in the build container only; imports the reference's second
model implementation, ddpm_exp/models/diffusion.py:191-341, which needs no shim).

Writes ddpm_original.npz: the forward output of the reference `Model` on seeded inputs, with weights generated from the
ORIGINAL parameter names by golden_common.det_param -- so the test can rebuild the identical weights without the reference,
convert them with checkpoint.convert_ddpm_original and compare -- and ddpm_original.json: the reference's key -> shape table."""
import json
import os
import sys
from types import SimpleNamespace as NS

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, '/root/reference/ddpm_exp')
import golden_common as gc                     # noqa: E402
from models.diffusion import Model             # noqa: E402

CFG = dict(ch=32, ch_mult=[1, 2, 2, 2], num_res_blocks=2, attn_resolutions=[8], image_size=16)


def main():
    config = NS(model=NS(type='simple', in_channels=3, out_ch=3, ch=CFG['ch'], ch_mult=CFG['ch_mult'],
                         num_res_blocks=CFG['num_res_blocks'], attn_resolutions=CFG['attn_resolutions'], dropout=0.0,
                         resamp_with_conv=True),
                data=NS(image_size=CFG['image_size']), diffusion=NS(num_diffusion_timesteps=1000))
    model = Model(config).eval()
    shapes = {}
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.copy_(torch.from_numpy(gc.det_param(n, tuple(p.shape), 21)))
            shapes[n] = list(p.shape)
    x = torch.from_numpy(gc.det_noise((2, 3, 16, 16), 31))
    t = torch.tensor([3, 500])
    with torch.no_grad():
        y = model(x, t)
    np.savez(os.path.join(HERE, 'ddpm_original.npz'), out=y.numpy())
    # the ddpm_exp flavour of the Diff-Pruning sweep (ddpm_exp/prune.py:236-258 with functions/losses.py:9-15): threshold
    # test BEFORE backward, loss = sum over C,H,W / mean over the batch, timesteps passed as float
    from functions.losses import noise_estimation_loss
    x0 = torch.from_numpy(gc.det_clean((2, 3, 16, 16), 41))
    e = torch.from_numpy(gc.det_noise((2, 3, 16, 16), 42))
    betas = torch.linspace(1e-4, 0.02, 1000)
    model.zero_grad()
    max_loss, losses, thr = 0, [], 0.99
    for step_k in range(1000):
        tt = torch.ones(2, dtype=torch.long) * step_k
        loss = noise_estimation_loss(model, x0, tt, e, betas)
        losses.append(float(loss))
        if loss > max_loss:
            max_loss = loss
        if loss < max_loss * thr:
            break
        loss.backward()
    gstats = {n: [float(p.grad.double().sum()), float(p.grad.double().abs().sum())] for n, p in model.named_parameters()}
    json.dump(dict(cfg=CFG, shapes=shapes, seed=21, input_seed=31, timesteps=[3, 500],
                   sweep=dict(thr=thr, clean_seed=41, noise_seed=42, losses=losses, grad_stats=gstats)),
              open(os.path.join(HERE, 'ddpm_original.json'), 'w'))
    print('twin sweep: %d losses (last step breaks before backward)' % len(losses))
    print('ddpm_original ok:', len(shapes), 'tensors, out', tuple(y.shape), float(y.abs().mean()))


if __name__ == '__main__':
    main()
