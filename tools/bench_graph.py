#!/usr/bin/env python3
"""Eager launches vs hipGraph replay of one sweep timestep (CIFAR UNet), per batch size."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
gc = importlib.import_module('diff-pruning_amd.synthetic')
unet = importlib.import_module('diff-pruning_amd.unet'); sweep = importlib.import_module('diff-pruning_amd.sweep')
diffusion = importlib.import_module('diff-pruning_amd.diffusion')
for B in (4, 16, 64, 256):
    res = {}
    for mode in ('eager', 'graph'):
        m = unet.UNet2DModel(**gc.CIFAR_CFG); gc.det_init_(m, 0); m = m.cuda().eval()
        c = torch.from_numpy(gc.det_clean((B, 3, 32, 32), 1)).cuda(); n = torch.from_numpy(gc.det_noise((B, 3, 32, 32), 2)).cuda()
        sweep.flatten_grads(m)
        st = sweep.HipSweepStep(m, diffusion.DDPMScheduler(), c, n, B * c[0].numel(), 'mse', B)
        st(0); st(1)
        if mode == 'graph':
            st.capture()
        st(2); torch.cuda.synchronize(); t0 = time.perf_counter()
        K = 10
        for k in range(K): st(3 + k)
        torch.cuda.synchronize(); res[mode] = (time.perf_counter() - t0) / K * 1e3
    print('B=%d eager %.2f ms/step  graph %.2f ms/step  (%.0f vs %.0f img-steps/s)' % (B, res['eager'], res['graph'], B / res['eager'] * 1e3, B / res['graph'] * 1e3), flush=True)
