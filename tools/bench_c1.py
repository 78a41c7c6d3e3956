#!/usr/bin/env python3
"""Config C1 of BASELINE.json on the GPU: CIFAR-10 UNet, batch 4, 8 timesteps, Taylor ratio 0.3 (the reference's CPU-runnable
case: 7.1 s cold on 8 CPU threads in the survey container)."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
gc = importlib.import_module('diff-pruning_amd.synthetic')
unet = importlib.import_module('diff-pruning_amd.unet'); sweep = importlib.import_module('diff-pruning_amd.sweep')
diffusion = importlib.import_module('diff-pruning_amd.diffusion')
for rep in range(2):
    m = unet.UNet2DModel(**gc.CIFAR_CFG); gc.det_init_(m, 0); m = m.cuda().eval()
    c = torch.from_numpy(gc.det_clean((4, 3, 32, 32), 1)).cuda(); n = torch.from_numpy(gc.det_noise((4, 3, 32, 32), 2)).cuda()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = sweep.taylor_sweep(m, diffusion.DDPMScheduler(), c, n, num_steps=8)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    pr = sweep.prune_model(m, 0.3)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print('C1 run %d: sweep 8 steps %.1f ms (%.0f image-steps/s), prune %.1f ms, total %.1f ms, params %d'
          % (rep, (t1 - t0) * 1e3, 32 / (t1 - t0), (t2 - t1) * 1e3, (t2 - t0) * 1e3, sum(p.numel() for p in m.parameters())))
