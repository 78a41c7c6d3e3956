#!/usr/bin/env python3
"""Where the host time of a latency-bound sweep goes: cProfile of 40 timesteps of config C1 (CIFAR UNet, batch 4), sorted by
self time.  The GPU is idle most of such a step (12 ms of kernels in a 16-20 ms step): this is the Python / ctypes side."""
import cProfile, importlib, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
gc = importlib.import_module('diff-pruning_amd.synthetic')
unet = importlib.import_module('diff-pruning_amd.unet'); sweep = importlib.import_module('diff-pruning_amd.sweep')
diffusion = importlib.import_module('diff-pruning_amd.diffusion')
m = unet.UNet2DModel(**gc.CIFAR_CFG); gc.det_init_(m, 0); m = m.cuda().eval()
c = torch.from_numpy(gc.det_clean((4, 3, 32, 32), 1)).cuda(); n = torch.from_numpy(gc.det_noise((4, 3, 32, 32), 2)).cuda()
sweep.taylor_sweep(m, diffusion.DDPMScheduler(), c, n, num_steps=8)
torch.cuda.synchronize(); t0 = time.perf_counter()
sweep.taylor_sweep(m, diffusion.DDPMScheduler(), c, n, num_steps=40)
torch.cuda.synchronize(); print('40 steps un-profiled: %.2f ms per step' % ((time.perf_counter() - t0) / 40 * 1e3))
pr = cProfile.Profile(); pr.enable()
sweep.taylor_sweep(m, diffusion.DDPMScheduler(), c, n, num_steps=40)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(45)
