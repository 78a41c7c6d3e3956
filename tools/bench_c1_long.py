#!/usr/bin/env python3
"""C1-size model (CIFAR-10 UNet, batch 4) over a long sweep: eager launches vs hipGraph replay (use_graph=True)."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
gc = importlib.import_module('diff-pruning_amd.synthetic')
unet = importlib.import_module('diff-pruning_amd.unet'); sweep = importlib.import_module('diff-pruning_amd.sweep')
diffusion = importlib.import_module('diff-pruning_amd.diffusion')
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for mode in (False, True, False, True):
    m = unet.UNet2DModel(**gc.CIFAR_CFG); gc.det_init_(m, 0); m = m.cuda().eval()
    c = torch.from_numpy(gc.det_clean((4, 3, 32, 32), 1)).cuda(); n = torch.from_numpy(gc.det_noise((4, 3, 32, 32), 2)).cuda()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = sweep.taylor_sweep(m, diffusion.DDPMScheduler(), c, n, num_steps=steps, use_graph=mode)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('C1-size sweep, %d timesteps, %s: %.2f s, %.2f ms/step, %.0f image-steps/s (capture included)'
          % (steps, 'hipGraph replay' if mode else 'eager', dt, dt / steps * 1e3, 4 * steps / dt))
