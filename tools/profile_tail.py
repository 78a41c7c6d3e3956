#!/usr/bin/env python3
"""cProfile of the prune tail (scoring + mask selection + slicing) on the CIFAR UNet."""
import cProfile, importlib, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests', 'golden')]
import golden_common as gc
unet = importlib.import_module('diff-pruning_amd.unet'); sweep = importlib.import_module('diff-pruning_amd.sweep')
diffusion = importlib.import_module('diff-pruning_amd.diffusion')
def fresh():
    m = unet.UNet2DModel(**gc.CIFAR_CFG); gc.det_init_(m, 0); m = m.cuda().eval()
    c = torch.from_numpy(gc.det_clean((4, 3, 32, 32), 1)).cuda(); n = torch.from_numpy(gc.det_noise((4, 3, 32, 32), 2)).cuda()
    sweep.taylor_sweep(m, diffusion.DDPMScheduler(), c, n, num_steps=2)
    torch.cuda.synchronize()
    return m
m = fresh(); t = time.perf_counter(); sweep.prune_model(m, 0.3); torch.cuda.synchronize(); print('warm-up tail ms', (time.perf_counter() - t) * 1e3)
m = fresh(); t = time.perf_counter(); sweep.prune_model(m, 0.3); torch.cuda.synchronize(); print('tail ms', (time.perf_counter() - t) * 1e3)
m = fresh()
pr = cProfile.Profile(); pr.enable(); sweep.prune_model(m, 0.3); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
