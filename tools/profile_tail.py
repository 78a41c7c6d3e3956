#!/usr/bin/env python3
"""cProfile of the sweep's tail (scoring + mask selection + slicing, sweep.prune_model) on the CIFAR UNet."""
import cProfile
import importlib
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT,):
    sys.path.insert(0, p)
gc = importlib.import_module('diff-pruning_amd.synthetic')      # configs + seeded init

unet = importlib.import_module('diff-pruning_amd.unet')
sweep = importlib.import_module('diff-pruning_amd.sweep')
dev = torch.device('cuda')


def fresh():
    model = unet.UNet2DModel(**gc.CIFAR_CFG)
    gc.det_init_(model, 0)
    model = model.to(dev).eval()
    flat = sweep.flatten_grads(model)
    g = torch.Generator(device='cpu').manual_seed(1)
    for p in model.parameters():
        p.grad.copy_(torch.randn(p.shape, generator=g).to(dev))
    torch.cuda.synchronize()
    return model


m = fresh()
t0 = time.perf_counter(); sweep.prune_model(m, 0.3); torch.cuda.synchronize(); print('first call %.1f ms' % ((time.perf_counter() - t0) * 1e3))
m = fresh()
t0 = time.perf_counter(); sweep.prune_model(m, 0.3); torch.cuda.synchronize(); print('second call %.1f ms' % ((time.perf_counter() - t0) * 1e3))
m = fresh()
pr = cProfile.Profile()
pr.enable(); sweep.prune_model(m, 0.3); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(35)
