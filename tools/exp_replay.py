#!/usr/bin/env python3
"""Eager launches vs native replay list (csrc/replay.hip) vs hipGraph replay of one sweep timestep; bit-identity of the
accumulated gradients.  usage: exp_replay.py [cifar|bedroom] B [modes...]"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
gc = importlib.import_module('diff-pruning_amd.synthetic')
unet = importlib.import_module('diff-pruning_amd.unet'); sweep = importlib.import_module('diff-pruning_amd.sweep')
diffusion = importlib.import_module('diff-pruning_amd.diffusion'); ops = importlib.import_module('diff-pruning_amd.ops')
which = sys.argv[1] if len(sys.argv) > 1 else 'cifar'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
modes = sys.argv[3:] or ['eager', 'native', 'graph']
cfg, hw = (gc.CIFAR_CFG, 32) if which == 'cifar' else (gc.BEDROOM_CFG, 256)
K = 20 if which == 'cifar' else 6
ref = None
for mode in modes:
    m = unet.UNet2DModel(**cfg); gc.det_init_(m, 0); m = m.cuda().eval()
    c = torch.from_numpy(gc.det_clean((B, 3, hw, hw), 1)).cuda(); n = torch.from_numpy(gc.det_noise((B, 3, hw, hw), 2)).cuda()
    flat = sweep.flatten_grads(m)
    st = sweep.HipSweepStep(m, diffusion.DDPMScheduler(), c, n, B * c[0].numel(), 'mse', B)
    st(0); st(1)
    if mode != 'eager':
        t0 = time.perf_counter()
        st.capture(native=(mode == 'native'))
        torch.cuda.synchronize()
        print(mode, 'capture %.0f ms' % ((time.perf_counter() - t0) * 1e3), getattr(st._replay, 'info', None), flush=True)
    flat.zero_()
    st(2); torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(K): l = st(3 + k)
    th = time.perf_counter() - t0
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K * 1e3
    same = None
    if ref is None:
        ref = flat.clone()
    else:
        same = bool(torch.equal(ref, flat))
    print('%s B=%d %-7s %.2f ms/step (host enqueue %.2f ms/step)  loss %.6f  grads bit-identical to first mode: %s'
          % (which, B, mode, dt, th / K * 1e3, float(l), same), flush=True)
