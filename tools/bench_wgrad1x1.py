#!/usr/bin/env python3
"""1x1 weight gradients (attention projections): split count sweep."""
import importlib, math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
ops = importlib.import_module('diff-pruning_amd.ops')
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / iters * 1e-3
B = 256
for (ci, co, h, k) in [(256, 256, 16, 1), (512, 256, 16, 1), (256, 256, 8, 3), (256, 256, 4, 3)]:
    x = torch.randn(B, ci, h, h, device='cuda'); dy = torch.randn(B, co, h, h, device='cuda'); gw = torch.zeros(co, ci, k, k, device='cuda')
    spec = ops.ConvSpec(k, 1, k // 2, 0)
    out = []
    for ms in (None, 256, 128, 64, 32, 16):
        t = timeit(lambda: ops.conv_wgrad(dy, x, None, gw, spec, accumulate=True, max_splits=ms))
        out.append('%s:%.0fus/%.0fTF' % (ms, t * 1e6, 2.0 * B * h * h * ci * co * k * k / t / 1e12))
    print((ci, co, h, k), ' '.join(out), flush=True)
