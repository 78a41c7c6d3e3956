#!/usr/bin/env python3
"""Go / no-go of round 6 (second half): the two-dimensional transposed Winograd F(3x3, 2x2) weight gradient (csrc/wgrad2d.hip) against the
one-dimensional F(3, 2) kernel (csrc/winograd.hip wgrad_wino) and the direct split-K kernel, batch 256, with each one's error against an
fp64 weight gradient.  Reference-equivalent TFLOP/s = 2*9*Cout*Cin*pixels / time."""
import importlib, math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module('diff-pruning_amd.ops')
ops.WGRAD_WINO_MIN_WORK = 0
ops.WGRAD_WINO_MIN_FILL = 0.0
ops.WGRAD_WINO2D_MIN_FILL = 0.0
B = int(os.environ.get('B', '256'))
CHECK = os.environ.get('CHECK', '1') == '1'


def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def err(y, ref):
    return float((y.double().cpu() - ref).abs().max() / ref.abs().max())


shapes = [(128, 0, 128, 32), (256, 0, 256, 16), (256, 0, 256, 8), (128, 128, 128, 32), (256, 256, 256, 16), (128, 256, 128, 32), (96, 0, 96, 32),
          (192, 0, 192, 16), (384, 0, 384, 32), (128, 0, 256, 16)]
print('shape                      direct ms   F(3,2) ms (ref-eq TF/s)   F(3x3,2x2) ms (ref-eq TF/s, executed TF/s)   2-D vs 1-D   err direct / 1-D / 2-D vs fp64')
for (ci, c2, co, h) in shapes:
    bb = B if ci < 384 else 12
    dev = torch.device('cuda')
    x = ops.empty_act((bb, ci, h, h), dev).normal_()
    x2 = ops.empty_act((bb, c2, h, h), dev).normal_() if c2 else None
    dy = ops.empty_act((bb, co, h, h), dev).normal_()
    spec = ops.ConvSpec(3, 1, 1, 0)
    gws = [torch.zeros(co, ci + c2, 3, 3, device=dev) for _ in range(3)]

    def run(k):
        ops.WGRAD_WINO = k >= 1
        ops.WGRAD_WINO2D = k == 2
        ops.conv_wgrad(dy, x, x2, gws[k], spec, accumulate=False)
    ts = [timeit(lambda k=k: run(k)) for k in range(3)]
    ops.WGRAD_WINO = ops.WGRAD_WINO2D = True
    es = ['-'] * 3
    if CHECK:
        nb = min(bb, 16)
        xin = (torch.cat([x[:nb], x2[:nb]], 1) if c2 else x[:nb]).double().cpu().requires_grad_(False)
        w = torch.zeros(co, ci + c2, 3, 3, dtype=torch.float64, requires_grad=True)
        torch.nn.functional.conv2d(xin, w, padding=1).backward(dy[:nb].double().cpu())
        ref = w.grad
        for k in range(3):
            g = torch.zeros(co, ci + c2, 3, 3, device=dev)
            ops.WGRAD_WINO = k >= 1
            ops.WGRAD_WINO2D = k == 2
            ops.conv_wgrad(dy[:nb].contiguous() if False else ops.empty_act((nb, co, h, h), dev).copy_(dy[:nb]),
                           ops.empty_act((nb, ci, h, h), dev).copy_(x[:nb]),
                           ops.empty_act((nb, c2, h, h), dev).copy_(x2[:nb]) if c2 else None, g, spec, accumulate=False)
            es[k] = '%.1e' % err(g, ref)
        ops.WGRAD_WINO = ops.WGRAD_WINO2D = True
    fl = 2.0 * bb * h * h * (ci + c2) * co * 9
    print('wgrad %3d+%-3d->%3d @%2dx%-2d  %.3f       %.3f (%.1f)           %.3f (%.1f, %.1f)                    %.2fx       %s / %s / %s' % (
        ci, c2, co, h, h, ts[0], ts[1], fl / ts[1] / 1e9, ts[2], fl / ts[2] / 1e9, fl * 4 / 9 / ts[2] / 1e9, ts[1] / ts[2], es[0], es[1], es[2]), flush=True)
