#!/usr/bin/env python3
"""Go / no-go gate of the round-3 verdict, item 5: the Winograd F(2, 3) convolution (csrc/winograd.hip) against the direct implicit
GEMM on the dominant CIFAR shapes at batch 256 (forward and input gradient), reference-equivalent TFLOP/s = 2*9*C*M*pixels / time."""
import importlib, math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module('diff-pruning_amd.ops')
ops.WINO_MIN_TILES = 0
ops.WGRAD_WINO_MIN_WORK = 0
B = int(os.environ.get('B', '256'))


def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


ops.WINO43, ops.WINO43_MIN_TILES = True, 0
print('shape                     direct ms (TF/s)    winograd ms (ref-eq TF/s, executed TF/s)   speedup   max rel diff   [fwd: F(4, 3) ms (ref-eq TF/s), vs F(2, 3), max rel diff]')
for (ci, c2, co, h) in [(256, 0, 256, 16), (128, 0, 128, 32), (128, 128, 128, 32), (256, 0, 256, 8), (192, 0, 192, 16), (96, 0, 96, 32), (384, 0, 384, 32)]:
    bb = B if ci < 384 else 12
    x = ops.empty_act((bb, ci, h, h), torch.device('cuda')).normal_()
    x2 = ops.empty_act((bb, c2, h, h), torch.device('cuda')).normal_() if c2 else None
    w = torch.randn(co, ci + c2, 3, 3, device='cuda') / math.sqrt((ci + c2) * 9)
    spec = ops.ConvSpec(3, 1, 1, 0)
    for mode, name in ((0, 'fwd'), (1, 'dgrad'), (2, 'wgrad')):
        if mode == 2:
            dy = ops.empty_act((bb, co, h, h), x.device).normal_()
            y1, y2 = torch.zeros_like(w), torch.zeros_like(w)
            def f_d():
                ops.WGRAD_WINO = False
                ops.conv_wgrad(dy, x, x2, y1, spec, accumulate=False)
            def f_w():
                ops.WGRAD_WINO = True
                ops.conv_wgrad(dy, x, x2, y2, spec, accumulate=False)
        if mode < 2:
            wp, ld = ops.pack_weight(w, mode)
            U = ops.pack_weight_wino(w, mode)
        if mode == 2:
            pass
        elif mode == 0:
            y1 = ops.empty_act((bb, co, h, h), x.device); y2 = ops.empty_act((bb, co, h, h), x.device)
            f_d = lambda: ops.conv_forward(x, x2, wp, ld, co, spec, out=y1)
            f_w = lambda: ops.conv_forward(x, x2, wp, ld, co, spec, out=y2, wino=U)
        else:
            if c2: continue
            dy = ops.empty_act((bb, co, h, h), x.device).normal_()
            y1 = ops.empty_act((bb, ci, h, h), x.device); y2 = ops.empty_act((bb, ci, h, h), x.device)
            f_d = lambda: ops.conv_dgrad(dy, wp, ld, ci, spec, (h, h), out=y1)
            f_w = lambda: ops.conv_dgrad(dy, wp, ld, ci, spec, (h, h), out=y2, wino=U)
        td, tw = timeit(f_d), timeit(f_w)
        fl = 2.0 * bb * h * h * (ci + c2) * co * 9
        diff = float((y1 - y2).abs().max() / y1.abs().max())
        extra = ''
        if mode == 0:                         # the go / no-go gate of round 4's verdict, item 5: F(4, 3) forward against F(2, 3)
            U4 = ops.pack_weight_wino43(w)
            y3 = ops.empty_act((bb, co, h, h), x.device)
            t4 = timeit(lambda: ops.conv_forward(x, x2, wp, ld, co, spec, out=y3, wino43=U4))
            extra = '   [%.3f (%.1f)  %.2fx  %.1e]' % (t4, fl / t4 / 1e9, tw / t4, float((y1 - y3).abs().max() / y1.abs().max()))
        print('%-5s %3d+%-3d->%3d @%2dx%-2d  %.3f (%.1f)   %.3f (%.1f, %.1f)   %.2fx   %.1e%s' % (name, ci, c2, co, h, h, td, fl / td / 1e9, tw, fl / tw / 1e9,
              fl * 2 / 3 / tw / 1e9, td / tw, diff, extra))
