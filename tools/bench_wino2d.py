#!/usr/bin/env python3
"""Go / no-go of round 6: the two-dimensional Winograd F(2x2, 3x3) convolution (csrc/winograd2d.hip) against the one-dimensional F(2, 3)
(csrc/winograd.hip) and the direct implicit GEMM -- forward and input gradient, batch 256 -- with the error of each against an fp64
convolution of the same operands.  Reference-equivalent TFLOP/s = 2*9*C*M*pixels / time."""
import importlib, math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module('diff-pruning_amd.ops')
ops.WINO_MIN_TILES = 0
ops.WINO2D_MIN_TILES = int(os.environ.get('W2D_MIN_TILES', '0'))
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
    return float((y.double() - ref).abs().max() / ref.abs().max())


shapes = [(128, 0, 128, 32), (256, 0, 256, 16), (256, 0, 256, 8), (128, 128, 128, 32), (256, 256, 256, 16), (256, 0, 256, 4), (96, 0, 96, 32),
          (192, 0, 192, 16), (128, 0, 256, 16), (384, 0, 384, 32), (64, 0, 64, 64), (128, 0, 128, 256), (128, 0, 128, 128), (256, 0, 256, 64),
          (128, 128, 128, 256)]
print('shape                      direct ms   F(2,3) ms (ref-eq TF/s)   F(2x2,3x3) ms (ref-eq TF/s, executed TF/s)   2-D vs 1-D   err direct / 1-D / 2-D vs fp64')
for (ci, c2, co, h) in shapes:
    bb = B if ci < 384 else 12
    if h == 64: bb = min(bb, 64)
    if h >= 128: bb = 4                            # bedroom-256 at 4 images per GPU
    dev = torch.device('cuda')
    x = ops.empty_act((bb, ci, h, h), dev).normal_()
    x2 = ops.empty_act((bb, c2, h, h), dev).normal_() if c2 else None
    w = torch.randn(co, ci + c2, 3, 3, device='cuda') / math.sqrt((ci + c2) * 9)
    bias = torch.randn(co, device='cuda')
    spec = ops.ConvSpec(3, 1, 1, 0)
    for mode, name in ((0, 'fwd'), (1, 'dgrad')):
        if mode == 1 and c2: continue
        wp, ld = ops.pack_weight(w, mode)
        U = ops.pack_weight_wino(w, mode)
        U2 = ('2d',) + tuple(ops.pack_weight_wino2d(w, mode))
        if mode == 0:
            ys = [ops.empty_act((bb, co, h, h), dev) for _ in range(3)]
            fns = [lambda: ops.conv_forward(x, x2, wp, ld, co, spec, out=ys[0], bias=bias),
                   lambda: ops.conv_forward(x, x2, wp, ld, co, spec, out=ys[1], bias=bias, wino=U),
                   lambda: ops.conv_forward(x, x2, wp, ld, co, spec, out=ys[2], bias=bias, wino=U2)]
        else:
            dy = ops.empty_act((bb, co, h, h), dev).normal_()
            ys = [ops.empty_act((bb, ci, h, h), dev) for _ in range(3)]
            fns = [lambda: ops.conv_dgrad(dy, wp, ld, ci, spec, (h, h), out=ys[0]),
                   lambda: ops.conv_dgrad(dy, wp, ld, ci, spec, (h, h), out=ys[1], wino=U),
                   lambda: ops.conv_dgrad(dy, wp, ld, ci, spec, (h, h), out=ys[2], wino=U2)]
        for y in ys: y.fill_(float('nan'))
        ts = [timeit(f) for f in fns]
        es = ['-', '-', '-']
        if CHECK:
            nb = min(bb, 8)
            if mode == 0:
                xin = torch.cat([x[:nb], x2[:nb]], 1) if c2 else x[:nb]
                ref = torch.nn.functional.conv2d(xin.double(), w.double(), bias.double(), padding=1)
            else:
                ref = torch.nn.functional.conv_transpose2d(dy[:nb].double(), w.double(), padding=1)
            es = ['%.1e' % err(y[:nb], ref) for y in ys]
            assert all(torch.isfinite(y).all() for y in ys), 'non-finite output (unwritten elements?)'
        fl = 2.0 * bb * h * h * (ci + c2) * co * 9
        print('%-5s %3d+%-3d->%3d @%2dx%-2d  %.3f       %.3f (%.1f)           %.3f (%.1f, %.1f)                    %.2fx       %s / %s / %s' % (
            name, ci, c2, co, h, h, ts[0], ts[1], fl / ts[1] / 1e9, ts[2], fl / ts[2] / 1e9, fl * 4 / 9 / ts[2] / 1e9, ts[1] / ts[2], es[0], es[1], es[2]), flush=True)
