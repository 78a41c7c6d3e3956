#!/usr/bin/env python3
"""Grid quantisation of dp_conv_wino: executed TFLOP/s of the forward Winograd F(2, 3) convolution against the number of workgroups
(batch sweep at fixed layer shape).  768 resident slots = 256 CUs x occupancy 3."""
import importlib, math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module('diff-pruning_amd.ops')
ops.WINO_MIN_TILES = 0


def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


print('shape                 batch   workgroups  rounds@768   ms      executed TF/s')
for (ci, co, h) in [(128, 128, 32), (256, 256, 16), (256, 256, 8)]:
    for bb in [int(v) for v in os.environ.get('BATCHES', '48,96,128,144,192,240,256,288,384,512').split(',')]:
        x = ops.empty_act((bb, ci, h, h), torch.device('cuda')).normal_()
        w = torch.randn(co, ci, 3, 3, device='cuda') / math.sqrt(ci * 9)
        spec = ops.ConvSpec(3, 1, 1, 0)
        wp, ld = ops.pack_weight(w, 0)
        U = ops.pack_weight_wino(w, 0)
        y = ops.empty_act((bb, co, h, h), x.device)
        t = timeit(lambda: ops.conv_forward(x, None, wp, ld, co, spec, out=y, wino=U))
        wgs = -(-co // 128) * -(-(bb * h * h) // 128)
        fl = 2.0 * bb * h * h * ci * co * 6
        print('%3d->%3d @%2dx%-2d       %4d   %6d      %5.2f      %.3f   %.1f' % (ci, co, h, h, bb, wgs, wgs / 768, t, fl / t / 1e9))
