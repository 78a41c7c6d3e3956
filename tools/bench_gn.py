#!/usr/bin/env python3
"""GroupNorm(+SiLU) forward / backward bandwidth on the CIFAR-32 shapes at B=256 (algorithmic bytes: 8 B/elem forward,
16 B/elem backward with one addend) -- run with DP_NO_GN_WAVE=1 for the workgroup-per-group kernels."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module('diff-pruning_amd.ops')


def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
tot_f = tot_b = 0.0
CIFAR = ((128, 32, 8), (256, 32, 2), (384, 32, 1), (128, 16, 1), (256, 16, 6), (384, 16, 1), (512, 16, 2), (256, 8, 7),
         (512, 8, 3), (256, 4, 11), (512, 4, 3))
BEDROOM = ((128, 256, 10), (256, 256, 2), (128, 128, 8), (256, 128, 4), (256, 64, 12), (512, 64, 3), (256, 32, 12), (512, 32, 3),
           (512, 16, 14), (1024, 16, 3), (512, 8, 12), (1024, 8, 3))     # python tools/bench_gn.py 4 bedroom: the split kernels
for C, H, cnt in (BEDROOM if len(sys.argv) > 2 and sys.argv[2] == 'bedroom' else CIFAR):
    x = ops.empty_act((B, C, H, H), 'cuda'); x.normal_()
    dz = ops.empty_act((B, C, H, H), 'cuda'); dz.normal_()
    add = ops.empty_act((B, C, H, H), 'cuda'); add.normal_()
    ga = torch.rand(C, device='cuda') + 0.5; be = torch.randn(C, device='cuda')
    y, st = ops.groupnorm_fwd(x, None, ga, be, 32, 1e-6, True)
    tf = timeit(lambda: ops.groupnorm_fwd(x, None, ga, be, 32, 1e-6, True))
    tb = timeit(lambda: ops.groupnorm_bwd(x, None, ga, be, st, dz, 32, True, add1=add, want_rows=True))
    n = x.numel() * 4
    tot_f += tf * cnt; tot_b += tb * cnt
    print('C=%3d %2dx%-2d x%-2d  fwd %6.1f us %5.2f TB/s   bwd %6.1f us %5.2f TB/s' % (C, H, H, cnt, tf * 1e6, 2 * n / tf / 1e12, tb * 1e6, 4 * n / tb / 1e12), flush=True)
print('weighted per-step total: fwd %.2f ms  bwd %.2f ms' % (tot_f * 1e3, tot_b * 1e3))
