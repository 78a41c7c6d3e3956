#!/usr/bin/env python3
"""Forward-conv micro-benchmark on the two dominant CIFAR shapes (B=256) -- used with experiment builds (DP_HIP_LIB=...)."""
import importlib, math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module('diff-pruning_amd.ops')
B = 256
for (ci, co, h, k) in [(256, 256, 16, 3), (128, 128, 32, 3), (256, 256, 16, 1), (256, 256, 8, 3), (192, 192, 16, 3), (96, 96, 32, 3)]:
    x = ops.empty_act((B, ci, h, h), torch.device('cuda')).normal_(); w = torch.randn(co, ci, k, k, device='cuda') / math.sqrt(ci * k * k)
    wp, ld = ops.pack_weight(w, 0); y = torch.empty(B, co, h, h, device='cuda'); spec = ops.ConvSpec(k, 1, k // 2, 0)
    for _ in range(5): ops.conv_forward(x, None, wp, ld, co, spec, out=y)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): ops.conv_forward(x, None, wp, ld, co, spec, out=y)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    print('conv %dx%d h%d k%d: %.3f ms  %.1f TFLOP/s' % (ci, co, h, k, ms, 2.0 * B * h * h * ci * co * k * k / ms / 1e9))
