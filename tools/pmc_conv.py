#!/usr/bin/env python3
"""One dominant shape, few launches: target for rocprofv3 --pmc passes (conv 256->256 @16x16, B=256: fwd, dgrad, wgrad)."""
import importlib, math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module('diff-pruning_amd.ops')
B, ci, co, h, k = 256, 256, 256, 16, 3
x = torch.randn(B, ci, h, h, device='cuda'); w = torch.randn(co, ci, k, k, device='cuda') / math.sqrt(ci * 9)
dy = torch.randn(B, co, h, h, device='cuda'); spec = ops.ConvSpec(3, 1, 1, 0)
wp, ld = ops.pack_weight(w, 0); wd, ldd = ops.pack_weight(w, 1)
y = torch.empty(B, co, h, h, device='cuda'); dx = torch.empty(B, ci, h, h, device='cuda'); gw = torch.zeros_like(w)
for _ in range(5):
    ops.conv_forward(x, None, wp, ld, co, spec, out=y)
    ops.conv_dgrad(dy, wd, ldd, ci, spec, (h, h), out=dx)
    ops.conv_wgrad(dy, x, None, gw, spec, accumulate=True)
torch.cuda.synchronize()
print('done')
