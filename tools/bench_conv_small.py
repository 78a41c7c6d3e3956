#!/usr/bin/env python3
"""Forward conv at the low-resolution levels (B=256): split-K target sweep."""
import importlib, math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
ops = importlib.import_module('diff-pruning_amd.ops')
B = 256
for blocks in (256, 512, 768, 1024, 1536, 2048):
    ops.CONV_SPLITK_BLOCKS = blocks
    out = []
    for (ci, co, h, k) in [(256, 256, 4, 3), (256, 256, 8, 3), (512, 256, 8, 3), (256, 256, 4, 1), (256, 256, 8, 1)]:
        x = ops.empty_act((B, ci, h, h), torch.device('cuda')).normal_(); w = torch.randn(co, ci, k, k, device='cuda') / math.sqrt(ci * k * k)
        wp, ld = ops.pack_weight(w, 0); y = torch.empty(B, co, h, h, device='cuda'); spec = ops.ConvSpec(k, 1, k // 2, 0)
        for _ in range(5): ops.conv_forward(x, None, wp, ld, co, spec, out=y)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): ops.conv_forward(x, None, wp, ld, co, spec, out=y)
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 20
        out.append('%dx%d@%d k%d %.0fus/%.0fTF' % (ci, co, h, k, ms * 1e3, 2.0 * B * h * h * ci * co * k * k / ms / 1e9))
    print(blocks, ' | '.join(out), flush=True)
