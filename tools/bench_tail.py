#!/usr/bin/env python3
"""A/B of the `_tail` instantiations of the two-dimensional Winograd kernels (csrc/winograd2d.hip, csrc/wgrad2d.hip): layers whose
output-channel count leaves the last 64-row tile half empty (the pruned models' 96 / 160-wide layers).  Run once per setting of
DP_WINO2D_TAIL (0 = the plain instantiations multiply the empty row block, default = skip it); prints ms per launch and a digest of every
output, which must not depend on the setting (rows >= M are never stored, the other rows' arithmetic is untouched)."""
import hashlib, importlib, math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module('diff-pruning_amd.ops')
ops.WINO_MIN_TILES = 0
ops.WINO2D_MIN_TILES = 0


def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def digest(t):
    return hashlib.sha1(t.detach().cpu().numpy().tobytes()).hexdigest()[:12]


print('DP_WINO2D_TAIL=%s' % os.environ.get('DP_WINO2D_TAIL', '(default: on)'))
print('shape                               fwd ms (ref-eq TF/s)   dgrad ms   wgrad ms (ref-eq TF/s)   digests fwd / dgrad / wgrad')
dev = torch.device('cuda')
for (B, ci, c2, co, h) in [(128, 96, 0, 96, 32), (256, 96, 0, 96, 32), (128, 96, 96, 96, 32), (128, 192, 96, 96, 32), (128, 192, 0, 96, 32),
                           (128, 96, 0, 192, 16), (128, 192, 0, 192, 16), (256, 128, 0, 160, 16), (256, 128, 0, 128, 32), (256, 256, 0, 256, 16)]:
    g = torch.Generator(device='cuda').manual_seed(1)
    x = ops.empty_act((B, ci, h, h), dev).normal_(generator=g)
    x2 = ops.empty_act((B, c2, h, h), dev).normal_(generator=g) if c2 else None
    w = torch.randn(co, ci + c2, 3, 3, device='cuda', generator=g) / math.sqrt((ci + c2) * 9)
    dy = ops.empty_act((B, co, h, h), dev).normal_(generator=g)
    spec = ops.ConvSpec(3, 1, 1, 0)
    wp, ld = ops.pack_weight(w, 0)
    wd, ldd = ops.pack_weight(w, 1)
    U0 = ('2d',) + tuple(ops.pack_weight_wino2d(w, 0))
    U1 = ('2d',) + tuple(ops.pack_weight_wino2d(w, 1))
    y = ops.empty_act((B, co, h, h), dev)
    dx = ops.empty_act((B, ci + c2, h, h), dev)
    gw = torch.zeros_like(w)
    t_f = timeit(lambda: ops.conv_forward(x, x2, wp, ld, co, spec, out=y, wino=U0))
    t_d = timeit(lambda: ops.conv_dgrad(dy, wd, ldd, ci + c2, spec, (h, h), out=dx, wino=U1)) if not c2 else float('nan')
    t_w = timeit(lambda: ops.conv_wgrad(dy, x, x2, gw, spec, alpha=1.0, accumulate=False))
    fl = 2.0 * B * h * h * (ci + c2) * co * 9
    print('B=%3d %3d+%-3d->%3d @%2dx%-2d            %.3f (%5.1f)         %.3f      %.3f (%5.1f)          %s / %s / %s' % (
        B, ci, c2, co, h, h, t_f, fl / t_f / 1e9, t_d, t_w, fl / t_w / 1e9, digest(y), digest(dx) if not c2 else '-', digest(gw)), flush=True)
