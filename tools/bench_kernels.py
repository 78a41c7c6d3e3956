#!/usr/bin/env python3
"""Micro-benchmark of the contraction kernels on the dominant CIFAR-32 shapes (B=256): TFLOP/s vs the 157.3 fp32 peak."""
import importlib, json, math, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module('diff-pruning_amd.ops')

def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3

def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    res = {}
    for (ci, co, h, k) in [(128, 128, 32, 3), (256, 256, 16, 3), (256, 256, 8, 3), (256, 256, 4, 3), (512, 256, 16, 3),
                           (384, 128, 32, 3), (256, 256, 16, 1), (512, 256, 16, 1)]:
        x = torch.randn(B, ci, h, h, device='cuda')
        w = torch.randn(co, ci, k, k, device='cuda') / math.sqrt(ci * k * k)
        dy = torch.randn(B, co, h, h, device='cuda')
        spec = ops.ConvSpec(k, 1, k // 2, 0)
        wp, ld = ops.pack_weight(w, 0)
        wd, ldd = ops.pack_weight(w, 1)
        y = torch.empty(B, co, h, h, device='cuda')
        dx = torch.empty(B, ci, h, h, device='cuda')
        gw = torch.zeros_like(w)
        fl = 2.0 * B * h * h * ci * co * k * k
        tf = timeit(lambda: ops.conv_forward(x, None, wp, ld, co, spec, out=y))
        td = timeit(lambda: ops.conv_dgrad(dy, wd, ldd, ci, spec, (h, h), out=dx))
        tw = timeit(lambda: ops.conv_wgrad(dy, x, None, gw, spec, accumulate=True))
        key = 'c%d_%d_h%d_k%d' % (ci, co, h, k)
        res[key] = dict(fwd_tflops=fl / tf / 1e12, dgrad_tflops=fl / td / 1e12, wgrad_tflops=fl / tw / 1e12,
                        fwd_ms=tf * 1e3, dgrad_ms=td * 1e3, wgrad_ms=tw * 1e3)
        print(key, json.dumps(res[key]), flush=True)
    # GN + SiLU bandwidth
    x = torch.randn(B, 128, 32, 32, device='cuda')
    gma = torch.ones(128, device='cuda'); bta = torch.zeros(128, device='cuda')
    t = timeit(lambda: ops.groupnorm_fwd(x, None, gma, bta, 32, 1e-6, True))
    res['gn_fwd_GBps'] = 2 * x.numel() * 4 / t / 1e9
    print('gn_fwd GB/s', res['gn_fwd_GBps'])
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, 'gpurun_out', 'bench_kernels.json'), 'w'), indent=1)

if __name__ == '__main__':
    main()
