"""Fused attention forward (csrc/attention.hip) against the three launches it replaces (QK^T -> softmax -> P.V), per shape of
the sampling forwards: HIP-event time per call and executed TFLOP/s (4 T^2 d FLOP per image and head).
    python tools/bench_attention.py [--iters 20] > gpurun_out/r3_attention.txt"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
ops = importlib.import_module('diff-pruning_amd.ops')

SHAPES = [
    # name, N, heads, d, H
    ('cifar256  T=256  d=256', 256, 1, 256, 16),
    ('cifar ddim T=256 d=256', 64, 1, 256, 16),
    ('bedroom   T=256  d=512', 16, 1, 512, 16),
    ('ldm cfg   T=1024 d=384', 12, 1, 384, 32),
    ('ldm cfg   T=256  d=576', 12, 1, 576, 16),
    ('ldm 4gpu  T=1024 d=384', 4, 1, 384, 32),
]


def timed(fn, iters):
    for _ in range(3):
        fn()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    st.record()
    for _ in range(iters):
        fn()
    en.record()
    torch.cuda.synchronize()
    return st.elapsed_time(en) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=20)
    a = ap.parse_args()
    print('%-26s %10s %10s %10s %8s %8s %9s' % ('shape', 'plain us', 'pipe us', '3-launch', 'speedup', 'TF/s', 'max diff'))
    for name, N, heads, d, H in SHAPES:
        T = H * H
        g = torch.Generator().manual_seed(N + d)
        qkv = torch.randn(N, 3 * heads * d, H, H, generator=g).cuda()
        q, k, v = qkv[:, :heads * d], qkv[:, heads * d:2 * heads * d], qkv[:, 2 * heads * d:]
        scale = float(d) ** -0.5
        Z = N * heads

        def fused(variant=0):
            return ops.attention_fwd(q, k, v, heads, scale, variant=variant)

        def three():
            s = ops.bmm_tn(q.view(Z, d, T), k.view(Z, d, T), alpha=scale)
            return ops.bmm_nt(v.view(Z, d, T), ops.softmax_fwd(s, out=s))
        diff = float((fused().view(Z, d, T) - three()).abs().max())
        t1, tp, t3 = timed(lambda: fused(1), a.iters), timed(lambda: fused(2), a.iters), timed(three, a.iters)
        tf = min(t1, tp)
        flop = 4.0 * Z * T * T * d
        print('%-26s %10.1f %10.1f %10.1f %8.2f %8.1f %9.2e' % (name, t1 * 1e3, tp * 1e3, t3 * 1e3, t3 / tf, flop / tf / 1e9, diff),
              flush=True)


if __name__ == '__main__':
    main()
