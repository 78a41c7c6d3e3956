#!/usr/bin/env python3
"""Timing of one LDM cin256-v2 UNet forward+backward (400.9 M params) at B=6 64x64 latents and of a CFG forward at B=12."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
gc = importlib.import_module('diff-pruning_amd.synthetic')
ldm = importlib.import_module('diff-pruning_amd.ldm'); ops = importlib.import_module('diff-pruning_amd.ops')
m = ldm.UNetModel(**gc.LDM_CIN256_CFG); gc.det_init_(m, 1); m = m.cuda().eval()
eng = m.engine(); grads = {n: torch.zeros_like(p) for n, p in m.named_parameters()}; eng.bind(eng.P, grads)
B = 6
x = torch.randn(B, 3, 64, 64, device='cuda'); ctx = torch.randn(B, 1, 512, device='cuda'); t = torch.full((B,), 500, device='cuda'); noise = torch.randn_like(x)
def step():
    y = eng.forward(x, t, ctx, save=True); n = y.numel()
    loss, dout = ops.mse_fwd_bwd(y, noise, 2.0 / n, 1.0 / n); eng.backward(dout); return loss
for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): l = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print('fwd+bwd B=6: %.1f ms  (%.1f TFLOP/s at 625 GFLOP/latent-step)  loss %.4f' % (dt * 1e3, 625e9 * B / dt / 1e12, float(l)))
x2 = torch.randn(12, 3, 64, 64, device='cuda'); c2 = torch.randn(12, 1, 512, device='cuda'); t2 = torch.full((12,), 500, device='cuda')
with torch.no_grad(), m.pin_weights():          # a sampling loop: the weights are frozen, pack once
    for _ in range(2): m(x2, t2, context=c2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): m(x2, t2, context=c2)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print('CFG forward B=12: %.1f ms (%.1f TFLOP/s at 208 GFLOP/latent)' % (dt * 1e3, 208.4e9 * 12 / dt / 1e12))
