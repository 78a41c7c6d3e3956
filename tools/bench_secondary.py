#!/usr/bin/env python3
"""Secondary metrics of SURVEY.md §8(d): finetune images/s on the pruned CIFAR UNet (config C4, batch 128 per GPU) and DDIM
image-steps/s (ddpm_sample.py shape: batch 256, pruned UNet, forward only).  Synthetic data, seeded weights."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
gc = importlib.import_module('diff-pruning_amd.synthetic')
unet = importlib.import_module('diff-pruning_amd.unet'); sweep = importlib.import_module('diff-pruning_amd.sweep')
diffusion = importlib.import_module('diff-pruning_amd.diffusion'); train = importlib.import_module('diff-pruning_amd.train')

dev = torch.device('cuda')
m = unet.UNet2DModel(**gc.CIFAR_CFG); gc.det_init_(m, 0); m = m.to(dev).eval()
c = torch.from_numpy(gc.det_clean((16, 3, 32, 32), 1)).to(dev); n = torch.from_numpy(gc.det_noise((16, 3, 32, 32), 2)).to(dev)
sweep.taylor_sweep(m, diffusion.DDPMScheduler(), c, n, num_steps=2)
sweep.prune_model(m, 0.3)
for p in m.parameters():
    p.grad = None
print('pruned params', sum(p.numel() for p in m.parameters()))

B = 128
sched = diffusion.DDPMScheduler()
ft = train.FinetuneEngine(m, sched, lr=2e-4)
clean = torch.from_numpy(gc.det_clean((B, 3, 32, 32), 3)).to(dev); noise = torch.from_numpy(gc.det_noise((B, 3, 32, 32), 4)).to(dev)
g = torch.Generator().manual_seed(0)
ts = [train.antithetic_timesteps(B, 1000, g).to(dev) for _ in range(12)]
for k in range(2): ft.step(clean, noise, ts[k])
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(2, 12): l = ft.step(clean, noise, ts[k])
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print('finetune (C4) pruned CIFAR UNet B=%d: %.1f ms/step, %.0f images/s, %.1f TFLOP/s (20.6 GFLOP/img-step), loss %.4f'
      % (B, dt * 1e3, B / dt, 20.6e9 * B / dt / 1e12, float(l)))

B = 256
x = torch.randn(B, 3, 32, 32, device=dev); t = torch.full((B,), 500, device=dev, dtype=torch.long)
with torch.no_grad(), m.pin_weights():          # a sampling loop: the weights are frozen, pack once
    for _ in range(2): m(x, t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): y = m(x, t).sample
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print('DDIM UNet forward, pruned CIFAR UNet B=%d: %.1f ms/step, %.0f image-steps/s, %.1f TFLOP/s (6.87 GFLOP/img forward)'
      % (B, dt * 1e3, B / dt, 6.87e9 * B / dt / 1e12))
pipe = diffusion.DDIMPipeline(m, diffusion.DDIMScheduler())
gen = torch.Generator().manual_seed(0)
torch.cuda.synchronize(); t0 = time.perf_counter()
img = pipe(batch_size=B, generator=gen, num_inference_steps=20, output_type='numpy').images
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print('DDIMPipeline 20 steps B=%d: %.2f s total, %.0f image-steps/s end to end, finite %s' % (B, dt, B * 20 / dt, bool((img == img).all())))
