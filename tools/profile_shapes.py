#!/usr/bin/env python3
"""Per-shape table of the contraction launches in one sweep timestep (HIP events on the launch stream):
which layer shapes lose the most time against a target rate.   python tools/profile_shapes.py [--batch 256]"""
import argparse
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT,):
    sys.path.insert(0, p)
gc = importlib.import_module('diff-pruning_amd.synthetic')      # configs + seeded init

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=256)
ap.add_argument('--target', type=float, default=135.0, help='TFLOP/s the "lost ms" column is priced against')
ap.add_argument('--config', default='cifar', choices=['cifar', 'bedroom', 'ldm', 'pruned'])
ap.add_argument('--forward-only', action='store_true')
args = ap.parse_args()
ops = importlib.import_module('diff-pruning_amd.ops')
unet = importlib.import_module('diff-pruning_amd.unet')
diffusion = importlib.import_module('diff-pruning_amd.diffusion')
sweep = importlib.import_module('diff-pruning_amd.sweep')
cg, nt = ops._cg_name, ops._nt_name
ops._cg_name = lambda p: 'cg M=%d C=%d N=%d taps=%d ks=%d z=%d' % (p.M, p.C, p.NPIX, p.ntaps, p.ksplit, p.batches)
ops._nt_name = lambda p: 'nt M=%d NC=%d P=%d taps=%d sp=%d z=%d' % (p.M, p.NCOLS, p.P, p.ntaps, p.splits, p.batches)
ops._wino_name = lambda p, bk, wr: 'wino M=%d C=%d N=%d W=%d ks=%d bk=%d wr=%d' % (p.M, p.C, p.NPIX, p.g.Wo, p.ksplit, bk, wr)
ops._wino43_name = lambda p: 'wino43 M=%d C=%d N=%d W=%d ks=%d' % (p.M, p.C, p.NPIX, p.g.Wo, p.ksplit)
ops._wgrad_wino_name = lambda p, bt: 'wgwino M=%d NC=%d P=%d sp=%d bt=%d' % (p.M, p.NCOLS, p.P, p.splits, bt)
dev = torch.device('cuda')
B = args.batch
if args.config == 'ldm':
    # LDM cin256-v2 UNet: CFG forward at B latents (sampling: 2 x 6 = 12) or scored forward + backward (B = 6)
    ldm = importlib.import_module('diff-pruning_amd.ldm')
    model = ldm.UNetModel(**gc.LDM_CIN256_CFG)
    gc.det_init_(model, 1)
    model = model.to(dev).eval()
    x = torch.randn(B, 3, 64, 64, device=dev); ctx = torch.randn(B, 1, 512, device=dev)
    t = torch.full((B,), 500, device=dev); nz = torch.randn_like(x)
    eng = model.engine()
    eng.bind(eng.P, {n: torch.zeros_like(p) for n, p in model.named_parameters()})
    eng.overlap_wgrad = False
    eng.packs.pin_depth += 1

    def step(k):
        y = eng.forward(x, t, ctx, save=not args.forward_only)
        if not args.forward_only:
            l, d = ops.mse_fwd_bwd(y, nz, 2.0 / y.numel(), 1.0 / y.numel())
            eng.backward(d)
else:
    cfg = gc.BEDROOM_CFG if args.config == 'bedroom' else gc.CIFAR_CFG
    H = cfg['sample_size']
    model = unet.UNet2DModel(**cfg)
    gc.det_init_(model, 0)
    model = model.to(dev).eval()
    if args.config == 'pruned':
        c0 = torch.from_numpy(gc.det_clean((16, 3, H, H), 1)).to(dev); n0 = torch.from_numpy(gc.det_noise((16, 3, H, H), 2)).to(dev)
        sweep.taylor_sweep(model, diffusion.DDPMScheduler(), c0, n0, num_steps=2)
        sweep.prune_model(model, 0.3)
    clean = torch.from_numpy(gc.det_clean((B, 3, H, H), 1)).to(dev)
    noise = torch.from_numpy(gc.det_noise((B, 3, H, H), 2)).to(dev)
    sweep.flatten_grads(model)
    if args.forward_only:
        tt = torch.full((B,), 500, device=dev)
        model.engine().packs.pin_depth += 1

        def step(k):
            with torch.no_grad():
                model(clean, tt)
    else:
        # ONE timestep pipeline and no weight-gradient side stream: one kernel on the GPU at a time, clean per-shape durations.  (Until
        # round 5 this tool inherited the sweep's default of two timesteps in flight for big shards: consecutive steps ran on two
        # streams and stretched each other's event-timed kernels -- the tables of rounds 3-5 UNDERSTATE the big shapes' TFLOP/s by
        # 10-25 %; an isolated loop of the same launch, profiles/round5_wino_isolated.txt, is the cross-check.)
        step = sweep.HipSweepStep(model, diffusion.DDPMScheduler(), clean, noise, B * 3 * H * H, 'mse', B, timestep_pipelines=1)
        step.eng.overlap_wgrad = False
step(0); step(1)
torch.cuda.synchronize()
agg = {}
for rep in range(3):
    ops._prof = []
    step(2 + rep)
    torch.cuda.synchronize()
    log, ops._prof = ops._prof, None
    for name, fl, st, en, ab in log:
        a = agg.setdefault(name, [0, 0.0, 0.0])
        a[0] += 1; a[1] += fl; a[2] += st.elapsed_time(en) * 1e-3
rows = []
for n, (c, fl, s) in agg.items():
    lost = (s - fl / (args.target * 1e12)) / 3 * 1e3
    rows.append((lost, n, c // 3, fl / s / 1e12, s / 3 * 1e3))
rows.sort(reverse=True)
print('%-52s %4s %8s %8s %8s' % ('shape', 'n', 'TFLOP/s', 'ms/step', 'lost ms'))
for lost, n, c, tf, ms in rows[:45]:
    print('%-52s %4d %8.1f %8.3f %8.3f' % (n, c, tf, ms, lost))
print('total contraction ms/step %.2f, lost vs %.0f TF: %.2f' % (sum(r[4] for r in rows), args.target, sum(r[0] for r in rows)))
