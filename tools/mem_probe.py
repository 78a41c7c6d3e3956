#!/usr/bin/env python3
"""How far does the host run ahead of the device in a long plain-Taylor sweep, and what does that do to device memory?
    python tools/mem_probe.py [--batch 256] [--steps 300] [--hog-gb 0] [--ahead N]
Samples hipMemGetInfo from a thread while the sweep is being enqueued; prints the peak of (total - free), torch's reserved /
allocated peaks, the allocator's retry count (every retry is a device synchronisation + a release of the whole cache) and the
host's enqueue time against the device's time.  --hog-gb: allocate that much first (a box with less free memory)."""
import argparse
import importlib
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
gc = importlib.import_module('diff-pruning_amd.synthetic')
unet = importlib.import_module('diff-pruning_amd.unet')
diffusion = importlib.import_module('diff-pruning_amd.diffusion')
sweep = importlib.import_module('diff-pruning_amd.sweep')

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=256)
ap.add_argument('--steps', type=int, default=300)
ap.add_argument('--hog-gb', type=float, default=0.0)
ap.add_argument('--ahead', type=int, default=None, help='override sweep.MAX_STEPS_AHEAD (0 = unbounded)')
a = ap.parse_args()
if a.ahead is not None and hasattr(sweep, 'MAX_STEPS_AHEAD'):
    sweep.MAX_STEPS_AHEAD = a.ahead

dev = torch.device('cuda:0')
model = unet.UNet2DModel(**gc.CIFAR_CFG)
gc.det_init_(model, 0)
model = model.to(dev).eval()
clean = torch.from_numpy(gc.det_clean((a.batch, 3, 32, 32), 11)).to(dev)
noise = torch.from_numpy(gc.det_noise((a.batch, 3, 32, 32), 12)).to(dev)
sched = diffusion.DDPMScheduler()
sweep.taylor_sweep(model, sched, clean, noise, num_steps=4)           # warm
torch.cuda.synchronize()
free0, total = torch.cuda.mem_get_info()
hog = torch.empty(int(a.hog_gb * (1 << 30)), dtype=torch.uint8, device=dev) if a.hog_gb > 0 else None
torch.cuda.reset_peak_memory_stats()
peak = [0]
stop = [False]


def sampler():
    while not stop[0]:
        f, t = torch.cuda.mem_get_info()
        peak[0] = max(peak[0], t - f)
        time.sleep(0.05)


th = threading.Thread(target=sampler, daemon=True)
th.start()
timings = {}
t0 = time.perf_counter()
res = sweep.taylor_sweep(model, sched, clean, noise, num_steps=a.steps, timings=timings)
torch.cuda.synchronize()
wall = time.perf_counter() - t0
stop[0] = True
th.join()
st = torch.cuda.memory_stats()
print('batch %d steps %d hog %.0f GB: device total %.1f GB, free before %.1f GB' % (a.batch, a.steps, a.hog_gb, total / 2**30, free0 / 2**30))
print('  host enqueue %.2f s, device done %.2f s (%.1f ms/step), max steps-ahead setting %s' % (
    timings['enqueue_s'], wall, 1e3 * wall / a.steps, getattr(sweep, 'MAX_STEPS_AHEAD', 'n/a')))
print('  peak used (hipMemGetInfo) %.1f GB; torch reserved peak %.1f GB, allocated peak %.1f GB; alloc retries %d, ooms %d' % (
    peak[0] / 2**30, st['reserved_bytes.all.peak'] / 2**30, st['allocated_bytes.all.peak'] / 2**30, st['num_alloc_retries'], st['num_ooms']))
print('  losses[0], [-1]:', res['losses'][0], res['losses'][-1])
