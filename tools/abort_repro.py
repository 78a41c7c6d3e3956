#!/usr/bin/env python3
"""The failure class behind the round-5 driver abort (GPUTEST_r05: rc 134 inside the 1000-timestep batch-256 test), on demand:
    DP_SIDE_RECORD_STREAM=1 DP_MAX_STEPS_AHEAD=0 AMD_LOG_LEVEL=1 python tools/abort_repro.py     # the round-5 behaviour
    AMD_LOG_LEVEL=1 python tools/abort_repro.py                                                  # the fixed behaviour
Phase 1: a long plain-Taylor sweep at batch 256 (Winograd dispatch).  With Tensor.record_stream on every side-stream operand the
caching allocator's reserved memory grows until hipMemGetInfo reports (almost) nothing free.  Phase 2: the same sweep on the direct
kernels, on fresh streams -- kernels with a private segment (conv_gemm_fast_kernel<128,128>: 12 bytes of scratch per lane) now need
the HIP runtime to allocate scratch, kernel-argument and signal memory from a device that has none left."""
import argparse
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
gc = importlib.import_module('diff-pruning_amd.synthetic')
unet = importlib.import_module('diff-pruning_amd.unet')
diffusion = importlib.import_module('diff-pruning_amd.diffusion')
sweep = importlib.import_module('diff-pruning_amd.sweep')
ops = importlib.import_module('diff-pruning_amd.ops')

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=1000)
ap.add_argument('--steps2', type=int, default=200)
ap.add_argument('--rounds', type=int, default=1)
a = ap.parse_args()
dev = torch.device('cuda:0')
clean = torch.from_numpy(gc.det_clean((256, 3, 32, 32), 11)).to(dev)
noise = torch.from_numpy(gc.det_noise((256, 3, 32, 32), 12)).to(dev)
sched = diffusion.DDPMScheduler()


def run(steps, tag):
    model = unet.UNet2DModel(**gc.CIFAR_CFG)
    gc.det_init_(model, 0)
    model = model.to(dev).eval()
    t0 = time.perf_counter()
    res = sweep.taylor_sweep(model, sched, clean, noise, num_steps=steps)
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    st = torch.cuda.memory_stats()
    print('%s: %d steps in %.1f s; free %.2f of %.0f GB; torch reserved %.1f GB, allocated peak %.1f GB, alloc retries %d; loss[-1] %.6f'
          % (tag, steps, time.perf_counter() - t0, free / 2**30, total / 2**30, st['reserved_bytes.all.current'] / 2**30,
             st['allocated_bytes.all.peak'] / 2**30, st['num_alloc_retries'], res['losses'][-1]), flush=True)
    return model


print('record_stream form: %s, max steps ahead: %s' % (os.environ.get('DP_SIDE_RECORD_STREAM') == '1', sweep.MAX_STEPS_AHEAD), flush=True)
keep = []
for r in range(a.rounds):
    # the sequence of the round-5 test: every sweep builds fresh streams, whose cached blocks no other stream can reuse
    ops.WINO = ops.WGRAD_WINO = True
    keep.append(run(a.steps, 'round %d phase 1 (winograd)' % r))
    ops.WINO = ops.WGRAD_WINO = False
    m2 = run(a.steps2, 'round %d phase 2 (direct kernels, fresh streams)' % r)
    sweep.prune_model(m2, 0.3)
    torch.cuda.synchronize()
print('survived', flush=True)
