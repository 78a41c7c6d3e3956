#!/usr/bin/env python3
"""Headline benchmark: importance-scored images/sec of the Taylor sweep on the ddpm-cifar10-32 UNet.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): CIFAR-10 DDPM UNet (tools/ddpm_cifar10_config.json, 35.7 M parameters, seeded
deterministic weights), batch 256 per GPU of synthetic 32x32 images, sweep timesteps t = 0..K-1 (forward + loss +
backward with gradient accumulation), then -- inside the timed region -- the whole tail of the job: all-reduce of the
accumulated gradients over the ranks (N > 1), fused |w*g| group scoring, mask selection and channel slicing at ratio 0.3.
A "step" is one sweep timestep over one batch.  value = (images processed by all ranks) / wall of the timed region;
ms_per_step is the sweep-only time per timestep.  fp32 everywhere (the reference's dtype).

Extra objects on the JSON line:
  roofline     dominant kernel (conv_gemm_fast_kernel<128,128,false>: conv3x3/1x1 forward + dgrad), algorithmic FLOP per
               launch / average launch duration, measured with HIP events on the launch stream in an instrumented step
               after the timed region (weight-gradient stream overlap switched off there, so every kernel is timed
               alone), against the 157.3 TFLOP/s fp32 MFMA peak.
  cpu_baseline the oracle (CPU restatement of the reference path) timed on this box's host cores, bounded sample.
"""
import argparse
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, 'tests', 'golden')):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import golden_common as gc       # noqa: E402  (configs + deterministic init; reference-free)

PEAK_F32_TFLOPS = 157.3          # MI355X_MICROARCH.md: fp32 MFMA (= vector) dense peak
FLOP_PER_IMG_STEP = 37.3e9       # CIFAR-32 UNet fwd+bwd, algorithmic


def cpu_baseline(B=16, steps=3):
    """Oracle sweep (plain PyTorch fp32 on the host cores): 1 warm-up + `steps` timed timesteps."""
    from oracle import unet_ref, diffusion_ref
    cfg = gc.CIFAR_CFG
    shapes = unet_ref.param_shapes(cfg)
    P = {n: torch.from_numpy(gc.det_param(n, s, 0)).requires_grad_(True) for n, s in shapes.items()}
    clean = torch.from_numpy(gc.det_clean((B, 3, 32, 32), 1))
    noise = torch.from_numpy(gc.det_noise((B, 3, 32, 32), 2))
    marks = []
    diffusion_ref.taylor_sweep(P, cfg, clean, noise, steps + 1, on_step=lambda k, l: marks.append(time.perf_counter()))
    dt = marks[-1] - marks[0]
    return dict(value=B * steps / dt, unit='images/s', cores=torch.get_num_threads(), kind='port',
                sample='oracle sweep, CIFAR-32 UNet, B=%d, 1 warm-up + %d timed timesteps (fwd+bwd), fp32 PyTorch CPU' % (B, steps))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=256, help='images per GPU')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--graph', action='store_true', help='replay the timestep from a captured hipGraph')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)

    ops = importlib.import_module('diff-pruning_amd.ops')
    unet = importlib.import_module('diff-pruning_amd.unet')
    diffusion = importlib.import_module('diff-pruning_amd.diffusion')
    sweep = importlib.import_module('diff-pruning_amd.sweep')
    ops._lib()                                   # fail loudly if the HIP library is missing

    cfg = gc.CIFAR_CFG
    B = args.batch
    model = unet.UNet2DModel(**cfg)
    gc.det_init_(model, 0)
    model = model.to(dev).eval()
    sched = diffusion.DDPMScheduler()
    clean = torch.from_numpy(gc.det_clean((B, 3, 32, 32), 100 + rank)).to(dev)
    noise = torch.from_numpy(gc.det_noise((B, 3, 32, 32), 200 + rank)).to(dev)
    flat = sweep.flatten_grads(model)
    step = sweep.HipSweepStep(model, sched, clean, noise, world * B * clean[0].numel(), 'mse', world * B)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    if args.graph:
        step.capture()
        step(0)
    flat.zero_()
    barrier()
    t0 = time.perf_counter()
    losses = []
    for k in range(args.steps):
        losses.append(step(k))
    t_enqueue = time.perf_counter() - t0          # host time to enqueue the K steps (GPU runs asynchronously)
    torch.cuda.synchronize()
    t_sweep = time.perf_counter() - t0
    if world > 1:
        dist.all_reduce(flat)                                   # the sweep's one exchange step
    pr = sweep.prune_model(model, 0.3)                          # scoring + mask selection + slicing
    barrier()
    t_total = time.perf_counter() - t0
    tt = torch.tensor([t_total, t_sweep], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    t_total, t_sweep = float(tt[0]), float(tt[1])
    n_params_after = sum(p.numel() for p in model.parameters())
    loss_vals = [float(l) for l in losses]

    roof = None
    if rank == 0 and not args.no_roofline:
        # instrumented timestep on a fresh (un-pruned) model: HIP events around every contraction launch
        model2 = unet.UNet2DModel(**cfg)
        gc.det_init_(model2, 0)
        model2 = model2.to(dev).eval()
        sweep.flatten_grads(model2)
        step2 = sweep.HipSweepStep(model2, sched, clean, noise, world * B * clean[0].numel(), 'mse', world * B)
        step2.eng.overlap_wgrad = False      # per-kernel durations: one kernel on the GPU at a time
        step2(0)
        torch.cuda.synchronize()
        ops._prof = []
        step2(1)
        torch.cuda.synchronize()
        log, ops._prof = ops._prof, None
        agg = {}
        for name, fl, st, en, ab in log:
            a = agg.setdefault(name, [0, 0.0, 0.0, 0.0])
            a[0] += 1
            a[1] += fl
            a[2] += st.elapsed_time(en) * 1e-3
            a[3] += ab
        dom = max(agg, key=lambda n: agg[n][2])
        cnt, fl, sec, ab = agg[dom]
        # HBM bytes per launch of the dominant kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE in separate runs of this same command, profiles/round1_pmc_bench_traffic.json, aggregated by tools/pmc_aggregate.py); KB -> bytes.
        # FETCH_SIZE is uncalibrated for 4-byte-per-lane buffer loads on gfx950 (MI355X_MICROARCH.md, HBM section).
        traffic = None
        try:
            pm = json.load(open(os.path.join(ROOT, 'profiles', 'round1_pmc_bench_traffic.json')))
            k = pm.get(dom)
            if k:
                traffic = (k['FETCH_SIZE']['avg_kb'] + k['WRITE_SIZE']['avg_kb']) * 1024.0
        except (OSError, KeyError, ValueError):
            traffic = None
        roof = dict(bound='mfma', kernel=dom, achieved=fl / sec / 1e12, peak=PEAK_F32_TFLOPS, unit='TFLOP/s',
                    frac=fl / sec / 1e12 / PEAK_F32_TFLOPS, traffic=traffic, algorithmic_bytes_per_launch=ab / cnt,
                    launches_per_step=cnt,
                    avg_launch_ms=sec / cnt * 1e3, flop_per_launch=fl / cnt,
                    step_share=sec / (t_sweep / args.steps),
                    kernels={n: dict(launches=v[0], tflops=v[1] / v[2] / 1e12, ms=v[2] * 1e3) for n, v in agg.items()},
                    step_tflops=FLOP_PER_IMG_STEP * B / (t_sweep / args.steps) / 1e12,
                    step_frac=FLOP_PER_IMG_STEP * B / (t_sweep / args.steps) / 1e12 / PEAK_F32_TFLOPS)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    if rank == 0:
        imgs = world * B * args.steps
        out = {
            'metric': 'importance-scored images/sec (UNet fwd+bwd+|w*dL/dw|)',
            'value': imgs / t_total, 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': t_sweep / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'ddpm-cifar10-32 UNet (35.7M params, seeded weights), batch %d/GPU, Taylor sweep '
                                   't=0..%d + grad all-reduce + |w*g| scoring + mask selection + slicing (ratio 0.3)'
                                   % (B, args.steps - 1),
                       'global_batch': world * B, 'image': '3x32x32', 'parallelism': 'dp%d (batch shards)' % world,
                       'sweep_only_images_per_s': imgs / t_sweep, 'tail_ms': (t_total - t_sweep) * 1e3,
                       'host_enqueue_ms_per_step': t_enqueue / args.steps * 1e3,
                       'wgrad_stream_overlap': bool(step.eng.overlap_wgrad), 'hipgraph': bool(args.graph),
                       'pruned_groups': len(pr.records), 'params_after': n_params_after,
                       'loss_first_last': [loss_vals[0], loss_vals[-1]]},
            'roofline': roof, 'cpu_baseline': cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
