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


def cpu_baseline(steps=8):
    """Oracle sweep (plain PyTorch fp32 on the host cores), SURVEY.md §8(d): 1 warm-up + 8 timed timesteps at B=4 (config
    C1's batch) and at B=16; `value` is the better of the two."""
    from oracle import unet_ref, diffusion_ref
    cfg = gc.CIFAR_CFG
    shapes = unet_ref.param_shapes(cfg)
    P = {n: torch.from_numpy(gc.det_param(n, s, 0)).requires_grad_(True) for n, s in shapes.items()}
    rates = {}
    for B in (4, 16):
        clean = torch.from_numpy(gc.det_clean((B, 3, 32, 32), 1))
        noise = torch.from_numpy(gc.det_noise((B, 3, 32, 32), 2))
        marks = []
        diffusion_ref.taylor_sweep(P, cfg, clean, noise, steps + 1, on_step=lambda k, l: marks.append(time.perf_counter()))
        rates[B] = B * steps / (marks[-1] - marks[0])
    best = max(rates, key=rates.get)
    return dict(value=rates[best], unit='images/s', cores=torch.get_num_threads(), kind='port',
                by_batch={'B=%d' % b: r for b, r in rates.items()},
                sample='oracle sweep, CIFAR-32 UNet, 1 warm-up + %d timed timesteps (fwd+bwd) at B=4 and at B=16 '
                       '(value = B=%d), fp32 PyTorch CPU' % (steps, best))


def _self_launch(args):
    """`python bench.py --gpus N` (N > 1) outside a launcher: start the N ranks ourselves, one per GPU, the way the driver
    does (torch.distributed.run on 127.0.0.1) -- a silent 1-rank run reporting n_gpus 1 must not happen."""
    import socket
    import subprocess
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus:
        sys.exit('bench.py: --gpus %d needs %d MI355X GPUs, this machine has %d' % (args.gpus, args.gpus, n_dev))
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    sys.exit(subprocess.call(cmd, env=env))


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

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        _self_launch(args)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        sys.exit('bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE)' % (args.gpus, world))
    if not torch.cuda.is_available():
        sys.exit('bench.py: no MI355X visible (torch.cuda.is_available() is False); the hot path has no CPU fallback')
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
        world = dist.get_world_size()
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
    step.finish()
    flat.zero_()
    lib = ops._lib()
    barrier()
    launches0 = lib.dp_launch_count()
    t0 = time.perf_counter()
    losses = []
    for k in range(args.steps):
        losses.append(step(k))
    t_enqueue = time.perf_counter() - t0          # host time to enqueue the K steps (GPU runs asynchronously)
    launches_per_step = (lib.dp_launch_count() - launches0) / args.steps
    step.finish()                                 # fold the second half-batch pipeline's gradients in (once per sweep)
    torch.cuda.synchronize()
    t_sweep = time.perf_counter() - t0
    t_allreduce = 0.0
    if world > 1:
        ta = time.perf_counter()
        dist.all_reduce(flat)                                   # the sweep's one exchange step (RCCL over xGMI)
        torch.cuda.synchronize()
        t_allreduce = time.perf_counter() - ta
    pr = sweep.prune_model(model, 0.3)                          # scoring + mask selection + slicing
    barrier()
    t_total = time.perf_counter() - t0
    tt = torch.tensor([t_total, t_sweep, t_allreduce], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    t_total, t_sweep, t_allreduce = float(tt[0]), float(tt[1]), float(tt[2])
    n_ranks = dist.get_world_size() if world > 1 else 1
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
        step2.eng.overlap_wgrad = False      # per-kernel durations: one kernel on the GPU at a time, at the shapes of the
        if step2._half is not None:          # timed region (half-batch kernels when two pipelines are used)
            step2._half['serial'] = True
            step2._half['eng'].overlap_wgrad = False
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
        # HBM bytes per launch of the dominant kernel come from rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE in
        # separate runs of this same command, aggregated by tools/pmc_aggregate.py; KB -> bytes) that cannot run inside
        # this process: they are read from the newest profiles/round*_pmc_bench_traffic.json, which records the git blob
        # of csrc/gemm.hip it was measured on.  A different blob today = stale counters = traffic null.
        # FETCH_SIZE is uncalibrated for 4-byte-per-lane buffer loads on gfx950 (MI355X_MICROARCH.md, HBM section).
        traffic, traffic_src = None, None
        try:
            import glob
            import hashlib
            cand = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'round*_pmc_bench_traffic.json')))
            src = open(os.path.join(ROOT, 'diff-pruning_amd', 'csrc', 'gemm.hip'), 'rb').read()
            blob = hashlib.sha1(b'blob %d\0' % len(src) + src).hexdigest()
            pm = json.load(open(cand[-1]))
            measured_on = pm.get('_gemm_hip_blob')
            traffic_src = dict(file=os.path.relpath(cand[-1], ROOT), measured_on_gemm_hip_blob=measured_on,
                               current_gemm_hip_blob=blob, stale=measured_on != blob)
            k = pm.get(dom)
            if k and measured_on == blob:
                traffic = (k['FETCH_SIZE']['avg_kb'] + k['WRITE_SIZE']['avg_kb']) * 1024.0
        except (OSError, KeyError, ValueError, IndexError):
            traffic = None
        roof = dict(bound='mfma', kernel=dom, achieved=fl / sec / 1e12, peak=PEAK_F32_TFLOPS, unit='TFLOP/s',
                    frac=fl / sec / 1e12 / PEAK_F32_TFLOPS, traffic=traffic, traffic_source=traffic_src,
                    algorithmic_bytes_per_launch=ab / cnt,
                    launches_per_step=cnt,
                    avg_launch_ms=sec / cnt * 1e3, flop_per_launch=fl / cnt,
                    step_share=sec / (t_sweep / args.steps),
                    kernels={n: dict(launches=v[0], tflops=v[1] / v[2] / 1e12, ms=v[2] * 1e3) for n, v in agg.items()},
                    # Whole-step rates.  `executed`: the multiply-adds the kernels of one timestep actually perform (sum over the
                    # instrumented launches) -- the number to hold against the MFMA peak.  `reference_equivalent`: SURVEY 8(d)'s
                    # count of the reference's own arithmetic (37.3 GFLOP per image-timestep) over the same time; it exceeds the
                    # executed count because the upsample convolutions run in their sub-pixel form (16 instead of 36
                    # multiply-adds per low-resolution pixel) -- a rate of useful work, not of hardware utilisation.
                    executed_flop_per_step=sum(v[1] for v in agg.values()),
                    step_tflops=sum(v[1] for v in agg.values()) / (t_sweep / args.steps) / 1e12,
                    step_frac=sum(v[1] for v in agg.values()) / (t_sweep / args.steps) / 1e12 / PEAK_F32_TFLOPS,
                    step_tflops_reference_equivalent=FLOP_PER_IMG_STEP * B / (t_sweep / args.steps) / 1e12)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    if rank == 0:
        imgs = n_ranks * B * args.steps
        out = {
            'metric': 'importance-scored images/sec (UNet fwd+bwd+|w*dL/dw|)',
            'value': imgs / t_total, 'unit': 'images/s', 'n_gpus': n_ranks, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': t_sweep / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'ddpm-cifar10-32 UNet (35.7M params, seeded weights), batch %d/GPU, Taylor sweep '
                                   't=0..%d + grad all-reduce + |w*g| scoring + mask selection + slicing (ratio 0.3)'
                                   % (B, args.steps - 1),
                       'global_batch': n_ranks * B, 'image': '3x32x32', 'parallelism': 'dp%d (batch shards)' % n_ranks,
                       'sweep_only_images_per_s': imgs / t_sweep, 'tail_ms': (t_total - t_sweep) * 1e3,
                       'host_enqueue_ms_per_step': t_enqueue / args.steps * 1e3,
                       'kernel_launches_per_step': launches_per_step, 'grad_allreduce_ms': t_allreduce * 1e3,
                       'wgrad_stream_overlap': bool(step.eng.overlap_wgrad), 'hipgraph': bool(args.graph),
                       'half_batch_pipelines': 2 if step._half is not None else 1,
                       'pruned_groups': len(pr.records), 'params_after': n_params_after,
                       'loss_first_last': [loss_vals[0], loss_vals[-1]]},
            'roofline': roof, 'cpu_baseline': cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
