#!/usr/bin/env python3
"""Benchmarks of the Diff-Pruning hot path on MI355X.  Default = the headline: importance-scored images/sec of the Taylor
sweep on the ddpm-cifar10-32 UNet.

    python bench.py --gpus 1 --steps K --warmup W [--config NAME]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W [--config NAME]

--config (each names one of BASELINE.json's configs; every line has the same JSON shape with its own `roofline`):
  cifar256 (default)  configs[1]: CIFAR-10 DDPM UNet (35.7 M parameters, seeded weights), batch 256 per GPU of synthetic 32x32
               images, Taylor sweep t = 0..K-1 (forward + loss + backward, gradients accumulating), then -- inside the timed
               region -- the whole tail of the job: the gradient all-reduce over the ranks (taylor_sweep's own exchange),
               fused |w*g| group scoring, mask selection and channel slicing at ratio 0.3.
  bedroom256   configs[2]: google/ddpm-ema-bedroom-256 topology (113.7 M parameters), 256x256 images, 4 images per GPU (batch
               32 over 8 GPUs), Diff-Pruning threshold 0.05 (on-device early-exit state, scalar-loss all-reduce per step), tail
               as above.
  c4_finetune  configs[3]: finetune step (ddpm_train.py) of the ratio-0.3 pruned CIFAR UNet (19.85 M parameters), batch 128 per
               GPU, dropout 0.1, clip + Adam + EMA, bucketed gradient all-reduce overlapped with the backward pass.
  ddim         ddpm_sample.py's inner loop: pruned CIFAR UNet, batch 256 per GPU, one DDIM step = UNet forward + scheduler step.
  ldm          configs[4]: LDM cin256-v2 UNet (400.9 M parameters) importance pass (prune_ldm.py:103-131): 6 latents of 3x64x64
               per step SHARDED over the ranks (strong scaling), 20-step CFG DDIM sampling + loss at t + backward per step, loss
               all-reduce per step, one gradient all-reduce at the end.
A "step" is one pass of the path over one batch (a sweep timestep / an optimizer step / a DDIM step / an importance step).
value = units processed by all ranks / wall of the timed region; fp32 everywhere (the reference's dtype).

Extra objects on the JSON line:
  roofline     the kernel with the largest share of an instrumented step (HIP events on the launch stream around every
               contraction launch, weight-gradient stream overlap switched off there so every kernel is timed alone):
               algorithmic FLOP per launch / average launch duration against the 157.3 TFLOP/s fp32 MFMA peak, plus the
               whole-step rates (executed FLOP and the reference's own arithmetic over the measured step time).
  cpu_baseline the oracle (CPU restatement of the reference path) timed on this box's host cores, bounded sample
               (cifar256 and bedroom256; null for the other configs).
"""
import argparse
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT,):
    if _p not in sys.path:
        sys.path.insert(0, _p)
gc = importlib.import_module('diff-pruning_amd.synthetic')      # configs + seeded init

PEAK_F32_TFLOPS = 157.3          # MI355X_MICROARCH.md: fp32 MFMA (= vector) dense peak
# the reference's own arithmetic per unit: forward + backward unless noted
FLOP_CIFAR_IMG_STEP = 37.3e9
FLOP_BEDROOM_IMG_STEP = 1.491e12
FLOP_PRUNED_CIFAR_IMG_STEP = 20.6e9
FLOP_PRUNED_CIFAR_FWD = 6.87e9
FLOP_LDM_FWD, FLOP_LDM_FWD_BWD = 208.4e9, 625e9


def pkg(sub):
    return importlib.import_module('diff-pruning_amd.' + sub)


def cpu_baseline(config, steps=8):
    """Oracle sweep (plain PyTorch fp32 on the host cores), SURVEY.md §8(d) -- the host's BEST, not its default: a 32x32 UNet at
    batch 16 is oversubscribed on all 128+ threads of the GPU box (round 4 read 2.7-5.7 images/s there against the 13.5 the survey
    measured on 8 threads), so the thread count is swept.  cifar256: B=16 at {8, 16, 32, 64, all} threads (1 warm-up + 2 timed
    timesteps each), then B=64 at the best thread count (1 warm-up + 1 timed); `value` = the best rate, `cores` = the threads it
    used, `host_cores` = what the box has.  bedroom256: one 256x256 image, {16, all} threads, 1 warm-up + 1 timed timestep."""
    from oracle import unet_ref, diffusion_ref
    host = os.cpu_count() or torch.get_num_threads()
    default_threads = torch.get_num_threads()
    if config == 'cifar256':
        cfg, hw = gc.CIFAR_CFG, 32
        plan = [(16, n, 2) for n in sorted({n for n in (8, 16, 32, 64, default_threads) if n <= max(host, default_threads)})]
    elif config == 'bedroom256':
        cfg, hw = gc.BEDROOM_CFG, 256
        plan = [(1, n, 1) for n in sorted({min(16, default_threads), default_threads})]
    else:
        return None
    shapes = unet_ref.param_shapes(cfg)
    P = {n: torch.from_numpy(gc.det_param(n, s, 0)).requires_grad_(True) for n, s in shapes.items()}
    rates = {}

    def timed(B, threads, k):
        torch.set_num_threads(threads)
        clean = torch.from_numpy(gc.det_clean((B, 3, hw, hw), 1))
        noise = torch.from_numpy(gc.det_noise((B, 3, hw, hw), 2))
        marks = []
        diffusion_ref.taylor_sweep(P, cfg, clean, noise, k + 1, on_step=lambda i, l: marks.append(time.perf_counter()))
        rates[(B, threads)] = B * k / (marks[-1] - marks[0])

    try:
        for B, threads, k in plan:
            timed(B, threads, k)
        if config == 'cifar256':
            timed(64, max(rates, key=rates.get)[1], 1)
    finally:
        torch.set_num_threads(default_threads)
    best = max(rates, key=rates.get)
    return dict(value=rates[best], unit='images/s', cores=best[1], host_cores=host, kind='port',
                by_batch_and_threads={'B=%d,threads=%d' % k: r for k, r in rates.items()},
                sample='oracle sweep, %dx%d UNet, fwd+bwd timesteps after 1 warm-up each, fp32 PyTorch CPU: %s; value = the best '
                       '(B=%d on %d threads)' % (hw, hw, ', '.join('B=%d x %d timed @ %d threads' % (b, k, n) for b, n, k in plan)
                                                 + (', then B=64 x 1 at the best thread count' if config == 'cifar256' else ''),
                                                 best[0], best[1]))


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


class Env:
    def __init__(self, args, dev, rank, world):
        self.args, self.dev, self.rank, self.world = args, dev, rank, world

    def barrier(self):
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()


def _pruned_cifar(dev):
    """The ratio-0.3 pruned CIFAR UNet (19 851 157 parameters) that configs[3] finetunes and ddpm_sample.py samples from."""
    unet, sweep, diffusion = pkg('unet'), pkg('sweep'), pkg('diffusion')
    m = unet.UNet2DModel(**gc.CIFAR_CFG)
    gc.det_init_(m, 0)
    m = m.to(dev).eval()
    c = torch.from_numpy(gc.det_clean((16, 3, 32, 32), 1)).to(dev)
    n = torch.from_numpy(gc.det_noise((16, 3, 32, 32), 2)).to(dev)
    sweep.taylor_sweep(m, diffusion.DDPMScheduler(), c, n, num_steps=2, reduce_grads=False)
    sweep.prune_model(m, 0.3)
    for p in m.parameters():
        p.grad = None
    return m


# ---- workloads: setup() -> run(K) -> dict(units, t_total, t_steps, extra), instrumented() -> one step for the roofline ----
class SweepWorkload:
    """cifar256 / bedroom256: the importance sweep + prune tail through sweep.taylor_sweep (its own exchange step)."""

    def __init__(self, env, name):
        self.env, self.name = env, name
        unet, diffusion, sweep = pkg('unet'), pkg('diffusion'), pkg('sweep')
        if name == 'cifar256':
            self.cfg, self.hw, self.B, self.thr = gc.CIFAR_CFG, 32, env.args.batch or 256, None
            self.flop_unit, self.label = FLOP_CIFAR_IMG_STEP, 'ddpm-cifar10-32 UNet (35.7M params, seeded weights)'
        else:
            self.cfg, self.hw, self.B, self.thr = gc.BEDROOM_CFG, 256, env.args.batch or 4, 0.05
            self.flop_unit, self.label = FLOP_BEDROOM_IMG_STEP, 'ddpm-ema-bedroom-256 topology (113.7M params, seeded weights)'
        self.metric = 'importance-scored images/sec (UNet fwd+bwd+|w*dL/dw|)'
        self.unit, self.scaling = 'images/s', 'weak'
        self.sched = diffusion.DDPMScheduler()
        B, hw, rank = self.B, self.hw, env.rank
        self.clean = torch.from_numpy(gc.det_clean((B, 3, hw, hw), 100 + rank)).to(env.dev)
        self.noise = torch.from_numpy(gc.det_noise((B, 3, hw, hw), 200 + rank)).to(env.dev)
        self.model = self._model()
        self.flat = sweep.flatten_grads(self.model)
        self.step = self._step(self.model)

    def _model(self):
        m = pkg('unet').UNet2DModel(**self.cfg)
        gc.det_init_(m, 0)
        return m.to(self.env.dev).eval()

    def _step(self, model, pipelines=None):
        n = self.env.world * self.B
        return pkg('sweep').HipSweepStep(model, self.sched, self.clean, self.noise, n * self.clean[0].numel(), 'mse', n,
                                         timestep_pipelines=pipelines)

    def warmup(self, W):
        # every timestep pipeline is created by its first step: a warm-up shorter than the number of pipelines (--warmup 1) would
        # leave that one-off (~0.3 s of allocations) inside the timed region, so W >= 1 warms each pipeline at least once
        for k in range(max(W, getattr(self.step, '_tp_want', 1)) if W else 0):
            self.step(k)
        if self.env.args.graph:
            self.step.capture()
            self.step(0)
        self.step.finish()
        self.flat.zero_()

    def run(self, K):
        sweep, lib = pkg('sweep'), pkg('ops')._lib()
        tm = {}
        launches0 = lib.dp_launch_count()
        t0 = time.perf_counter()
        # the product's own driver: K timesteps, (Diff-Pruning: stream-ordered loss all-reduce + on-device early exit,) the one
        # gradient all-reduce of the sweep -- the code path tests/test_dist_cpu.py covers is the one timed
        res = sweep.taylor_sweep(self.model, self.sched, self.clean, self.noise, num_steps=K, thr=self.thr, step_fn=self.step,
                                 flat_grads=self.flat, timings=tm, use_graph=False)
        launches = lib.dp_launch_count() - launches0
        pr = sweep.prune_model(self.model, 0.3)                          # scoring + mask selection + slicing
        self.env.barrier()
        t_total = time.perf_counter() - t0
        return dict(units=self.env.world * self.B * res['steps'], steps_done=res['steps'], t_total=t_total, t_steps=tm['sweep_s'],
                    extra={'sweep_only_images_per_s': self.env.world * self.B * res['steps'] / tm['sweep_s'],
                           'tail_ms': (t_total - tm['sweep_s']) * 1e3,
                           # the host's own time per timestep (Python + ctypes + hipLaunchKernel); what it spends WAITING for the
                           # device -- sweep.MAX_STEPS_AHEAD bounds how far it runs ahead -- is reported beside it.  Before round 6
                           # the figure included the time the HIP runtime blocks a launch once the queue is full (37-50 ms then).
                           'host_enqueue_ms_per_step': tm['enqueue_s'] / res['steps'] * 1e3,
                           'host_wait_for_device_ms_per_step': tm.get('throttle_wait_s', 0.0) / res['steps'] * 1e3,
                           'max_steps_ahead': sweep.MAX_STEPS_AHEAD,
                           'kernel_launches_per_step': launches / res['steps'],
                           'grad_allreduce_ms': tm.get('allreduce_s', 0.0) * 1e3,
                           'wgrad_stream_overlap': bool(self.step.eng._overlap_now), 'hipgraph': bool(self.env.args.graph),
                           'timestep_pipelines': 1 + len(getattr(self.step, '_tp', None) or []) if self.thr is None else 1,
                           'diff_pruning_threshold': self.thr,
                           'pruned_groups': len(pr.records), 'params_after': sum(p.numel() for p in self.model.parameters()),
                           'loss_first_last': [res['losses'][0], res['losses'][-1]]})

    def workload(self, K):
        return ('%s, batch %d/GPU, %s sweep t=0..%d + grad all-reduce + |w*g| scoring + mask selection + slicing (ratio 0.3)'
                % (self.label, self.B, 'Taylor' if self.thr is None else 'Diff-Pruning (thr %.2f)' % self.thr, K - 1))

    def config(self):
        return {'global_batch': self.env.world * self.B, 'image': '3x%dx%d' % (self.hw, self.hw),
                'parallelism': 'dp%d (batch shards)' % self.env.world}

    def instrumented(self):
        model2 = self._model()
        pkg('sweep').flatten_grads(model2)
        step2 = self._step(model2, pipelines=1)
        step2.eng.overlap_wgrad = False      # per-kernel durations: one kernel on the GPU at a time, at the shapes of the
        if step2._half is not None:          # timed region
            step2._half['serial'] = True
            step2._half['eng'].overlap_wgrad = False
        return lambda k: step2(k)


class FinetuneWorkload:
    def __init__(self, env):
        self.env = env
        train, diffusion = pkg('train'), pkg('diffusion')
        self.B = env.args.batch or 128
        self.metric, self.unit, self.scaling = 'finetune images/sec (pruned UNet fwd+bwd+clip+Adam+EMA)', 'images/s', 'weak'
        self.flop_unit = FLOP_PRUNED_CIFAR_IMG_STEP
        self.model = _pruned_cifar(env.dev)
        self.sched = diffusion.DDPMScheduler()
        self.ft = train.FinetuneEngine(self.model, self.sched, lr=2e-4, dropout=0.1, dropout_seed=1)
        B = self.B
        self.clean = torch.from_numpy(gc.det_clean((B, 3, 32, 32), 300 + env.rank)).to(env.dev)
        self.noise = torch.from_numpy(gc.det_noise((B, 3, 32, 32), 400 + env.rank)).to(env.dev)
        self.gen = torch.Generator().manual_seed(env.rank)
        self.k = 0

    def _ts(self):
        return pkg('train').antithetic_timesteps(self.B, 1000, self.gen).to(self.env.dev, non_blocking=True)

    def warmup(self, W):
        for _ in range(W):
            self.ft.step(self.clean, self.noise, self._ts())

    def run(self, K):
        lib = pkg('ops')._lib()
        ts = [self._ts() for _ in range(K)]
        torch.cuda.synchronize()
        launches0 = lib.dp_launch_count()
        t0 = time.perf_counter()
        for k in range(K):
            loss = self.ft.step(self.clean, self.noise, ts[k])
        t_enq = time.perf_counter() - t0
        self.env.barrier()
        t_total = time.perf_counter() - t0
        return dict(units=self.env.world * self.B * K, steps_done=K, t_total=t_total, t_steps=t_total,
                    extra={'host_enqueue_ms_per_step': t_enq / K * 1e3,
                           'kernel_launches_per_step': ((self.ft._cap['call'].info.get('kernels', 0) if self.ft._cap else 0)
                                                        + (lib.dp_launch_count() - launches0) / K),
                           'step_replayed_natively': self.ft._cap is not None,
                           'replay_info': dict(self.ft._cap['call'].info) if self.ft._cap else None, 'dropout': 0.1,
                           'params': sum(p.numel() for p in self.model.parameters()), 'last_local_loss': float(loss)})

    def workload(self, K):
        return ('pruned ddpm-cifar10 UNet (19.85M params) finetune step (ddpm_train.py), batch %d/GPU, dropout 0.1, '
                'grad all-reduce (bucketed, overlapped) + clip + Adam + EMA' % self.B)

    def config(self):
        return {'global_batch': self.env.world * self.B, 'image': '3x32x32', 'parallelism': 'dp%d (batch shards)' % self.env.world}

    def instrumented(self):
        if self.env.world > 1:
            return None            # the finetune step itself contains the gradient collectives: rank 0 cannot run one alone
        eng = self.model.engine()
        eng.overlap_wgrad = False
        ts = self._ts()

        def one(k):
            self.ft.replay = False                     # HIP events around every launch need the eager step
            self.model._engine.overlap_wgrad = False
            self.ft.step(self.clean, self.noise, ts)
        return one


class DdimWorkload:
    def __init__(self, env):
        self.env = env
        self.B = env.args.batch or 256
        self.metric, self.unit, self.scaling = 'DDIM image-steps/sec (pruned UNet forward + scheduler step)', 'image-steps/s', 'weak'
        self.flop_unit = FLOP_PRUNED_CIFAR_FWD
        self.model = _pruned_cifar(env.dev)
        diffusion = pkg('diffusion')
        self.sched = diffusion.DDIMScheduler()
        self.sched.set_timesteps(100)
        self.x = torch.from_numpy(gc.det_noise((self.B, 3, 32, 32), 500 + env.rank)).to(env.dev)

    def _steps(self, n, fwd):
        """n steps of ddpm_sample.py's inner loop as DDIMPipeline.__call__ runs them: eps = UNet(x, t) through `fwd`
        (UNet2DModel.sampling_forward: the captured forward replayed natively, or the eager pinned call), then the scheduler step."""
        x = self.x
        ts = [int(v) for v in self.sched.timesteps.tolist()]
        with torch.no_grad():
            for i in range(n):
                t = ts[i % len(ts)]
                e = fwd(x, t)
                x = self.sched.step(e, t, x, eta=0.0).prev_sample
        return x

    def warmup(self, W):
        self.fwd = self.model.sampling_forward(tuple(self.x.shape), 100)       # what a 100-step DDIMPipeline call gets
        self._steps(W, self.fwd)

    def run(self, K):
        lib = pkg('ops')._lib()
        if getattr(self.fwd, '_pin', None) is None:                             # closed by an earlier run(): a fresh (captured) forward
            self.fwd = self.model.sampling_forward(tuple(self.x.shape), 100)
            self._steps(1, self.fwd)
            torch.cuda.synchronize()
        launches0 = lib.dp_launch_count()
        t0 = time.perf_counter()
        x = self._steps(K, self.fwd)
        t_enq = time.perf_counter() - t0
        self.env.barrier()
        t_total = time.perf_counter() - t0
        replayed = type(self.fwd).__name__ == '_CapturedForward'
        info = dict(self.fwd.call.info) if replayed else {}
        self.fwd.close()
        return dict(units=self.env.world * self.B * K, steps_done=K, t_total=t_total, t_steps=t_total,
                    extra={'host_enqueue_ms_per_step': t_enq / K * 1e3,
                           'kernel_launches_per_step': (info.get('kernels', 0) + (lib.dp_launch_count() - launches0) / K) if replayed
                           else (lib.dp_launch_count() - launches0) / K,
                           'forward_replayed_natively': replayed, 'replay_nodes': info.get('nodes'),
                           'finite': bool(torch.isfinite(x).all())})

    def workload(self, K):
        return 'pruned ddpm-cifar10 UNet (19.85M params) DDIM sampling loop (ddpm_sample.py), batch %d/GPU, eta 0' % self.B

    def config(self):
        return {'global_batch': self.env.world * self.B, 'image': '3x32x32',
                'parallelism': 'dp%d (independent batches per rank)' % self.env.world}

    def instrumented(self):
        eager = self.model.sampling_forward(tuple(self.x.shape), 1, replay=False)     # HIP events need the eager launches
        return lambda k: self._steps(1, eager)


class LdmWorkload:
    def __init__(self, env):
        self.env = env
        ldm, ldm_sweep = pkg('ldm'), pkg('ldm_sweep')
        self.n = env.args.batch or 6
        self.metric, self.unit, self.scaling = 'importance-scored latents/sec (20-step CFG DDIM sampling + UNet fwd+bwd)', 'latents/s', 'strong'
        self.ddim_steps = 20
        self.flop_unit = 2 * self.ddim_steps * FLOP_LDM_FWD + FLOP_LDM_FWD_BWD
        m = ldm.UNetModel(**gc.LDM_CIN256_CFG)
        gc.det_init_(m, 1)
        self.model = m.to(env.dev).eval()
        emb = ldm_sweep.ClassEmbedder(512, 1001)
        with torch.no_grad():
            emb.embedding.weight.copy_(torch.from_numpy(gc.det_param('embedding.weight', (1001, 512), 61)))
        self.emb = emb.to(env.dev)
        # the weights are frozen from here to the end of the process (warm-up pass, timed pass, instrumented step): the packed
        # operands survive between the passes, as they do inside one 1000-step importance pass
        self._pin = self.model.pin_weights()
        self._pin.__enter__()

    def _pass(self, K, seed):
        import random
        return pkg('ldm_sweep').ldm_importance_sweep(self.model, self.emb, num_steps=K, thr=0.1, n_samples=self.n,
                                                     ddim_steps=self.ddim_steps, latent_shape=(3, 64, 64),
                                                     class_rng=random.Random(seed), seed=seed)

    def warmup(self, W):
        if W:
            self._pass(W, 1)

    def run(self, K):
        lib = pkg('ops')._lib()
        launches0 = lib.dp_launch_count()
        t0 = time.perf_counter()
        res = self._pass(K, 2)
        self.env.barrier()
        t_total = time.perf_counter() - t0
        return dict(units=self.n * res['steps'], steps_done=res['steps'], t_total=t_total, t_steps=t_total,
                    extra={'kernel_launches_per_step': (lib.dp_launch_count() - launches0) / max(res['steps'], 1),
                           'latents_this_rank': res['shard'][1] - res['shard'][0], 'accumulated': res['accumulated'],
                           'loss_first_last': [res['losses'][0], res['losses'][-1]],
                           'params': sum(p.numel() for p in self.model.parameters())})

    def workload(self, K):
        return ('LDM cin256-v2 UNet (400.9M params, seeded weights) importance pass (prune_ldm.py): %d latents 3x64x64 per step '
                'sharded over the ranks, %d-step CFG DDIM sampling (scale 3.0) + loss at t + backward, t=0..%d, thr 0.1'
                % (self.n, self.ddim_steps, K - 1))

    def config(self):
        return {'global_batch': self.n, 'image': 'latent 3x64x64', 'parallelism': 'dp%d (latents sharded: strong scaling)' % self.env.world}

    def instrumented(self):
        ldm_sweep, sweep = pkg('ldm_sweep'), pkg('sweep')
        dev = self.env.dev
        lo, hi = ldm_sweep.shard_bounds(self.n, self.env.rank, self.env.world)
        nl = hi - lo
        sched = ldm_sweep.LdmSchedule()
        sweep.flatten_grads(self.model)
        step = ldm_sweep.LdmSweepStep(self.model, sched, global_numel=self.n * 3 * 64 * 64)
        xT = torch.randn(nl, 3, 64, 64, device=dev)
        noise = torch.randn(nl, 3, 64, 64, device=dev)
        c = self.emb(torch.arange(nl, device=dev))
        uc = self.emb(torch.full((nl,), 1000, device=dev))
        tt = torch.full((nl,), 500, dtype=torch.long, device=dev)

        def one(k):
            self.model.engine().overlap_wgrad = False
            with self.model.pin_weights():
                x0 = ldm_sweep.ddim_sample_cfg(self.model, sched, xT, c, uc, S=self.ddim_steps, scale=3.0)
                step.eng.overlap_wgrad = False
                step.loss(x0, tt, c, noise)
                step.backward()
        return one


def kernel_switches(ops):
    """Which kernel families the measured process had active (module-level switches of ops.py / engine.py, i.e. the DP_*
    environment as it was read at import) and every DP_* variable set in the environment: a stray DP_WINO=0 must show on the line."""
    eng, sweep = pkg('engine'), pkg('sweep')
    return {'wino': bool(ops.WINO), 'wgrad_wino': bool(ops.WINO and ops.WGRAD_WINO), 'wino_min_tiles': ops.WINO_MIN_TILES,
            'splitk_fold': bool(ops.SPLITK_FOLD), 'splitk_fold_max': ops.SPLITK_FOLD_MAX,
            'wino43_no_grad_forwards': getattr(ops, 'WINO43', False),
            'wino2d': bool(ops.WINO and getattr(ops, 'WINO2D', False)), 'wino2d_min_tiles': getattr(ops, 'WINO2D_MIN_TILES', None),
            'fused_attn': ops.FUSED_ATTN, 'ups_subpixel': bool(eng.UPS_SUBPIXEL), 's2_parity': bool(eng.S2_PARITY),
            'timestep_pipelines_default': sweep.TIMESTEP_PIPELINES,
            'env': {k: v for k, v in sorted(os.environ.items()) if k.startswith('DP_')}}


def _pmc_traffic(config_name, dom):
    """HBM bytes per launch of the dominant kernel from the rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE in separate runs of
    THIS config's bench command, aggregated by tools/pmc_aggregate.py; KB -> bytes; they cannot run inside this process).  Read from
    the newest profiles/round*_pmc_bench_traffic[_<config>].json whose `_config` equals the config being timed and whose recorded
    hash of the contraction kernels' sources (csrc/gemm.hip + csrc/winograd.hip + csrc/winograd2d.hip + csrc/wgrad2d.hip + the two *_kloop.inc) equals today's; anything else = null.
    FETCH_SIZE is uncalibrated for 4-byte-per-lane buffer loads on gfx950 (MI355X_MICROARCH.md, HBM section)."""
    import glob
    import hashlib
    try:
        sfx = '' if config_name == 'cifar256' else '_' + config_name
        cand = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'round*_pmc_bench_traffic%s.json' % sfx)))
        if not cand:
            return None, dict(file=None, reason='no PMC pass of config %r under profiles/' % config_name)
        src = b''.join(open(os.path.join(ROOT, 'diff-pruning_amd', 'csrc', f), 'rb').read() for f in ('gemm.hip', 'winograd.hip', 'winograd2d.hip', 'winograd2d_kloop.inc', 'wgrad2d.hip', 'wgrad2d_kloop.inc'))
        blob = hashlib.sha1(b'blob %d\0' % len(src) + src).hexdigest()          # the contraction kernels' sources, concatenated
        pm = json.load(open(cand[-1]))
        measured_on, measured_cfg = pm.get('_gemm_hip_blob'), pm.get('_config', 'cifar256')
        src_info = dict(file=os.path.relpath(cand[-1], ROOT), measured_on_gemm_hip_blob=measured_on, current_gemm_hip_blob=blob,
                        stale=measured_on != blob, config=measured_cfg)
        k = pm.get(dom)
        if k and measured_on == blob and measured_cfg == config_name:
            return (k['FETCH_SIZE']['avg_kb'] + k['WRITE_SIZE']['avg_kb']) * 1024.0, src_info
        return None, src_info
    except (OSError, KeyError, ValueError, IndexError):
        return None, None


def roofline(ops, one_step, step_seconds, flop_reference_per_step, config_name='cifar256'):
    """HIP events around every contraction launch of one instrumented step (after one un-instrumented pass of the same step)."""
    one_step(0)
    torch.cuda.synchronize()
    ops._prof = []
    one_step(1)
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
    traffic, traffic_src = _pmc_traffic(config_name, dom)
    executed = sum(v[1] for v in agg.values())
    # The Winograd F(2, 3) kernels execute 2/3 of the multiply-adds of the convolution they compute (6 instead of 9 per output,
    # channel pair and pixel): `achieved` / `frac` are EXECUTED FLOPs over time (the number to hold against the matrix pipe);
    # the same launches in the direct form's arithmetic -- the reference's -- are 1.5x that.
    # The F(2x2, 3x3) kernel (round 6) executes 4/9: 16 multiplies per 2x2 output tile and channel pair instead of 36 -> x 2.25.
    ref_eq = 2.25 if 'wino2d' in dom else 1.5 if 'wino' in dom else 1.0
    return dict(bound='mfma', kernel=dom, achieved=fl / sec / 1e12, peak=PEAK_F32_TFLOPS, unit='TFLOP/s',
                frac=fl / sec / 1e12 / PEAK_F32_TFLOPS, achieved_in_reference_arithmetic=ref_eq * fl / sec / 1e12,
                traffic=traffic, traffic_source=traffic_src,
                algorithmic_bytes_per_launch=ab / cnt, launches_per_step=cnt, avg_launch_ms=sec / cnt * 1e3,
                flop_per_launch=fl / cnt, step_share=sec / step_seconds,
                kernels={n: dict(launches=v[0], tflops=v[1] / v[2] / 1e12, ms=v[2] * 1e3) for n, v in agg.items()},
                # Whole-step rates.  `executed`: the multiply-adds the kernels of one step actually perform (sum over the
                # instrumented launches) -- the number to hold against the MFMA peak.  `reference_equivalent`: SURVEY 8(d)'s count
                # of the reference's own arithmetic over the same time; it exceeds the executed count because the upsample
                # convolutions run in their sub-pixel form, stride-2 input gradients by parity classes and the 3x3 layers as
                # Winograd F(2, 3) -- a rate of useful work, not of hardware utilisation (it may exceed 1.0 of the peak).
                executed_flop_per_step=executed, step_tflops=executed / step_seconds / 1e12,
                step_frac=executed / step_seconds / 1e12 / PEAK_F32_TFLOPS,
                step_tflops_reference_equivalent=flop_reference_per_step / step_seconds / 1e12,
                step_frac_reference_arithmetic=flop_reference_per_step / step_seconds / 1e12 / PEAK_F32_TFLOPS)


DEFAULT_STEPS = {'cifar256': (20, 2), 'bedroom256': (20, 2), 'c4_finetune': (20, 3), 'ddim': (40, 5), 'ldm': (4, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None)
    ap.add_argument('--warmup', type=int, default=None)
    ap.add_argument('--config', default='cifar256', choices=sorted(DEFAULT_STEPS))
    ap.add_argument('--batch', type=int, default=None, help='images per GPU (ldm: latents per step, global)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--graph', action='store_true', help='replay the timestep from a captured hipGraph (sweep configs)')
    ap.add_argument('--backend', default='nccl', help="process-group backend; 'gloo' with ranks sharing a GPU is a logic check of the N > 1 path on a 1-GPU box, not a measurement")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = DEFAULT_STEPS[args.config][0]
    if args.warmup is None:
        args.warmup = DEFAULT_STEPS[args.config][1]

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
        if args.backend == 'nccl':
            torch.cuda.set_device(local_rank)
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            local_rank = local_rank % torch.cuda.device_count()
            dist.init_process_group(args.backend)
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
        world = dist.get_world_size()
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)
    ops = pkg('ops')
    ops._lib()                                   # fail loudly if the HIP library is missing
    env = Env(args, dev, rank, world)

    if args.config in ('cifar256', 'bedroom256'):
        wl = SweepWorkload(env, args.config)
    elif args.config == 'c4_finetune':
        wl = FinetuneWorkload(env)
    elif args.config == 'ddim':
        wl = DdimWorkload(env)
    else:
        wl = LdmWorkload(env)

    wl.warmup(args.warmup)
    env.barrier()
    r = wl.run(args.steps)                        # starts its clock right after the barrier, ends it behind another one
    tt = torch.tensor([r['t_total'], r['t_steps']], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    t_total, t_steps = float(tt[0]), float(tt[1])
    steps_done = r['steps_done']
    step_seconds = t_steps / max(steps_done, 1)
    units_per_step = r['units'] / max(steps_done, 1)

    roof = None
    if rank == 0 and not args.no_roofline:
        # per-rank work of one step against this rank's kernels (weak scaling: 1/world of the units; ldm: this rank's latents)
        per_rank_units = units_per_step / world if wl.scaling == 'weak' else r['extra'].get('latents_this_rank', units_per_step)
        one_step = wl.instrumented()
        roof = roofline(ops, one_step, step_seconds, wl.flop_unit * per_rank_units, args.config) if one_step is not None else None
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.config)

    if rank == 0:
        cfg = {'workload': wl.workload(args.steps), 'name': args.config}
        cfg.update(wl.config())
        cfg.update(r['extra'])
        cfg['steps_executed'] = steps_done
        cfg['kernels'] = kernel_switches(ops)
        out = {'metric': wl.metric, 'value': r['units'] / t_total, 'unit': wl.unit, 'n_gpus': world, 'steps': args.steps,
               'warmup': args.warmup, 'ms_per_step': step_seconds * 1e3, 'higher_is_better': True, 'scaling': wl.scaling,
               'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic', 'config': cfg, 'roofline': roof, 'cpu_baseline': cpu}
        if args.backend != 'nccl':
            out['config']['backend'] = args.backend + ' (logic check, not a measurement)'
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
