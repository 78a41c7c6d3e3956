"""Taylor / Diff-Pruning gradient sweep (ddpm_prune.py:94-106) on the HIP engine, batch-sharded over ranks.

For t = 0, 1, 2, ...: noisy = add_noise(clean, noise, t); out = UNet(noisy, t); L_t = mse(out, noise);
d L_t / d w is ACCUMULATED into the parameter gradients (never zeroed inside the sweep).  Diff-Pruning stops
once L_t < thr * max_s L_s -- after accumulating the breaking step, exactly as the reference does.

Data parallelism (new w.r.t. the reference): every rank walks the same timestep sequence on
its own slice of the image batch.  L_t is a mean over the *global* batch, hence linear in per-image terms:
each rank scales its loss / gradient by 1/numel_global, the early-exit test uses the all-reduced scalar loss
(so every rank stops at the same t), and the gradients are summed ONCE at the end of the sweep with a single
all-reduce of the flat gradient buffer (RCCL over xGMI when the process group is 'nccl').  Importance is
non-linear in the gradient, so scores are only ever computed from the reduced gradients.
"""
import os

import torch

from . import ops
from .engine import UNetEngine, shared_stream


GRAPH_AUTO_PIXELS = 8192        # taylor_sweep(use_graph=None): shards up to this many pixels (CIFAR: batch <= 8) ...
# [measured, round 3, CIFAR UNet batch 256, one box] ms per timestep with 1 / 2 / 3 / 4 timestep pipelines: 78.8 / 76.9 / 77.5 / 76.7
# (without the weight-gradient side streams: 79.1 with two).  Two is the default for plain Taylor sweeps.
TIMESTEP_PIPELINES = 2
GRAPH_AUTO_STEPS = 64           # ... swept for at least this many timesteps replay one captured timestep (native replay list)
# A plain Taylor sweep never reads anything back, so nothing stops the host from enqueueing hundreds of timesteps ahead of the
# device: every cross-stream event, kernel-argument block and allocator block of those timesteps is then outstanding at once
# (round 6: the allocator's reserved memory grew to the whole HBM within a 1000-timestep sweep at batch 256 -- DESIGN.md section 5
# "Round 6").  HipSweepStep.__call__ lets the host run at most this many timesteps ahead (over all timestep pipelines): it waits
# for the event recorded behind timestep k - MAX_STEPS_AHEAD before enqueueing timestep k.  The device never runs dry (4 steps =
# 45 ... 250 ms of queued work) and a GPU-bound sweep loses nothing.  0 = unbounded.
MAX_STEPS_AHEAD = int(os.environ.get('DP_MAX_STEPS_AHEAD', '4'))


def dist_active(group=None):
    """True when the exchange steps of the data-parallel path run: an initialised process group of more than one rank -- or of
    ONE rank under DP_FORCE_DIST=1, which is how the `-m gpu` suite drives every collective of the path (scalar-loss all-reduce,
    flat-gradient all-reduce, finetune buckets, FID statistics) through RCCL on the single GPU of the test box."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or os.environ.get('DP_FORCE_DIST') == '1'


class StepThrottle:
    """`mark(stream)` behind every enqueued step: records 'step enqueued' on the stream the step ran on and first waits (host) for
    the mark `max_ahead` steps back -- see MAX_STEPS_AHEAD.  Events are reused round-robin; `wait_s` accumulates the host's waits."""

    def __init__(self, max_ahead=None):
        self.max_ahead = MAX_STEPS_AHEAD if max_ahead is None else max_ahead
        self.ring, self.i, self.wait_s = [], 0, 0.0

    def mark(self, stream=None):
        n = self.max_ahead
        if n <= 0 or not torch.cuda.is_available() or torch.cuda.is_current_stream_capturing():
            return
        slot = self.i % n
        if len(self.ring) < n:
            self.ring.append(torch.cuda.Event())
        elif not self.ring[slot].query():                # recorded n steps ago
            import time
            t0 = time.perf_counter()
            self.ring[slot].synchronize()
            self.wait_s += time.perf_counter() - t0
        self.ring[slot].record(stream if stream is not None else torch.cuda.current_stream())
        self.i += 1


def flatten_grads(model):
    """Point every parameter's .grad at a slice of one zero-initialised flat fp32 buffer; returns the buffer."""
    params = [p for p in model.parameters()]
    total = sum(p.numel() for p in params)
    flat = torch.zeros(total, dtype=torch.float32, device=params[0].device)
    off = 0
    for p in params:
        n = p.numel()
        p.grad = flat[off:off + n].view_as(p)
        off += n
    return flat


class HipSweepStep:
    """One timestep of the sweep on the local image shard (forward + loss + backward on the HIP kernels).
    `micro` (images): the shard is walked in micro-batches of that size within every timestep (gradients accumulate, the
    per-image loss scaling is the global one, so the result is the same sum) -- for shards whose activations would
    exceed the 2 GiB-per-tensor limit of the buffer descriptors (e.g. 256x256 images at batch >= 64 per GPU)."""
    _graph = None
    _replay = None
    micro = None
    _half = None
    stop_state = None          # device [loss_max, stopped, steps] of the on-device Diff-Pruning early exit (taylor_sweep)
    _tp = None
    _tp_want = 1
    _tp_count = 0
    sequential = False         # True: the caller reads every step's loss before the next (host-side threshold test): one pipeline

    def __init__(self, model, scheduler, clean, noise, global_numel, loss_kind='mse', global_batch=None, halves=None,
                 timestep_pipelines=None):
        if clean.device.type != 'cuda':
            raise RuntimeError('the sweep runs on the MI355X HIP kernels only (no CPU fallback)')
        self.model, self.scheduler = model, scheduler
        self.clean, self.noise = clean.contiguous().float(), noise.contiguous().float()
        self.B = clean.shape[0]
        if loss_kind == 'mse':                       # F.mse_loss: mean over every element of the global batch
            self.gscale, self.lscale = 2.0 / global_numel, 1.0 / global_numel
        else:                                        # sum over C,H,W then mean over the global batch
            gb = global_batch if global_batch is not None else self.B
            self.gscale, self.lscale = 2.0 / gb, 1.0 / gb
        self.eng = model.engine()
        self._P = {n: p.detach() for n, p in model.named_parameters()}
        self._G = {n: p.grad for n, p in model.named_parameters()}
        self.eng.bind(self._P, self._G)
        self.acp = scheduler._acp_on(clean.device)
        # Two half-batch pipelines on two HIP streams (the gradient is linear in the images, exactly the property the
        # multi-GPU path relies on): the second half runs through its own engine -- same parameters, own context, own weight-
        # gradient stream -- into a SECOND flat gradient buffer that is added to the first once, in finish().  Every
        # convolution of one half is a single round of workgroups with a memory-bound ramp and tail; two independent
        # kernel streams fill each other's ramps and hide the HBM-bound GroupNorm / reduction kernels of one half under the
        # MFMA kernels of the other.  Same kernels, fixed order: run-to-run bit-identical; vs. one pipeline the sums are
        # re-associated like a 2-rank data-parallel run.
        # [measured, round 2, B=256 CIFAR] 92.9 ms/step with two pipelines vs 89.6 with one: the half-size launches
        # (512 workgroups = two per CU) run at 107 instead of 122 TFLOP/s and the overlap does not buy that back, so the
        # default stays ONE pipeline; `halves=2` / DP_HALVES=2 remains selectable (tested: same masks, bit-reproducible).
        if halves is None:
            halves = int(os.environ.get('DP_HALVES', '1'))
        if halves == 2 and self.B >= 2 and isinstance(self.eng, UNetEngine) and type(self.eng) is UNetEngine:
            self._setup_second_half()
        # Two TIMESTEPS in flight (plain Taylor only: no step depends on another).  Timesteps of odd position run through a second
        # engine -- same parameters, own context, own flat gradient buffer, own stream -- at the FULL batch, so the kernels keep
        # their size (the half-batch pipelines above lost to their smaller launches) and the two independent kernel streams fill
        # each other's ramps, tails and HBM-bound phases.  No cross-stream edge between the pipelines until finish().
        if timestep_pipelines is None:
            # small shards are bound by the host's launch rate, not by the GPU (batch 4: 10.90 vs 10.96 ms): one pipeline, which
            # also keeps them bit-identical to the replayed form
            from .engine import OVERLAP_MIN_WORK
            width = self.eng.cfg.get('block_out_channels', [self.eng.cfg.get('model_channels', 128)])[0]
            big = clean.shape[0] * clean.shape[2] * clean.shape[3] * width >= OVERLAP_MIN_WORK
            timestep_pipelines = int(os.environ.get('DP_TIMESTEP_PIPELINES', str(TIMESTEP_PIPELINES if big else 1)))
        # created lazily by the first call that can use them (never with a threshold, micro-batches or a captured step), so a
        # sweep that cannot run two timesteps at once pays neither the second activation context nor the gradient buffer
        self._tp = None
        self._tp_want = timestep_pipelines if (self._half is None and type(self.eng) is UNetEngine) else 1
        self._tp_count = 0

    def _make_timestep_pipelines(self):
        """The extra pipelines of `timestep_pipelines` >= 2; on an allocation failure the sweep stays on one pipeline."""
        self._tp = []
        try:
            for _ in range(self._tp_want - 1):
                self._setup_second_half()
                tp, self._half = self._half, None
                tp['eng'].set_dropout(self.eng.dropout, self.eng.drop_seed, self.eng.drop_step, self.eng.drop_n_off)
                tp['synced'] = False
                self._tp.append(tp)
        except torch.cuda.OutOfMemoryError:
            import warnings
            warnings.warn('no memory for a second timestep pipeline (gradient buffer): sweeping with one')
            self._half, self._tp = None, []
        return self._tp

    def _setup_second_half(self):
        dev = self.clean.device
        total = sum(g.numel() for g in self._G.values())
        flat2 = torch.zeros(total, dtype=torch.float32, device=dev)
        G2, off = {}, 0
        for n, g in self._G.items():
            G2[n] = flat2[off:off + g.numel()].view_as(g)
            off += g.numel()
        eng2 = UNetEngine(self.eng.cfg)
        slot = 1 + len(self._tp or [])                    # pipeline k of this step: stream slot k (engine.shared_stream)
        eng2.stream_slot = slot
        eng2.packs = self.eng.packs                       # frozen weights: one set of packed operands for both halves
        eng2.bind(self._P, G2)
        eng2.set_dropout(self.eng.dropout, self.eng.drop_seed, self.eng.drop_step, self.eng.drop_n_off + self.B // 2)
        self.eng.bind(self._P, self._G)
        self.eng.prepare_packs()                          # packed before the second stream's first read
        self._half = dict(eng=eng2, G=G2, flat=flat2, stream=shared_stream(dev, 'pipeline', slot), h=self.B // 2, serial=False)

    def finish(self):
        """Fold the second pipeline's gradients into the parameters' .grad buffers (once per sweep)."""
        others = [self._half] if self._half is not None else (getattr(self, '_tp', None) or [])
        for second in others:                                  # fixed order: deterministic sums
            if second.get('synced') is False:
                continue                                       # a timestep pipeline that never ran (sweep with a threshold)
            cur = torch.cuda.current_stream()
            cur.wait_stream(second['stream'])
            flat = self._flat_of_grads()
            if flat is not None:                               # .grad buffers are consecutive views of one flat buffer: one launch
                ops.axpby(second['flat'], 1.0, flat, 1.0)
            else:
                for n, g in self._G.items():
                    g2 = second['G'][n]
                    ops.axpby(g2.reshape(-1), 1.0, g.reshape(-1), 1.0)
            second['flat'].zero_()

    def _flat_of_grads(self):
        """The flat buffer behind the parameters' .grad views (flatten_grads), when they tile it in order; else None."""
        gs = list(self._G.values())
        base = getattr(gs[0], '_base', None)
        total = sum(g.numel() for g in gs)
        if base is None or base.dim() != 1 or base.numel() != total or not base.is_contiguous():
            return None
        off = base.data_ptr()
        for g in gs:
            if g.data_ptr() != off or not g.is_contiguous():
                return None
            off += g.numel() * 4
        return base

    def _step(self, t):
        if self._half is not None and self.micro is None:
            return self._two_half_step(t)
        if self.micro is None or self.B <= self.micro:
            return self._micro_step(self.clean, self.noise, t)
        total = None
        for lo in range(0, self.B, self.micro):
            hi = min(lo + self.micro, self.B)
            l = self._micro_step(self.clean[lo:hi], self.noise[lo:hi], t[lo:hi])
            total = l if total is None else ops.axpby(l, 1.0, total, 1.0)
        return total

    def _two_half_step(self, t):
        hf = self._half
        h = hf['h']
        cur = torch.cuda.current_stream()
        s1 = cur if hf['serial'] else hf['stream']        # serial: both halves on one stream (per-kernel timing in bench.py)
        s1.wait_stream(cur)
        t.record_stream(s1)
        la = self._micro_step(self.clean[:h], self.noise[:h], t[:h])
        with torch.cuda.stream(s1):
            eng = hf['eng']
            noisy = ops.add_noise(self.clean[h:], self.noise[h:], self.acp, t[h:])
            out = eng.forward(noisy, t[h:], save=True)
            lb, dout = ops.mse_fwd_bwd(out, self.noise[h:], self.gscale, self.lscale, stop_state=self.stop_state)
            eng.backward(dout)
        cur.wait_stream(s1)
        lb.record_stream(cur)
        return ops.axpby(lb, 1.0, la, 1.0)

    def _micro_step(self, clean, noise, t):
        # any model(...) / model.engine() call between two sweep steps (a no-grad evaluation, the autograd bridge) re-binds
        # the engine without -- or with temporary -- gradient buffers: bind ours again
        self.eng.bind(self._P, self._G)
        noisy = ops.add_noise(clean, noise, self.acp, t)
        out = self.eng.forward(noisy, t, save=True)
        loss, dout = ops.mse_fwd_bwd(out, noise, self.gscale, self.lscale, stop_state=self.stop_state)
        self.eng.backward(dout)
        return loss

    # ---- two-phase step for the ddpm_exp / LDM flavour of Diff-Pruning, whose threshold test sits BEFORE the backward
    #      (ddpm_exp/prune.py:249-256): the breaking timestep contributes no gradient
    def forward_loss(self, k):
        if self.micro is not None and self.B > self.micro:
            raise NotImplementedError('break-before-backward keeps one forward context: not combined with micro-batches')
        t = torch.full((self.B,), int(k), dtype=torch.long, device=self.clean.device)
        self.eng.bind(self._P, self._G)
        noisy = ops.add_noise(self.clean, self.noise, self.acp, t)
        out = self.eng.forward(noisy, t, save=True)
        loss, self._dout = ops.mse_fwd_bwd(out, self.noise, self.gscale, self.lscale, stop_state=self.stop_state)
        return loss

    def backward_pending(self, cancel_if_stopped=False):
        if cancel_if_stopped and self.stop_state is not None:
            ops.zero_if_stopped(self._dout, self.stop_state)      # the breaking step itself contributes no gradient
        self.eng.backward(self._dout)
        self._dout = None

    def discard_pending(self):
        self.eng.ctx = None
        self._dout = None

    def capture(self, native=None):
        """Record one timestep (~700 kernel launches) into a hipGraph; afterwards every step is: write t, replay.
        Removes the ~60 ms of Python/ctypes launch overhead per step -- the launch-bound regime at small batch.
        native (default: True unless DP_REPLAY=graph): the captured graph is not instantiated; its nodes are read back and
        re-issued from the library's C loop (csrc/replay.hip: dp_replay_build / dp_replay_launch) -- hipGraphLaunch itself costs
        more host time than the eager launches on this stack."""
        if native is None:
            native = os.environ.get('DP_REPLAY', 'native') != 'graph'
        self.eng.prepare_packs()
        ops._workspace(1 << 26, self.clean.device)          # split-K workspace must exist before capture
        self._t = torch.zeros(self.B, dtype=torch.long, device=self.clean.device)
        # one eager pass before the capture (code objects loaded, workspaces and allocator pools warm) that leaves the
        # accumulated gradients untouched: with the early-exit state set to "stopped" the loss kernel emits dOut = 0 and
        # every += epilogue adds an exact zero
        real_state = self.stop_state
        self.stop_state = torch.tensor([0.0, 1.0, 0.0], dtype=torch.float32, device=self.clean.device)
        self._step(self._t)
        self.stop_state = real_state
        torch.cuda.synchronize()
        if native:
            try:
                g = torch.cuda.CUDAGraph(keep_graph=True)
            except TypeError:                              # a torch without keep_graph: no raw graph to read back
                import warnings
                warnings.warn('torch.cuda.CUDAGraph has no keep_graph: the captured timestep replays through hipGraphLaunch')
                native, g = False, torch.cuda.CUDAGraph()
        else:
            g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._loss = self._step(self._t)
        self._graph = g
        self._replay = None
        if native:
            from ._lib import DpHipError
            try:
                self._replay = ops.ReplayList(g)
            except DpHipError as e:
                if e.code != 801:                          # anything but hipErrorNotSupported is a real error of the build
                    raise
                import warnings                            # a node the list cannot re-issue: replay through hipGraphLaunch instead
                warnings.warn('native replay refused the captured timestep (a node it cannot re-issue): falling back to '
                              'hipGraphLaunch, measured ~3x slower per timestep')
                g.instantiate()
        return self

    def _second_pipeline_step(self, k, tp):
        """Timestep k on another pipeline's stream and engine (full batch, own gradient buffer)."""
        if not tp['synced']:             # once: inputs, schedule table and every packed operand (also the ones the first timestep
            tp['stream'].wait_stream(torch.cuda.current_stream())      # packed lazily on the main stream) are complete
            tp['synced'] = True
        with torch.cuda.stream(tp['stream']):
            t = torch.full((self.B,), int(k), dtype=torch.long, device=self.clean.device)
            eng = tp['eng']
            noisy = ops.add_noise(self.clean, self.noise, self.acp, t)
            out = eng.forward(noisy, t, save=True)
            loss, dout = ops.mse_fwd_bwd(out, self.noise, self.gscale, self.lscale)
            eng.backward(dout)
        return loss

    def _throttle(self, stream=None):
        if self.clean.device.type != 'cuda':
            return
        th = self.__dict__.get('_th')
        if th is None:
            th = self._th = StepThrottle()
        th.mark(stream)
        self.throttle_wait_s = th.wait_s

    throttle_wait_s = 0.0

    def __call__(self, k):
        if self._tp_want >= 2 and self.stop_state is None and self._graph is None and self.micro is None and not self.sequential:
            tps = self._tp if self._tp is not None else self._make_timestep_pipelines()
            i = self._tp_count % (len(tps) + 1)                # round-robin: main pipeline first
            self._tp_count += 1
            if i:
                loss = self._second_pipeline_step(k, tps[i - 1])
                self._throttle(tps[i - 1]['stream'])
                return loss
        if self._graph is not None:
            self._t.fill_(int(k))
            if self._replay is not None:
                self._replay.launch(self.eng.replay_side_stream(self.clean.device))
            else:
                self._graph.replay()
            loss = self._loss.clone()
            self._throttle()
            return loss
        t = torch.full((self.B,), int(k), dtype=torch.long, device=self.clean.device)
        loss = self._step(t)          # [1] device tensor: this rank's share of L_t
        self._throttle()
        return loss


def _f32_lt_prod(loss, loss_max, thr):
    """`loss < loss_max * thr` as the reference evaluates it: 0-d fp32 tensors, the product rounded to fp32."""
    import numpy as np
    return bool(np.float32(loss) < np.float32(loss_max) * np.float32(thr))


def taylor_sweep(model, scheduler, clean_images, noise, num_steps=1000, thr=None, loss_kind='mse', group=None,
                 step_fn=None, flat_grads=None, reduce_grads=True, use_graph=None, micro_batch=None,
                 accumulate_breaking_step=True, device_exit=True, poll_every=8, timings=None):
    """Runs the sweep; returns dict(losses=[python floats of the GLOBAL loss per executed step], steps=int).

    thr=None: plain Taylor.  thr=x: Diff-Pruning early exit.  accumulate_breaking_step=True is ddpm_prune.py:102-106
    (backward, then the threshold test); False is the ddpm_exp flavour (ddpm_exp/prune.py:249-256: test, break, else
    backward), to be combined with loss_kind='sum' (functions/losses.py:15).

    clean_images / noise: this rank's shard.  `group`: torch.distributed process group (None = default group if
    torch.distributed is initialised, single process otherwise).  `step_fn(k) -> local loss tensor` lets the
    multi-process CPU tests drive the same control flow with a different per-step engine.
    device_exit / poll_every: keep the Diff-Pruning early-exit state on the device and read the stop flag every
    `poll_every` timesteps (False: read the loss on the host after every step, as the reference does).
    use_graph: capture one timestep once and replay it (bit-identical; works with the on-device early exit).  The replay is the
    library's native list (csrc/replay.hip: the captured graph's nodes re-issued with hipLaunchKernel), not hipGraphLaunch.
    [measured, round 3, CIFAR UNet, batch 4] eager with the weight-gradient side stream 15.4 ms per timestep, eager without it
    11.2, hipGraph replay of the two-stream capture 27.6, native replay of a one-stream capture **8.5**: what made graph replay
    slow in rounds 1-2 was the ~80 cross-stream edges per timestep (each costs tens of microseconds of cross-queue
    synchronisation on this stack), not the node count.  At batch 16 replay and eager tie (11.8 / 12.0), from batch 64 on the
    step is GPU-bound.  None = automatic: shards of <= GRAPH_AUTO_PIXELS pixels swept for >= GRAPH_AUTO_STEPS timesteps
    (DP_GRAPH=off disables, DP_REPLAY=graph replays through hipGraphLaunch instead).
    timings: a dict that receives host wall-clock marks (bench.py): 'enqueue_s' (all timesteps enqueued), 'sweep_s' (device
    done with them) and 'allreduce_s' (the gradient exchange alone; only then is the exchange followed by a device sync)."""
    import time
    import torch.distributed as dist
    t_start = time.perf_counter()
    poll_wait = 0.0
    throttle_wait0 = float(getattr(step_fn, 'throttle_wait_s', 0.0))
    use_dist = dist_active(group)
    B_local = clean_images.shape[0]
    per_img = clean_images[0].numel()
    if use_dist:
        cnt = torch.tensor([float(B_local)], device=clean_images.device)
        dist.all_reduce(cnt, group=group)
        B_global = int(round(float(cnt)))
    else:
        B_global = B_local
    if flat_grads is None and step_fn is None:
        flat_grads = flatten_grads(model)
    two_phase = thr is not None and not accumulate_breaking_step
    own_step = step_fn is None
    if use_graph is None:
        use_graph = (os.environ.get('DP_GRAPH', 'auto') == 'auto' and own_step and not two_phase and micro_batch is None
                     and clean_images.device.type == 'cuda' and num_steps >= GRAPH_AUTO_STEPS
                     and B_local * clean_images.shape[2] * clean_images.shape[3] <= GRAPH_AUTO_PIXELS)
    stop_state = losses_dev = None
    if own_step:
        step_fn = HipSweepStep(model, scheduler, clean_images, noise, B_global * per_img, loss_kind, B_global)
        step_fn.micro = micro_batch
    # a threshold makes the steps sequential (the loss of step k decides whether step k + 1 runs, on the device or on the
    # host): timesteps of odd position must not run on another stream, whose loss the main stream's reads would not wait for
    if isinstance(step_fn, HipSweepStep):
        step_fn.sequential = thr is not None
        if use_graph:
            if thr is not None and device_exit and not two_phase:
                # the captured loss kernel reads the early-exit state: it has to exist before the capture
                stop_state = torch.zeros(3, dtype=torch.float32, device=clean_images.device)
                losses_dev = torch.zeros(num_steps, dtype=torch.float32, device=clean_images.device)
                step_fn.stop_state = stop_state
            step_fn.capture()
    losses = []
    pending = []
    loss_max = 0.0
    steps = 0
    if two_phase and use_graph:
        raise ValueError('use_graph replays forward + backward as one unit: not available with accumulate_breaking_step=False')
    on_device = (thr is not None and device_exit and (not use_graph or stop_state is not None)
                 and isinstance(step_fn, HipSweepStep) and hasattr(ops, 'early_exit_update')
                 and (clean_images.device.type == 'cuda' or getattr(ops, 'IS_MOCK', False)))
    if on_device:
        # The early-exit state lives on the device: no host read of the loss per step.  Timesteps enqueued after the stop are
        # exact no-ops (dOut = 0), the host looks at the flag every `poll_every` steps, and the scalar-loss all-reduce of the
        # data-parallel path is stream-ordered (RCCL), so nothing blocks the host between polls.
        dev = clean_images.device
        state = stop_state if stop_state is not None else torch.zeros(3, dtype=torch.float32, device=dev)
        if losses_dev is None:
            losses_dev = torch.zeros(num_steps, dtype=torch.float32, device=dev)
        step_fn.stop_state = state
        k = 0
        while k < num_steps:
            l = step_fn.forward_loss(k) if two_phase else step_fn(k)
            if use_dist:
                dist.all_reduce(l, group=group)
            ops.early_exit_update(l, thr, state, losses_dev)
            if two_phase:
                step_fn.backward_pending(cancel_if_stopped=True)
            k += 1
            if k % poll_every == 0 or k == num_steps:
                t_poll = time.perf_counter()
                stopped = float(state[1]) != 0.0            # one host sync per poll_every timesteps
                poll_wait += time.perf_counter() - t_poll
                if stopped:
                    break
        step_fn.stop_state = None
        steps = int(float(state[2]))
        losses = [float(v) for v in losses_dev[:steps].cpu()]
    for k in range(0 if not on_device else num_steps, num_steps):
        if two_phase:
            l = step_fn.forward_loss(k)
            steps += 1
            if use_dist:
                dist.all_reduce(l, group=group)
            lv = float(l)
            losses.append(lv)
            if lv > loss_max:
                loss_max = lv
            if _f32_lt_prod(lv, loss_max, thr):
                step_fn.discard_pending()
                break
            step_fn.backward_pending()
            continue
        l = step_fn(k)
        steps += 1
        if use_dist and (thr is not None):
            dist.all_reduce(l, group=group)
        if thr is not None:
            lv = float(l)                       # host sync: the reference's `if loss > loss_max` (ddpm_prune.py:104-106)
            losses.append(lv)
            if lv > loss_max:
                loss_max = lv
            if _f32_lt_prod(lv, loss_max, thr):
                break
        else:
            pending.append(l)
    if timings is not None:
        # host time spent ENQUEUEING: the waits inside the polls of the on-device early exit (which drain the queue) are not part
        # of it -- with them the figure is just the step time (bedroom-256: 58.07 of a 58.08 ms step in round 3)
        throttle_wait = float(getattr(step_fn, 'throttle_wait_s', 0.0)) - throttle_wait0     # the host waiting for the device (MAX_STEPS_AHEAD) is not enqueue work
        timings['enqueue_s'] = time.perf_counter() - t_start - poll_wait - throttle_wait
        timings['poll_wait_s'] = poll_wait
        timings['throttle_wait_s'] = throttle_wait
    if hasattr(step_fn, 'finish'):
        step_fn.finish()
    if pending:
        stacked = torch.cat([p.reshape(1) for p in pending])
        if use_dist:
            dist.all_reduce(stacked, group=group)
        losses = [float(v) for v in stacked.cpu()]
    if timings is not None:
        torch.cuda.synchronize()
        timings['sweep_s'] = time.perf_counter() - t_start
    if use_dist and reduce_grads and flat_grads is not None:
        dist.all_reduce(flat_grads, group=group)      # the one exchange step of the sweep (sum of per-shard grads)
        if timings is not None:
            torch.cuda.synchronize()
            timings['allreduce_s'] = time.perf_counter() - t_start - timings['sweep_s']
    return dict(losses=losses, steps=steps, global_batch=B_global)


def prune_model(model, pruning_ratio=0.3, importance=None, ignored_layers=None, channel_groups=None):
    """ddpm_prune.py:79-116: build the pruner, run `pruner.step(interactive=True)` pruning every yielded group,
    then fix the static `channels` attributes.  Returns the pruner (its `.records` hold scores and masks)."""
    from . import pruning
    imp = importance if importance is not None else pruning.TaylorImportance()
    pr = pruning.MagnitudePruner(model, None, importance=imp, iterative_steps=1, channel_groups=channel_groups or {},
                                 ch_sparsity=pruning_ratio,
                                 ignored_layers=ignored_layers if ignored_layers is not None else [model.conv_out])
    for g in pr.step(interactive=True):
        g.prune()
    pruning.fix_static_attributes(model)
    if getattr(model, '_engine', None) is not None:
        model._engine.packs.clear()
    return pr
