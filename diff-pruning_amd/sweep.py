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
import torch

from . import ops


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
    micro = None

    def __init__(self, model, scheduler, clean, noise, global_numel, loss_kind='mse', global_batch=None):
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

    def _step(self, t):
        if self.micro is None or self.B <= self.micro:
            return self._micro_step(self.clean, self.noise, t)
        total = None
        for lo in range(0, self.B, self.micro):
            hi = min(lo + self.micro, self.B)
            l = self._micro_step(self.clean[lo:hi], self.noise[lo:hi], t[lo:hi])
            total = l if total is None else ops.axpby(l, 1.0, total, 1.0)
        return total

    def _micro_step(self, clean, noise, t):
        # any model(...) / model.engine() call between two sweep steps (a no-grad evaluation, the autograd bridge) re-binds
        # the engine without -- or with temporary -- gradient buffers: bind ours again
        self.eng.bind(self._P, self._G)
        noisy = ops.add_noise(clean, noise, self.acp, t)
        out = self.eng.forward(noisy, t, save=True)
        loss, dout = ops.mse_fwd_bwd(out, noise, self.gscale, self.lscale)
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
        loss, self._dout = ops.mse_fwd_bwd(out, self.noise, self.gscale, self.lscale)
        return loss

    def backward_pending(self):
        self.eng.backward(self._dout)
        self._dout = None

    def discard_pending(self):
        self.eng.ctx = None
        self._dout = None

    def capture(self):
        """Record one timestep (~900 kernel launches) into a hipGraph; afterwards every step is: write t, replay.
        Removes the ~60 ms of Python/ctypes launch overhead per step -- the launch-bound regime at small batch."""
        self.eng.prepare_packs()
        ops._workspace(1 << 26, self.clean.device)          # split-K workspace must exist before capture
        self._t = torch.zeros(self.B, dtype=torch.long, device=self.clean.device)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._loss = self._step(self._t)
        self._graph = g
        return self

    def __call__(self, k):
        if self._graph is not None:
            self._t.fill_(int(k))
            self._graph.replay()
            return self._loss.clone()
        t = torch.full((self.B,), int(k), dtype=torch.long, device=self.clean.device)
        return self._step(t)          # [1] device tensor: this rank's share of L_t


def taylor_sweep(model, scheduler, clean_images, noise, num_steps=1000, thr=None, loss_kind='mse', group=None,
                 step_fn=None, flat_grads=None, reduce_grads=True, use_graph=False, micro_batch=None,
                 accumulate_breaking_step=True):
    """Runs the sweep; returns dict(losses=[python floats of the GLOBAL loss per executed step], steps=int).

    thr=None: plain Taylor.  thr=x: Diff-Pruning early exit.  accumulate_breaking_step=True is ddpm_prune.py:102-106
    (backward, then the threshold test); False is the ddpm_exp flavour (ddpm_exp/prune.py:249-256: test, break, else
    backward), to be combined with loss_kind='sum' (functions/losses.py:15).

    clean_images / noise: this rank's shard.  `group`: torch.distributed process group (None = default group if
    torch.distributed is initialised, single process otherwise).  `step_fn(k) -> local loss tensor` lets the
    multi-process CPU tests drive the same control flow with a different per-step engine."""
    import torch.distributed as dist
    use_dist = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
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
    if step_fn is None:
        step_fn = HipSweepStep(model, scheduler, clean_images, noise, B_global * per_img, loss_kind, B_global)
        step_fn.micro = micro_batch
        if use_graph:
            step_fn.capture()
    losses = []
    pending = []
    loss_max = 0.0
    steps = 0
    two_phase = thr is not None and not accumulate_breaking_step
    if two_phase and use_graph:
        raise ValueError('use_graph replays forward + backward as one unit: not available with accumulate_breaking_step=False')
    for k in range(num_steps):
        if two_phase:
            l = step_fn.forward_loss(k)
            steps += 1
            if use_dist:
                dist.all_reduce(l, group=group)
            lv = float(l)
            losses.append(lv)
            if lv > loss_max:
                loss_max = lv
            if lv < loss_max * thr:
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
            if lv < loss_max * thr:
                break
        else:
            pending.append(l)
    if pending:
        stacked = torch.cat([p.reshape(1) for p in pending])
        if use_dist:
            dist.all_reduce(stacked, group=group)
        losses = [float(v) for v in stacked.cpu()]
    if use_dist and reduce_grads and flat_grads is not None:
        dist.all_reduce(flat_grads, group=group)      # the one exchange step of the sweep (sum of per-shard grads)
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
