"""Post-prune finetune step (ddpm_train.py:426-471) on the HIP engine, data-parallel over ranks.

One step = add_noise -> UNet forward -> eps-loss (sum over C,H,W, mean over the batch) -> hand-written backward
-> gradient all-reduce (one flat buffer; RCCL over xGMI) -> global-norm clip (1.0) -> Adam -> EMA (constant decay).
Parameters, gradients, Adam moments and the EMA copy live in flat fp32 buffers so that the optimizer is ONE
HBM-bound kernel launch (csrc/optim.hip) and the all-reduce is ONE collective.
Dropout: the reference finetunes with dropout 0.1 (scripts/finetune_ddpm_cifar10.sh); this engine implements
p = 0 only and raises otherwise (RNG-stream parity of dropout masks is not reproducible across backends anyway).
"""
import torch

from . import ops


def antithetic_timesteps(bsz, num_train_timesteps, generator=None):
    """ddpm_train.py:446-449 -- generated on the CPU (RNG-stream parity), then moved by the caller."""
    t = torch.randint(low=0, high=num_train_timesteps, size=(bsz // 2 + 1,), generator=generator)
    return torch.cat([t, num_train_timesteps - t - 1], dim=0)[:bsz]


def _require_hip_device(dev):
    if dev.type != 'cuda':
        raise RuntimeError('finetune runs on the MI355X HIP kernels only')


class FinetuneEngine:
    def __init__(self, model, scheduler, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, ema_decay=0.9999, max_grad_norm=1.0,
                 use_ema=True, group=None, dropout=0.0):
        if dropout != 0.0:
            raise NotImplementedError('dropout > 0 is not implemented in the HIP engine')
        self.model, self.scheduler = model, scheduler
        self.lr, self.betas, self.eps = lr, betas, eps
        self.ema_decay, self.max_grad_norm, self.group = ema_decay, max_grad_norm, group
        params = list(model.parameters())
        dev = params[0].device
        _require_hip_device(dev)
        total = sum(p.numel() for p in params)
        self.flat_p = torch.empty(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        for p in params:                     # re-home parameters into the flat buffer (views keep nn.Module semantics)
            n = p.numel()
            self.flat_p[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat_p[off:off + n].view_as(p)
            p.grad = self.flat_g[off:off + n].view_as(p)
            off += n
        self.m = torch.zeros_like(self.flat_p)
        self.v = torch.zeros_like(self.flat_p)
        self.ema = self.flat_p.clone() if use_ema else None
        self.step_count = 0
        self.acp = scheduler._acp_on(dev)
        self.last_grad_norm = None

    def ema_state(self):
        """EMA parameters as a {name: tensor} dict (views of the flat EMA buffer)."""
        out, off = {}, 0
        for n, p in self.model.named_parameters():
            out[n] = self.ema[off:off + p.numel()].view_as(p)
            off += p.numel()
        return out

    # ---- EMAModel.store / copy_to / restore (training_utils.py:220-262) as used around checkpoints and evaluation
    #      (ddpm_train.py:387-401,489-514): swap the EMA weights into the live model and back
    def _weights_changed(self):
        eng = getattr(self.model, '_engine', None)
        if eng is not None:
            eng.packs.clear()                              # packed operands are stale

    def ema_store(self):
        self._stash = self.flat_p.clone()

    def ema_copy_to(self):
        if self.ema is None:
            raise RuntimeError('FinetuneEngine was built with use_ema=False')
        self.flat_p.copy_(self.ema)
        self._weights_changed()

    def ema_restore(self):
        if getattr(self, '_stash', None) is None:
            raise RuntimeError('This ExponentialMovingAverage has no `store()`ed weights to `restore()`')
        self.flat_p.copy_(self._stash)
        self._stash = None
        self._weights_changed()

    def step(self, clean, noise, timesteps, global_batch=None):
        """Returns the (local share of the) loss as a [1] device tensor; no host synchronisation."""
        import torch.distributed as dist
        use_dist = dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1
        B = clean.shape[0]
        gb = global_batch if global_batch is not None else (B * dist.get_world_size(self.group) if use_dist else B)
        model = self.model
        eng = model.engine()
        eng.bind({n: p.detach() for n, p in model.named_parameters()}, {n: p.grad for n, p in model.named_parameters()})
        t = timesteps.to(device=clean.device, dtype=torch.long)
        noisy = ops.add_noise(clean.contiguous(), noise.contiguous(), self.acp, t)
        self.flat_g.zero_()                               # optimizer.zero_grad()
        out = eng.forward(noisy, t, save=True)
        loss, dout = ops.mse_fwd_bwd(out, noise.contiguous(), 2.0 / gb, 1.0 / gb)
        eng.backward(dout)
        if use_dist:
            dist.all_reduce(self.flat_g, group=self.group)      # sum of per-shard gradients of the global-mean loss
        partial = ops.sumsq_partials(self.flat_g)
        nc = ops.clip_coef(partial, self.max_grad_norm)
        self.last_grad_norm = nc[0:1]
        self.step_count += 1
        ops.adam_ema(self.flat_p, self.flat_g, self.m, self.v, self.ema, nc[1:2], self.lr, self.betas[0], self.betas[1],
                     self.eps, self.step_count, self.ema_decay)
        eng.packs.clear()                                  # weights changed: packed operands are stale
        return loss
