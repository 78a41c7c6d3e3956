"""Post-prune finetune step (ddpm_train.py:426-471) on the HIP engine, data-parallel over ranks.

One step = add_noise -> UNet forward -> eps-loss (sum over C,H,W, mean over the batch) -> hand-written backward
-> gradient all-reduce (one flat buffer; RCCL over xGMI) -> global-norm clip (1.0) -> Adam -> EMA (constant decay).
Parameters, gradients, Adam moments and the EMA copy live in flat fp32 buffers so that the optimizer is ONE
HBM-bound kernel launch (csrc/optim.hip) and the all-reduce is ONE collective.
Dropout: the reference finetunes with dropout 0.1 (scripts/finetune_ddpm_cifar10.sh:16 -> utils.set_dropout, utils.py:26-29,
ddpm_train.py:380-382: EVERY nn.Dropout of the model, i.e. ResnetBlock2D.dropout and Attention.to_out[1]).  The masks are
Philox functions of (seed, layer, optimizer step, global element index): fused into the GroupNorm+SiLU kernels, regenerated
in the backward pass, independent of how the batch is sharded over ranks (csrc/dp_common.h).
LR schedule: diffusers/optimization.py:282 `get_scheduler` (LambdaLR multipliers), stepped once per optimizer step
(ddpm_train.py:340-346,464).
"""
import math

import torch

from . import ops
from .sweep import StepThrottle, dist_active


def antithetic_timesteps(bsz, num_train_timesteps, generator=None):
    """ddpm_train.py:446-449 -- generated on the CPU (RNG-stream parity), then moved by the caller."""
    t = torch.randint(low=0, high=num_train_timesteps, size=(bsz // 2 + 1,), generator=generator)
    return torch.cat([t, num_train_timesteps - t - 1], dim=0)[:bsz]


class LambdaLR:
    """torch.optim.lr_scheduler.LambdaLR semantics for ONE parameter group (host scalar arithmetic): the lr in force is
    base_lr * f(last_epoch); construction evaluates f(0), every step() advances last_epoch by one."""

    def __init__(self, base_lr, lr_lambda, last_epoch=-1):
        self.base_lr, self.lr_lambda = float(base_lr), lr_lambda
        self.last_epoch = last_epoch
        self.step()

    def step(self):
        self.last_epoch += 1
        self._last_lr = [self.base_lr * self.lr_lambda(self.last_epoch)]

    def get_last_lr(self):
        return self._last_lr

    def state_dict(self):
        return dict(base_lr=self.base_lr, last_epoch=self.last_epoch)

    def load_state_dict(self, sd):
        self.base_lr, self.last_epoch = sd['base_lr'], sd['last_epoch'] - 1
        self.step()


SCHEDULER_TYPES = ('linear', 'cosine', 'cosine_with_restarts', 'polynomial', 'constant', 'constant_with_warmup',
                   'piecewise_constant')


def get_scheduler(name, base_lr, step_rules=None, num_warmup_steps=None, num_training_steps=None, num_cycles=1, power=1.0,
                  last_epoch=-1, lr_end=1e-7):
    """diffusers/optimization.py:282-354 with the optimizer replaced by its learning rate (one parameter group)."""
    if name not in SCHEDULER_TYPES:
        raise ValueError('%r is not a valid SchedulerType' % (name,))
    if name == 'constant':                                        # optimization.py:40-53
        return LambdaLR(base_lr, lambda _: 1, last_epoch)
    if name == 'piecewise_constant':
        # reference behaviour kept: optimization.py:321 calls get_piecewise_constant_schedule(optimizer, rules=...) whose
        # parameter is named step_rules, so this branch of get_scheduler raises; the schedule itself is reachable directly
        raise TypeError("get_piecewise_constant_schedule() got an unexpected keyword argument 'rules'")
    if num_warmup_steps is None:
        raise ValueError('%s requires `num_warmup_steps`, please provide that argument.' % name)
    W = num_warmup_steps
    if name == 'constant_with_warmup':                            # optimization.py:56-78
        return LambdaLR(base_lr, lambda k: float(k) / float(max(1.0, W)) if k < W else 1.0, last_epoch)
    if num_training_steps is None:
        raise ValueError('%s requires `num_training_steps`, please provide that argument.' % name)
    T = num_training_steps
    if name == 'linear':                                          # optimization.py:123-149
        def f(k):
            if k < W:
                return float(k) / float(max(1, W))
            return max(0.0, float(T - k) / float(max(1, T - W)))
    elif name == 'cosine':                                        # optimization.py:152-183 (num_cycles = 0.5)
        def f(k):
            if k < W:
                return float(k) / float(max(1, W))
            progress = float(k - W) / float(max(1, T - W))
            return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(0.5) * 2.0 * progress)))
    elif name == 'cosine_with_restarts':                          # optimization.py:186-218
        def f(k):
            if k < W:
                return float(k) / float(max(1, W))
            progress = float(k - W) / float(max(1, T - W))
            if progress >= 1.0:
                return 0.0
            return max(0.0, 0.5 * (1.0 + math.cos(math.pi * ((float(num_cycles) * progress) % 1.0))))
    else:                                                         # polynomial, optimization.py:221-268
        lr_init = base_lr
        if not (lr_init > lr_end):
            raise ValueError('lr_end (%s) must be be smaller than initial lr (%s)' % (lr_end, lr_init))

        def f(k):
            if k < W:
                return float(k) / float(max(1, W))
            if k > T:
                return lr_end / lr_init
            pct_remaining = 1 - (k - W) / (T - W)
            return ((lr_init - lr_end) * pct_remaining ** power + lr_end) / lr_init
    return LambdaLR(base_lr, f, last_epoch)


def get_piecewise_constant_schedule(base_lr, step_rules, last_epoch=-1):
    """optimization.py:81-120: step_rules = "1:10,0.1:20,0.01:30,0.005" -> multiplier 1 before step 10, 0.1 before 20, ..."""
    rules, rule_list = {}, step_rules.split(',')
    for rule in rule_list[:-1]:
        value, steps = rule.split(':')
        rules[int(steps)] = float(value)
    last = float(rule_list[-1])

    def piecewise(step):
        for s in sorted(rules):
            if step < s:
                return rules[s]
        return last
    return LambdaLR(base_lr, piecewise, last_epoch)


def set_dropout(model, p):
    """utils.py:26-29."""
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = p


def _require_hip_device(dev):
    if dev.type != 'cuda':
        raise RuntimeError('finetune runs on the MI355X HIP kernels only')


class FinetuneEngine:
    def __init__(self, model, scheduler, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, ema_decay=0.9999, max_grad_norm=1.0,
                 use_ema=True, group=None, dropout=None, lr_scheduler=None, dropout_seed=0, replay=None):
        """dropout: None keeps whatever `set_dropout(model, p)` has set on the nn.Dropout holders; a float sets it.
        lr_scheduler: a `LambdaLR` from `get_scheduler` (its base_lr is the learning rate) or None (constant `lr`).
        replay: True / False / None (automatic: single-process steps on a cuda device, DP_FINETUNE_REPLAY=0 disables) -- the step is
        stream-captured ONCE (second call; the first runs eagerly) and afterwards re-issued from the library's C loop
        (ops.CapturedCall): ~750 launches without Python / ctypes per launch.  What changes from step to step lives on the device:
        inputs and timesteps in static buffers, {lr, Adam bias corrections, optimizer step (= the dropout masks' step)} in a
        4-word buffer written by ONE by-value launch per step (ops.set_step_scalars); the weight re-packing is part of the
        captured step.  Same kernels, arguments and order as the eager step -> the same bits."""
        self.replay = replay
        self._caps = None            # {capture key: captured step} (_step_replayed)
        self._seen = {}              # {batch shape: eager steps run at it}
        self._throttle = StepThrottle()
        if dropout is not None:
            set_dropout(model, float(dropout))
        self.model, self.scheduler = model, scheduler
        self.lr_scheduler = lr_scheduler
        self.dropout_seed = dropout_seed
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
        # gradient buckets of the data-parallel step: contiguous ranges of the flat gradient buffer that become final at the three
        # milestones of the backward pass (output head + up blocks, mid block, down blocks) and at its end (conv_in, time embedding)
        self._buckets = {'up': [], 'mid': [], 'down': [], 'rest': []}
        off = 0
        for name, p in model.named_parameters():
            seg = ('up' if name.startswith(('up_blocks.', 'conv_norm_out.', 'conv_out.')) else
                   'mid' if name.startswith('mid_block.') else 'down' if name.startswith('down_blocks.') else 'rest')
            rs = self._buckets[seg]
            if rs and rs[-1][1] == off:
                rs[-1][1] = off + p.numel()
            else:
                rs.append([off, off + p.numel()])
            off += p.numel()
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

    REPLAY_OVERLAP = None        # weight-gradient side stream inside the captured step: None = the engine's own rule (by step size)
    MAX_CAPTURES = 2             # captured steps kept alive at once (full batch + the partial last batch of an epoch)

    @property
    def _cap(self):
        """The most recently built captured step, or None."""
        return next(reversed(self._caps.values())) if self._caps else None

    def _replay_wanted(self, use_dist, dev):
        import os
        if use_dist or dev.type != 'cuda' or getattr(ops, 'IS_MOCK', False) or not hasattr(ops, 'CapturedCall'):
            return False                                 # the data-parallel step interleaves collectives with the backward pass
        if self.replay is None:
            return os.environ.get('DP_FINETUNE_REPLAY', '1') != '0'
        return bool(self.replay)

    def _step_replayed(self, clean, noise, timesteps, gb, image_offset):
        """One optimizer step through the captured step (built on first use for this batch shape)."""
        import os
        model, dev = self.model, self.flat_p.device
        table = getattr(model, 'dropout_table', dict)()
        # everything the captured launches carry as immediate arguments: a change of any of them builds a new capture
        key = (tuple(clean.shape), int(image_offset), int(gb), tuple(sorted(table.items())), int(self.dropout_seed),
               self.max_grad_norm, self.eps, self.ema_decay, tuple(self.betas))
        if self._caps is None:
            self._caps = {}
        cap = self._caps.get(key)
        if cap is None:
            # One capture per key, at most MAX_CAPTURES alive (the partial last batch of an epoch is a second shape: with
            # drop_last=False it alternates with the full one, and re-capturing at every epoch boundary would hold the old pool, the
            # new pool and the eager step's cache at once).  Each capture's private pool pins every activation of a training
            # step, so the oldest goes -- and its pool is returned to the device -- BEFORE the new one is built.
            while len(self._caps) >= self.MAX_CAPTURES:
                old_key = next(iter(self._caps))
                torch.cuda.synchronize(dev)               # nothing of the old capture may still be in flight when its pool goes
                del self._caps[old_key]
                import gc
                gc.collect()
                torch.cuda.empty_cache()
            hyper = torch.zeros(4, dtype=torch.float32, device=dev)
            st = dict(key=key, hyper=hyper, clean=ops.empty_act(tuple(clean.shape), dev), noise=ops.empty_act(tuple(noise.shape), dev),
                      t=torch.zeros(clean.shape[0], dtype=torch.long, device=dev))
            eng = model.engine()
            P = {n: p.detach() for n, p in model.named_parameters()}
            G = {n: p.grad for n, p in model.named_parameters()}
            ov = os.environ.get('DP_FINETUNE_REPLAY_OVERLAP')
            overlap = self.REPLAY_OVERLAP if ov is None else (ov != '0')

            def body():
                eng.bind(P, G)
                eng.set_dropout(table, self.dropout_seed, 0, image_offset, step_dev=hyper.data_ptr() + 12)
                if overlap is not None:
                    eng.overlap_wgrad = overlap
                eng.prepare_packs()                       # the optimizer update of the previous replay invalidated every operand
                noisy = ops.add_noise(st['clean'], st['noise'], self.acp, st['t'])
                self.flat_g.zero_()
                out = eng.forward(noisy, st['t'], save=True)
                loss, dout = ops.mse_fwd_bwd(out, st['noise'], 2.0 / gb, 1.0 / gb)
                eng.backward(dout)
                nc = ops.clip_coef(ops.sumsq_partials(self.flat_g), self.max_grad_norm)
                ops.adam_ema_dev(self.flat_p, self.flat_g, self.m, self.v, self.ema, nc[1:2], hyper, self.betas[0], self.betas[1],
                                 self.eps, self.ema_decay)
                eng.packs.clear()                         # host bookkeeping: nothing packed here outlives the step
                return loss, nc
            saved_overlap = eng.overlap_wgrad
            try:
                st['call'] = ops.CapturedCall(body, side_stream=eng.replay_side_stream(dev))
            finally:
                eng.overlap_wgrad = saved_overlap
                eng.set_dropout(None)
            self._caps[key] = cap = st
        if clean.data_ptr() != cap['clean'].data_ptr():
            cap['clean'].copy_(clean)
        if noise.data_ptr() != cap['noise'].data_ptr():
            cap['noise'].copy_(noise)
        cap['t'].copy_(timesteps)
        self.step_count += 1
        lr = self.lr_scheduler.get_last_lr()[0] if self.lr_scheduler is not None else self.lr
        self.last_lr = lr
        ops.set_step_scalars(cap['hyper'], lr, self.betas[0], self.betas[1], self.step_count)
        loss, nc = cap['call'].launch()
        self.last_grad_norm = nc[0:1].clone()              # (the captured tensors are overwritten by the next step)
        if self.lr_scheduler is not None:
            self.lr_scheduler.step()                       # ddpm_train.py:464
        eng = getattr(model, '_engine', None)
        if eng is not None:
            eng.packs.clear()
        loss = loss.clone()                                # the captured tensor is overwritten by the next step
        self._throttle.mark()
        return loss

    def step(self, clean, noise, timesteps, global_batch=None, image_offset=None):
        """Returns the (local share of the) loss as a [1] device tensor; no host synchronisation.
        image_offset: global index of this rank's first image (default rank * B): the dropout masks are functions of the
        GLOBAL element index, so a sharded step draws the masks of the un-sharded one."""
        import torch.distributed as dist
        use_dist = dist_active(self.group)
        B = clean.shape[0]
        gb = global_batch if global_batch is not None else (B * dist.get_world_size(self.group) if use_dist else B)
        if image_offset is None:
            image_offset = dist.get_rank(self.group) * B if use_dist else 0
        model = self.model
        model.train()                                     # ddpm_train.py:430
        dev = self.flat_p.device
        # a batch shape runs eagerly the first time it is seen (lazy operands, streams, code objects: CapturedCall's precondition)
        shape_key = tuple(clean.shape)
        seen = self._seen.get(shape_key, 0)
        self._seen[shape_key] = seen + 1
        if seen >= 1 and self.step_count >= 1 and self._replay_wanted(use_dist, dev):
            return self._step_replayed(clean.to(dev, torch.float32).contiguous(), noise.to(dev, torch.float32).contiguous(),
                                       timesteps.to(device=dev, dtype=torch.long).contiguous(), gb, image_offset)
        clean = clean.to(dev, torch.float32).contiguous()
        noise = noise.to(dev, torch.float32).contiguous()
        eng = model.engine()
        eng.bind({n: p.detach() for n, p in model.named_parameters()}, {n: p.grad for n, p in model.named_parameters()})
        eng.set_dropout(getattr(model, 'dropout_table', dict)(), self.dropout_seed, self.step_count + 1, image_offset)
        if hasattr(ops, 'pack_weight_batch'):
            eng.prepare_packs()          # the last optimizer step invalidated every packed operand: re-pack in a few launches
        # the batched time-embedding backward finalises the time_emb_proj gradients only at the END of the backward pass: it is
        # switched off when gradient buckets are all-reduced at the segment milestones
        eng.temb_batch = eng.temb_batch and not use_dist
        t = timesteps.to(device=dev, dtype=torch.long).contiguous()
        noisy = ops.add_noise(clean, noise, self.acp, t)
        self.flat_g.zero_()                               # optimizer.zero_grad()
        out = eng.forward(noisy, t, save=True)
        loss, dout = ops.mse_fwd_bwd(out, noise, 2.0 / gb, 1.0 / gb)
        pending = []
        if use_dist:
            # bucketed all-reduce overlapped with the backward pass: each bucket's collective (RCCL over xGMI) is enqueued as
            # soon as its gradients are final and runs while the remaining layers' MFMA kernels execute
            def reduce_segment(seg):
                for lo, hi in self._buckets[seg]:
                    pending.append(dist.all_reduce(self.flat_g[lo:hi], group=self.group, async_op=True))
            eng.segment_hook = reduce_segment
        try:
            eng.backward(dout)
        finally:
            eng.segment_hook = None
        if use_dist:
            reduce_segment('rest')
            for w in pending:                                   # stream-ordered wait for RCCL (blocks the host for gloo)
                w.wait()
        partial = ops.sumsq_partials(self.flat_g)
        nc = ops.clip_coef(partial, self.max_grad_norm)
        self.last_grad_norm = nc[0:1]
        self.step_count += 1
        lr = self.lr_scheduler.get_last_lr()[0] if self.lr_scheduler is not None else self.lr
        self.last_lr = lr
        ops.adam_ema(self.flat_p, self.flat_g, self.m, self.v, self.ema, nc[1:2], lr, self.betas[0], self.betas[1],
                     self.eps, self.step_count, self.ema_decay)
        if self.lr_scheduler is not None:
            self.lr_scheduler.step()                       # ddpm_train.py:464
        eng.packs.clear()                                  # weights changed: packed operands are stale
        self._throttle.mark()                              # no host read-back in a step: bound how far the host runs ahead
        return loss
