"""ORACLE (test infrastructure only) -- CPU restatement of the reference's diffusion glue.

Parity pinned against tests/golden/schedule.npz, ddim.npz, tiny_unet.npz, cifar_c1.json (outputs of
the reference itself, see tests/golden/make_golden.py).  Never imported by the product path.

Reference lines followed (relative to /root/reference):
  diffusers/schedulers/scheduling_ddpm.py:141-158     beta / alpha-bar tables (linear schedule, fp32)
  diffusers/schedulers/scheduling_ddpm.py:408-429     add_noise
  diffusers/schedulers/scheduling_ddim.py:239-268     set_timesteps (modified: skip_type uniform|quad)
  diffusers/schedulers/scheduling_ddim.py:194-202,324-370  _get_variance / step (eta, clip_sample)
  diffusers/pipelines/ddim/pipeline_ddim.py:98-117    sampling loop + image post-processing
  ddpm_prune.py:94-106                                Taylor / Diff-Pruning gradient sweep
  ddpm_train.py:446-459                               finetune loss (antithetic t, sum-CHW mean-B)
"""
import numpy as np
import torch
import torch.nn.functional as F

from .unet_ref import unet_forward


def alphas_cumprod(num_train_timesteps=1000, beta_start=1e-4, beta_end=0.02):
    betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
    return torch.cumprod(1.0 - betas, dim=0)


def add_noise(acp, x0, noise, timesteps):
    a = acp.to(device=x0.device, dtype=x0.dtype)[timesteps]
    sa = (a ** 0.5).flatten()
    sb = ((1 - a) ** 0.5).flatten()
    while sa.dim() < x0.dim():
        sa = sa.unsqueeze(-1)
        sb = sb.unsqueeze(-1)
    return sa * x0 + sb * noise


def ddim_timesteps(num_inference_steps, num_train_timesteps=1000, skip_type='uniform', steps_offset=0):
    n, T = num_inference_steps, num_train_timesteps
    if skip_type == 'uniform':
        ratio = (T - 1) / (n - 1)
        ts = (np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64)
    elif skip_type == 'quad':
        ratio = (T - 1) / (n - 1) ** 2
        ts = (np.arange(0, n) ** 2 * ratio).round()[::-1].copy().astype(np.int64)
    else:
        raise NotImplementedError(skip_type)
    return torch.from_numpy(ts) + steps_offset


def ddim_step(acp, model_output, t, sample, num_inference_steps, num_train_timesteps=1000, eta=0.0,
              clip_sample=True, variance_noise=None):
    """scheduling_ddim.py:324-370 (epsilon prediction, set_alpha_to_one=True)."""
    t = int(t)
    prev_t = t - num_train_timesteps // num_inference_steps      # NB: not the next visited timestep (quirk)
    a_t = acp[t]
    a_prev = acp[prev_t] if prev_t >= 0 else torch.tensor(1.0)
    b_t = 1 - a_t
    x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
    if clip_sample:
        x0 = x0.clamp(-1.0, 1.0)
    var = ((1 - a_prev) / (1 - a_t)) * (1 - a_t / a_prev)
    std = eta * var ** 0.5
    direction = (1 - a_prev - std ** 2) ** 0.5 * model_output
    prev = a_prev ** 0.5 * x0 + direction
    if eta > 0:
        prev = prev + std * variance_noise
    return prev


def ddpm_timesteps(num_inference_steps, num_train_timesteps=1000):
    """scheduling_ddpm.py:231-236 (equal spacing, integer step ratio)."""
    ratio = num_train_timesteps // num_inference_steps
    return torch.from_numpy((np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64))


def ddpm_step(acp, model_output, t, sample, num_inference_steps=None, num_train_timesteps=1000, variance_noise=None,
              clip_sample=True, clip_sample_range=1.0, variance_type='fixed_small'):
    """scheduling_ddpm.py:312-406 (epsilon prediction; variance types fixed_small / fixed_large, _get_variance :238-280)."""
    t = int(t)
    n_inf = num_inference_steps if num_inference_steps else num_train_timesteps
    prev_t = t - num_train_timesteps // n_inf
    one = torch.tensor(1.0)
    a_t = acp[t]
    a_prev = acp[prev_t] if prev_t >= 0 else one
    b_t = 1 - a_t
    b_prev = 1 - a_prev
    cur_a = a_t / a_prev
    cur_b = 1 - cur_a
    x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
    if clip_sample:
        x0 = x0.clamp(-clip_sample_range, clip_sample_range)
    c_x0 = (a_prev ** 0.5 * cur_b) / b_t
    c_xt = cur_a ** 0.5 * b_prev / b_t
    prev = c_x0 * x0 + c_xt * sample
    if t > 0:
        var = torch.clamp((1 - a_prev) / (1 - a_t) * cur_b, min=1e-20)
        if variance_type == 'fixed_large':
            var = cur_b
        elif variance_type != 'fixed_small':
            raise NotImplementedError(variance_type)
        prev = prev + (var ** 0.5) * variance_noise
    return prev


@torch.no_grad()
def ddim_sample(P, cfg, x_T, num_inference_steps, skip_type='uniform', eta=0.0, first_n=None):
    acp = alphas_cumprod()
    ts = ddim_timesteps(num_inference_steps, skip_type=skip_type)
    x = x_T
    trace = []
    for i, t in enumerate(ts):
        if first_n is not None and i >= first_n:
            break
        eps = unet_forward(P, cfg, x, t)
        x = ddim_step(acp, eps, t, x, num_inference_steps, eta=eta)
        trace.append(x)
    return x, trace


def to_image(x):
    """pipeline_ddim.py:114-115."""
    return (x / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1)


def taylor_sweep(P, cfg, clean, noise, steps, thr=None, loss_kind='mse', on_step=None, accumulate_breaking_step=True):
    """ddpm_prune.py:94-106.  P: dict of leaf tensors with requires_grad; grads accumulate into .grad.

    thr=None -> plain Taylor (all `steps`); thr=x -> Diff-Pruning early exit; the breaking step IS
    accumulated (backward happens before the threshold test, ddpm_prune.py:102-106).
    accumulate_breaking_step=False -> the ddpm_exp flavour (ddpm_exp/prune.py:249-256): threshold test first, the
    breaking step is NOT accumulated (pinned by tests/golden/ddpm_original.json 'sweep').
    Returns the list of per-step losses (python floats)."""
    acp = alphas_cumprod()
    for p in P.values():
        p.grad = None
    B = clean.shape[0]
    losses = []
    loss_max = 0.0
    for k in range(steps):
        t = torch.full((B,), k, dtype=torch.long)
        noisy = add_noise(acp, clean, noise, t)
        out = unet_forward(P, cfg, noisy, t)
        if loss_kind == 'mse':
            loss = F.mse_loss(out, noise)
        else:                                   # ddpm_exp/functions/losses.py:15, ddpm_train.py:459
            loss = (noise - out).square().sum(dim=(1, 2, 3)).mean(dim=0)
        lv = float(loss.detach())
        if thr is not None and not accumulate_breaking_step:
            losses.append(lv)
            if lv > loss_max:
                loss_max = lv
            if lv < loss_max * thr:
                break
            loss.backward()
            continue
        loss.backward()
        losses.append(lv)
        if on_step is not None:
            on_step(k, lv)
        if thr is not None:
            if lv > loss_max:
                loss_max = lv
            if lv < loss_max * thr:
                break
    return losses


def finetune_loss(P, cfg, clean, noise, t, drop=None):
    """ddpm_train.py:453-459.  drop: philox_ref.DropSpec when the model trains with dropout (ddpm_train.py:380-382)."""
    acp = alphas_cumprod()
    out = unet_forward(P, cfg, add_noise(acp, clean, noise, t), t, drop)
    return (noise - out).square().sum(dim=(1, 2, 3)).mean(dim=0)


def adam_ema_step(params, grads, m, v, ema, step, lr=2e-4, b1=0.9, b2=0.999, eps=1e-8, ema_decay=0.9999,
                  max_norm=1.0):
    """ddpm_train.py:462-469: clip_grad_norm_(1.0) -> Adam (torch.optim.Adam defaults of :331-337, wd 0) ->
    EMAModel.step with constant decay (training_utils.py:201,215-216).  Lists of tensors, updated in place.
    Pinned by tests/golden/optim.json (three steps of the reference's own clip + torch Adam + vendored EMAModel).
    Returns the pre-clip global gradient norm."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    for p, g, mi, vi, e in zip(params, grads, m, v, ema):
        g = g * coef
        mi.mul_(b1).add_(g, alpha=1 - b1)
        vi.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (vi.sqrt() / (bc2 ** 0.5)).add_(eps)
        p.addcdiv_(mi, denom, value=-lr / bc1)
        e.copy_((1 - ema_decay) * p + ema_decay * e)
    return float(total)
