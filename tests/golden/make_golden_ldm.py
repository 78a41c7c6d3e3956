#!/usr/bin/env python3
"""Golden vectors for the LDM (CompVis) UNet -- build container only; imports the reference's own UNetModel
(ldm_exp/ldm/modules/diffusionmodules/openaimodel.py) through the omegaconf stub of SURVEY.md App. E.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_ldm.py

Writes tests/golden/ldm_unet.npz (reduced-width config: forward output, loss, selected full gradients) and
ldm_unet_stats.json (per-parameter gradient statistics, parameter count of the full cin256-v2 config)."""
import json, os, sys, types
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import numpy as np
import torch
import golden_common as gc

stub = types.ModuleType('omegaconf.listconfig'); stub.ListConfig = type('ListConfig', (list,), {})
oc = types.ModuleType('omegaconf'); oc.listconfig = stub
sys.modules.setdefault('omegaconf', oc); sys.modules.setdefault('omegaconf.listconfig', stub)
sys.path.insert(0, '/root/reference/ldm_exp')
from ldm.modules.diffusionmodules.openaimodel import UNetModel      # noqa: E402 (reference)

torch.set_num_threads(8)


def run(cfg, seed, B, tag):
    m = UNetModel(**cfg).eval()
    gc.det_init_(m, seed)
    H = cfg['image_size']
    x = torch.from_numpy(gc.det_noise((B, cfg['in_channels'], H, H), 31))
    ctx = torch.from_numpy(gc.det_noise((B, 1, cfg['context_dim']), 32))
    noise = torch.from_numpy(gc.det_noise((B, cfg['in_channels'], H, H), 33))
    t = torch.tensor([7, 640][:B])
    y = m(x, t, context=ctx)
    loss = (y - noise).square().mean(dim=(1, 2, 3)).mean()
    loss.backward()
    P = dict(m.named_parameters())
    stats = {n: [float(p.grad.double().sum()), float(p.grad.double().abs().sum())] for n, p in P.items()}
    full = {}
    for n in ['input_blocks.0.0.weight', 'input_blocks.4.1.transformer_blocks.0.attn1.to_q.weight',
              'input_blocks.4.1.transformer_blocks.0.attn2.to_v.weight', 'input_blocks.4.1.transformer_blocks.0.ff.net.0.proj.weight',
              'input_blocks.4.1.transformer_blocks.0.norm2.weight', 'input_blocks.4.1.proj_out.weight', 'input_blocks.3.0.op.weight',
              'middle_block.1.transformer_blocks.0.ff.net.2.weight', 'output_blocks.2.1.conv.weight',
              'output_blocks.5.0.skip_connection.weight', 'time_embed.0.weight', 'out.2.bias']:
        if n in P:
            full['grad::' + n] = P[n].grad.numpy().copy()
    np.savez(os.path.join(HERE, 'ldm_unet%s.npz' % tag), fwd_out=y.detach().numpy(), loss=np.array(float(loss)), **full)
    return stats, {n: list(p.shape) for n, p in P.items()}


if __name__ == '__main__' and len(sys.argv) == 1:
    stats, shapes = run(gc.LDM_TINY_CFG, 9, 2, '')
    full = UNetModel(**gc.LDM_CIN256_CFG)
    json.dump(dict(grad_stats=stats, shapes=shapes, cin256_params=sum(p.numel() for p in full.parameters()),
                   cin256_shapes={n: list(p.shape) for n, p in full.named_parameters()}),
              open(os.path.join(HERE, 'ldm_unet_stats.json'), 'w'))
    print('ok', len(stats))


LDM_VARIANTS = {      # other members of the configuration family (same block kinds, other depths / widths / attention levels)
    'small_2lvl': dict(gc.LDM_TINY_CFG, image_size=8, model_channels=32, channel_mult=[1, 2], attention_resolutions=[1, 2],
                       num_res_blocks=1),
    'deep_3lvl': dict(gc.LDM_TINY_CFG, image_size=16, model_channels=32, channel_mult=[1, 1, 3], attention_resolutions=[4],
                      num_res_blocks=3),
}


def ldm_groups(which=None, out_name='ldm_groups.json'):
    """Group table of the LDM UNet under the reference's vendored torch_pruning (prune_ldm.py:70-100 setup)."""
    sys.path.insert(0, '/root/reference/ddpm_exp')
    os.makedirs('/tmp/golden_scratch', exist_ok=True)
    os.chdir('/tmp/golden_scratch')
    import torch_pruning as tp
    out = {}
    for tag, cfg in (which or (('tiny', gc.LDM_TINY_CFG),)):
        m = UNetModel(**cfg).eval()
        gc.det_init_(m, 9)
        H = cfg['image_size']
        ex = {'x': torch.randn(2, 3, H, H), 'timesteps': torch.full((2,), 1, dtype=torch.long),
              'context': torch.randn(2, 1, cfg['context_dim'])}
        pr = tp.pruner.MagnitudePruner(m, ex, importance=tp.importance.TaylorImportance(), iterative_steps=1,
                                       channel_groups={}, ch_sparsity=0.3, ignored_layers=[m.out], round_to=2)
        names = {mod: n for n, mod in m.named_modules()}
        f = tp.function
        table = []
        for g in pr.DG.get_all_groups(ignored_layers=pr.ignored_layers, root_module_types=pr.root_module_types):
            mem = []
            for dep, idxs in g:
                mod = dep.target.module
                if mod not in names:
                    continue
                h = dep.handler
                kind = ('out' if h in (f.prune_conv_out_channels, f.prune_linear_out_channels) else
                        'in' if h in (f.prune_conv_in_channels, f.prune_linear_in_channels) else
                        'gn' if h == f.prune_groupnorm_out_channels else
                        'ln' if h == f.prune_layernorm_out_channels else 'other')
                rng = []
                for i in sorted(idxs):
                    if rng and rng[-1][1] == i:
                        rng[-1][1] = i + 1
                    else:
                        rng.append([i, i + 1])
                mem.append([names[mod], kind, rng])
            table.append(dict(ch_groups=int(pr.get_channel_groups(g)), members=mem))
        out[tag] = table
        print(tag, 'groups', len(table))
    json.dump(out, open(os.path.join(HERE, out_name), 'w'))


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'groups':
    ldm_groups()
if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'groups_more':
    ldm_groups(tuple(LDM_VARIANTS.items()), 'ldm_groups_more.json')


def ldm_prune():
    """One backward of the golden loss, then the reference's interactive prune (vendored TaylorImportance, ratio 0.3,
    round_to=2, head groups for to_q/k/v as in prune_ldm.py:75-81) recording every group's scores and mask."""
    sys.path.insert(0, '/root/reference/ddpm_exp')
    os.makedirs('/tmp/golden_scratch', exist_ok=True)
    os.chdir('/tmp/golden_scratch')
    import torch_pruning as tp
    from ldm.modules.attention import CrossAttention
    cfg = gc.LDM_TINY_CFG
    m = UNetModel(**cfg).eval()
    gc.det_init_(m, 9)
    H = cfg['image_size']
    x = torch.from_numpy(gc.det_noise((2, 3, H, H), 31))
    ctx = torch.from_numpy(gc.det_noise((2, 1, cfg['context_dim']), 32))
    noise = torch.from_numpy(gc.det_noise((2, 3, H, H), 33))
    t = torch.tensor([7, 640])
    ex = {'x': torch.randn(2, 3, H, H), 'timesteps': torch.full((2,), 1, dtype=torch.long), 'context': torch.randn(2, 1, cfg['context_dim'])}
    channel_groups = {}
    for mod in m.modules():
        if isinstance(mod, CrossAttention):
            channel_groups[mod.to_q] = mod.heads
            channel_groups[mod.to_k] = mod.heads
            channel_groups[mod.to_v] = mod.heads
    pr = tp.pruner.MagnitudePruner(m, ex, importance=tp.importance.TaylorImportance(), iterative_steps=1,
                                   channel_groups=channel_groups, ch_sparsity=0.3, ignored_layers=[m.out], round_to=2)
    m.zero_grad()
    loss = (m(x, t, context=ctx) - noise).square().mean(dim=(1, 2, 3)).mean()
    loss.backward()
    pr.current_step += 1
    names = {mod: n for n, mod in m.named_modules()}
    rec = []
    for group in pr.DG.get_all_groups(ignored_layers=pr.ignored_layers, root_module_types=pr.root_module_types):
        if not pr._check_sparsity(group):
            continue
        module, fn = group[0][0].target.module, group[0][0].handler
        ch_groups = pr.get_channel_groups(group)
        imp = pr.estimate_importance(group, ch_groups=ch_groups)
        if imp is None:
            continue
        cur = pr.DG.get_out_channels(module)
        n_pruned = cur - int(pr.layer_init_out_ch[module] * (1 - pr.get_target_sparsity(module)))
        if pr.round_to:
            n_pruned = n_pruned - (n_pruned % pr.round_to)
        if n_pruned <= 0:
            continue
        if ch_groups > 1:
            gs, per = cur // ch_groups, n_pruned // ch_groups
            idxs = torch.cat([torch.argsort(imp[c * gs:(c + 1) * gs])[:per] + c * gs for c in range(ch_groups)], 0)
        else:
            idxs = torch.argsort(imp)[:(n_pruned // ch_groups)]
        g2 = pr.DG.get_pruning_group(module, fn, idxs.tolist())
        ok = pr.DG.check_pruning_group(g2)
        rec.append(dict(root=names[module], ch_groups=int(ch_groups), cur=int(cur), n_pruned=int(n_pruned),
                        pruned=sorted(int(i) for i in idxs.tolist()), score=gc.f32_to_b64(imp.detach().float().numpy()), ok=bool(ok)))
        if ok:
            g2.prune()
    with torch.no_grad():
        y2 = m(x, t, context=ctx)
    json.dump(dict(prune=rec, shapes_after={n: list(p.shape) for n, p in m.named_parameters()},
                   params_after=sum(p.numel() for p in m.parameters()), fwd_after=gc.f32_to_b64(y2.numpy())),
              open(os.path.join(HERE, 'ldm_prune.json'), 'w'))
    print('ldm prune groups', len(rec), 'params after', sum(p.numel() for p in m.parameters()))


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'prune':
    ldm_prune()


def ldm_sampler():
    """Pins the CFG DDIM sampler of the LDM importance pass (ldm_exp/prune_ldm.py:111-118 -> DDIMSampler.sample,
    ldm/models/diffusion/ddim.py:57-203) and the latent-diffusion noise schedule (ldm/modules/diffusionmodules/util.py
    make_beta_schedule 'linear' with the cin256-v2 linear_start / linear_end): the reference's own DDIMSampler driven over the
    reference's own UNetModel (tiny config, seeded weights), 20 steps, guidance 3.0, eta 0, fixed x_T.
    LatentDiffusion itself needs pytorch_lightning (absent); the sampler only reads a handful of attributes of it, which
    the stand-in below provides from the reference's own schedule function."""
    from ldm.models.diffusion.ddim import DDIMSampler
    from ldm.modules.diffusionmodules.util import make_beta_schedule
    cfg = gc.LDM_TINY_CFG
    unet = UNetModel(**cfg).eval()
    gc.det_init_(unet, 9)
    betas = make_beta_schedule('linear', 1000, linear_start=0.0015, linear_end=0.0195, cosine_s=8e-3)
    alphas_cumprod = np.cumprod(1.0 - betas, axis=0)

    class Host:                                   # what DDIMSampler reads from LatentDiffusion
        num_timesteps = 1000
        device = torch.device('cpu')
        parameterization = 'eps'

        def __init__(self):
            self.betas = torch.tensor(betas, dtype=torch.float32)
            self.alphas_cumprod = torch.tensor(alphas_cumprod, dtype=torch.float32)
            self.alphas_cumprod_prev = torch.tensor(np.append(1.0, alphas_cumprod[:-1]), dtype=torch.float32)

        def apply_model(self, x, t, c):           # ddpm.py:840-860 with conditioning_key 'crossattn': unet(x, t, context=c)
            return unet(x, t, context=c)

    class CpuSampler(DDIMSampler):
        def register_buffer(self, name, attr):    # the reference moves every buffer to 'cuda' here; no arithmetic
            setattr(self, name, attr)

    B, H = 2, cfg['image_size']
    x_T = torch.from_numpy(gc.det_noise((B, cfg['in_channels'], H, H), 51))
    cond = torch.from_numpy(gc.det_noise((B, 1, cfg['context_dim']), 52))
    uncond = torch.from_numpy(gc.det_noise((B, 1, cfg['context_dim']), 53))
    sampler = CpuSampler(Host())
    with torch.no_grad():
        samples, inter = sampler.sample(S=20, conditioning=cond, batch_size=B, shape=[cfg['in_channels'], H, H], verbose=False,
                                        unconditional_guidance_scale=3.0, unconditional_conditioning=uncond, eta=0.0, x_T=x_T,
                                        log_every_t=5)
    np.savez(os.path.join(HERE, 'ldm_sampler.npz'), samples=samples.numpy(), ddim_timesteps=np.asarray(sampler.ddim_timesteps),
             ddim_alphas=np.asarray(sampler.ddim_alphas, dtype=np.float64), alphas_cumprod=alphas_cumprod.astype(np.float64),
             x_inter=torch.stack(inter['x_inter']).numpy())
    print('ldm sampler ok:', tuple(samples.shape), 'steps', list(sampler.ddim_timesteps)[:4], '...', float(samples.abs().mean()))


def _latent_diffusion():
    """The reference's LatentDiffusion object used by ldm_loss / ldm_driver (see ldm_loss for what runs and what is stubbed)."""
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules.setdefault(name, m)

    class LightningModule(torch.nn.Module):
        @property
        def device(self):
            return torch.device('cpu')

    mod('pytorch_lightning', LightningModule=LightningModule)
    mod('pytorch_lightning.utilities')
    mod('pytorch_lightning.utilities.distributed', rank_zero_only=lambda f: f)
    mod('torchvision'); mod('torchvision.utils', make_grid=None)
    for n in ('taming', 'taming.modules', 'taming.modules.vqvae'):
        mod(n)
    mod('taming.modules.vqvae.quantize', VectorQuantizer2=object)
    mod('clip'); mod('kornia')
    from ldm.models.diffusion.ddpm import DDPM, LatentDiffusion
    from ldm.modules.encoders.modules import ClassEmbedder

    cfg = gc.LDM_TINY_CFG
    ld = LatentDiffusion.__new__(LatentDiffusion)
    ld.num_timesteps_cond = 1              # LatentDiffusion.__init__ default, read by its register_schedule override (ddpm.py:498)
    DDPM.__init__(ld, unet_config={'target': 'ldm.modules.diffusionmodules.openaimodel.UNetModel', 'params': dict(cfg)},
                  conditioning_key='crossattn', linear_start=0.0015, linear_end=0.0195, timesteps=1000, use_ema=False,
                  monitor=None, image_size=cfg['image_size'], channels=cfg['in_channels'])
    ld.cond_stage_trainable, ld.shorten_cond_schedule, ld.cond_stage_forward = True, False, None
    ld.cond_stage_key = 'class_label'
    ld.cond_stage_model = ClassEmbedder(cfg['context_dim'], n_classes=1001, key='class_label')
    ld.eval()
    gc.det_init_(ld.model.diffusion_model, 9)
    with torch.no_grad():
        ld.cond_stage_model.embedding.weight.copy_(torch.from_numpy(gc.det_param('embedding.weight', (1001, cfg['context_dim']), 61)))
    return ld, cfg


def ldm_loss():
    """Pins get_loss_at_t / p_losses / q_sample / get_learned_conditioning of the LDM importance pass
    (ldm_exp/prune_ldm.py:122-123 -> ldm/models/diffusion/ddpm.py:881-889, 1022-1056, 274-277, 553-565) and ClassEmbedder
    (ldm/modules/encoders/modules.py:21-33) by running the reference's OWN methods.  `ddpm.py` imports pytorch_lightning,
    torchvision.utils, taming, clip and kornia at module level (all absent here); empty stand-in modules satisfy those import
    statements -- none of their code is on this path (LightningModule only contributes `nn.Module` + a `.device` property).
    The object is a LatentDiffusion whose base-class __init__ (DDPM.__init__: the reference's DiffusionWrapper around the
    reference's UNetModel, register_schedule with the cin256-v2 linear_start / linear_end, logvar, loss weights) really runs;
    LatentDiffusion.__init__ itself (first-stage autoencoder, checkpoint plumbing) is skipped and the five attributes
    get_loss_at_t reads are set as configs/latent-diffusion/cin256-v2.yaml sets them."""
    ld, cfg = _latent_diffusion()
    assert float(ld.logvar.abs().max()) == 0.0 and ld.l_simple_weight == 1.0 and ld.original_elbo_weight == 0.0
    B, H = 3, cfg['image_size']
    x = torch.from_numpy(gc.det_noise((B, cfg['in_channels'], H, H), 62))
    xc = torch.tensor([3, 500, 1000])
    out = dict(class_ids=xc.numpy(), context=ld.get_learned_conditioning({'class_label': xc}).detach().numpy(),
               sqrt_acp=ld.sqrt_alphas_cumprod.numpy().astype(np.float64),
               sqrt_1macp=ld.sqrt_one_minus_alphas_cumprod.numpy().astype(np.float64))
    ts = [0, 1, 250, 999]
    losses, noisy = [], []
    for k, t in enumerate(ts):
        noise = torch.from_numpy(gc.det_noise((B, cfg['in_channels'], H, H), 70 + k))
        tt = torch.full((B,), t, dtype=torch.long)
        ld.zero_grad()
        loss = ld.get_loss_at_t(x, {'class_label': xc}, tt, noise=noise)
        losses.append(float(loss[0]))
        noisy.append(ld.q_sample(x, tt, noise).numpy())
        if t == 250:
            loss[0].backward()
            gs = {n: p.grad for n, p in ld.model.diffusion_model.named_parameters()}
            out['grad_abs_sum_names'] = np.array(sorted(gs))
            out['grad_abs_sum'] = np.array([float(gs[n].abs().sum()) for n in sorted(gs)], dtype=np.float64)
            out['grad_embedding_rows'] = ld.cond_stage_model.embedding.weight.grad[xc].numpy()
    out.update(ts=np.array(ts), losses=np.array(losses, dtype=np.float64), x_noisy=np.stack(noisy))
    np.savez(os.path.join(HERE, 'ldm_loss_at_t.npz'), **out)
    print('ldm loss ok: t', ts, 'losses', losses)


def ldm_driver(K=4):
    """Pins the importance-pass loop of the prune_ldm.py SCRIPT (ldm_exp/prune_ldm.py: from `import random` to `loss.backward()`,
    lines 103-127) by EXECUTING those source lines, read from the reference file at generation time, in a namespace that holds
    the reference LatentDiffusion object of ldm_loss, the reference DDIMSampler, and the script's own preamble values at reduced
    size (2 samples per class, 2 DDIM steps; pruner 'diff-pruning').  Randomness is made replayable by replacing the three
    sources the loop consumes -- random.sample, torch.randn, torch.randn_like -- with counter-based deterministic draws
    (golden_common.det_noise(shape, 5000 + call index)).  All 1000 iterations run (the 0.1 threshold never fires on a random
    UNet: recorded, so the restatement's no-break path is what is pinned).  Recorded: every loss, the class ids and draw
    indices of the first K iterations, and per-parameter |grad| sums after K backward passes (snapshot taken inside the
    script's own print of iteration K, which precedes that iteration's backward)."""
    import builtins
    import random as _random
    from ldm.models.diffusion.ddim import DDIMSampler
    ld, cfg = _latent_diffusion()

    class CpuSampler(DDIMSampler):
        def register_buffer(self, name, attr):    # the reference moves every buffer to 'cuda' here; no arithmetic
            setattr(self, name, attr)

    n = 2
    calls, class_draws, printed, snap = [], [], [], {}

    scale_draw = {}                            # draw index -> factor (the break scenario shrinks one loss-noise draw)

    def det(shape):
        shape = tuple(int(v) for v in shape)
        calls.append(shape)
        return torch.from_numpy(gc.det_noise(shape, 5000 + len(calls) - 1)) * scale_draw.get(len(calls) - 1, 1.0)

    def det_randn(*size, **kw):
        return det(size[0] if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else size)

    def det_sample(population, k):
        c = len(class_draws)
        ids = [(37 * c + 101 * i + 11) % len(population) for i in range(k)]
        class_draws.append((ids, len(calls)))
        return ids

    def rec_print(*a):
        printed.append([float(v) for v in a])
        if int(a[0]) == K:
            for name, p in ld.model.diffusion_model.named_parameters():
                snap[name] = float(p.grad.abs().sum())

    src = open('/root/reference/ldm_exp/prune_ldm.py').read().splitlines()
    i0 = src.index('import random')
    i1 = next(i for i in range(i0, len(src)) if src[i].strip() == 'loss.backward()')
    block = '\n'.join(src[i0:i1 + 1])
    ns = dict(model=ld, sampler=CpuSampler(ld), args=types.SimpleNamespace(pruner='diff-pruning'), n_samples_per_class=n,
              ddim_steps=2, scale=3.0, ddim_eta=0.0, torch=torch, print=rec_print,
              uc=ld.get_learned_conditioning({ld.cond_stage_key: torch.tensor(n * [1000])}))
    ld.zero_grad()
    keep = (torch.randn, torch.randn_like, _random.sample)
    torch.randn, torch.randn_like, _random.sample = det_randn, lambda x: det(x.shape), det_sample
    try:
        exec(compile(block, 'prune_ldm.py[%d:%d]' % (i0 + 1, i1 + 1), 'exec'), ns)
    finally:
        torch.randn, torch.randn_like, _random.sample = keep
    losses = [p[2] for p in printed]
    assert len(losses) == 1000 and len(snap) > 0
    per_iter = len(calls) // len(class_draws)
    first = []
    for k in range(K):
        ids, c0 = class_draws[k]
        first.append(dict(class_ids=ids, x_T_draw=5000 + c0, noise_draw=5000 + c0 + per_iter - 1,
                          shapes=[list(c) for c in calls[c0:c0 + per_iter]]))
    rec = dict(n_samples=n, ddim_steps=2, scale=3.0, thr=0.1, lines=[i0 + 1, i1 + 1], iterations=len(losses), losses=losses,
               max_loss=float(ns['max_loss']), first=first, K=K, grad_abs_sum_after_K=snap, draws_per_iteration=per_iter)
    # Second scenario, for the branch the run above never takes: the threshold break.  The UNet's output convolution is zeroed
    # (eps_hat = 0, so loss = mean(noise^2)) and the loss-noise draw of iteration 2 is scaled by 0.3 -> loss ratio 0.09 < 0.1:
    # the script must stop at t = 2 WITHOUT that iteration's backward (gradients = two passes).
    with torch.no_grad():
        for pn, pp in ld.model.diffusion_model.named_parameters():
            if pn.startswith('out.2.'):
                pp.zero_()
    ld.zero_grad()
    del calls[:], class_draws[:], printed[:]
    scale_draw[2 * per_iter + per_iter - 1] = 0.3
    ns.update(sampler=CpuSampler(ld))
    ns.pop('max_loss', None)
    torch.randn, torch.randn_like, _random.sample = det_randn, lambda x: det(x.shape), det_sample
    try:
        exec(compile(block, 'prune_ldm.py[%d:%d]' % (i0 + 1, i1 + 1), 'exec'), ns)
    finally:
        torch.randn, torch.randn_like, _random.sample = keep
    assert len(printed) == 2 and len(class_draws) == 3, (len(printed), len(class_draws))       # broke inside iteration 2, before its print
    rec['break_case'] = dict(zeroed_prefix='out.2.', scaled_draw=5000 + 3 * per_iter - 1, factor=0.3, printed_losses=[p[2] for p in printed],
                             breaking_loss=float(ns['loss']), max_loss=float(ns['max_loss']), stopped_at=int(ns['t']),
                             class_ids=[c[0] for c in class_draws],
                             grad_abs_sum={pn: float(pp.grad.abs().sum()) for pn, pp in ld.model.diffusion_model.named_parameters()
                                           if pp.grad is not None and float(pp.grad.abs().sum()) > 0})
    json.dump(rec, open(os.path.join(HERE, 'ldm_driver.json'), 'w'))
    print('break case: stopped at t =', ns['t'], 'loss', float(ns['loss']), 'ratio', float(ns['loss']) / float(ns['max_loss']))
    print('ldm driver ok: lines', i0 + 1, i1 + 1, 'iterations', len(losses), 'losses', losses[:3], 'min ratio', min(losses) / rec['max_loss'])



LDM_HEAD_VARIANTS = {   # multi-head / deeper SpatialTransformers (openaimodel.py:542-559, attention.py:196-258): (config, context tokens)
    'h2d2': (dict(gc.LDM_TINY_CFG, num_heads=2, transformer_depth=2), 1),                       # 2 heads, 2 blocks, the class token
    'hc16_L3': (dict(gc.LDM_TINY_CFG, num_heads=-1, num_head_channels=16, attention_resolutions=[4, 2]), 3),   # 6 / 10 heads of 16, 3 tokens
}


def ldm_heads():
    """Multi-head (`num_heads` > 1 / `num_head_channels`) and `transformer_depth` > 1 members of the LDM UNet family under the
    reference's own UNetModel + vendored torch_pruning: forward, loss, gradients, the group table, and the interactive Taylor prune
    with the head channel groups of prune_ldm.py:78-82 (scores, masks, shapes and forward after)."""
    sys.path.insert(0, '/root/reference/ddpm_exp')
    os.makedirs('/tmp/golden_scratch', exist_ok=True)
    os.chdir('/tmp/golden_scratch')
    import torch_pruning as tp
    from ldm.modules.attention import CrossAttention
    f = tp.function
    rec_all, arrays = {}, {}
    for tag, (cfg, L_ctx) in LDM_HEAD_VARIANTS.items():
        m = UNetModel(**cfg).eval()
        gc.det_init_(m, 9)
        H = cfg['image_size']
        x = torch.from_numpy(gc.det_noise((2, 3, H, H), 31))
        ctx = torch.from_numpy(gc.det_noise((2, L_ctx, cfg['context_dim']), 32))
        noise = torch.from_numpy(gc.det_noise((2, 3, H, H), 33))
        t = torch.tensor([7, 640])
        ex = {'x': torch.randn(2, 3, H, H), 'timesteps': torch.full((2,), 1, dtype=torch.long), 'context': torch.randn(2, L_ctx, cfg['context_dim'])}
        channel_groups = {}
        for mod in m.modules():
            if isinstance(mod, CrossAttention):
                channel_groups[mod.to_q] = channel_groups[mod.to_k] = channel_groups[mod.to_v] = mod.heads
        pr = tp.pruner.MagnitudePruner(m, ex, importance=tp.importance.TaylorImportance(), iterative_steps=1,
                                       channel_groups=channel_groups, ch_sparsity=0.3, ignored_layers=[m.out], round_to=2)
        names = {mod: n for n, mod in m.named_modules()}
        table = []
        for g in pr.DG.get_all_groups(ignored_layers=pr.ignored_layers, root_module_types=pr.root_module_types):
            mem = []
            for dep, idxs in g:
                mod = dep.target.module
                if mod not in names:
                    continue
                h = dep.handler
                kind = ('out' if h in (f.prune_conv_out_channels, f.prune_linear_out_channels) else
                        'in' if h in (f.prune_conv_in_channels, f.prune_linear_in_channels) else
                        'gn' if h == f.prune_groupnorm_out_channels else
                        'ln' if h == f.prune_layernorm_out_channels else 'other')
                rng = []
                for i in sorted(idxs):
                    if rng and rng[-1][1] == i:
                        rng[-1][1] = i + 1
                    else:
                        rng.append([i, i + 1])
                mem.append([names[mod], kind, rng])
            table.append(dict(ch_groups=int(pr.get_channel_groups(g)), members=mem))
        m.zero_grad()
        y = m(x, t, context=ctx)
        loss = (y - noise).square().mean(dim=(1, 2, 3)).mean()
        loss.backward()
        P = dict(m.named_parameters())
        stats = {n: [float(p.grad.double().sum()), float(p.grad.double().abs().sum())] for n, p in P.items()}
        shapes = {n: list(p.shape) for n, p in P.items()}
        arrays[tag + '::fwd_out'] = y.detach().numpy().copy()
        arrays[tag + '::loss'] = np.array(float(loss))
        st = next(n for n in P if n.endswith('.transformer_blocks.0.attn1.to_q.weight'))[:-len('.transformer_blocks.0.attn1.to_q.weight')]
        last = 'transformer_blocks.%d' % (cfg.get('transformer_depth', 1) - 1)
        for n in [st + '.transformer_blocks.0.attn1.to_q.weight', st + '.' + last + '.attn1.to_k.weight', st + '.' + last + '.attn2.to_q.weight',
                  st + '.' + last + '.attn2.to_v.weight', st + '.' + last + '.norm2.weight', st + '.' + last + '.ff.net.0.proj.weight',
                  'middle_block.1.' + last + '.attn1.to_v.weight', 'middle_block.1.proj_in.weight', 'input_blocks.0.0.weight']:
            arrays[tag + '::grad::' + n] = P[n].grad.numpy().copy()
        pr.current_step += 1
        rec = []
        for group in pr.DG.get_all_groups(ignored_layers=pr.ignored_layers, root_module_types=pr.root_module_types):
            if not pr._check_sparsity(group):
                continue
            module, fn = group[0][0].target.module, group[0][0].handler
            ch_groups = pr.get_channel_groups(group)
            imp = pr.estimate_importance(group, ch_groups=ch_groups)
            if imp is None:
                continue
            cur = pr.DG.get_out_channels(module)
            n_pruned = cur - int(pr.layer_init_out_ch[module] * (1 - pr.get_target_sparsity(module)))
            if pr.round_to:
                n_pruned = n_pruned - (n_pruned % pr.round_to)
            if n_pruned <= 0:
                continue
            if ch_groups > 1:
                gs, per = cur // ch_groups, n_pruned // ch_groups
                idxs = torch.cat([torch.argsort(imp[c * gs:(c + 1) * gs])[:per] + c * gs for c in range(ch_groups)], 0)
            else:
                idxs = torch.argsort(imp)[:(n_pruned // ch_groups)]
            g2 = pr.DG.get_pruning_group(module, fn, idxs.tolist())
            ok = pr.DG.check_pruning_group(g2)
            rec.append(dict(root=names[module], ch_groups=int(ch_groups), cur=int(cur), n_pruned=int(n_pruned),
                            pruned=sorted(int(i) for i in idxs.tolist()), score=gc.f32_to_b64(imp.detach().float().numpy()), ok=bool(ok)))
            if ok:
                g2.prune()
        with torch.no_grad():
            y2 = m(x, t, context=ctx)
        arrays[tag + '::fwd_after'] = y2.numpy().copy()
        rec_all[tag] = dict(cfg=cfg, context_tokens=L_ctx, shapes=shapes, grad_stats=stats, groups=table, prune=rec,
                            heads={names[mod]: int(mod.heads) for mod in m.modules() if isinstance(mod, CrossAttention)},
                            shapes_after={n: list(p.shape) for n, p in m.named_parameters()},
                            params_after=sum(p.numel() for p in m.parameters()))
        print(tag, 'groups', len(table), 'pruned groups', len(rec), 'loss', float(loss), 'params after', rec_all[tag]['params_after'])
    json.dump(rec_all, open(os.path.join(HERE, 'ldm_heads.json'), 'w'))
    np.savez(os.path.join(HERE, 'ldm_heads.npz'), **arrays)


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'heads':
    ldm_heads()
    sys.exit(0)

if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'driver':
    ldm_driver()
    sys.exit(0)

if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'loss':
    ldm_loss()
    sys.exit(0)

if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'sampler':
    ldm_sampler()
