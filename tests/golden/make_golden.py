#!/usr/bin/env python3
"""Golden-vector generator (runs ONLY in the build container, never on the GPU box).

Imports the *reference's own code* from /root/reference (vendored diffusers 0.17.0.dev0 and the
vendored ddpm_exp/torch_pruning) through the compatibility shim of SURVEY.md App. C, runs the hot
path on seeded inputs and writes small data fixtures into tests/golden/.  Only DATA is written:
inputs are regenerated from `numpy.random.default_rng` seeds by `golden_common.py`, so no reference
source text ends up in this repository.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py            # all fixtures
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py tiny ddim  # a subset

Fixtures:
  schedule.npz      G6 alphas_cumprod table + add_noise outputs; G8 timestep embeddings
  ddim.npz          G7 DDIM timestep lists (uniform/quad) + 5 consecutive step() outputs
  tiny_unet.npz     G1 tiny UNet forward output, loss, per-parameter gradient statistics + a few full grads
  tiny_prune.json   G3/G4 for the tiny UNet: group tables, scores, pruned indices (ratio 0.3), early-exit step count
  cifar_groups.json G3 group table for the CIFAR-10 UNet (50 groups) and the bedroom-256 topology
  cifar_c1.npz/json G2/G4/G5 config C1: CIFAR UNet, B=4, 8 timesteps, Taylor, ratio 0.3
  tiny_long_sweep.json  1000-step sweep + prune of the tiny UNet (accumulation length of config C2)
  cifar_long_sweep.json the same on the CIFAR-10 UNet at B=4 (~8 min of the reference on 8 CPU threads)
  lr_schedules.json     diffusers get_scheduler: lr per step for every schedule type
  ddpm.npz              DDPMScheduler.step sequences + a DDPMPipeline call
  tiny_dropout.json     finetune loss / gradients of the reference UNet in train mode with reproducible (Philox) masks
"""
import os, sys, json, time, base64, io

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import numpy as np
import torch
import golden_common as gc

# ---------------------------------------------------------------------------------------------
# reference import shim
# ---------------------------------------------------------------------------------------------
import huggingface_hub, huggingface_hub.constants as _c, importlib.util as _iu
_c.hf_cache_home = getattr(_c, 'hf_cache_home', os.path.expanduser('~/.cache/huggingface'))
class _HfFolder:
    get_token = staticmethod(lambda: None)
for _n, _v in (('HfFolder', _HfFolder), ('cached_download', lambda *a, **k: (_ for _ in ()).throw(RuntimeError('offline')))):
    if not hasattr(huggingface_hub, _n):
        setattr(huggingface_hub, _n, _v)
_orig = _iu.find_spec
_iu.find_spec = lambda name, package=None: None if name.split('.')[0] in {
    'transformers', 'flax', 'jax', 'onnxruntime', 'k_diffusion', 'xformers', 'tensorflow'} else _orig(name, package)
sys.path[:0] = ['/root/reference', '/root/reference/ddpm_exp']
os.makedirs('/tmp/golden_scratch', exist_ok=True)
os.chdir('/tmp/golden_scratch')          # vendored MetaPruner writes run/pruning_logs/*.png into cwd

import diffusers                                                    # noqa: E402  (reference)
import torch_pruning as tp                                          # noqa: E402  (reference, vendored)
from diffusers import UNet2DModel, DDPMScheduler, DDIMScheduler     # noqa: E402
from diffusers.models.attention_processor import Attention, AttnProcessor  # noqa: E402
from diffusers.models.embeddings import get_timestep_embedding      # noqa: E402
from diffusers.models.resnet import Upsample2D, Downsample2D        # noqa: E402

torch.set_num_threads(8)


def build_ref_unet(cfg, seed):
    m = UNet2DModel(**cfg).eval()
    gc.det_init_(m, seed)
    return m


def name_of(model):
    return {mod: n for n, mod in model.named_modules()}


def compress(idxs):
    """list of ints -> list of [start, stop) ranges (order preserved for sorted input)."""
    out = []
    for i in idxs:
        if out and out[-1][1] == i:
            out[-1][1] = i + 1
        else:
            out.append([i, i + 1])
    return out


def kind_of(dep):
    h = dep.handler
    f = tp.function
    if h in (f.prune_conv_out_channels, f.prune_linear_out_channels):
        return 'out'
    if h in (f.prune_conv_in_channels, f.prune_linear_in_channels):
        return 'in'
    if h == f.prune_groupnorm_out_channels:
        return 'gn'
    if h in (f.prune_batchnorm_out_channels, f.prune_batchnorm_in_channels):
        return 'bn'
    if h in (f.prune_layernorm_out_channels, f.prune_layernorm_in_channels):
        return 'ln'
    if h in (f.prune_depthwise_conv_out_channels, f.prune_depthwise_conv_in_channels):
        return 'out'
    if h in (f.prune_instancenorm_out_channels, f.prune_instancenorm_in_channels):
        return 'inorm'
    if h in (f.prune_prelu_out_channels, f.prune_prelu_in_channels):
        return 'prelu'
    if h in (f.prune_embedding_out_channels, f.prune_embedding_in_channels):
        return 'embed'
    return 'other'


def dump_group(group, names):
    mem = []
    for dep, idxs in group:
        mod = dep.target.module
        if mod not in names:
            continue          # elementwise / concat / reshape pseudo-ops
        mem.append([names[mod], kind_of(dep), compress(sorted(idxs))])
    return mem


def make_pruner(model, H, ratio=0.3, imp=None):
    ex = {'sample': torch.randn(1, model.config.in_channels, H, H), 'timestep': torch.ones((1,)).long()}
    imp = imp if imp is not None else tp.importance.TaylorImportance()
    return tp.pruner.MagnitudePruner(model, ex, importance=imp, iterative_steps=1, channel_groups={},
                                     ch_sparsity=ratio, ignored_layers=[model.conv_out]), ex


def sweep(model, sched, clean, noise, steps, thr=None):
    """ddpm_prune.py:94-106 loop, verbatim semantics."""
    model.zero_grad()
    model.eval()
    losses = []
    loss_max = 0
    B = clean.shape[0]
    for step_k in range(steps):
        timesteps = (step_k * torch.ones((B,))).long()
        noisy = sched.add_noise(clean, noise, timesteps)
        out = model(noisy, timesteps).sample
        loss = torch.nn.functional.mse_loss(out, noise)
        loss.backward()
        losses.append(float(loss))
        if thr is not None:
            if loss > loss_max:
                loss_max = loss
            if loss < loss_max * thr:
                break
    return losses


def grad_stats(model):
    st = {}
    for n, p in model.named_parameters():
        g = p.grad.detach().double()
        st[n] = [float(g.sum()), float(g.abs().sum()), float((g * g).sum())]
    return st


def prune_run(model, H, ratio=0.3, imp=None):
    """pruner.step(interactive=True) loop of ddpm_prune.py:108-116, recording every group."""
    pruner, ex = make_pruner(model, H, ratio, imp)
    pruner.current_step += 1          # what MetaPruner.step() does first (metapruner.py:155)
    names = name_of(model)
    rec = []
    # re-implement prune_local's bookkeeping around the reference calls so that scores can be recorded
    for group in pruner.DG.get_all_groups(ignored_layers=pruner.ignored_layers,
                                          root_module_types=pruner.root_module_types):
        if not pruner._check_sparsity(group):
            continue
        module = group[0][0].target.module
        fn = group[0][0].handler
        ch_groups = pruner.get_channel_groups(group)
        imp = pruner.estimate_importance(group, ch_groups=ch_groups)
        if imp is None:
            continue
        cur = pruner.DG.get_out_channels(module)
        n_pruned = cur - int(pruner.layer_init_out_ch[module] * (1 - pruner.get_target_sparsity(module)))
        if n_pruned <= 0:
            continue
        if ch_groups > 1:
            gs = cur // ch_groups
            per = n_pruned // ch_groups
            idxs = []
            for chg in range(ch_groups):
                sub = imp[chg * gs:(chg + 1) * gs]
                idxs.append(torch.argsort(sub)[:per] + chg * gs)
            idxs = torch.cat(idxs, 0)
        else:
            idxs = torch.argsort(imp)[:(n_pruned // ch_groups)]
        members = dump_group(group, names)
        g2 = pruner.DG.get_pruning_group(module, fn, idxs.tolist())
        ok = pruner.DG.check_pruning_group(g2)
        rec.append(dict(root=names[module], ch_groups=int(ch_groups), cur=int(cur), n_pruned=int(n_pruned),
                        pruned=sorted(int(i) for i in idxs.tolist()), members=members,
                        score=gc.f32_to_b64(imp.detach().float().numpy()), ok=bool(ok)))
        if ok:
            g2.prune()
    for m in model.modules():
        if isinstance(m, (Upsample2D, Downsample2D)):
            m.channels = m.conv.in_channels
    return rec


def group_table(model, H):
    pruner, ex = make_pruner(model, H)
    names = name_of(model)
    table = []
    for group in pruner.DG.get_all_groups(ignored_layers=pruner.ignored_layers,
                                          root_module_types=pruner.root_module_types):
        table.append(dict(ch_groups=int(pruner.get_channel_groups(group)), members=dump_group(group, names)))
    return table


# ---------------------------------------------------------------------------------------------
def do_schedule():
    s = DDPMScheduler(num_train_timesteps=1000)
    x0 = torch.from_numpy(gc.det_clean((2, 3, 4, 4), 11))
    eps = torch.from_numpy(gc.det_noise((2, 3, 4, 4), 12))
    ts = [0, 1, 7, 500, 999]
    outs = [s.add_noise(x0, eps, torch.tensor([t, t])).numpy() for t in ts]
    emb = get_timestep_embedding(torch.tensor([0, 1, 999]), 128, flip_sin_to_cos=False, downscale_freq_shift=1).numpy()
    emb_f = get_timestep_embedding(torch.tensor([0.0, 1.0, 999.0]), 32, flip_sin_to_cos=True, downscale_freq_shift=0).numpy()
    np.savez(os.path.join(HERE, 'schedule.npz'), alphas_cumprod=s.alphas_cumprod.numpy(), ts=np.array(ts),
             add_noise=np.stack(outs), temb_128=emb, temb_32_flip=emb_f)
    print('schedule.npz ok')


def do_ddim():
    cfg = gc.TINY_CFG
    model = build_ref_unet(cfg, 5)
    out = {}
    for skip in ('uniform', 'quad'):
        sch = DDIMScheduler(num_train_timesteps=1000)
        sch.skip_type = skip
        sch.set_timesteps(100)
        out['timesteps_' + skip] = sch.timesteps.numpy().copy()
    sch = DDIMScheduler(num_train_timesteps=1000)
    sch.skip_type = 'uniform'
    sch.set_timesteps(100)
    x = torch.from_numpy(gc.det_noise((2, 3, 16, 16), 21))
    xs = []
    with torch.no_grad():
        for t in sch.timesteps[:5]:
            eps = model(x, t).sample
            x = sch.step(eps, t, x, eta=0.0).prev_sample
            xs.append(x.numpy().copy())
    out['x_steps'] = np.stack(xs)
    # a 10-step full chain + final image post-processing (pipeline_ddim.py:103-117)
    sch.set_timesteps(10)
    x = torch.from_numpy(gc.det_noise((2, 3, 16, 16), 22))
    with torch.no_grad():
        for t in sch.timesteps:
            x = sch.step(model(x, t).sample, t, x, eta=0.0).prev_sample
    out['chain10_image'] = (x / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).numpy()
    np.savez(os.path.join(HERE, 'ddim.npz'), **out)
    print('ddim.npz ok')


def do_tiny():
    cfg = gc.TINY_CFG
    H = cfg['sample_size']
    model = build_ref_unet(cfg, 5)
    sched = DDPMScheduler(num_train_timesteps=1000)
    B = 2
    clean = torch.from_numpy(gc.det_clean((B, 3, H, H), 1))
    noise = torch.from_numpy(gc.det_noise((B, 3, H, H), 2))
    t = torch.tensor([3, 500])
    with torch.no_grad():
        y = model(sched.add_noise(clean, noise, t), t).sample.numpy()
    losses = sweep(model, sched, clean, noise, 4)
    st = grad_stats(model)
    full = {}
    for n in ['conv_in.weight', 'down_blocks.1.attentions.0.to_q.weight', 'mid_block.resnets.0.conv1.weight',
              'up_blocks.2.resnets.0.conv_shortcut.weight', 'up_blocks.0.upsamplers.0.conv.weight',
              'down_blocks.0.downsamplers.0.conv.weight', 'time_embedding.linear_1.weight', 'conv_norm_out.weight',
              'down_blocks.1.attentions.0.group_norm.bias', 'conv_out.bias']:
        full['grad::' + n] = dict(model.named_parameters())[n].grad.numpy().copy()
    np.savez(os.path.join(HERE, 'tiny_unet.npz'), fwd_out=y, losses=np.array(losses), **full)
    table = group_table(model, H)
    rec = prune_run(model, H, 0.3)
    nparams = sum(p.numel() for p in model.parameters())
    for m in model.modules():
        if isinstance(m, Attention):
            m.set_processor(AttnProcessor())
    with torch.no_grad():
        y2 = model(sched.add_noise(clean, noise, t), t).sample.numpy()
    # diff-pruning early exit on a fresh model
    model2 = build_ref_unet(cfg, 5)
    losses2 = sweep(model2, sched, clean, noise, 1000, thr=0.99)
    json.dump(dict(grad_stats=st, groups=table, prune=rec, params_after=int(nparams),
                   shapes_after={n: list(p.shape) for n, p in model.named_parameters()},
                   fwd_after=gc.f32_to_b64(y2), early_exit=dict(thr=0.99, steps=len(losses2), losses=losses2)),
              open(os.path.join(HERE, 'tiny_prune.json'), 'w'))
    print('tiny ok: groups', len(table), 'pruned groups', len(rec), 'params', nparams, 'early steps', len(losses2))


def do_tiny_bedroom():
    """Bedroom/church-256 topology (6 levels, attention in the 5th down / 2nd up block) at tiny widths: forward output, loss and
    per-parameter gradient statistics of one sweep step from the reference UNet2DModel."""
    cfg = dict(gc.BEDROOM_CFG, block_out_channels=[16, 16, 32, 32, 64, 64], sample_size=32, norm_num_groups=8)
    model = build_ref_unet(cfg, 3)
    sched = DDPMScheduler(num_train_timesteps=1000)
    clean = torch.from_numpy(gc.det_clean((1, 3, 32, 32), 5))
    noise = torch.from_numpy(gc.det_noise((1, 3, 32, 32), 6))
    t = torch.tensor([250])
    with torch.no_grad():
        y = model(sched.add_noise(clean, noise, t), t).sample.numpy()
    losses = sweep(model, sched, clean, noise, 2)
    np.savez(os.path.join(HERE, 'tiny_bedroom.npz'), fwd_out=y, losses=np.array(losses))
    json.dump(dict(cfg={k: (list(v) if isinstance(v, (list, tuple)) else v) for k, v in cfg.items()}, grad_stats=grad_stats(model)),
              open(os.path.join(HERE, 'tiny_bedroom.json'), 'w'))
    print('tiny bedroom ok', y.shape, losses)


def do_optim():
    """ddpm_train.py:453-469 update rule on small seeded tensors, 3 consecutive steps: clip_grad_norm_(1.0) -> torch.optim.Adam
    with the script's argparse defaults (ddpm_train.py:137-159: betas (0.9, 0.999), weight_decay 0, eps 1e-8) and the lr of
    scripts/finetune_ddpm_cifar10.sh (2e-4) -> vendored EMAModel.step (ema_max_decay 0.9999 from the same script)."""
    from diffusers.training_utils import EMAModel
    shapes = [(4, 3, 3, 3), (4,), (5, 4)]
    params = [torch.nn.Parameter(torch.from_numpy(gc.det_param('p%d.weight' % i, s, 61))) for i, s in enumerate(shapes)]
    ema = EMAModel(params, decay=0.9999, use_ema_warmup=False, inv_gamma=1.0, power=0.75)
    hp = dict(lr=2e-4, betas=(0.9, 0.999), weight_decay=0.0, eps=1e-8)
    opt = torch.optim.Adam(params, **hp)
    rec = []
    for step in range(3):
        opt.zero_grad()
        for i, p_ in enumerate(params):
            p_.grad = torch.from_numpy(gc.det_param('g%d_%d.weight' % (i, step), shapes[i], 62)) * (3.0 if step == 0 else 0.05)
        norm = torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        ema.step(params)
        rec.append(dict(norm=float(norm), params=[gc.f32_to_b64(p_.detach().numpy()) for p_ in params],
                        ema=[gc.f32_to_b64(e.detach().numpy()) for e in ema.shadow_params]))
    json.dump(dict(shapes=[list(s) for s in shapes], hp=dict(lr=hp['lr'], betas=list(hp['betas']), weight_decay=hp['weight_decay'],
                                                            eps=hp['eps']), ema_decay=0.9999, steps=rec),
              open(os.path.join(HERE, 'optim.json'), 'w'))
    print('optim ok', [r['norm'] for r in rec])


def do_pretrained():
    """A Diffusers pipeline directory as the reference writes it (ddpm_prune.py:50,131: DDPMPipeline.from_pretrained /
    save_pretrained): micro UNet (9 k parameters) with seeded weights -> tests/golden/pretrained_micro/."""
    import shutil
    from diffusers import DDPMPipeline
    cfg = dict(gc.TINY_CFG, block_out_channels=[8, 8], down_block_types=['DownBlock2D', 'AttnDownBlock2D'],
               up_block_types=['AttnUpBlock2D', 'UpBlock2D'], norm_num_groups=4, sample_size=8, layers_per_block=1)
    model = build_ref_unet(cfg, 71)
    out = os.path.join(HERE, 'pretrained_micro')
    shutil.rmtree(out, ignore_errors=True)
    DDPMPipeline(unet=model, scheduler=DDPMScheduler(num_train_timesteps=1000)).save_pretrained(out)
    x = torch.from_numpy(gc.det_noise((1, 3, 8, 8), 72))
    with torch.no_grad():
        y = model(x, torch.tensor([10])).sample
    np.savez(os.path.join(out, 'expected.npz'), fwd_out=y.numpy())
    for root, _, files in os.walk(out):
        for f in files:
            print(os.path.relpath(os.path.join(root, f), out), os.path.getsize(os.path.join(root, f)))


def do_criteria():
    """The sibling criteria selectable in ddpm_exp/prune.py:193-208 on the tiny UNet after a 4-step sweep:
    per-group score vectors and pruned index lists of the whole sequential prune, one run per criterion."""
    cfg = gc.TINY_CFG
    H = cfg['sample_size']
    sched = DDPMScheduler(num_train_timesteps=1000)
    clean = torch.from_numpy(gc.det_clean((2, 3, H, H), 1))
    noise = torch.from_numpy(gc.det_noise((2, 3, H, H), 2))
    crits = dict(full1=lambda: tp.importance.FullTaylorImportance(order=1),
                 full2=lambda: tp.importance.FullTaylorImportance(order=2),
                 abs=lambda: tp.importance.AbsTaylorImportance(),
                 fisher=lambda: tp.importance.FisherImportance(),
                 magnitude=lambda: tp.importance.MagnitudeImportance())
    out = {}
    for name, mk in crits.items():
        model = build_ref_unet(cfg, 5)
        sweep(model, sched, clean, noise, 4)
        rec = prune_run(model, H, 0.3, mk())
        out[name] = dict(groups=[dict(root=r['root'], ch_groups=r['ch_groups'], pruned=compress(r['pruned']),
                                      score=r['score']) for r in rec],
                         params_after=int(sum(p.numel() for p in model.parameters())))
        print('criterion', name, 'groups', len(rec), 'params', out[name]['params_after'])
    json.dump(out, open(os.path.join(HERE, 'tiny_criteria.json'), 'w'))


def do_groups():
    out = {}
    cfg = gc.CIFAR_CFG
    model = build_ref_unet(cfg, 0)
    out['cifar'] = group_table(model, 32)
    print('cifar groups', len(out['cifar']))
    # bedroom topology at reduced width and resolution (same block types / layer counts => same graph)
    cfgb = dict(gc.BEDROOM_CFG)
    cfgb['block_out_channels'] = [32, 32, 64, 64, 128, 128]
    cfgb['sample_size'] = 64
    model = build_ref_unet(cfgb, 0)
    out['bedroom_topology'] = group_table(model, 64)
    print('bedroom groups', len(out['bedroom_topology']))
    json.dump(out, open(os.path.join(HERE, 'groups.json'), 'w'))


def do_groups_more():
    """Further UNet2DModel topologies for the group enumeration (the graph is generic over block types, depths and widths):
    all-attention blocks, 3 levels with layers_per_block 1 / 3, the Diffusers default head dim (attention_head_dim 8),
    non-uniform widths."""
    base = dict(gc.TINY_CFG)
    variants = {
        'all_attn_3lvl_l1': dict(base, block_out_channels=[16, 32, 32], layers_per_block=1, norm_num_groups=8, sample_size=16,
                                 down_block_types=['AttnDownBlock2D'] * 3, up_block_types=['AttnUpBlock2D'] * 3),
        'no_attn_2lvl_l3': dict(base, block_out_channels=[16, 48], layers_per_block=3, norm_num_groups=8, sample_size=16,
                                down_block_types=['DownBlock2D'] * 2, up_block_types=['UpBlock2D'] * 2),
        'heads8_4lvl': dict(base, block_out_channels=[16, 32, 48, 64], layers_per_block=2, norm_num_groups=8, sample_size=16,
                            attention_head_dim=8,
                            down_block_types=['DownBlock2D', 'AttnDownBlock2D', 'AttnDownBlock2D', 'DownBlock2D'],
                            up_block_types=['UpBlock2D', 'AttnUpBlock2D', 'AttnUpBlock2D', 'UpBlock2D']),
    }
    out = {}
    for name, cfg in variants.items():
        model = build_ref_unet(cfg, 0)
        out[name] = dict(cfg={k: (list(v) if isinstance(v, (list, tuple)) else v) for k, v in cfg.items()},
                         groups=group_table(model, cfg['sample_size']))
        print(name, 'groups', len(out[name]['groups']))
    # ldm_prune.py:62-90 on the multi-head variant: MagnitudeImportance + channel_groups = attention heads (q/k/v selected
    # per head), ratio 0.3 -- the pruned index list of every group
    cfg = variants['heads8_4lvl']
    model = build_ref_unet(cfg, 4)
    channel_groups = {}
    for m in model.modules():
        if isinstance(m, Attention):
            channel_groups[m.to_q] = channel_groups[m.to_k] = channel_groups[m.to_v] = m.heads
    ex = {'sample': torch.randn(1, 3, 16, 16), 'timestep': torch.ones((1,)).long()}
    pruner = tp.pruner.MagnitudePruner(model, ex, importance=tp.importance.MagnitudeImportance(), iterative_steps=1,
                                       channel_groups=channel_groups, ch_sparsity=0.3, ignored_layers=[model.conv_out])
    names = name_of(model)
    rec = []
    for g in pruner.step(interactive=True):
        rec.append(dict(root=names[g[0][0].target.module], ch_groups=int(pruner.get_channel_groups(g)),
                        pruned=compress(sorted(int(i) for i in g[0][1]))))
        g.prune()
    out['heads8_4lvl']['magnitude_prune'] = dict(seed=4, records=rec, params_after=int(sum(p.numel() for p in model.parameters())),
                                                 shapes_after={n: list(p.shape) for n, p in model.named_parameters()})
    print('heads8 magnitude prune groups', len(rec), 'head-grouped', sum(1 for r in rec if r['ch_groups'] > 1 and r['ch_groups'] != 8))
    json.dump(out, open(os.path.join(HERE, 'groups_more.json'), 'w'))


def do_tiny_heads():
    """Multi-head attention (attention_head_dim 8) on the 'heads8_4lvl' topology: forward, two sweep steps, gradient stats."""
    cfg = json.load(open(os.path.join(HERE, 'groups_more.json')))['heads8_4lvl']['cfg']
    model = build_ref_unet(cfg, 4)
    sched = DDPMScheduler(num_train_timesteps=1000)
    clean = torch.from_numpy(gc.det_clean((2, 3, 16, 16), 81))
    noise = torch.from_numpy(gc.det_noise((2, 3, 16, 16), 82))
    t = torch.tensor([5, 700])
    with torch.no_grad():
        y = model(sched.add_noise(clean, noise, t), t).sample.numpy()
    losses = sweep(model, sched, clean, noise, 2)
    np.savez(os.path.join(HERE, 'tiny_heads.npz'), fwd_out=y, losses=np.array(losses))
    json.dump(dict(grad_stats=grad_stats(model)), open(os.path.join(HERE, 'tiny_heads.json'), 'w'))
    print('tiny heads ok', y.shape, losses)


def do_c1():
    cfg = gc.CIFAR_CFG
    model = build_ref_unet(cfg, 0)
    sched = DDPMScheduler(num_train_timesteps=1000)
    B = 4
    clean = torch.from_numpy(gc.det_clean((B, 3, 32, 32), 1))
    noise = torch.from_numpy(gc.det_noise((B, 3, 32, 32), 2))
    t0 = time.time()
    losses = sweep(model, sched, clean, noise, 8)
    print('C1 sweep %.1fs' % (time.time() - t0), losses)
    st = grad_stats(model)
    base_macs, base_params = tp.utils.count_ops_and_params(
        model, {'sample': torch.randn(1, 3, 32, 32), 'timestep': torch.ones((1,)).long()})
    rec = prune_run(model, 32, 0.3)
    for m in model.modules():
        if isinstance(m, Attention):
            m.set_processor(AttnProcessor())
    macs, params = tp.utils.count_ops_and_params(
        model, {'sample': torch.randn(1, 3, 32, 32), 'timestep': torch.ones((1,)).long()})
    print('params %d -> %d, macs %.4fG -> %.4fG' % (base_params, params, base_macs / 1e9, macs / 1e9))
    json.dump(dict(losses=losses, grad_stats=st, prune=rec, base_params=int(base_params), params_after=int(params),
                   base_macs=float(base_macs), macs_after=float(macs),
                   shapes_after={n: list(p.shape) for n, p in model.named_parameters()}),
              open(os.path.join(HERE, 'cifar_c1.json'), 'w'))


def do_long_sweep():
    """1000 accumulated backward passes (the C2 sweep length; SURVEY.md §7 'hard parts': error growth over the accumulation is
    the risk to bit-exact masks): tiny UNet, B=2, plain Taylor over t = 0..999, then the whole sequential prune."""
    cfg = gc.TINY_CFG
    H = cfg['sample_size']
    model = build_ref_unet(cfg, 5)
    sched = DDPMScheduler(num_train_timesteps=1000)
    clean = torch.from_numpy(gc.det_clean((2, 3, H, H), 1))
    noise = torch.from_numpy(gc.det_noise((2, 3, H, H), 2))
    t0 = time.time()
    losses = sweep(model, sched, clean, noise, 1000)
    print('1000-step sweep %.1fs' % (time.time() - t0))
    st = grad_stats(model)
    rec = prune_run(model, H, 0.3)
    json.dump(dict(losses=losses, grad_stats=st,
                   prune=[dict(root=r['root'], ch_groups=r['ch_groups'], cur=r['cur'], pruned=r['pruned'], score=r['score'])
                          for r in rec],
                   params_after=int(sum(p.numel() for p in model.parameters()))),
              open(os.path.join(HERE, 'tiny_long_sweep.json'), 'w'))
    print('long sweep ok: groups', len(rec), 'loss[0], loss[999]', losses[0], losses[-1])


def do_c1_long():
    """C2's accumulation length on the C1-size model: CIFAR-10 UNet (35.7 M parameters), B=4, plain Taylor over t = 0..999,
    then the whole sequential prune at ratio 0.3 (scores + masks of all 50 groups)."""
    cfg = gc.CIFAR_CFG
    model = build_ref_unet(cfg, 0)
    sched = DDPMScheduler(num_train_timesteps=1000)
    clean = torch.from_numpy(gc.det_clean((4, 3, 32, 32), 1))
    noise = torch.from_numpy(gc.det_noise((4, 3, 32, 32), 2))
    t0 = time.time()
    losses = sweep(model, sched, clean, noise, 1000)
    print('C1-size 1000-step sweep %.1fs' % (time.time() - t0))
    st = grad_stats(model)
    rec = prune_run(model, 32, 0.3)
    json.dump(dict(losses=losses, grad_stats=st,
                   prune=[dict(root=r['root'], ch_groups=r['ch_groups'], cur=r['cur'], pruned=r['pruned'], score=r['score'])
                          for r in rec],
                   params_after=int(sum(p.numel() for p in model.parameters()))),
              open(os.path.join(HERE, 'cifar_long_sweep.json'), 'w'))
    print('c1 long ok: groups', len(rec), 'params', sum(p.numel() for p in model.parameters()))


def do_fid():
    """fid_score.py:182-262 on seeded feature matrices: the reference's own calculate_frechet_distance (imported with stub
    `torchvision` / `inception` modules: only numpy / scipy code is executed) and its np.mean / np.cov statistics."""
    import types
    tv = types.ModuleType('torchvision')
    tv.transforms = types.ModuleType('torchvision.transforms')
    inc = types.ModuleType('inception')
    inc.InceptionV3 = type('InceptionV3', (), dict(BLOCK_INDEX_BY_DIM={64: 0, 192: 1, 768: 2, 2048: 3}))
    for n, m in (('torchvision', tv), ('torchvision.transforms', tv.transforms), ('inception', inc)):
        sys.modules.setdefault(n, m)
    import fid_score
    out = []
    for dims, n1, n2, seed in ((64, 300, 250, 1), (192, 400, 400, 2), (48, 40, 60, 3)):
        r = np.random.default_rng(seed)
        mix = r.standard_normal((dims, dims)) / np.sqrt(dims)
        a = np.maximum(r.standard_normal((n1, dims)) @ mix + 0.3, 0).astype(np.float32)          # non-negative, correlated features
        b = np.maximum(r.standard_normal((n2, dims)) @ (mix * 1.1) + 0.35, 0).astype(np.float32)
        m1, s1 = np.mean(a.astype(np.float64), axis=0), np.cov(a.astype(np.float64), rowvar=False)
        m2, s2 = np.mean(b.astype(np.float64), axis=0), np.cov(b.astype(np.float64), rowvar=False)
        out.append(dict(dims=dims, n1=n1, n2=n2, seed=seed, fid=float(fid_score.calculate_frechet_distance(m1, s1, m2, s2)),
                        fid_self=float(fid_score.calculate_frechet_distance(m1, s1, m1, s1)),
                        mu1_sum=float(m1.sum()), trace1=float(np.trace(s1))))
    json.dump(out, open(os.path.join(HERE, 'fid.json'), 'w'))
    print('fid ok', [o['fid'] for o in out])


def do_lr():
    """diffusers/optimization.py:282 get_scheduler: the lr in force at each of 40 optimizer steps, every schedule type."""
    from diffusers.optimization import get_scheduler
    out = {}
    cases = dict(constant=dict(), constant_with_warmup=dict(num_warmup_steps=5),
                 linear=dict(num_warmup_steps=5, num_training_steps=30), cosine=dict(num_warmup_steps=5, num_training_steps=30),
                 cosine_with_restarts=dict(num_warmup_steps=5, num_training_steps=30, num_cycles=3),
                 polynomial=dict(num_warmup_steps=5, num_training_steps=30, power=2.0),
                 piecewise_constant=dict(step_rules='1:10,0.1:20,0.01:30,0.005'))
    for name, kw in cases.items():
        w = torch.nn.Parameter(torch.zeros(2))
        opt = torch.optim.Adam([w], lr=2e-4)
        if name == 'piecewise_constant':
            # get_scheduler('piecewise_constant') itself raises TypeError in the reference (optimization.py:321 passes
            # `rules=` to a function whose parameter is `step_rules`): record the working direct call
            from diffusers.optimization import get_piecewise_constant_schedule
            sch = get_piecewise_constant_schedule(opt, **kw)
        else:
            sch = get_scheduler(name, optimizer=opt, **kw)
        lrs = []
        for _ in range(40):
            lrs.append(sch.get_last_lr()[0])
            w.grad = torch.ones(2)
            opt.step()
            sch.step()
        out[name] = dict(kwargs=kw, lrs=lrs)
    json.dump(dict(base_lr=2e-4, cases=out), open(os.path.join(HERE, 'lr_schedules.json'), 'w'))
    print('lr ok', {k: v['lrs'][6] for k, v in out.items()})


def do_ddpm():
    """DDPMScheduler.step / DDPMPipeline (scheduling_ddpm.py:312-406, pipeline_ddpm.py:24-105) on the tiny UNet: 4 consecutive
    ancestral steps from t = 999 (1000 inference steps), 4 with set_timesteps(50), fixed_large variance, and a 6-step pipeline
    call; the variance noise comes from a seeded CPU generator (randn_tensor)."""
    from diffusers import DDPMPipeline
    cfg = gc.TINY_CFG
    model = build_ref_unet(cfg, 5)
    out = {}
    for tag, n_inf, vt in (('full', 1000, 'fixed_small'), ('s50', 50, 'fixed_small'), ('large', 50, 'fixed_large')):
        sch = DDPMScheduler(num_train_timesteps=1000, variance_type=vt)
        sch.set_timesteps(n_inf)
        gen = torch.Generator().manual_seed(123)
        x = torch.from_numpy(gc.det_noise((2, 3, 16, 16), 23))
        xs = []
        with torch.no_grad():
            for t in sch.timesteps[:4]:
                x = sch.step(model(x, t).sample, t, x, generator=gen).prev_sample
                xs.append(x.numpy().copy())
        out['x_' + tag] = np.stack(xs)
        out['timesteps_' + tag] = sch.timesteps.numpy().copy()
    # last step (t = 0: no noise) from a seeded sample
    sch = DDPMScheduler(num_train_timesteps=1000)
    x = torch.from_numpy(gc.det_noise((2, 3, 16, 16), 24))
    with torch.no_grad():
        out['x_t0'] = sch.step(model(x, 0).sample, 0, x, generator=torch.Generator().manual_seed(5)).prev_sample.numpy()
    pipe = DDPMPipeline(unet=model, scheduler=DDPMScheduler(num_train_timesteps=1000))
    pipe.set_progress_bar_config(disable=True)
    out['pipe6'] = pipe(batch_size=2, generator=torch.Generator().manual_seed(9), num_inference_steps=6, output_type='numpy').images
    np.savez(os.path.join(HERE, 'ddpm.npz'), **out)
    print('ddpm ok', {k: v.shape for k, v in out.items()})


def do_dropout():
    """Where dropout sits in the reference graph, with masks the build can reproduce: the reference UNet2DModel in train()
    mode after utils.set_dropout(model, 0.1) (utils.py:26-29: every nn.Dropout, i.e. ResnetBlock2D.dropout AND
    Attention.to_out[1]), each nn.Dropout.forward replaced by a multiplication with the Philox mask of oracle/philox_ref.py
    (logical index = channel-major element of the [N, C, H, W] activation; the attention dropout acts on [N, T, C] tokens, so it
    is transposed around the multiplication).  Records the finetune loss (ddpm_train.py:453-459) and gradient statistics."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import philox_ref
    cfg = gc.TINY_CFG
    model = build_ref_unet(cfg, 5)
    for m in model.modules():                       # utils.set_dropout(model, 0.1)
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.1
    model.train()
    for m in model.modules():
        if isinstance(m, Attention):
            m.set_processor(AttnProcessor())
    seed, step = 77, 3
    n_sites = 0
    for name, m in model.named_modules():
        if isinstance(m, torch.nn.Dropout):
            n_sites += 1
            spec = philox_ref.DropSpec({name: m.p}, seed, step, 0)
            if name.endswith('to_out.1'):
                m.forward = (lambda x, spec=spec, name=name:
                             spec.apply(name, x.transpose(1, 2).contiguous()).transpose(1, 2))
            else:
                m.forward = (lambda x, spec=spec, name=name: spec.apply(name, x))
    sched = DDPMScheduler(num_train_timesteps=1000)
    B = 4
    clean = torch.from_numpy(gc.det_clean((B, 3, 16, 16), 3))
    noise = torch.from_numpy(gc.det_noise((B, 3, 16, 16), 4))
    t = torch.tensor([1, 250, 500, 998])
    out = model(sched.add_noise(clean, noise, t), t).sample
    loss = (noise - out).square().sum(dim=(1, 2, 3)).mean(dim=0)
    loss.backward()
    json.dump(dict(p=0.1, seed=seed, step=step, sites=n_sites, loss=float(loss), fwd_out=gc.f32_to_b64(out.detach().numpy()),
                   grad_stats=grad_stats(model)), open(os.path.join(HERE, 'tiny_dropout.json'), 'w'))
    print('dropout ok: sites', n_sites, 'loss', float(loss))


def do_traced():
    """Row f2: the reference's DependencyGraph (autograd trace) and MagnitudePruner on the toy networks of toy_nets.py:
    group tables in visiting order, then a whole pruning pass (ratio 0.5, deterministic index scores) with its
    pruning history, the parameter shapes afterwards and the output shapes of the pruned network."""
    import toy_nets
    res = {}
    for name in toy_nets.NETS:
        model, inputs, ignored = toy_nets.build(name)
        names = name_of(model)
        pruner = tp.pruner.MagnitudePruner(model, inputs, importance=toy_nets.IndexScore(), iterative_steps=1,
                                           ch_sparsity=0.5, ignored_layers=ignored)
        table = [dict(ch_groups=int(pruner.get_channel_groups(g)), members=dump_group(g, names))
                 for g in pruner.DG.get_all_groups(ignored_layers=pruner.ignored_layers,
                                                   root_module_types=pruner.root_module_types)]
        pruned = []
        for g in pruner.step(interactive=True):
            pruned.append(dump_group(g, names))
            g.prune()
        with torch.no_grad():
            out = model(*inputs)
        outs = [list(o.shape) for o in (out if isinstance(out, tuple) else (out,))]
        hist = [[n, bool(o), sorted(int(i) for i in ix)] for n, o, ix in pruner.DG.pruning_history()]
        res[name] = dict(groups=table, pruned_groups=pruned, history=hist, out_shapes=outs,
                         shapes={k: list(v.shape) for k, v in model.state_dict().items()})
        print('traced', name, len(table), 'groups,', len(pruned), 'pruned, params', sum(p.numel() for p in model.parameters()))
    # Pin of the tracer itself, only possible here: THIS repository's tracer applied to the REFERENCE's Diffusers
    # UNet2DModel modules must enumerate exactly the groups the reference's DependencyGraph enumerates on them.
    import importlib
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    own_trace = importlib.import_module('diff-pruning_amd.trace')
    own_graph = importlib.import_module('diff-pruning_amd.graph')
    own_pruning = importlib.import_module('diff-pruning_amd.pruning')
    pinned = {}
    for cfgname, cfg, H in (('tiny', gc.TINY_CFG, 16), ('cifar', gc.CIFAR_CFG, 32)):
        model = build_ref_unet(cfg, 0)
        ref = [t['members'] for t in group_table(model, H)]
        tg = own_trace.TracedGraph(model, {'sample': torch.randn(1, 3, H, H), 'timestep': torch.ones((1,)).long()})
        n2m = dict(model.named_modules())
        chan = own_graph.ChannelView(lambda name: own_pruning._out_channels(n2m[name]))
        mine = [[[m.name, m.kind, compress(m.idxs)] for m in members]
                for _, members in own_graph.all_groups(tg, lambda: chan, ('conv_out',))]
        assert mine == ref, cfgname
        pinned[cfgname] = len(mine)
    res['_own_tracer_on_reference_unet2d_groups_equal'] = pinned
    print('own tracer on the reference UNet2DModel: groups equal', pinned)
    json.dump(res, open(os.path.join(HERE, 'traced_groups.json'), 'w'))


def do_script_loop():
    """Provenance of `sweep()` above: the accumulation loop of the ddpm_prune.py SCRIPT (the `if args.pruner in ['taylor',
    'diff-pruning']:` block, lines 94-106) is module-level code inside `if __name__ == '__main__'`, so the fixtures drive the
    reference model with the restatement `sweep()`.  Here the script's own source lines are read from the reference file,
    dedented and EXECUTED on the tiny UNet -- plain Taylor over all 1000 timesteps and Diff-Pruning with the fixture's threshold --
    and every accumulated gradient must be bit-identical to what `sweep()` accumulates (same stop step).  The verdict is stored
    in script_loop_check.json."""
    import textwrap
    src = open('/root/reference/ddpm_prune.py').read().splitlines()
    i1 = next(i for i in range(len(src)) if 'if loss<loss_max * args.thr: break' in src[i])
    i0 = max(i for i in range(i1) if src[i].strip().startswith("if args.pruner in ['taylor', 'diff-pruning']"))
    block = textwrap.dedent('\n'.join(src[i0:i1 + 1]))
    cfg = gc.TINY_CFG
    H = cfg['sample_size']
    sched = DDPMScheduler(num_train_timesteps=1000)
    clean = torch.from_numpy(gc.det_clean((2, 3, H, H), 1))
    noise = torch.from_numpy(gc.det_noise((2, 3, H, H), 2))
    out = dict(lines=[i0 + 1, i1 + 1])
    for pruner_name, thr in (('taylor', None), ('diff-pruning', 0.99)):
        ours = build_ref_unet(cfg, 5)
        losses = sweep(ours, sched, clean, noise, 1000, thr=thr)
        theirs = build_ref_unet(cfg, 5)
        theirs.zero_grad(); theirs.eval()
        ns = dict(args=types_ns(pruner=pruner_name, thr=thr, batch_size=2), tqdm=lambda it: it, torch=torch, clean_images=clean,
                  noise=noise, scheduler=sched, model=theirs, print=lambda *a: None)
        exec(compile(block, 'ddpm_prune.py[%d:%d]' % (i0 + 1, i1 + 1), 'exec'), ns)
        same = all(torch.equal(a.grad, b.grad) for a, b in zip(ours.parameters(), theirs.parameters()))
        assert same and ns['step_k'] + 1 == len(losses), (pruner_name, ns['step_k'], len(losses))
        out[pruner_name] = dict(steps=len(losses), gradients_bit_identical_to_sweep=bool(same))
    json.dump(out, open(os.path.join(HERE, 'script_loop_check.json'), 'w'))
    print('script loop ok:', out)


def do_train_loop(K=3):
    """The finetune loop of the ddpm_train.py SCRIPT (the body of `for step, batch in enumerate(train_dataloader):`, lines 427-475:
    noise / antithetic timestep draw, add_noise, zero_grad, forward, per-image summed loss, backward, clip, Adam step, LR-scheduler
    step, EMA step) EXECUTED from the reference file's own source lines for K batches, with the objects the script builds around
    it (ddpm_train.py:318-350): the reference UNet2DModel (tiny config, dropout 0) and DDPMScheduler, torch.optim.Adam with the
    argparse defaults and the lr of scripts/finetune_ddpm_cifar10.sh, diffusers' get_scheduler('cosine', warm-up 2 of 10 steps --
    so the lr changes every step), the vendored EMAModel, and a real accelerate.Accelerator(cpu=True) after prepare().
    torch.randn / torch.randint are replaced by counter-based replayable draws.  Recorded per step: loss, lr after the step, the
    timesteps; at the end every parameter and EMA tensor's |.| sum and three full tensors."""
    import textwrap
    import accelerate
    from diffusers.optimization import get_scheduler
    from diffusers.training_utils import EMAModel
    src = open('/root/reference/ddpm_train.py').read().splitlines()
    i0 = next(i for i, l in enumerate(src) if l.strip() == 'for step, batch in enumerate(train_dataloader):')
    i1 = next(i for i in range(i0, len(src)) if l_strip(src[i]) == 'logs["ema_decay"] = ema_model.cur_decay_value')
    block = textwrap.dedent('\n'.join(src[i0:i1 + 1]))
    cfg = gc.TINY_CFG
    H, B = cfg['sample_size'], 4
    model = build_ref_unet(cfg, 5)
    noise_scheduler = DDPMScheduler(num_train_timesteps=1000)
    args = types_ns(resume_from_checkpoint=None, gradient_accumulation_steps=1, use_ema=True, learning_rate=2e-4, adam_beta1=0.9,
                    adam_beta2=0.999, adam_weight_decay=0.0, adam_epsilon=1e-8, ema_max_decay=0.9999, ema_inv_gamma=1.0,
                    ema_power=0.75)
    ema_model = EMAModel(model.parameters(), decay=args.ema_max_decay, use_ema_warmup=False, inv_gamma=args.ema_inv_gamma,
                         power=args.ema_power, model_cls=UNet2DModel, model_config=model.config)
    optimizer = torch.optim.Adam(model.parameters(), lr=args.learning_rate, betas=(args.adam_beta1, args.adam_beta2),
                                 weight_decay=args.adam_weight_decay, eps=args.adam_epsilon)
    lr_scheduler = get_scheduler('cosine', optimizer=optimizer, num_warmup_steps=2, num_training_steps=10)
    batches = [torch.from_numpy(gc.det_clean((B, 3, H, H), 10 + k)) for k in range(K)]
    accelerator = accelerate.Accelerator(cpu=True)
    model, optimizer, lr_scheduler = accelerator.prepare(model, optimizer, lr_scheduler)
    calls, steps = [], []

    def det_randn(*size, **kw):
        shape = tuple(size[0]) if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else tuple(size)
        calls.append(('randn', shape))
        return torch.from_numpy(gc.det_noise(shape, 7000 + len(calls) - 1))

    def det_randint(low=0, high=None, size=None, **kw):
        calls.append(('randint', tuple(size)))
        n = size[0]
        return (torch.arange(n) * 389 + 173 * len(calls) + 7) % (high - low) + low

    class Bar:
        def update(self, n):
            steps.append(dict(loss=float(ns['loss'].detach()), lr_after=float(ns['lr_scheduler'].get_last_lr()[0]),
                              timesteps=[int(t) for t in ns['timesteps']], noise_draw=7000 + len(calls) - 2))

    ns = dict(train_dataloader=batches, model=model, args=args, epoch=0, first_epoch=0, resume_step=0, progress_bar=Bar(), torch=torch,
              noise_scheduler=noise_scheduler, accelerator=accelerator, optimizer=optimizer, lr_scheduler=lr_scheduler,
              ema_model=ema_model, global_step=0)
    keep = (torch.randn, torch.randint)
    torch.randn, torch.randint = det_randn, det_randint
    try:
        exec(compile(block, 'ddpm_train.py[%d:%d]' % (i0 + 1, i1 + 1), 'exec'), ns)
    finally:
        torch.randn, torch.randint = keep
    assert len(steps) == K and ns['global_step'] == K and model.training
    names = [n for n, _ in model.named_parameters()]
    full = ['conv_in.weight', 'mid_block.attentions.0.to_q.weight', 'conv_out.bias']
    P = dict(model.named_parameters())
    E = dict(zip(names, ema_model.shadow_params))
    json.dump(dict(lines=[i0 + 1, i1 + 1], batch=B, steps=steps, lr=args.learning_rate, ema_decay=args.ema_max_decay,
                   scheduler=dict(name='cosine', num_warmup_steps=2, num_training_steps=10),
                   param_abs_sum={n: float(P[n].detach().abs().sum()) for n in names},
                   ema_abs_sum={n: float(E[n].detach().abs().sum()) for n in names},
                   full={n: gc.f32_to_b64(P[n].detach().numpy()) for n in full},
                   full_ema={n: gc.f32_to_b64(E[n].detach().numpy()) for n in full}),
              open(os.path.join(HERE, 'train_loop.json'), 'w'))
    print('train loop ok: lines', i0 + 1, i1 + 1, [(round(s_['loss'], 4), s_['lr_after']) for s_ in steps])


def l_strip(line):
    return line.strip()


def types_ns(**kw):
    import types
    return types.SimpleNamespace(**kw)


def do_b2():
    """Boundary B2 (ddpm_prune.py:79-87): the REFERENCE's own pruner -- vendored `tp.pruner.MagnitudePruner`, whose
    DependencyGraph traces `model(**example_inputs)` through forward hooks and grad_fns (dependency.py:631-690) -- is handed THIS
    repository's drop-in `UNet2DModel` (on the CPU, where only its structure-only meta forward can run).  Recorded:
      * the group tables the reference's tracer enumerates on the product model == the tables it enumerates on its own
        Diffusers model (groups.json), member for member, for the CIFAR and bedroom topologies and the tiny UNet;
      * config C1 end to end: the reference's sweep on its own model supplies the gradients (same parameter names), the
        reference's pruner then scores, selects and SLICES the product model's holder modules: pruned index lists, parameter
        shapes and count equal cifar_c1.json.
    Written to b2_reference_pruner_on_product_model.json (booleans and counts only)."""
    import importlib
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    own_unet = importlib.import_module('diff-pruning_amd.unet')
    own_pruning = importlib.import_module('diff-pruning_amd.pruning')
    res = {}
    want = json.load(open(os.path.join(HERE, 'groups.json')))
    cfgb = dict(gc.BEDROOM_CFG, block_out_channels=[32, 32, 64, 64, 128, 128], sample_size=64)
    for key, cfg, H, ref_table in (('cifar', gc.CIFAR_CFG, 32, want['cifar']), ('bedroom_topology', cfgb, 64, want['bedroom_topology']),
                                   ('tiny', gc.TINY_CFG, 16, None)):
        own = own_unet.UNet2DModel(**cfg).eval()
        gc.det_init_(own, 0)
        table = group_table(own, H)                                 # the reference's tracer on the product model
        if ref_table is None:
            ref_table = group_table(build_ref_unet(cfg, 0), H)
        assert table == ref_table, key
        res[key] = dict(groups=len(table), equal_to_reference_model=True)
        print('b2', key, len(table), 'groups: reference tracer on the product model == on the reference model')
    # C1 end to end with the reference's pruner driving the product model
    cfg = gc.CIFAR_CFG
    ref = build_ref_unet(cfg, 0)
    sched = DDPMScheduler(num_train_timesteps=1000)
    clean = torch.from_numpy(gc.det_clean((4, 3, 32, 32), 1))
    noise = torch.from_numpy(gc.det_noise((4, 3, 32, 32), 2))
    sweep(ref, sched, clean, noise, 8)
    own = own_unet.UNet2DModel(**cfg).eval()
    gc.det_init_(own, 0)
    grads = {n: p.grad for n, p in ref.named_parameters()}
    for n, p in own.named_parameters():
        p.grad = grads[n].clone()
    rec = prune_run(own, 32, 0.3)
    own_pruning.fix_static_attributes(own)
    c1 = json.load(open(os.path.join(HERE, 'cifar_c1.json')))
    assert [r['root'] for r in rec] == [r['root'] for r in c1['prune']]
    assert [r['pruned'] for r in rec] == [r['pruned'] for r in c1['prune']]
    assert {n: list(p.shape) for n, p in own.named_parameters()} == c1['shapes_after']
    n_after = sum(p.numel() for p in own.parameters())
    assert n_after == c1['params_after'] == 19851157
    res['c1_reference_pruner_on_product_model'] = dict(pruned_groups=len(rec), masks_equal=True, shapes_equal=True,
                                                        params_after=int(n_after))
    print('b2 C1: the reference pruner pruned the product model to', n_after, 'parameters, masks equal')
    # ddpm_prune.py:89,118: `tp.utils.count_ops_and_params(model, example_inputs)` -- the reference's hook-based MAC counter --
    # on the product model (hooks fire in its shape-only pass; the sample itself comes from the engine, whose kernels are the
    # CPU stand-ins of tests/mock_ops.py in this GPU-less container) against the same call on the reference's own model,
    # before the prune and after it.
    sys.path.insert(0, os.path.dirname(HERE))
    import mock_ops
    own_engine = importlib.import_module('diff-pruning_amd.engine')
    own_engine.ops = mock_ops

    def _cpu_engine(self):
        if self._engine is None:
            self._engine = own_engine.UNetEngine(self.config)
        self._engine.packs.rebind()
        self._engine.bind({n: p.detach() for n, p in self.named_parameters()}, None)
        self._engine.set_dropout(None, 0, 0)
        return self._engine
    own_unet.UNet2DModel.engine = _cpu_engine
    ex = {'sample': torch.randn(1, 3, 32, 32), 'timestep': torch.ones((1,)).long()}
    fresh = own_unet.UNet2DModel(**cfg).eval()
    gc.det_init_(fresh, 0)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        base_own = tp.utils.count_ops_and_params(fresh, ex)
        after_own = tp.utils.count_ops_and_params(own, ex)
        out = fresh(**ex).sample
    base_ref = tp.utils.count_ops_and_params(build_ref_unet(cfg, 0), ex)
    assert tuple(out.shape) == (1, 3, 32, 32)
    assert tuple(base_own) == tuple(base_ref) == (c1['base_macs'], c1['base_params']), (base_own, base_ref)
    assert tuple(after_own) == (c1['macs_after'], c1['params_after']), after_own
    res['count_ops_and_params'] = dict(count_ops_equal=True, base_macs=float(base_own[0]), base_params=int(base_own[1]),
                                       macs_after=float(after_own[0]), params_after=int(after_own[1]))
    print('b2 count_ops_and_params (reference counter on the product model):', base_own, '->', after_own)
    json.dump(res, open(os.path.join(HERE, 'b2_reference_pruner_on_product_model.json'), 'w'))


if __name__ == '__main__':
    what = sys.argv[1:] or ['schedule', 'ddim', 'tiny', 'groups', 'c1', 'criteria', 'tiny_bedroom', 'optim', 'pretrained', 'groups_more', 'tiny_heads', 'long_sweep', 'lr', 'ddpm', 'dropout', 'fid']
    for w in what:
        globals()['do_' + w]()
