"""Pruned-model checkpoints (reference: ddpm_prune.py:132-135, ddpm_train.py:289-300,484-498,
ddpm_exp/torch_pruning/dependency.py:278-293).

The reference hands a pruned UNet to the finetune / sampling scripts as a whole-module pickle (`torch.save(model,
'unet_pruned.pth')`): the module *is* the record of the pruned shapes.  Here the same hand-over works three ways:

* `torch.save(model, path)` / `torch.load(path)` -- the whole-module pickle of this package's UNet2DModel / UNetModel
  (the HIP engine, its packed operands and its streams are dropped from the pickle and rebuilt on first use);
* `save_pruned(model, directory)` / `load_pruned(directory)` -- pickle-free: weights as safetensors plus a JSON with the
  constructor config and the replayable `pruning_history` ([root module name, is_out_channel_pruning, indices] per
  group, exactly the reference's DependencyGraph.pruning_history() format); loading builds the un-pruned module, replays
  the history (structure only) and loads the weights strictly;
* `adopt_state_dict(model, state_dict)` -- shape-aware load of a pruned state dict that comes WITHOUT a history (e.g. the
  `state_dict()` of the reference's own `unet_pruned.pth`, whose keys are identical): every Conv2d / Linear / GroupNorm /
  LayerNorm takes the shapes found in the checkpoint, then the coupling graph is checked for consistency.

Host-side only: tensors are plain torch tensors on any device; nothing here touches the HIP kernels."""
import json
import os

import torch
import torch.nn as nn

from . import pruning

FORMAT_VERSION = 1
WEIGHTS, META = 'unet_pruned.safetensors', 'unet_pruned.json'


def _config_dict(model):
    cfg = dict(model.config)
    return {k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()}


def _build(kind, cfg):
    if kind == 'UNetModel':
        from .ldm import UNetModel
        return UNetModel(**cfg)
    from .unet import UNet2DModel
    return UNet2DModel(**cfg)


def save_pruned(model, directory, pruning_history):
    """Write `unet_pruned.safetensors` + `unet_pruned.json` into `directory`.  `pruning_history`: the list returned by
    `pruner.pruning_history()` (or `pruner.DG.pruning_history()`) after the groups were pruned."""
    from safetensors.torch import save_file
    os.makedirs(directory, exist_ok=True)
    sd = {k: v.detach().to('cpu').contiguous() for k, v in model.state_dict().items()}
    save_file(sd, os.path.join(directory, WEIGHTS))
    meta = dict(format_version=FORMAT_VERSION, model_class=type(model).__name__, config=_config_dict(model),
                pruning_history=[[n, bool(o), [int(i) for i in ix]] for n, o, ix in pruning_history],
                shapes={k: list(v.shape) for k, v in sd.items()}, num_parameters=int(sum(v.numel() for v in sd.values())))
    with open(os.path.join(directory, META), 'w') as f:
        json.dump(meta, f)
    return meta


def load_pruned(directory, device=None):
    """Rebuild the pruned module from `save_pruned` output: construct, replay the pruning history, load the weights."""
    from safetensors.torch import load_file
    with open(os.path.join(directory, META)) as f:
        meta = json.load(f)
    if meta.get('format_version') != FORMAT_VERSION:
        raise ValueError('unsupported pruned-checkpoint format %r' % (meta.get('format_version'),))
    model = _build(meta['model_class'], meta['config'])
    pruning.DependencyGraph(model).load_pruning_history(meta['pruning_history'])
    pruning.fix_static_attributes(model)
    sd = load_file(os.path.join(directory, WEIGHTS))
    got = {k: list(v.shape) for k, v in model.state_dict().items()}
    if got != meta['shapes']:
        bad = [k for k in meta['shapes'] if got.get(k) != meta['shapes'][k]][:5]
        raise ValueError('replayed pruning history does not reproduce the checkpoint shapes (first mismatches: %s)' % bad)
    model.load_state_dict(sd, strict=True)
    model.eval()
    return model.to(device) if device is not None else model


def adopt_state_dict(model, state_dict):
    """Give every prunable layer of the UN-pruned `model` the shapes found in `state_dict` (same keys), load the weights
    and verify that the result is a consistent pruned network (every coupled member agrees on its channel count)."""
    mods = dict(model.named_modules())
    for key, w in state_dict.items():
        name, _, attr = key.rpartition('.')
        m = mods.get(name)
        if m is None or attr not in ('weight', 'bias'):
            raise KeyError('unexpected key in pruned state dict: %s' % key)
        old = getattr(m, attr)
        if old is None or tuple(old.shape) == tuple(w.shape):
            continue
        setattr(m, attr, nn.Parameter(torch.empty(tuple(w.shape), dtype=old.dtype, device=old.device)))
    for m in mods.values():
        if isinstance(m, nn.Conv2d):
            m.out_channels, m.in_channels = m.weight.shape[0], m.weight.shape[1]
        elif isinstance(m, nn.Linear):
            m.out_features, m.in_features = m.weight.shape
        elif isinstance(m, nn.GroupNorm):
            m.num_channels = m.weight.shape[0]
        elif isinstance(m, nn.LayerNorm):
            m.normalized_shape = (m.weight.shape[0],)
    model.load_state_dict(state_dict, strict=True)
    pruning.fix_static_attributes(model)
    # consistency: walking every coupling group with all of its root's channels must tile each member dimension exactly
    # (a producer narrower or wider than its consumers -- or a concat whose parts do not add up -- leaves holes / overflow)
    dg = pruning.DependencyGraph(model)
    covered = {}
    for group in dg.get_all_groups(ignored_layers=()):
        for dep, idxs in group:
            covered.setdefault((dep.target.name, 'in' if dep.kind == 'in' else 'out'), set()).update(idxs)
    for (name, side), got in covered.items():
        m = dg.name2module[name]
        dim = pruning._in_channels(m) if side == 'in' else pruning._out_channels(m)
        if got != set(range(dim)):
            raise ValueError('inconsistent pruned shapes at %s (%s channels: %d, coupling graph covers %d)'
                             % (name, side, dim, len(got)))
    return model


# --------------------------------------------------------------------------------------------------------
# Original-DDPM ("ddpm_exp") checkpoints <-> Diffusers UNet2DModel keys
# --------------------------------------------------------------------------------------------------------
# The reference's second code path (ddpm_exp/prune.py, finetune_simple.py, runners/diffusion.py) works on the original
# DDPM `Model` class (ddpm_exp/models/diffusion.py:191-341), whose checkpoints (`ckpt.pth`, the `pretrained/` download)
# name the same tensors differently.  The key correspondence below restates the layout that class builds
# (models/diffusion.py:218-305) against unet_2d.py:84-217; the reference's own converter for it is
# tools/convert_ddpm_original_checkpoint_to_diffusers_cifar10.py:100-240.  Differences in content, not only in names:
# attention projections are 1x1 Conv2d there ([C, C, 1, 1]) and Linear here ([C, C]); `up.{i}` is indexed by resolution
# level (0 = full resolution), `up_blocks.{j}` by execution order (0 = lowest resolution).
_RES_O2D = (('norm1', 'norm1'), ('conv1', 'conv1'), ('temb_proj', 'time_emb_proj'), ('norm2', 'norm2'), ('conv2', 'conv2'),
            ('nin_shortcut', 'conv_shortcut'))
_ATT_O2D = (('norm', 'group_norm'), ('q', 'to_q'), ('k', 'to_k'), ('v', 'to_v'), ('proj_out', 'to_out.0'))


def ddpm_original_key_map(keys):
    """{original key: diffusers key} for the parameter names of a ddpm_exp `Model` state dict."""
    keys = list(keys)
    levels = 1 + max(int(k.split('.')[1]) for k in keys if k.startswith('down.'))
    fixed = {'temb.dense.0': 'time_embedding.linear_1', 'temb.dense.1': 'time_embedding.linear_2', 'conv_in': 'conv_in',
             'norm_out': 'conv_norm_out', 'conv_out': 'conv_out', 'mid.block_1': 'mid_block.resnets.0',
             'mid.block_2': 'mid_block.resnets.1', 'mid.attn_1': 'mid_block.attentions.0'}
    out = {}
    for k in keys:
        stem, _, leaf = k.rpartition('.')                      # leaf = weight | bias
        parts = stem.split('.')
        new = None
        if parts[0] in ('down', 'up'):
            i = int(parts[1])
            blk = ('down_blocks.%d' % i) if parts[0] == 'down' else ('up_blocks.%d' % (levels - 1 - i))
            if parts[2] == 'block':
                new = '%s.resnets.%s.%s' % (blk, parts[3], dict(_RES_O2D)[parts[4]])
            elif parts[2] == 'attn':
                new = '%s.attentions.%s.%s' % (blk, parts[3], dict(_ATT_O2D)[parts[4]])
            elif parts[2] == 'downsample':
                new = blk + '.downsamplers.0.conv'
            elif parts[2] == 'upsample':
                new = blk + '.upsamplers.0.conv'
        else:
            for old, rep in fixed.items():
                if stem == old:
                    new = rep
                elif stem.startswith(old + '.'):
                    sub = stem[len(old) + 1:]
                    new = rep + '.' + (dict(_ATT_O2D)[sub] if 'attn' in old else dict(_RES_O2D)[sub])
        if new is None:
            raise KeyError('not a ddpm_exp Model parameter: %s' % k)
        out[k] = new + '.' + leaf
    return out


def convert_ddpm_original(state_dict):
    """Original-DDPM `Model` state dict -> state dict with this package's / Diffusers' UNet2DModel keys."""
    kmap = ddpm_original_key_map(state_dict.keys())
    out = {}
    for k, v in state_dict.items():
        nk = kmap[k]
        if '.attentions.' in nk and not nk.rsplit('.', 2)[-2].startswith('group_norm') and v.dim() == 4:
            v = v.reshape(v.shape[0], v.shape[1])              # 1x1 Conv2d -> Linear
        out[nk] = v
    return out


def convert_to_ddpm_original(state_dict):
    """Inverse of convert_ddpm_original (to hand a pruned / finetuned model back to the ddpm_exp scripts)."""
    levels = 1 + max(int(k.split('.')[1]) for k in state_dict if k.startswith('down_blocks.'))
    res_d2o = {b: a for a, b in _RES_O2D}
    att_d2o = {b: a for a, b in _ATT_O2D}
    fixed = {'time_embedding.linear_1': 'temb.dense.0', 'time_embedding.linear_2': 'temb.dense.1', 'conv_in': 'conv_in',
             'conv_norm_out': 'norm_out', 'conv_out': 'conv_out'}
    out = {}
    for k, v in state_dict.items():
        stem, _, leaf = k.rpartition('.')
        p = stem.split('.')
        if stem in fixed:
            new = fixed[stem]
        elif p[0] == 'mid_block':
            new = ('mid.block_%d.%s' % (int(p[2]) + 1, res_d2o[p[3]])) if p[1] == 'resnets' else \
                  ('mid.attn_1.' + att_d2o['.'.join(p[3:])])
        elif p[0] in ('down_blocks', 'up_blocks'):
            i = int(p[1])
            blk = ('down.%d' % i) if p[0] == 'down_blocks' else ('up.%d' % (levels - 1 - i))
            if p[2] == 'resnets':
                new = '%s.block.%s.%s' % (blk, p[3], res_d2o[p[4]])
            elif p[2] == 'attentions':
                new = '%s.attn.%s.%s' % (blk, p[3], att_d2o['.'.join(p[4:])])
            elif p[2] == 'downsamplers':
                new = blk + '.downsample.conv'
            elif p[2] == 'upsamplers':
                new = blk + '.upsample.conv'
            else:
                raise KeyError(k)
        else:
            raise KeyError(k)
        if ('.attn' in new) and not new.endswith('.norm') and v.dim() == 2:
            v = v.reshape(v.shape[0], v.shape[1], 1, 1)        # Linear -> 1x1 Conv2d
        out[new + '.' + leaf] = v
    return out


def unet2d_config_from_ddpm_original(ch, ch_mult, num_res_blocks, attn_resolutions, image_size, in_channels=3, out_ch=3):
    """UNet2DModel kwargs equivalent to a ddpm_exp `Model` config (ddpm_exp/configs/*.yml: model.ch, ch_mult,
    num_res_blocks, attn_resolutions; data.image_size)."""
    res, down, up = image_size, [], []
    for lvl in range(len(ch_mult)):
        down.append('AttnDownBlock2D' if res in attn_resolutions else 'DownBlock2D')
        up.insert(0, 'AttnUpBlock2D' if res in attn_resolutions else 'UpBlock2D')
        if lvl != len(ch_mult) - 1:
            res //= 2
    return dict(sample_size=image_size, in_channels=in_channels, out_channels=out_ch, layers_per_block=num_res_blocks,
                block_out_channels=tuple(ch * m for m in ch_mult), down_block_types=tuple(down), up_block_types=tuple(up),
                norm_num_groups=32, norm_eps=1e-6, downsample_padding=0, flip_sin_to_cos=False, freq_shift=1,
                attention_head_dim=None, act_fn='silu')


# --------------------------------------------------------------------------------------------------------
# Diffusers pipeline directories (ddpm_prune.py:50 DDPMPipeline.from_pretrained, :131 pipeline.save_pretrained;
# ddpm_train.py:297-306,494-498; ddpm_sample.py:54-62)
#   <dir>/model_index.json                         {"_class_name": ..., "scheduler": ["diffusers", cls], "unet": [...]}
#   <dir>/unet/config.json                         constructor kwargs (+ "_class_name", "_diffusers_version")
#   <dir>/unet/diffusion_pytorch_model.bin|.safetensors
#   <dir>/scheduler/scheduler_config.json
# Layout pinned by tests/golden/pretrained_micro/ (written by the vendored diffusers 0.17.0.dev0).
# --------------------------------------------------------------------------------------------------------
DIFFUSERS_VERSION = '0.17.0.dev0'
_UNET_WEIGHTS = ('diffusion_pytorch_model.safetensors', 'diffusion_pytorch_model.bin')


def _read_json(path):
    with open(path) as f:
        return json.load(f)


def _write_json(path, obj):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, 'w') as f:
        json.dump(obj, f, indent=2, sort_keys=True)
        f.write('\n')


def load_unet(directory, subfolder=None):
    """UNet2DModel.from_pretrained: `directory` (or its `subfolder`) holds config.json + the weight file."""
    from .unet import UNet2DModel
    d = os.path.join(directory, subfolder) if subfolder else directory
    if not os.path.exists(os.path.join(d, 'config.json')) and os.path.exists(os.path.join(d, 'unet', 'config.json')):
        d = os.path.join(d, 'unet')
    cfg = {k: v for k, v in _read_json(os.path.join(d, 'config.json')).items() if not k.startswith('_')}
    model = UNet2DModel(**cfg)
    for name in _UNET_WEIGHTS:
        p = os.path.join(d, name)
        if os.path.exists(p):
            if name.endswith('.safetensors'):
                from safetensors.torch import load_file
                sd = load_file(p)
            else:
                sd = torch.load(p, map_location='cpu', weights_only=True)
            model.load_state_dict(sd, strict=True)
            return model.eval()
    raise FileNotFoundError('no %s in %s' % (' / '.join(_UNET_WEIGHTS), d))


def save_unet(model, directory, safe_serialization=False):
    """UNet2DModel.save_pretrained layout (the vendored diffusers writes the .bin pickle of the state dict by default)."""
    os.makedirs(directory, exist_ok=True)
    cfg = dict(_config_dict(model), _class_name='UNet2DModel', _diffusers_version=DIFFUSERS_VERSION)
    _write_json(os.path.join(directory, 'config.json'), cfg)
    sd = {k: v.detach().to('cpu').contiguous() for k, v in model.state_dict().items()}
    if safe_serialization:
        from safetensors.torch import save_file
        save_file(sd, os.path.join(directory, _UNET_WEIGHTS[0]))
    else:
        torch.save(sd, os.path.join(directory, _UNET_WEIGHTS[1]))


def load_scheduler(cls, directory, subfolder=None):
    d = os.path.join(directory, subfolder) if subfolder else directory
    if not os.path.exists(os.path.join(d, 'scheduler_config.json')) and os.path.exists(os.path.join(d, 'scheduler')):
        d = os.path.join(d, 'scheduler')
    cfg = {k: v for k, v in _read_json(os.path.join(d, 'scheduler_config.json')).items() if not k.startswith('_')}
    if cfg.get('thresholding') or cfg.get('trained_betas') is not None:
        raise NotImplementedError('thresholding / trained_betas schedulers are not on the hot path')
    import inspect
    accepted = set(inspect.signature(cls.__init__).parameters) - {'self'}
    return cls(**{k: v for k, v in cfg.items() if k in accepted})


def save_scheduler(scheduler, directory):
    cfg = {k: v for k, v in vars(scheduler.config).items()}
    cfg.update(_class_name=type(scheduler).__name__, _diffusers_version=DIFFUSERS_VERSION)
    _write_json(os.path.join(directory, 'scheduler_config.json'), cfg)


def load_pipeline(cls, directory):
    from . import diffusion
    index = _read_json(os.path.join(directory, 'model_index.json'))
    sched_name = index.get('scheduler', [None, 'DDPMScheduler'])[1]
    sched_cls = getattr(diffusion, sched_name, None)
    if sched_cls is None:
        raise NotImplementedError('scheduler class %s' % sched_name)
    return cls(unet=load_unet(directory, 'unet'), scheduler=load_scheduler(sched_cls, directory, 'scheduler'))


def save_pipeline(pipeline, directory, safe_serialization=False):
    _write_json(os.path.join(directory, 'model_index.json'),
                dict(_class_name=type(pipeline).__name__, _diffusers_version=DIFFUSERS_VERSION,
                     scheduler=['diffusers', type(pipeline.scheduler).__name__], unet=['diffusers', 'UNet2DModel']))
    save_unet(pipeline.unet, os.path.join(directory, 'unet'), safe_serialization)
    save_scheduler(pipeline.scheduler, os.path.join(directory, 'scheduler'))
