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
