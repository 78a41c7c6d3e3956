"""Importance criteria, group pruner and channel-slicing functions with the `torch_pruning` call surface.

Mirrors the part of (vendored) torch_pruning the reference's prune scripts use:
  importance.TaylorImportance / MagnitudeImportance / RandomImportance   importance.py:11-16,59-126,221-225,332-434
  pruner.MagnitudePruner (= MetaPruner).step(interactive) / prune_local  pruner/algorithms/metapruner.py:11-254
  Group.prune()                                                          dependency.py:157-185
  function.prune_{conv,linear}_{out,in}_channels, prune_groupnorm_out_channels   pruner/function.py:85-146,168-207,274-302
Scores are computed on the device by the fused |w*g| channel-reduction kernel (csrc/importance.hip); the
argsort over the <= 1024 scores of a group and the channel slicing itself are host/plumbing work.

`TaylorImportance.__call__(group, ch_groups=1)` accepts both this module's groups and groups produced by a real
`torch_pruning.DependencyGraph` (items `(dep, idxs)` with `dep.target.module` / `dep.handler`), so it can be handed
to `tp.pruner.MagnitudePruner(importance=...)` unchanged (drop-in boundary B1 of SURVEY.md §8b).
"""
import abc
import os
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .graph import UNetGraph, LdmGraph, ChannelView, coupled_members, all_groups


# --------------------------------------------------------------------------------------------------------
# pruning functions (function.py:85-146, 168-207, 274-302): slice weights AND their accumulated grads
# --------------------------------------------------------------------------------------------------------
_index_cache = {}


def _index_tensor(key, build, device):
    """Small host->device index vectors recur across the members of one group (same channel count, same dropped
    set): build and upload each once.  The cache is bounded and only ever holds immutable index tensors."""
    k = (key, str(device))
    t = _index_cache.get(k)
    if t is None:
        if len(_index_cache) > 256:
            _index_cache.clear()
        t = torch.tensor(build(), dtype=torch.long, device=device)
        _index_cache[k] = t
    return t


def _keep(n, idxs, device):
    drop = frozenset(int(i) for i in idxs)
    return _index_tensor(('keep', n, drop), lambda: [i for i in range(n) if i not in drop], device)


def _slice_param(layer, attr, dim, keep):
    p = getattr(layer, attr)
    if p is None:
        return
    g = p.grad.data.index_select(dim, keep) if p.grad is not None else None
    newp = nn.Parameter(p.data.index_select(dim, keep))
    newp.grad = g
    setattr(layer, attr, newp)


def prune_conv_out_channels(layer, idxs):
    keep = _keep(layer.out_channels, idxs, layer.weight.device)
    layer.out_channels = layer.out_channels - len(set(idxs))
    _slice_param(layer, 'weight', 0, keep)
    _slice_param(layer, 'bias', 0, keep)
    return layer


def prune_conv_in_channels(layer, idxs):
    keep = _keep(layer.in_channels, idxs, layer.weight.device)
    layer.in_channels = layer.in_channels - len(set(idxs))
    _slice_param(layer, 'weight', 1, keep)
    return layer


def prune_linear_out_channels(layer, idxs):
    keep = _keep(layer.out_features, idxs, layer.weight.device)
    layer.out_features = layer.out_features - len(set(idxs))
    _slice_param(layer, 'weight', 0, keep)
    _slice_param(layer, 'bias', 0, keep)
    return layer


def prune_linear_in_channels(layer, idxs):
    keep = _keep(layer.in_features, idxs, layer.weight.device)
    layer.in_features = layer.in_features - len(set(idxs))
    _slice_param(layer, 'weight', 1, keep)
    return layer


def prune_groupnorm_out_channels(layer, idxs):
    keep = _keep(layer.num_channels, idxs, layer.weight.device)
    layer.num_channels = layer.num_channels - len(set(idxs))
    if layer.affine:
        _slice_param(layer, 'weight', 0, keep)
        _slice_param(layer, 'bias', 0, keep)
    return layer


prune_groupnorm_in_channels = prune_groupnorm_out_channels


def prune_layernorm_out_channels(layer, idxs):
    """function.py LayerNormPruner: slice the normalised (last) dimension of weight / bias."""
    n = layer.normalized_shape[-1]
    keep = _keep(n, idxs, layer.weight.device)
    layer.normalized_shape = tuple(layer.normalized_shape[:-1]) + (n - len(set(idxs)),)
    if layer.elementwise_affine:
        _slice_param(layer, 'weight', 0, keep)
        _slice_param(layer, 'bias', 0, keep)
    return layer

def prune_batchnorm_out_channels(layer, idxs):
    """function.py BatchnormPruner: slice the running statistics and (if affine) weight / bias."""
    keep = _keep(layer.num_features, idxs, layer.running_mean.device if layer.running_mean is not None else layer.weight.device)
    layer.num_features = layer.num_features - len(set(idxs))
    if layer.track_running_stats:
        layer.running_mean = layer.running_mean.data.index_select(0, keep)
        layer.running_var = layer.running_var.data.index_select(0, keep)
    if layer.affine:
        _slice_param(layer, 'weight', 0, keep)
        _slice_param(layer, 'bias', 0, keep)
    return layer


def prune_conv_transpose_out_channels(layer, idxs):
    """function.py ConvPruner, `transposed` branch: the weight is [Cin, Cout, k, k]."""
    keep = _keep(layer.out_channels, idxs, layer.weight.device)
    layer.out_channels = layer.out_channels - len(set(idxs))
    _slice_param(layer, 'weight', 1, keep)
    _slice_param(layer, 'bias', 0, keep)
    return layer


def prune_conv_transpose_in_channels(layer, idxs):
    keep = _keep(layer.in_channels, idxs, layer.weight.device)
    layer.in_channels = layer.in_channels - len(set(idxs))
    _slice_param(layer, 'weight', 0, keep)
    return layer


def prune_instancenorm_out_channels(layer, idxs):
    """function.py InstanceNormPruner (running statistics, when tracked, are sliced too so that the module stays usable)."""
    dev = layer.weight.device if layer.affine else (layer.running_mean.device if layer.running_mean is not None else 'cpu')
    keep = _keep(layer.num_features, idxs, dev)
    layer.num_features = layer.num_features - len(set(idxs))
    if layer.affine:
        _slice_param(layer, 'weight', 0, keep)
        _slice_param(layer, 'bias', 0, keep)
    if layer.running_mean is not None:
        layer.running_mean = layer.running_mean.data.index_select(0, keep)
        layer.running_var = layer.running_var.data.index_select(0, keep)
    return layer


def prune_prelu_out_channels(layer, idxs):
    """function.py PReLUPruner: a single shared slope is not touched."""
    if layer.num_parameters == 1:
        return layer
    keep = _keep(layer.num_parameters, idxs, layer.weight.device)
    layer.num_parameters = layer.num_parameters - len(set(idxs))
    _slice_param(layer, 'weight', 0, keep)
    return layer


def prune_embedding_out_channels(layer, idxs):
    """function.py EmbeddingPruner: the embedding dimension (columns of the table)."""
    keep = _keep(layer.embedding_dim, idxs, layer.weight.device)
    layer.embedding_dim = layer.embedding_dim - len(set(idxs))
    _slice_param(layer, 'weight', 1, keep)
    return layer


def prune_depthwise_conv_out_channels(layer, idxs):
    """function.py DepthwiseConvPruner: one filter per channel, so in_channels, out_channels and groups shrink together."""
    keep = _keep(layer.out_channels, idxs, layer.weight.device)
    n = layer.out_channels - len(set(idxs))
    layer.out_channels = layer.in_channels = layer.groups = n
    _slice_param(layer, 'weight', 0, keep)
    _slice_param(layer, 'bias', 0, keep)
    return layer


function = SimpleNamespace(
    prune_batchnorm_out_channels=prune_batchnorm_out_channels, prune_batchnorm_in_channels=prune_batchnorm_out_channels,
    prune_depthwise_conv_out_channels=prune_depthwise_conv_out_channels,
    prune_depthwise_conv_in_channels=prune_depthwise_conv_out_channels,
    prune_instancenorm_out_channels=prune_instancenorm_out_channels, prune_instancenorm_in_channels=prune_instancenorm_out_channels,
    prune_prelu_out_channels=prune_prelu_out_channels, prune_prelu_in_channels=prune_prelu_out_channels,
    prune_embedding_out_channels=prune_embedding_out_channels, prune_embedding_in_channels=prune_embedding_out_channels,
    prune_conv_out_channels=prune_conv_out_channels, prune_conv_in_channels=prune_conv_in_channels,
    prune_linear_out_channels=prune_linear_out_channels, prune_linear_in_channels=prune_linear_in_channels,
    prune_groupnorm_out_channels=prune_groupnorm_out_channels, prune_groupnorm_in_channels=prune_groupnorm_in_channels,
    prune_layernorm_out_channels=prune_layernorm_out_channels)


def _handler_for(layer, kind):
    if kind == 'ln':
        return prune_layernorm_out_channels
    if kind == 'gn':
        return prune_groupnorm_out_channels
    if kind == 'bn':
        return prune_batchnorm_out_channels
    if kind == 'inorm':
        return prune_instancenorm_out_channels
    if kind == 'prelu':
        return prune_prelu_out_channels
    if kind == 'embed':
        return prune_embedding_out_channels
    if isinstance(layer, nn.modules.conv._ConvNd) and layer.transposed:
        return prune_conv_transpose_out_channels if kind == 'out' else prune_conv_transpose_in_channels
    if isinstance(layer, nn.modules.conv._ConvNd) and layer.groups > 1:
        return prune_depthwise_conv_out_channels
    if isinstance(layer, nn.Linear):
        return prune_linear_out_channels if kind == 'out' else prune_linear_in_channels
    return prune_conv_out_channels if kind == 'out' else prune_conv_in_channels


def _out_channels(layer):
    if isinstance(layer, nn.Linear):
        return layer.out_features
    if isinstance(layer, nn.GroupNorm):
        return layer.num_channels
    if isinstance(layer, nn.LayerNorm):
        return layer.normalized_shape[-1]
    if isinstance(layer, (nn.modules.batchnorm._BatchNorm, nn.modules.instancenorm._InstanceNorm)):
        return layer.num_features
    if isinstance(layer, nn.PReLU):
        return layer.num_parameters if layer.num_parameters > 1 else None
    if isinstance(layer, nn.Embedding):
        return layer.embedding_dim
    return layer.out_channels


def _in_channels(layer):
    if isinstance(layer, nn.Linear):
        return layer.in_features
    if isinstance(layer, nn.GroupNorm):
        return layer.num_channels
    if isinstance(layer, nn.LayerNorm):
        return layer.normalized_shape[-1]
    if isinstance(layer, (nn.modules.batchnorm._BatchNorm, nn.modules.instancenorm._InstanceNorm)):
        return layer.num_features
    if isinstance(layer, nn.PReLU):
        return layer.num_parameters if layer.num_parameters > 1 else None
    if isinstance(layer, nn.Embedding):
        return layer.embedding_dim
    return layer.in_channels


# --------------------------------------------------------------------------------------------------------
# groups
# --------------------------------------------------------------------------------------------------------
class Dependency:
    """One member of a group: `handler(target.module, idxs)` prunes it (dependency.py:91-140 surface)."""
    __slots__ = ('target', 'handler', 'kind')

    def __init__(self, module, name, kind):
        self.target = SimpleNamespace(module=module, name=name)
        self.handler = _handler_for(module, kind)
        self.kind = kind

    def __call__(self, idxs):
        return self.handler(self.target.module, idxs)


class Group:
    """Iterates as (dep, idxs), root first (dependency.py:143-191)."""

    def __init__(self, items, dg=None, aux=()):
        self._items = items
        self._DG = dg
        self._aux = list(aux)       # traced graphs: (slice node, channels of it in this group), see graph.coupled_members

    def __iter__(self):
        return iter(self._items)

    def __len__(self):
        return len(self._items)

    def __getitem__(self, k):
        return self._items[k]

    def prune(self, idxs=None, record_history=True):
        """dependency.py:160-188: prune every member; the root goes into the graph's replayable pruning history as
        [root module name, is_out_channel_pruning, root indices]."""
        if idxs is not None:
            raise NotImplementedError('re-indexing an enumerated group: ask the graph for get_pruning_group(module, fn, idxs)')
        if not _prune_group_batched(self._items):
            for dep, ix in self._items:
                dep(ix)
        for node, n in self._aux:                  # the outputs of a torch.split shrink with the tensor that is split
            j, info = node.part
            info.sizes[j] -= n
        self._aux = []
        if record_history and self._DG is not None and self._items:
            root, ridx = self._items[0]
            self._DG._pruning_history.append([root.target.name, root.kind != 'in', [int(i) for i in ridx]])

    def details(self):
        return ['%s:%s(%d)' % (dep.kind, dep.target.name, len(idxs)) for dep, idxs in self._items]


PRUNE_BATCH = not os.environ.get('DP_NO_PRUNE_BATCH')
_BATCH_RULES = {}        # handler -> (weight dim, slices the bias too, attribute that holds the channel count)


def _batch_rules():
    if not _BATCH_RULES:
        _BATCH_RULES.update({
            prune_conv_out_channels: (0, True, 'out_channels'), prune_conv_in_channels: (1, False, 'in_channels'),
            prune_linear_out_channels: (0, True, 'out_features'), prune_linear_in_channels: (1, False, 'in_features'),
            prune_groupnorm_out_channels: (0, True, 'num_channels')})
    return _BATCH_RULES


def _prune_group_batched(items):
    """function.py:85-146,168-207,274-302 for ALL members of a group in one kernel launch (ops.slice_batch): every weight, bias
    and accumulated gradient keeps its un-pruned channels; the layers' channel attributes are updated as the per-member
    functions do.  Returns False (nothing touched) when a member is not a plain Conv2d / Linear / GroupNorm on the device, or
    when one tensor would be sliced twice in the group -- the caller then prunes member by member."""
    if not PRUNE_BATCH or not hasattr(ops, 'slice_batch') or not items:
        return False
    rules = _batch_rules()
    plan, seen = [], set()
    for dep, idxs in items:
        rule = rules.get(dep.handler)
        layer = dep.target.module
        if rule is None or getattr(layer, 'transposed', False) or getattr(layer, 'groups', 1) != 1:
            return False
        dim, with_bias, attr = rule
        if isinstance(layer, nn.GroupNorm) and not layer.affine:
            return False
        names = ['weight'] + (['bias'] if with_bias and getattr(layer, 'bias', None) is not None else [])
        for nm in names:
            if (id(layer), nm) in seen or (getattr(layer, nm).device.type != 'cuda' and not getattr(ops, 'IS_MOCK', False)):
                return False          # host-side structure edits (history replay on a CPU model) slice member by member
            seen.add((id(layer), nm))
        plan.append((layer, dim, names, attr, idxs))
    keep_host, slices, assign, off_next = [], [], [], 0
    for layer, dim, names, attr, idxs in plan:
        n = getattr(layer, attr)
        mask = np.ones(n, dtype=bool)
        mask[np.asarray(idxs, dtype=np.int64)] = False
        keep = np.nonzero(mask)[0]
        if not len(keep):
            return False
        off = off_next
        off_next += len(keep)
        keep_host.append(keep)
        for nm in names:
            p = getattr(layer, nm)
            d = dim if nm == 'weight' else 0
            shp = list(p.shape)
            R = shp[0]
            Cc = shp[1] if len(shp) > 1 else 1
            T = 1
            for v in shp[2:]:
                T *= v
            new_shape = list(shp)
            new_shape[d] = len(keep)
            outs = []
            for src in (p.data, p.grad.data if p.grad is not None else None):
                if src is None:
                    outs.append(None)
                    continue
                dst = torch.empty(new_shape, dtype=src.dtype, device=src.device)
                slices.append((src.contiguous(), dst, R, Cc, T, d, len(keep), off))
                outs.append(dst)
            assign.append((layer, nm, outs[0], outs[1]))
    dev = plan[0][0].weight.device
    keep_dev = torch.from_numpy(np.concatenate(keep_host)).to(dev)
    ops.slice_batch(slices, keep_dev)
    for layer, nm, w, g in assign:
        newp = nn.Parameter(w)
        newp.grad = g
        layer._parameters[nm] = newp            # what Module.__setattr__ does for an existing parameter, minus its checks
    for layer, dim, names, attr, idxs in plan:
        object.__setattr__(layer, attr, getattr(layer, attr) - len(set(idxs)))
    return True


def _member_kind(dep):
    """Resolve 'out' / 'in' / 'gn' for this module's Dependency objects and for real torch_pruning ones."""
    k = getattr(dep, 'kind', None)
    if k is not None:
        return k
    fn = dep.handler
    owner = type(getattr(fn, '__self__', None)).__name__
    name = getattr(fn, '__name__', '')
    if 'LayerNorm' in owner or 'layernorm' in name:
        return 'ln'
    if 'GroupNorm' in owner or 'groupnorm' in name:
        return 'gn'
    if owner in ('ConvPruner', 'LinearPruner') or 'conv' in name or 'linear' in name:
        if 'out_channels' in name:
            return 'out'
        if 'in_channels' in name:
            return 'in'
    return None


# --------------------------------------------------------------------------------------------------------
# importance criteria
# --------------------------------------------------------------------------------------------------------
class Importance(abc.ABC):
    @abc.abstractclassmethod
    def __call__(self, group):
        raise NotImplementedError


_MODES = {'sum_sq': 0, 'sum_abs': 1, 'abs_sum': 2}
_NO_TERM = ('ln', 'bn', 'inorm', 'prelu', 'embed')                     # member kinds the vendored criteria have no branch for
_OUT_KINDS = ('out', 'gn', 'ln', 'bn', 'inorm', 'prelu', 'embed')      # members pruned through an out-channel pruning function
_F_SQ, _F_ABS, _F_SIGNED, _F_GN_ABS, _F_GRAD_SQ, _F_SUM = 0, 1, 2, 3, 4, 5      # dp_wg_reduce modes (include/dp_hip.h)


class TaylorImportance(Importance):
    """First-order Taylor importance on the device.

    multivariable=None  -> the reference's vendored criterion (importance.py:375-434): conv/linear members
                           contribute sum((w*g)^2) over the other dims, GroupNorm members |w*g|, members whose length
                           differs from the root's are dropped, plain sum over members, no normalisation ('sum_sq').
    multivariable=True / False -> the pip torch_pruning criterion named by ddpm_prune.py:60,66: |sum w*g| / sum |w*g|
                           per member, mean over members, mean-normalised (that package is absent from the reference
                           tree and unpinned in requirements.txt:7: PARITY UNPINNED for these two modes).
    groupnorm_term      -> whether GroupNorm members add |w*g|.  The vendored criterion has that branch
                           (importance.py:412-418) -> default True for multivariable=None.  The pip criterion of that
                           era only had a BatchNorm branch (prune_batchnorm_out_channels), GroupNorm members matched no
                           branch and added nothing -> default False for the multivariable modes (recalled, unpinned).
    `sweep.prune_model` defaults to the vendored ('sum_sq') criterion: the only one pinned by reference outputs."""

    def __init__(self, group_reduction='mean', normalizer='mean', multivariable=None, groupnorm_term=None):
        self.group_reduction, self.normalizer, self.multivariable = group_reduction, normalizer, multivariable
        self.mode = 'sum_sq' if multivariable is None else ('abs_sum' if multivariable else 'sum_abs')
        self.groupnorm_term = (multivariable is None) if groupnorm_term is None else bool(groupnorm_term)
        self._scratch = None

    def _member_score(self, layer, kind, idxs, out_len):
        w = layer.weight.data
        g = layer.weight.grad
        if g is None:
            raise RuntimeError('TaylorImportance needs accumulated gradients (run the sweep before pruner.step())')
        g = g.data
        dev = w.device
        if kind == 'gn':
            full = torch.empty(w.shape[0], dtype=torch.float32, device=dev)
            ops.wg_reduce(w, g, 0, 3, full, False)
            n_full = w.shape[0]
        else:
            dim = 0 if kind == 'out' else 1
            if getattr(layer, 'transposed', False):        # ConvTranspose: weight [Cin, Cout, k, k] (importance.py:390-392,404-406)
                dim = 1 - dim
            n_full = w.shape[dim]
            full = torch.empty(n_full, dtype=torch.float32, device=dev)
            if dim == 1:
                need = w.numel() // w.shape[0]
                if self._scratch is None or self._scratch.numel() < need or self._scratch.device != dev:
                    self._scratch = torch.empty(max(need, 1 << 16), dtype=torch.float32, device=dev)
            ops.wg_reduce(w.contiguous(), g.contiguous(), dim, _MODES[self.mode], full, False, self._scratch)
        return full, n_full

    def _score_batched(self, terms, n0, dev):
        """All members of the group through ops.group_score (two launches): the same per-member reductions and the same
        member order as the loop below, hence the same bits.  Returns (score, members used) or None (not applicable)."""
        members, idx_host, need = [], [], 0
        for layer, kind, idxs in terms:
            if len(idxs) != n0:             # importance.py:422-426: mis-sized members are dropped
                continue
            w, g = layer.weight.data, layer.weight.grad
            if g is None:
                raise RuntimeError('TaylorImportance needs accumulated gradients (run the sweep before pruner.step())')
            g = g.data
            if not (w.is_contiguous() and g.is_contiguous()) or w.device != dev:
                return None
            if kind == 'gn':
                m = dict(R=w.shape[0], C=1, T=1, dim=0, mode=3)
                n_full = w.shape[0]
            else:
                dim = 0 if kind == 'out' else 1
                if getattr(layer, 'transposed', False):
                    dim = 1 - dim
                T = 1
                for v in w.shape[2:]:
                    T *= v
                m = dict(R=w.shape[0], C=w.shape[1] if w.dim() > 1 else 1, T=T, dim=dim, mode=_MODES[self.mode])
                n_full = w.shape[dim]
            m.update(w=w, g=g, full_off=0, col_off=0, idx_off=-1)
            if m['mode'] != 3 and m['dim'] == 1:
                m['col_off'] = need
                need += m['C'] * m['T']
            else:
                m['full_off'] = need
                need += n_full
            if not (n_full == n0 and idxs[0] == 0 and idxs[-1] == n0 - 1):
                m['idx_off'] = len(idx_host)
                idx_host.extend(idxs)
            members.append(m)
        if not members:
            return None
        if self._scratch is None or self._scratch.numel() < need or self._scratch.device != dev:
            self._scratch = torch.empty(max(need, 1 << 16), dtype=torch.float32, device=dev)
        idx_dev = torch.from_numpy(np.asarray(idx_host, dtype=np.int64)).to(dev) if idx_host else None
        score = torch.empty(n0, dtype=torch.float32, device=dev)
        ops.group_score(members, n0, idx_dev, self._scratch, score)
        return score, len(members)

    @torch.no_grad()
    def __call__(self, group, ch_groups=1):
        terms = []
        for dep, idxs in group:
            idxs.sort()
            kind = _member_kind(dep)
            if kind is None or kind in _NO_TERM:      # LayerNorm / BatchNorm / ... members carry no term (importance.py:383-418)
                continue
            layer = dep.target.module
            if kind == 'gn' and (not layer.affine or not self.groupnorm_term):
                continue
            terms.append((layer, kind, idxs))
        if not terms:
            return None
        n0 = len(terms[0][2])
        dev = terms[0][0].weight.device
        batched = self._score_batched(terms, n0, dev) if PRUNE_BATCH and hasattr(ops, 'group_score') else None      # None: member by member
        score = torch.zeros(n0, dtype=torch.float32, device=dev) if batched is None else batched[0]
        used = 0 if batched is None else batched[1]
        for layer, kind, idxs in (terms if batched is None else ()):
            if len(idxs) != n0:             # importance.py:422-426: mis-sized members are dropped
                continue
            full, n_full = self._member_score(layer, kind, idxs, n0)
            if n_full == n0 and idxs[0] == 0 and idxs[-1] == n0 - 1:
                ops.axpby(full, 1.0, score, 1.0)
            else:
                ops.gather_add(full, _index_tensor(('idx', tuple(idxs)), lambda: idxs, dev), score)
            used += 1
        if self.multivariable is not None:
            if self.group_reduction == 'mean':
                ops.axpby(score, 0.0, score, 1.0 / used)
            if self.normalizer == 'mean':
                mean = float(score.mean())
                ops.axpby(score, 0.0, score, 1.0 / mean)
        return score


class _GradCriterion(Importance):
    """Shared driver of the gradient criteria selectable in ddpm_exp/prune.py:193-208: every member contributes
    sum over its other dims of f(w, g) (one or two dp_wg_reduce passes), length-mismatched members are dropped, members
    are summed, optionally |.| at the end.  conv_modes / gn_modes: dp_wg_reduce modes accumulated per member."""
    conv_modes = (_F_SQ,)
    gn_modes = (_F_ABS,)
    final_abs = False

    def __init__(self, group_reduction='mean', normalizer='mean'):      # accepted and unused, as in the reference
        self.group_reduction, self.normalizer = group_reduction, normalizer
        self._scratch = None

    def _member_score(self, layer, kind):
        w, g = layer.weight.data, layer.weight.grad
        if g is None:
            raise RuntimeError('%s needs accumulated gradients (run the sweep before pruner.step())' % type(self).__name__)
        g = g.data
        if kind == 'gn':
            w, g, dim, modes = w.reshape(-1, 1), g.reshape(-1, 1), 0, self.gn_modes
        else:
            dim, modes = (0 if kind == 'out' else 1), self.conv_modes
            if getattr(layer, 'transposed', False):
                dim = 1 - dim
        n_full = w.shape[dim]
        full = torch.empty(n_full, dtype=torch.float32, device=w.device)
        if dim == 1:
            need = w.numel() // w.shape[0]
            if self._scratch is None or self._scratch.numel() < need or self._scratch.device != w.device:
                self._scratch = torch.empty(max(need, 1 << 16), dtype=torch.float32, device=w.device)
        for i, m in enumerate(modes):
            ops.wg_reduce(w.contiguous(), g.contiguous(), dim, m, full, i > 0, self._scratch)
        return full, n_full

    @torch.no_grad()
    def __call__(self, group, ch_groups=1):
        terms = []
        for dep, idxs in group:
            idxs.sort()
            kind = _member_kind(dep)
            if kind is None or kind in _NO_TERM:
                continue
            layer = dep.target.module
            if kind == 'gn' and (not layer.affine or not self.gn_modes):
                continue
            terms.append((layer, kind, idxs))
        if not terms:
            return None
        n0 = len(terms[0][2])
        dev = terms[0][0].weight.device
        score = torch.zeros(n0, dtype=torch.float32, device=dev)
        self._used = 0
        for layer, kind, idxs in terms:
            if len(idxs) != n0:
                continue
            full, n_full = self._member_score(layer, kind)
            if n_full == n0 and idxs[0] == 0 and idxs[-1] == n0 - 1:
                ops.axpby(full, 1.0, score, 1.0)
            else:
                ops.gather_add(full, _index_tensor(('idx', tuple(idxs)), lambda: idxs, dev), score)
            self._used += 1
        if self.final_abs:           # |score|: signed row "sum" over one column with the abs-after-sum mode
            out = torch.empty_like(score)
            ops.wg_reduce(score.view(-1, 1), torch.ones_like(score).view(-1, 1), 0, _F_SIGNED, out, False)
            score = out
        return score


class FullTaylorImportance(_GradCriterion):
    """importance.py:482-548: order 1: sum w*g; order 2: sum w*g + sum (w*g)^2; |.| after the members are summed."""
    final_abs = True

    def __init__(self, order=1, group_reduction='mean', normalizer='mean'):
        super().__init__(group_reduction, normalizer)
        if order not in (1, 2):
            raise ValueError('order must be 1 or 2')
        self.order = order
        self.conv_modes = (_F_SUM,) if order == 1 else (_F_SUM, _F_SQ)
        self.gn_modes = self.conv_modes


class AbsTaylorImportance(_GradCriterion):
    """importance.py:611-670: sum |w*g| per member, plain sum over members."""
    conv_modes = (_F_ABS,)
    gn_modes = (_F_ABS,)

    def __init__(self, order=1, group_reduction='mean', normalizer='mean'):
        super().__init__(group_reduction, normalizer)


class FisherImportance(_GradCriterion):
    """importance.py:715-781: conv / linear members sum g^2, GroupNorm members (w*g)^2."""
    conv_modes = (_F_GRAD_SQ,)
    gn_modes = (_F_SQ,)


class MagnitudeImportance(_GradCriterion):
    """importance.py:59-126: sum |w|^p per conv / linear member (GroupNorm members carry no term: the vendored class only
    matches BatchNorm), mean over members, divided by its mean.  p = 2 runs on the fused reduction with g := w."""
    conv_modes = (_F_ABS,)
    gn_modes = ()

    def __init__(self, p=2, group_reduction='mean', normalizer='mean'):
        super().__init__(group_reduction, normalizer)
        if p != 2:
            raise NotImplementedError('MagnitudeImportance: only p = 2 (the reference default) is built')
        self.p = p

    def _member_score(self, layer, kind):
        w = layer.weight.data.contiguous()
        dim = 0 if kind == 'out' else 1
        n_full = w.shape[dim]
        full = torch.empty(n_full, dtype=torch.float32, device=w.device)
        if dim == 1:
            need = w.numel() // w.shape[0]
            if self._scratch is None or self._scratch.numel() < need or self._scratch.device != w.device:
                self._scratch = torch.empty(max(need, 1 << 16), dtype=torch.float32, device=w.device)
        ops.wg_reduce(w, w, dim, _F_ABS, full, False, self._scratch)          # |w*w| = |w|^2
        return full, n_full

    @torch.no_grad()
    def __call__(self, group, ch_groups=1):
        score = super().__call__(group, ch_groups)
        if score is None:
            return None
        if self.group_reduction == 'mean':
            ops.axpby(score, 0.0, score, 1.0 / self._used)
        if self.normalizer == 'mean':
            ops.axpby(score, 0.0, score, 1.0 / float(score.mean()))
        return score


class RandomImportance(Importance):
    @torch.no_grad()
    def __call__(self, group, ch_groups=1):
        return torch.rand(len(group[0][1]))


importance = SimpleNamespace(Importance=Importance, TaylorImportance=TaylorImportance,
                             FullTaylorImportance=FullTaylorImportance, AbsTaylorImportance=AbsTaylorImportance,
                             FisherImportance=FisherImportance,
                             MagnitudeImportance=MagnitudeImportance, RandomImportance=RandomImportance)


# --------------------------------------------------------------------------------------------------------
# pruner
# --------------------------------------------------------------------------------------------------------
def linear_scheduler(ch_sparsity, steps):
    return [((i) / float(steps)) * ch_sparsity for i in range(steps + 1)]


class DependencyGraph:
    """Group enumeration (dependency.py:259-527).  For the two model families of this package the op graph is written
    down from the model configuration (graph.py); any other module is traced through autograd with `example_inputs`
    (trace.py), as `tp.DependencyGraph().build_dependency(model, example_inputs=...)` does."""

    def __init__(self, model=None, example_inputs=None, forward_fn=None, output_transform=None):
        self._pruning_history = []
        if model is not None:
            self.build_dependency(model, example_inputs, forward_fn, output_transform)

    def build_dependency(self, model, example_inputs=None, forward_fn=None, output_transform=None):
        """dependency.py:295-383.  Returns self."""
        self.model = model
        cfg = getattr(model, 'config', None)
        cfg = dict(cfg) if cfg is not None and hasattr(cfg, 'keys') else {}
        # The written-down graphs are for exactly two architectures (this package's classes or the reference's own of the same
        # name); anything else -- including other Diffusers models that share config keys -- is traced.
        cls = type(model).__name__
        if cls == 'UNetModel' and 'model_channels' in cfg:
            self.graph = LdmGraph(cfg)
        elif cls == 'UNet2DModel' and 'block_out_channels' in cfg and 'down_block_types' in cfg:
            self.graph = UNetGraph(cfg)
        else:
            if example_inputs is None:
                raise ValueError('example_inputs are needed to trace %s' % type(model).__name__)
            from .trace import TracedGraph
            self.graph = TracedGraph(model, example_inputs, forward_fn, output_transform)
        self.name2module = dict(model.named_modules())
        self.module2name = {m: n for n, m in self.name2module.items()}
        return self

    def pruning_history(self):
        """dependency.py:278-279: [[root module name, is_out_channel_pruning, idxs], ...] in application order."""
        return self._pruning_history

    def load_pruning_history(self, pruning_history):
        """dependency.py:281-293: replay a recorded history on this (un-pruned) model -- structure only, no gradients
        needed; afterwards the parameter shapes match the pruned checkpoint."""
        self._pruning_history = [[n, bool(o), [int(i) for i in ix]] for n, o, ix in pruning_history]
        for name, is_out, idxs in self._pruning_history:
            if not is_out:
                raise NotImplementedError('in-channel roots do not occur in histories written by the prune scripts')
            module = self.name2module[name]
            self.get_pruning_group(module, _handler_for(module, 'out'), idxs).prune(record_history=False)

    def _chan(self):
        n2m = self.name2module
        return ChannelView(lambda name: _out_channels(n2m[name]))

    def _group(self, members, aux=()):
        return Group([(Dependency(self.name2module[m.name], m.name, m.kind), list(m.idxs)) for m in members], self, aux)

    def get_pruning_group(self, module, pruning_fn, idxs):
        name = self.module2name[module]
        aux = []
        return self._group(coupled_members(self.graph, self._chan(), name, list(idxs), aux), aux)

    def get_all_groups(self, ignored_layers=(), root_module_types=(nn.Conv2d, nn.Linear)):
        ignored = tuple(self.module2name[m] for m in ignored_layers if m in self.module2name)
        for _, members in all_groups(self.graph, self._chan, ignored):
            yield self._group(members)

    def check_pruning_group(self, group):
        for dep, idxs in group:
            n = _out_channels(dep.target.module) if dep.kind in _OUT_KINDS else _in_channels(dep.target.module)
            if n is None:                        # a PReLU with one shared slope has no channel count
                continue
            if n <= len(idxs):
                return False
        return True

    get_out_channels = staticmethod(_out_channels)
    get_in_channels = staticmethod(_in_channels)


class MetaPruner:
    """metapruner.py:11-254 (local pruning path used by the reference; `global_pruning` is not on the hot path)."""

    def __init__(self, model, example_inputs=None, importance=None, global_pruning=False, ch_sparsity=0.5,
                 ch_sparsity_dict=None, max_ch_sparsity=1.0, iterative_steps=1,
                 iterative_sparsity_scheduler=linear_scheduler, ignored_layers=None, channel_groups=None, round_to=None,
                 root_module_types=(nn.Conv2d, nn.Linear)):
        if global_pruning:
            raise NotImplementedError('global pruning is not used by the reference prune scripts')
        self.model, self.importance = model, importance
        self.ch_sparsity, self.max_ch_sparsity = ch_sparsity, max_ch_sparsity
        self.round_to, self.root_module_types = round_to, root_module_types
        self.channel_groups = dict(channel_groups) if channel_groups else {}
        self.DG = DependencyGraph(model, example_inputs)
        self.ignored_layers = []
        for layer in (ignored_layers or []):
            self.ignored_layers.extend(list(layer.modules()))
        self.iterative_steps, self.current_step = iterative_steps, 0
        self.layer_init_out_ch, self.layer_init_in_ch = {}, {}
        for m in model.modules():
            if isinstance(m, (nn.modules.conv._ConvNd, nn.Linear, nn.GroupNorm, nn.LayerNorm, nn.modules.batchnorm._BatchNorm,
                              nn.modules.instancenorm._InstanceNorm, nn.PReLU, nn.Embedding)) and _out_channels(m) is not None:
                self.layer_init_out_ch[m] = _out_channels(m)
                self.layer_init_in_ch[m] = _in_channels(m)
        self.per_step_ch_sparsity = iterative_sparsity_scheduler(ch_sparsity, iterative_steps)
        self.ch_sparsity_dict = {}
        for module, sp in (ch_sparsity_dict or {}).items():
            for sub in module.modules():
                if isinstance(sub, (nn.Conv2d, nn.Linear, nn.GroupNorm)):
                    self.ch_sparsity_dict[sub] = iterative_sparsity_scheduler(sp, iterative_steps)
        for m in model.modules():                       # metapruner.py:118-124
            if isinstance(m, nn.GroupNorm):
                self.channel_groups[m] = m.num_groups
        self.records = []          # per pruned group: (root name, ch_groups, score tensor, pruned idxs) for reporting

    def get_target_sparsity(self, layer):
        return self.ch_sparsity_dict.get(layer, self.per_step_ch_sparsity)[self.current_step]

    def reset(self):
        self.current_step = 0

    def step(self, interactive=False):
        self.current_step += 1
        if interactive:
            return self.prune_local()
        for group in self.prune_local():
            group.prune()

    def pruning_history(self):
        return self.DG.pruning_history()

    def load_pruning_history(self, pruning_history):
        self.DG.load_pruning_history(pruning_history)

    def estimate_importance(self, group, ch_groups=1):
        return self.importance(group, ch_groups=ch_groups)

    def _check_sparsity(self, group):
        for dep, _ in group:
            m = dep.target.module
            if m not in self.layer_init_out_ch:
                continue                             # e.g. a PReLU with one shared slope
            if dep.kind in _OUT_KINDS:
                n = _out_channels(m)
                if n < self.layer_init_out_ch[m] * (1 - self.max_ch_sparsity) or n == 1:
                    return False
            else:
                n = _in_channels(m)
                if n < self.layer_init_in_ch[m] * (1 - self.max_ch_sparsity) or n == 1:
                    return False
        return True

    def get_channel_groups(self, group):
        if isinstance(self.channel_groups, int):
            return self.channel_groups
        for dep, _ in group:
            if dep.target.module in self.channel_groups:
                return self.channel_groups[dep.target.module]
        return 1

    def prune_local(self):
        if self.current_step > self.iterative_steps:
            return
        for group in self.DG.get_all_groups(self.ignored_layers, self.root_module_types):
            if not self._check_sparsity(group):
                continue
            module = group[0][0].target.module
            fn = group[0][0].handler
            ch_groups = self.get_channel_groups(group)
            imp = self.estimate_importance(group, ch_groups=ch_groups)
            if imp is None:
                continue
            cur = _out_channels(module)
            n_pruned = cur - int(self.layer_init_out_ch[module] * (1 - self.get_target_sparsity(module)))
            if self.round_to:
                n_pruned = n_pruned - (n_pruned % self.round_to)
            if n_pruned <= 0:
                continue
            imp_host = imp.detach().float().cpu()          # <= 1024 scores: argsort on the host (metapruner.py:237-249)
            if ch_groups > 1:
                gs = cur // ch_groups
                per = n_pruned // ch_groups
                order = torch.argsort(imp_host[:gs * ch_groups].view(ch_groups, gs), dim=1)[:, :per]
                idxs = (order + torch.arange(ch_groups).view(-1, 1) * gs).reshape(-1)
            else:
                idxs = torch.argsort(imp_host)[:(n_pruned // ch_groups)]
            idxs = idxs.tolist()
            pg = self.DG.get_pruning_group(module, fn, idxs)
            if self.DG.check_pruning_group(pg):
                self.records.append((group[0][0].target.name, ch_groups, imp_host, sorted(idxs)))
                yield pg


MagnitudePruner = MetaPruner
pruner = SimpleNamespace(MetaPruner=MetaPruner, MagnitudePruner=MagnitudePruner)
utils = None        # filled below (count_ops_and_params is defined after the pruner classes)


def fix_static_attributes(model):
    """ddpm_prune.py:111-116: Up/Downsample2D keep a `channels` attribute their forward asserts on."""
    for m in model.modules():
        if hasattr(m, 'channels') and hasattr(m, 'conv') and isinstance(m.conv, nn.Conv2d):
            m.channels = m.conv.in_channels


def count_ops_and_params(model, example_inputs=None):
    """`tp.utils.count_ops_and_params(model, example_inputs)` (ddpm_prune.py:89,118, ddpm_sample.py:51) for UNet2DModel:
    returns (MACs-style count per image, parameter count) with the vendored counter's conventions
    (utils/op_counter.py:53-101,248-298): Conv2d = k*k*Cin*Cout*positions + Cout*positions (bias), Linear =
    numel(input)*out + out (bias, once), GroupNorm = 2*numel(input); activations, interpolation and the attention
    matmuls are not counted.  The reference obtains the shapes with forward hooks; here they follow from the block
    structure (this model has no PyTorch forward to hook).  `example_inputs` only supplies the spatial size."""
    cfg = getattr(model, 'config', None)
    if cfg is None or 'block_out_channels' not in cfg:
        raise NotImplementedError('count_ops_and_params is implemented for UNet2DModel')
    H = W = cfg['sample_size'] if isinstance(cfg['sample_size'], int) else None
    if H is None:
        H, W = cfg['sample_size']
    if example_inputs is not None:
        x = example_inputs['sample'] if isinstance(example_inputs, dict) else example_inputs[0]
        H, W = int(x.shape[-2]), int(x.shape[-1])
    levels = len(cfg['block_out_channels'])

    def hw(level):
        return (H >> level) * (W >> level)

    def positions(name, module):
        p = name.split('.')
        if p[0] == 'down_blocks':
            lvl = int(p[1])
            return hw(lvl + 1) if p[2] == 'downsamplers' else hw(lvl)
        if p[0] == 'up_blocks':
            lvl = levels - 1 - int(p[1])
            return hw(lvl - 1) if p[2] == 'upsamplers' else hw(lvl)
        if p[0] == 'mid_block':
            return hw(levels - 1)
        return H * W                                            # conv_in, conv_norm_out, conv_out

    total = 0
    for name, m in model.named_modules():
        if isinstance(m, nn.Conv2d):
            pos = positions(name, m)
            total += m.kernel_size[0] * m.kernel_size[1] * m.in_channels * m.out_channels * pos
            if m.bias is not None:
                total += m.out_channels * pos
        elif isinstance(m, nn.Linear):
            tokens = positions(name, m) if '.attentions.' in name else 1      # q/k/v/out see [1, T, C]; the rest [1, C]
            total += tokens * m.in_features * m.out_features + (m.out_features if m.bias is not None else 0)
        elif isinstance(m, nn.GroupNorm):
            total += 2 * m.num_channels * positions(name, m)
    return float(total), count_params(model)


def count_params(model):
    return sum(p.numel() for p in model.parameters())


utils = SimpleNamespace(count_ops_and_params=count_ops_and_params, count_params=count_params)
