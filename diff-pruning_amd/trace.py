"""Generic group enumeration: build the channel-coupling graph of an ARBITRARY torch module by tracing autograd.

Row f2 of SURVEY.md section 8.  `graph.UNetGraph` / `graph.LdmGraph` write the op graph of the two model families down
from their configuration; this module obtains the same structure (GNode lists with autograd's input order) for any
`nn.Module` built from plain PyTorch ops, the way the reference's DependencyGraph does it
(ddpm_exp/torch_pruning/dependency.py:636-811): forward hooks on the prunable leaf modules record which autograd node
is the output of which module, then the autograd graph is walked from the model outputs with a stack-DFS whose node
creation order is the order `get_all_groups` later visits root layers in (dependency.py:498-527, 761-806).  The walk
produces `graph.GNode`s, so `graph.coupled_members` / `graph.all_groups` (propagation, merge rule, visiting order) are
shared with the symbolic graphs.

Host logic only (it runs wherever the traced module runs; tracing needs no gradients to flow, only `grad_fn`s).

Where the reference infers shapes from neighbouring layers, this tracer reads them off the autograd nodes
(`_input_metadata`, `_saved_self_sym_sizes`, `_saved_dim`), which removes three approximations of the reference:
  * torch.cat / split / chunk carry index offsets only when they act on the FEATURE dimension (dim 1 of >= 4-D tensors,
    the last dim of 2-D / 3-D ones); a token- or batch-wise cat is an element-wise node (dependency.py:690-705 special-cases
    one such ViT pattern by hand);
  * the consumers of a split know WHICH output they read (the `input_nr` of the autograd edge), so every output is its
    own `slice` node with exact sizes (dependency.py:825-853 infers them from the consumers' in_channels);
  * flatten (N,C,H,W) -> (N, C*H*W) in front of a Linear and its inverse are recognised from the view's own sizes
    (dependency.py:883-944 compares inferred channel counts and gives up on models with 3-D Linear outputs).
On graphs where the reference's inference is right, both give the same groups: tests/golden/traced_groups.json
(written by the reference's DependencyGraph on the toy networks of tests/helpers.py) pins that.

Supported prunable leaves: Conv1d/2d/3d (groups == 1, or depthwise), ConvTranspose (groups == 1), Linear, BatchNorm1d/2d/3d,
InstanceNorm, GroupNorm, LayerNorm, PReLU, Embedding.  Anything else that owns parameters (grouped convolutions, LSTM,
MultiheadAttention, parameters used outside a module) raises: silently treating them as element-wise would enumerate wrong
groups.
"""
import torch
from torch import nn

from .graph import GNode, _GraphBase

_NORM_KINDS = {nn.GroupNorm: 'gn', nn.LayerNorm: 'ln'}


def is_depthwise(m):
    return isinstance(m, nn.modules.conv._ConvNd) and m.groups > 1 and m.groups == m.in_channels == m.out_channels


def _module_kind(m):
    """Node kind of a prunable leaf module, or None."""
    if isinstance(m, nn.modules.conv._ConvNd):
        if m.transposed:
            return 'convT' if m.groups == 1 else None
        if m.groups == 1:
            return 'conv'
        return 'dw' if is_depthwise(m) else None
    if isinstance(m, nn.Linear):
        return 'linear'
    if isinstance(m, nn.modules.batchnorm._BatchNorm):
        return 'bn'
    if isinstance(m, nn.modules.instancenorm._InstanceNorm):
        return 'inorm'
    if isinstance(m, nn.PReLU):
        return 'prelu'
    if isinstance(m, nn.Embedding):
        return 'embed'
    for t, k in _NORM_KINDS.items():
        if isinstance(m, t):
            return k
    return None


def _flatten_outputs(out):
    """Tensors of a model output: tensor | tuple/list | dict | dataclass-like (Diffusers' `.sample`, `.to_tuple()`)."""
    if isinstance(out, torch.Tensor):
        return [out]
    if isinstance(out, (tuple, list)):
        return [t for o in out for t in _flatten_outputs(o)]
    if isinstance(out, dict):
        return [t for o in out.values() for t in _flatten_outputs(o)]
    if hasattr(out, 'to_tuple'):
        return _flatten_outputs(out.to_tuple())
    if hasattr(out, '__dict__'):
        return [t for o in vars(out).values() for t in _flatten_outputs(o)]
    return []


def _feature_dim(rank):
    if rank >= 4:
        return 1
    if rank >= 2:
        return rank - 1
    return None


def _norm_dim(d, rank):
    d = int(d)
    if d >= 1 << 63:                # negative dims are saved as wrapped uint64
        d -= 1 << 64
    return d % rank


def _shape(meta):
    return tuple(int(s) for s in meta.shape)


class SplitInfo:
    """Current channel counts of the outputs of one feature-dim split (mutable: pruning a group shrinks them)."""
    __slots__ = ('sizes',)

    def __init__(self, sizes):
        self.sizes = [int(s) for s in sizes]

    def range(self, j):
        off = sum(self.sizes[:j])
        return off, self.sizes[j]


class TracedGraph(_GraphBase):
    """Op graph of `model(*example_inputs)` with the interface of the symbolic graphs: `.layers` (module name -> node),
    `.order` (node creation order of the reference's trace), `.out` (root nodes)."""

    def __init__(self, model, example_inputs, forward_fn=None, output_transform=None):
        self.model = model
        self._n = 0
        self.layers = {}
        self.splits = []
        outs = self._run(model, example_inputs, forward_fn, output_transform)
        self.out = []
        self.order = []
        self._fn2node, self._slice_nodes = {}, {}
        created = set()
        for o in outs:                                          # dependency.py:684-686: one walk per output tensor
            if o.grad_fn is None:
                continue
            self.out.append(self._walk(o.grad_fn, created))
        if not self.out:
            raise RuntimeError('no model output carries a grad_fn: nothing to trace (are all parameters frozen?)')
        del self._fn2module, self._fn2node, self._slice_nodes, outs      # drop the autograd graph (and the activations it saved)

    # ---- forward pass with hooks (dependency.py:636-676) ---------------------------------------------
    def _run(self, model, example_inputs, forward_fn, output_transform):
        name_of = {m: n for n, m in model.named_modules()}
        self._fn2module, calls = {}, {}
        unsupported = []
        hooks = []

        def record(module, inputs, output):
            calls[module] = calls.get(module, 0) + 1
            if isinstance(output, (tuple, list)):
                output = output[0]
            self._fn2module[output.grad_fn] = module

        for m in model.modules():
            if _module_kind(m) is not None:
                hooks.append(m.register_forward_hook(record))
            elif any(True for _ in m.parameters(recurse=False)):
                unsupported.append('%s (%s)' % (name_of[m], type(m).__name__))
        if unsupported:
            raise NotImplementedError('modules with parameters the tracer has no pruning rule for: ' + ', '.join(unsupported))
        frozen = [p for p in model.parameters() if not p.requires_grad]
        for p in frozen:
            p.requires_grad_(True)
        tensors = example_inputs.values() if isinstance(example_inputs, dict) else (
            example_inputs if isinstance(example_inputs, (tuple, list)) else [example_inputs])
        for t in tensors:                                       # leaf inputs become visible to the walk (see _walk)
            if isinstance(t, torch.Tensor) and t.is_floating_point() and t.is_leaf and not t.requires_grad and t.dim() >= 2:
                t.requires_grad_(True)
                frozen.append(t)
        self._input_ids = {id(t) for t in tensors if isinstance(t, torch.Tensor)}
        was_training = model.training
        model.eval()                                            # dependency.py:639
        try:
            with torch.enable_grad():
                if forward_fn is not None:
                    out = forward_fn(model, example_inputs)
                elif isinstance(example_inputs, dict):
                    out = model(**example_inputs)
                elif isinstance(example_inputs, (tuple, list)):
                    out = model(*example_inputs)
                else:
                    out = model(example_inputs)
        finally:
            for h in hooks:
                h.remove()
            for p in frozen:
                p.requires_grad_(False)
            model.train(was_training)
        reused = [name_of[m] for m, c in calls.items() if c > 1]
        if reused:
            raise NotImplementedError('modules called more than once in one forward (shared layers): ' + ', '.join(reused))
        self._name_of = name_of
        if output_transform is not None:
            out = output_transform(out)
        return _flatten_outputs(out)

    # ---- autograd walk (dependency.py:707-811) --------------------------------------------------------
    def _new(self, kind, name, part=None):
        self._n += 1
        n = GNode(kind, name, [], self._n)
        n.part = part
        if name is not None:
            self.layers[name] = n
        return n

    def _classify(self, fn):
        """Node for the autograd function `fn` (all of its outputs, except feature-dim splits: see _node_of)."""
        module = self._fn2module.get(fn)
        if module is not None:
            return self._new(_module_kind(module), self._name_of[module])
        name = fn.name().lower() if hasattr(fn, 'name') else ''
        meta = getattr(fn, '_input_metadata', None)
        oshape = _shape(meta[0]) if meta else None
        if 'catbackward' in name and oshape is not None:
            fd = _feature_dim(len(oshape))
            dim = _norm_dim(fn._saved_dim, len(oshape)) if len(oshape) else None
            return self._new('cat' if fd is not None and dim == fd else 'ew', None)
        if ('view' in name or 'reshape' in name) and oshape is not None and hasattr(fn, '_saved_self_sym_sizes'):
            ishape = tuple(int(s) for s in fn._saved_self_sym_sizes)
            if len(ishape) == 4 and len(oshape) == 2 and oshape[1] == ishape[1] * ishape[2] * ishape[3] and ishape[2] * ishape[3] > 1:
                return self._new('flatten', None, ishape[2] * ishape[3])
            if len(ishape) == 2 and len(oshape) == 4 and ishape[1] == oshape[1] * oshape[2] * oshape[3] and oshape[2] * oshape[3] > 1:
                return self._new('unflatten', None, oshape[2] * oshape[3])
        return self._new('ew', None)

    def _split_info(self, fn):
        """SplitInfo when `fn` is a split / chunk along the feature dimension, else None."""
        name = fn.name().lower() if hasattr(fn, 'name') else ''
        if 'split' not in name or fn in self._fn2module or not hasattr(fn, '_saved_self_sym_sizes'):
            return None
        ishape = tuple(int(s) for s in fn._saved_self_sym_sizes)
        fd = _feature_dim(len(ishape))
        if fd is None or _norm_dim(fn._saved_dim, len(ishape)) != fd:
            return None
        return [_shape(m)[fd] for m in fn._input_metadata]

    def _node_of(self, fn, out_nr):
        """(node, is_new) for output `out_nr` of autograd function `fn`."""
        key = fn
        if fn not in self._fn2node:
            sizes = self._split_info(fn)
            if sizes is not None:
                info = SplitInfo(sizes)
                self.splits.append(info)
                self._fn2node[fn] = info
            else:
                self._fn2node[fn] = self._classify(fn)
                return self._fn2node[fn], True
        entry = self._fn2node[fn]
        if isinstance(entry, SplitInfo):
            key = (fn, out_nr)
            if key not in self._slice_nodes:
                self._slice_nodes[key] = self._new('slice', None, (out_nr, entry))
                return self._slice_nodes[key], True
            return self._slice_nodes[key], False
        return entry, False

    def _walk(self, root_fn, created):
        def create(fn, nr):
            node, _ = self._node_of(fn, nr)
            if node.uid not in created:
                created.add(node.uid)
                self.order.append(node)
            return node

        root = create(root_fn, 0)
        stack, visited = [(root_fn, 0)], set()
        while stack:
            fn, nr = stack.pop()
            node = create(fn, nr)
            if node.uid in visited:
                continue
            visited.add(node.uid)
            for nxt, in_nr in getattr(fn, 'next_functions', ()):
                if nxt is None:
                    continue
                if hasattr(nxt, 'name') and 'accumulategrad' in nxt.name().lower():
                    # a leaf: weight / bias of a module, or a model input.  A model input concatenated with features
                    # (cat([image, h], 1)) still shifts the channel offsets of the inputs after it: keep its width.
                    if node.kind == 'cat' and id(nxt.variable) in self._input_ids:
                        v = nxt.variable
                        const = self._new('const', None, int(v.shape[_feature_dim(v.dim())]))
                        created.add(const.uid)
                        self.order.append(const)
                        node.inputs.append(const)
                        const.outputs.append(node)
                    continue
                inp = create(nxt, in_nr)
                if inp not in node.inputs:                      # Node.add_input(allow_dumplicated=False)
                    node.inputs.append(inp)
                if node not in inp.outputs:
                    inp.outputs.append(node)
                stack.append((nxt, in_nr))
        return root
