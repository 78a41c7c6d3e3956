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
    the last dim of 2-D ones; for 3-D tensors the layout is read off the PRODUCING leaf modules -- Conv1d / BatchNorm1d /
    InstanceNorm1d / GroupNorm outputs are [N, C, L], Linear / LayerNorm / Embedding outputs are [B, T, C] -- and a 3-D cat or
    split whose producers cannot be identified raises); a token- or batch-wise cat is an element-wise node
    (dependency.py:690-705 special-cases one such ViT pattern by hand).  Every input of a cat keeps its slot -- also tensors
    autograd does not track (integer- or buffer-derived maps: `None` edges) and model inputs: their widths are recorded from
    the cat call itself during the traced forward, so later inputs keep their channel offsets;
  * the consumers of a split know WHICH output they read (the `input_nr` of the autograd edge), so every output is its
    own `slice` node with exact sizes (dependency.py:825-853 infers them from the consumers' in_channels);
  * flatten (N, C, *spatial) -> (N, C*prod(spatial)) in front of a Linear (any number of spatial dims) and its inverse are
    recognised from the view's own sizes (dependency.py:883-944 compares inferred channel counts and gives up on models
    with 3-D Linear outputs).
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
    """Channel dimension of a tensor of that rank when the rank alone decides it (3-D: see TracedGraph._feature_dim3)."""
    if rank >= 4:
        return 1
    if rank >= 2:
        return rank - 1
    return None


_CHANNEL_FIRST = ('conv', 'convT', 'dw', 'bn', 'inorm', 'gn', 'prelu')        # leaves whose 3-D outputs are [N, C, L]
_CHANNEL_LAST = ('linear', 'ln', 'embed')                                       # ... and [B, T, C]
_LAYOUT_CHANGING = ('view', 'reshape', 'transpose', 'permute', 'squeeze', 'expand', 'unsafeview', 'tbackward', 'unfold',
                    'movedim', 'select', 'sum', 'mean', 'bmm', 'mm', 'unbind', 'stack', 'repeat')


class _CatRecorder(torch.overrides.TorchFunctionMode):
    """Records the input shapes of every torch.cat of the traced forward, keyed by the result's autograd node (CatBackward
    saves only `dim`): the widths of inputs autograd does not track are needed for the channel offsets of the others."""

    def __init__(self):
        super().__init__()
        self.sizes = {}

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        out = func(*args, **kwargs)
        if func in (torch.cat, torch.concat, torch.concatenate) and isinstance(out, torch.Tensor) and out.grad_fn is not None:
            tensors = args[0] if args else kwargs.get('tensors')
            self.sizes[out.grad_fn] = [tuple(int(d) for d in t.shape) for t in tensors]
        return out


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
        self._fn2node, self._slice_nodes, self._views3 = {}, {}, []
        created = set()
        for o in outs:                                          # dependency.py:684-686: one walk per output tensor
            if o.grad_fn is None:
                continue
            self.out.append(self._walk(o.grad_fn, created))
        self._resolve_views3()
        if not self.out:
            raise RuntimeError('no model output carries a grad_fn: nothing to trace (are all parameters frozen?)')
        del self._fn2module, self._fn2node, self._slice_nodes, self._cat_sizes, outs   # drop the autograd graph (and the activations it saved)

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
            with torch.enable_grad(), _CatRecorder() as rec:
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
        self._cat_sizes = rec.sizes
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
            dim = _norm_dim(fn._saved_dim, len(oshape)) if len(oshape) else None
            fd = self._feature_dim_of(fn, len(oshape), dim, 'cat')
            return self._new('cat' if fd is not None and dim == fd else 'ew', None)
        if ('view' in name or 'reshape' in name) and oshape is not None and hasattr(fn, '_saved_self_sym_sizes'):
            ishape = tuple(int(s) for s in fn._saved_self_sym_sizes)
            # (N, C, *spatial) <-> (N, C * prod(spatial)): Conv1d / 2d / 3d features in front of (behind) a Linear
            if len(ishape) >= 3 and len(oshape) == 2 and ishape[0] == oshape[0]:
                sp = 1
                for d in ishape[2:]:
                    sp *= d
                if sp > 1 and oshape[1] == ishape[1] * sp:
                    lay = self._feature_dim_of(fn, 3, 1, 'flatten', required=False) if len(ishape) == 3 else 1
                    if lay is None:
                        return self._new('ew', None)          # no prunable producer in front of it: no channels to map
                    if lay != 1:
                        if ishape[1] == 1:
                            return self._new('ew', None)      # [B, 1, C] -> [B, C]: the one token's channels, index for index
                        raise NotImplementedError('flatten of a token-major [B, T, C] tensor: the channel index map of the '
                                                  'tracer assumes [N, C, *spatial]')
                    return self._new('flatten', None, sp)
            if len(ishape) == 2 and len(oshape) >= 3 and ishape[0] == oshape[0]:
                sp = 1
                for d in oshape[2:]:
                    sp *= d
                if sp > 1 and ishape[1] == oshape[1] * sp:
                    if len(oshape) == 3:
                        # [N, C, L] (Conv1d behind it) or [B, T, C] (tokens)?  The CONSUMERS decide, and they are only known
                        # once the walk is complete: _resolve_views3
                        node = self._new('ew', None)
                        self._views3.append((node, oshape))
                        return node
                    return self._new('unflatten', None, sp)
        return self._new('ew', None)

    def _resolve_views3(self):
        """Views [N, X] -> [N, A, B] with X == A * B met during the walk: behind a channel-first consumer (Conv1d, BatchNorm1d,
        ...) they are an `unflatten` with B positions per channel; in front of token-wise consumers (Linear, LayerNorm) they
        are element-wise when A == 1 and have no rule here otherwise."""
        for node, oshape in self._views3:
            found, seen, stack = set(), set(), list(node.outputs)
            while stack:
                c = stack.pop()
                if c.uid in seen:
                    continue
                seen.add(c.uid)
                if c.kind in _CHANNEL_FIRST:
                    found.add(1)
                elif c.kind in _CHANNEL_LAST:
                    found.add(2)
                elif c.kind == 'ew':
                    stack.extend(c.outputs)
            if found == {1}:
                if oshape[2] > 1:
                    node.kind, node.part = 'unflatten', oshape[2]
            elif found == {2} or not found:
                if oshape[1] != 1 and found:
                    raise NotImplementedError('view of a [B, T*C] matrix as [B, T, C] tokens: no channel index map for it')
            else:
                raise NotImplementedError('a 3-D view feeds both channel-first and token-wise layers: layout undecidable')
        del self._views3

    def _feature_dim_of(self, fn, rank, dim, what, required=True):
        """Channel dimension of the tensors entering autograd function `fn` (rank-3 tensors: decided by the leaf modules that
        produce them, found through shape-preserving element-wise ops).  `dim`: the dimension the op acts on -- batch-wise
        ops need no answer."""
        if rank != 3:
            return _feature_dim(rank)
        if dim == 0:
            return 2                                  # a batch-wise cat / split is element-wise under either layout
        found = set()
        seen = set()
        stack = [f for f, _ in getattr(fn, 'next_functions', ()) if f is not None]
        while stack:
            f = stack.pop()
            if f in seen:
                continue
            seen.add(f)
            module = self._fn2module.get(f)
            if module is not None:
                kind = _module_kind(module)
                found.add(1 if kind in _CHANNEL_FIRST else 2)
                continue
            fname = f.name().lower() if hasattr(f, 'name') else ''
            if 'accumulategrad' in fname or any(k in fname for k in _LAYOUT_CHANGING):
                continue                              # a leaf, or an op behind which the layout is no longer the producer's
            meta = getattr(f, '_input_metadata', None)
            if meta and len(_shape(meta[0])) != 3:
                continue
            stack.extend(g for g, _ in getattr(f, 'next_functions', ()) if g is not None)
        if not found and not required:
            return None
        if len(found) != 1:
            raise NotImplementedError('%s along dim %d of a 3-D tensor whose layout ([N, C, L] or [B, T, C]) cannot be read off '
                                      'its producers (%s)' % (what, dim, 'none found' if not found else 'conflicting'))
        return found.pop()

    def _split_info(self, fn):
        """SplitInfo when `fn` is a split / chunk along the feature dimension, else None."""
        name = fn.name().lower() if hasattr(fn, 'name') else ''
        if 'split' not in name or fn in self._fn2module or not hasattr(fn, '_saved_self_sym_sizes'):
            return None
        ishape = tuple(int(s) for s in fn._saved_self_sym_sizes)
        dim = _norm_dim(fn._saved_dim, len(ishape)) if len(ishape) else None
        fd = self._feature_dim_of(fn, len(ishape), dim, 'split')
        if fd is None or dim != fd:
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
            nexts = getattr(fn, 'next_functions', ())
            tracked = 0
            for slot, (nxt, in_nr) in enumerate(nexts):
                leaf = nxt is None or (hasattr(nxt, 'name') and 'accumulategrad' in nxt.name().lower())
                if leaf:
                    # `None`: a tensor autograd does not track (integer- / buffer-derived maps, no-grad tensors); AccumulateGrad:
                    # a weight / bias of a module or a model input.  Concatenated with features (cat([image, h], 1)) either
                    # one still shifts the channel offsets of the inputs after it: EVERY cat input keeps its slot, with the
                    # width the cat call itself was given.
                    if node.kind == 'cat':
                        sizes = self._cat_sizes.get(fn)
                        if sizes is None or len(sizes) != len(nexts):
                            raise NotImplementedError('a concatenation has an input autograd does not track and its input '
                                                      'sizes were not recorded (cat called through a path torch functions '
                                                      'do not see): the channel offsets of its inputs are unknown')
                        shp = sizes[slot]
                        if len(shp) == 1 and shp[0] == 0:
                            continue                            # legacy empty tensor: contributes no channels
                        const = self._new('const', None, int(shp[_norm_dim(fn._saved_dim, len(shp))]))
                        created.add(const.uid)
                        self.order.append(const)
                        node.inputs.append(const)
                        const.outputs.append(node)
                    continue
                tracked += 1
                inp = create(nxt, in_nr)
                if inp not in node.inputs:                      # Node.add_input(allow_dumplicated=False)
                    node.inputs.append(inp)
                if node not in inp.outputs:
                    inp.outputs.append(node)
                stack.append((nxt, in_nr))
            if tracked == 0 and node.kind == 'ew' and fn not in self._fn2module:
                # computed from model inputs / untracked tensors only (cat([cond * 2, h], 1)): a constant-width source
                meta = getattr(fn, '_input_metadata', None)
                shp = _shape(meta[nr]) if meta and len(meta) > nr else None
                if shp is not None and len(shp) >= 2:
                    fd = _feature_dim(len(shp)) if len(shp) != 3 else None
                    node.kind = 'const'
                    node.part = int(shp[fd]) if fd is not None else None
        return root
