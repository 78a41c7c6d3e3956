"""Channel-coupling graph of UNet2DModel and the enumeration of pruning groups.

Host logic (pure Python, no tensors).  It reproduces what the reference obtains from
`torch_pruning.DependencyGraph` (ddpm_exp/torch_pruning/dependency.py) for this model family:

  * which layers' channel dimensions are coupled (dependency.py:433-496: propagation through
    element-wise ops, GroupNorm, and `torch.cat` with its index offsets, _helpers.py:34-52), and
  * the ORDER in which root layers are visited (dependency.py:498-527 iterates `module2node`, whose
    insertion order is the stack-DFS over the autograd graph of dependency.py:761-806).  The order matters:
    pruning is interleaved with scoring (metapruner.py:205-254), so later groups are scored on tensors
    already sliced by earlier ones.

Instead of tracing autograd, the graph is built symbolically from the model configuration, listing for
every op its inputs in the order autograd's `next_functions` lists them for the reference forward
(unet_2d.py:219-316, resnet.py:589-639, attention_processor.py:870-935).  Pinned against group tables
recorded from the reference (tests/golden/groups.json, tiny_prune.json, cifar_c1.json).
"""


class GNode:
    __slots__ = ('kind', 'name', 'inputs', 'outputs', 'uid', 'part')

    def __init__(self, kind, name, inputs, uid):
        self.kind, self.name, self.uid = kind, name, uid
        self.inputs = [i for i in inputs if i is not None]
        self.outputs = []
        for i in self.inputs:
            if self not in i.outputs:
                i.outputs.append(self)

    def __repr__(self):
        return '%s(%s)' % (self.kind, self.name or self.uid)


class _GraphBase:
    def __init__(self, cfg):
        self.cfg = cfg
        self._n = 0
        self.layers = {}
        self.out = self._build()
        self.order = self._trace_order(self.out)

    # ---- builders --------------------------------------------------------------------------------
    def _node(self, kind, name, inputs):
        self._n += 1
        n = GNode(kind, name, inputs, self._n)
        if name is not None:
            self.layers[name] = n
        return n

    def ew(self, *inputs):
        return self._node('ew', None, list(inputs))

    def conv(self, name, x):
        return self._node('conv', name, [x])

    def gn(self, name, x):
        return self._node('gn', name, [x])

    def linear(self, name, x):
        # AddmmBackward0.next_functions = (bias [AccumulateGrad, skipped], input, TBackward0(weight))
        return self._node('linear', name, [x, self.ew()])

    def ln(self, name, x):
        return self._node('ln', name, [x])

    def slice(self, x, part, nparts):
        """One output of torch.chunk / split along channels (SplitBackward0): channels [part*D, (part+1)*D) of x."""
        n = self._node('slice', None, [x])
        n.part = (part, nparts)
        return n

    # ---- dependency.py:761-806 -------------------------------------------------------------------
    @staticmethod
    def _trace_order(root):
        order, seen_nodes = [], set()

        def create(n):
            if n.uid not in seen_nodes:
                seen_nodes.add(n.uid)
                order.append(n)

        stack, visited = [root], set()
        relinked = set()
        while stack:
            f = stack.pop()
            if f.uid in visited:
                continue
            create(f)
            for inp in f.inputs:
                create(inp)
                if inp.uid not in relinked:
                    relinked.add(inp.uid)
                    inp.outputs = []
                if f not in inp.outputs:             # Node.add_output at the time the CONSUMER is expanded
                    inp.outputs.append(f)            # (dependency.py:796-797): consumers in trace order
                stack.append(inp)
            visited.add(f.uid)
        return order


class UNetGraph(_GraphBase):
    """Symbolic op graph of UNet2DModel.forward for a given config."""

    def resnet(self, pre, x, emb, has_shortcut):
        h = self.ew(self.gn(pre + '.norm1', x))                 # silu(norm1(x))
        h = self.conv(pre + '.conv1', h)
        t = self.linear(pre + '.time_emb_proj', self.ew(emb))   # nonlinearity(temb) is re-evaluated per block
        t = self.ew(self.ew(t))                                 # [:, :, None, None]
        h = self.ew(h, t)                                       # hidden_states + temb
        h = self.ew(self.gn(pre + '.norm2', h))
        h = self.conv(pre + '.conv2', h)                        # dropout is the identity in eval mode
        if has_shortcut:
            x = self.conv(pre + '.conv_shortcut', x)
        return self.ew(self.ew(x, h))                           # (input_tensor + hidden_states) / scale

    def attention(self, pre, x):
        h = self.ew(self.ew(x))                                 # view, transpose
        h = self.ew(self.gn(pre + '.group_norm', self.ew(h)))   # transpose, group_norm, transpose
        q = self.ew(self.ew(self.linear(pre + '.to_q', h)))     # view + transpose
        k = self.ew(self.ew(self.linear(pre + '.to_k', h)))
        v = self.ew(self.ew(self.linear(pre + '.to_v', h)))
        a = self.ew(q, k, v)                                    # scaled_dot_product_attention(q, k, v)
        o = self.ew(self.ew(a))                                 # transpose, reshape
        o = self.linear(pre + '.to_out.0', o)
        o = self.ew(self.ew(o))                                 # transpose, reshape
        return self.ew(self.ew(o, x))                           # (hidden_states + residual) / rescale

    def _build(self):
        cfg = self.cfg
        boc = list(cfg['block_out_channels'])
        L = cfg['layers_per_block']
        nb = len(boc)
        emb = self.linear('time_embedding.linear_2', self.ew(self.linear('time_embedding.linear_1', None)))
        x = self.conv('conv_in', None)
        skips = [x]
        out_c = boc[0]
        for i, bt in enumerate(cfg['down_block_types']):
            in_c, out_c = out_c, boc[i]
            pre = 'down_blocks.%d' % i
            for j in range(L):
                x = self.resnet('%s.resnets.%d' % (pre, j), x, emb, (in_c if j == 0 else out_c) != out_c)
                if bt == 'AttnDownBlock2D':
                    x = self.attention('%s.attentions.%d' % (pre, j), x)
                skips.append(x)
            if i != nb - 1:
                if cfg['downsample_padding'] == 0:
                    x = self.ew(x)                              # F.pad
                x = self.conv(pre + '.downsamplers.0.conv', x)
                skips.append(x)
        x = self.resnet('mid_block.resnets.0', x, emb, False)
        if cfg.get('add_attention', True):
            x = self.attention('mid_block.attentions.0', x)
        x = self.resnet('mid_block.resnets.1', x, emb, False)
        for i, bt in enumerate(cfg['up_block_types']):
            pre = 'up_blocks.%d' % i
            for j in range(L + 1):
                x = self._node('cat', None, [x, skips.pop()])
                x = self.resnet('%s.resnets.%d' % (pre, j), x, emb, True)
                if bt == 'AttnUpBlock2D':
                    x = self.attention('%s.attentions.%d' % (pre, j), x)
            if i != nb - 1:
                x = self.conv(pre + '.upsamplers.0.conv', self.ew(x))      # F.interpolate
        x = self.ew(self.gn('conv_norm_out', x))
        return self.conv('conv_out', x)


class LdmGraph(_GraphBase):
    """Symbolic op graph of the CompVis UNetModel.forward (ldm_exp/ldm/modules/diffusionmodules/openaimodel.py:710-742,
    ResBlock._forward :236-275, attention.py:37-46,168-212,246-258) with autograd's next_functions input order."""

    def res(self, pre, x, emb, has_skip):
        h = self.conv(pre + '.in_layers.2', self.ew(self.gn(pre + '.in_layers.0', x)))
        e = self.linear(pre + '.emb_layers.1', self.ew(emb))
        e = self.ew(self.ew(e))                                   # emb_out[..., None] twice
        h = self.ew(h, e)                                         # h + emb_out
        h = self.conv(pre + '.out_layers.3', self.ew(self.gn(pre + '.out_layers.0', h)))
        if has_skip:
            x = self.conv(pre + '.skip_connection', x)
        return self.ew(x, h)                                      # skip_connection(x) + h

    def cross_attn(self, pre, x, ctx_is_x):
        q = self.ew(self.ew(self.linear(pre + '.to_q', x)))       # rearrange 'b n (h d) -> (b h) n d'
        k = self.ew(self.ew(self.linear(pre + '.to_k', x if ctx_is_x else None)))
        v = self.ew(self.ew(self.linear(pre + '.to_v', x if ctx_is_x else None)))
        sim = self.ew(self.ew(q, k))                              # einsum (bmm) then * scale
        attn = self.ew(sim)                                       # softmax
        out = self.ew(self.ew(self.ew(attn, v)))                  # einsum (bmm), rearrange back
        return self.linear(pre + '.to_out.0', out)                # Dropout is the identity

    def transformer(self, pre, x):
        h = self.gn(pre + '.norm', x)
        h = self.conv(pre + '.proj_in', h)
        h = self.ew(self.ew(h))                                   # rearrange 'b c h w -> b (h w) c'
        for d in range(self.cfg.get('transformer_depth', 1)):     # attention.py:253-254
            tb = pre + '.transformer_blocks.%d' % d
            h = self.ew(self.cross_attn(tb + '.attn1', self.ln(tb + '.norm1', h), True), h)
            h = self.ew(self.cross_attn(tb + '.attn2', self.ln(tb + '.norm2', h), False), h)
            p = self.linear(tb + '.ff.net.0.proj', self.ln(tb + '.norm3', h))
            gl = self.ew(self.slice(p, 0, 2), self.ew(self.slice(p, 1, 2)))      # x * gelu(gate)
            h = self.ew(self.linear(tb + '.ff.net.2', gl), h)
        h = self.ew(self.ew(h))                                   # rearrange back
        return self.ew(self.conv(pre + '.proj_out', h), x)        # x + x_in

    def _build(self):
        from .ldm import ldm_blocks
        cfg = self.cfg
        inp, out, mid = ldm_blocks(cfg)
        emb = self.linear('time_embed.2', self.ew(self.linear('time_embed.0', None)))
        hs = []
        h = None
        for bi, items in enumerate(inp):
            for li, it in enumerate(items):
                pre = 'input_blocks.%d.%d' % (bi, li)
                if it[0] == 'conv_in':
                    h = self.conv(pre, None)
                elif it[0] == 'res':
                    h = self.res(pre, h, emb, it[1] != it[2])
                elif it[0] == 'st':
                    h = self.transformer(pre, h)
                else:
                    h = self.conv(pre + '.op', h)
            hs.append(h)
        h = self.res('middle_block.0', h, emb, False)
        h = self.transformer('middle_block.1', h)
        h = self.res('middle_block.2', h, emb, False)
        for bi, items in enumerate(out):
            h = self._node('cat', None, [h, hs.pop()])
            for li, it in enumerate(items):
                pre = 'output_blocks.%d.%d' % (bi, li)
                if it[0] == 'res':
                    h = self.res(pre, h, emb, True)
                elif it[0] == 'st':
                    h = self.transformer(pre, h)
                else:
                    h = self.conv(pre + '.conv', self.ew(h))      # F.interpolate
        return self.conv('out.2', self.ew(self.gn('out.0', h)))


# ---------------------------------------------------------------------------------------------------
class ChannelView:
    """Current channel counts: either a {param name: shape} dict or a callable layer name -> out channels
    (read lazily off the live modules, so nothing is rebuilt while the caller prunes between groups)."""

    def __init__(self, shapes):
        if callable(shapes):
            self._out = shapes
        else:
            self._out = lambda name: shapes[name + '.weight'][0]

    def layer_out(self, name):
        return self._out(name)

    def out_channels(self, node):
        if node.kind in ('conv', 'linear', 'convT', 'gn', 'ln', 'bn', 'dw', 'inorm', 'embed'):
            return self._out(node.name)
        if node.kind == 'prelu':
            c = self._out(node.name)
            if c is not None:                      # per-channel slopes; a single shared slope says nothing about the width
                return c
        if node.kind == 'cat':
            return sum(self.out_channels(i) for i in node.inputs)
        if node.kind == 'const':
            return node.part
        if node.kind == 'slice':
            return slice_range(self, node)[1]
        if node.kind in ('flatten', 'unflatten'):
            c = self.out_channels(node.inputs[0])
            return None if c is None else (c * node.part if node.kind == 'flatten' else c // node.part)
        for i in node.inputs:           # element-wise: same as (any) input
            c = self.out_channels(i)
            if c is not None:
                return c
        return None


def slice_range(chan, node):
    """(first channel, channel count) of a `slice` node inside its input tensor.  Symbolic graphs: part = (index, number of
    equal parts).  Traced graphs: part = (index, trace.SplitInfo) with the live sizes of every output of that split."""
    j, parts = node.part
    if isinstance(parts, int):
        d = chan.out_channels(node.inputs[0]) // parts
        return j * d, d
    return parts.range(j)


class Member:
    __slots__ = ('name', 'kind', 'idxs')

    def __init__(self, name, kind, idxs):
        self.name, self.kind, self.idxs = name, kind, idxs

    def __iter__(self):          # unpack like the fixture triples
        return iter((self.name, self.kind, self.idxs))

    def __repr__(self):
        return 'Member(%s, %s, %d idxs)' % (self.name, self.kind, len(self.idxs))


def _fns(node):
    """(in-channel pruning fn, out-channel pruning fn) of a node.  Layers with a weight matrix have two different ones;
    everything else prunes 'its channels' with ONE function (function.py: prune_in_channels = prune_out_channels for the
    norms and the depthwise convolution, ops.py dummy pruners for element-wise / concat / split / reshape)."""
    return ('in', 'out') if node.kind in ('conv', 'linear', 'convT') else ('p', 'p')


def _to_input(chan, node, k, ii):
    """Indices `ii` on the output of `node`, expressed on the output of its k-th input."""
    if node.kind == 'cat':
        off = sum(chan.out_channels(i) for i in node.inputs[:k])
        n_in = chan.out_channels(node.inputs[k])
        return [i - off for i in ii if off <= i < off + n_in]
    if node.kind == 'slice':
        off = slice_range(chan, node)[0]
        return [i + off for i in ii]
    if node.kind == 'flatten':
        return sorted({i // node.part for i in ii})
    if node.kind == 'unflatten':
        return [i * node.part + k2 for i in ii for k2 in range(node.part)]
    return ii


def _to_output(chan, node, c, ii):
    """Indices `ii` on the output of `node`, expressed on the output (or, for a layer, the input channels) of consumer c."""
    if c.kind == 'cat':
        off = 0
        for inp in c.inputs:
            if inp is node:
                break
            off += chan.out_channels(inp)
        return [i + off for i in ii]
    if c.kind == 'slice':
        off, d = slice_range(chan, c)
        return [i - off for i in ii if off <= i < off + d]
    if c.kind == 'flatten':
        return [i * c.part + k for i in ii for k in range(c.part)]
    if c.kind == 'unflatten':
        return sorted({i // c.part for i in ii})
    return ii


def coupled_members(graph, chan, root_name, idxs, aux=None):
    """dependency.py:433-496.  Propagate the out-channel set `idxs` of layer `root_name` through the graph; returns an
    ordered list of Members (root first), index sets reaching the same (layer, pruning fn) merged
    (Group.add_and_merge, dependency.py:492-494).

    The order is the reference's: a (node, fn, idxs) operation is APPENDED when it is discovered and EXPANDED in LIFO
    order; expanding lists the node's dependencies inputs first, then outputs (dependency.py:603-627), of which a layer
    triggers only one side (out-channel pruning -> its consumers' in-channels; in-channel pruning -> its producers'
    out-channels) and every other node both.  A dependency is skipped when its index set is empty or when its target was
    already expanded and the identical operation is already in the group (dependency.py:476-479).  Members are summed in
    this order by the importance criteria, so the order is part of bit-exactness.

    Index maps (_helpers.py:17-79): cat <-> input k shifts by the input's channel offset; slice (one output of a
    split / chunk) <-> its source shifts by the slice's offset; flatten / unflatten map channel c <-> features
    [c*s, (c+1)*s).  BatchNorm ('bn') and depthwise convolutions ('dw', member kind 'out') pass indices through like
    GroupNorm.  `aux`, when given, receives (slice node, number of its channels in the group) for the traced splits whose
    sizes the caller must shrink after pruning."""
    root = graph.layers[root_name]
    root_fn = _fns(root)[1]
    sliced = {}

    # A slice is looked THROUGH, in both directions: the reference has one split node per torch.split, living in the
    # coordinates of the tensor that is split, whose dependencies lead straight to the consumers of its outputs; the
    # per-output slice nodes of this graph are finer than that and must not add a level to the LIFO expansion order.
    def out_deps(node, c, ii):
        jj = _to_output(chan, node, c, ii)
        if c.kind == 'slice':
            if not jj:
                return []
            sliced.setdefault(c.uid, (c, set()))[1].update(jj)
            return [d for c2 in c.outputs for d in out_deps(c, c2, jj)]
        return [(c, _fns(c)[0], jj)]

    def in_deps(node, k, inp, ii):
        jj = _to_input(chan, node, k, ii)
        if inp.kind == 'slice':
            if not jj:
                return []
            sliced.setdefault(inp.uid, (inp, set()))[1].update(jj)
            return in_deps(inp, 0, inp.inputs[0], jj)
        return [(inp, _fns(inp)[1], jj)]

    idxs = list(idxs)
    entries = [(root, root_fn, idxs)]
    # Element-wise edges hand the SAME list object on, so most of the operations of a big group (hundreds of dependencies x
    # hundreds of indices) are repeats: the hashable form of a list is computed once per object (the memo keeps the object alive,
    # so its id cannot be recycled), and a list that was already merged into a member is not merged again.
    frozen = {}

    def key_of(jj):
        hit = frozen.get(id(jj))
        if hit is None:
            hit = frozen[id(jj)] = (jj, tuple(jj))
        return hit[1]

    present = {(root.uid, root_fn): {key_of(idxs)}}
    visited = set()
    stack = [entries[0]]
    while stack:
        node, fn, ii = stack.pop()
        visited.add(node.uid)
        in_fn, out_fn = _fns(node)
        deps = []
        if fn == in_fn:
            for k, inp in enumerate(node.inputs):
                deps.extend(in_deps(node, k, inp, ii))
        if fn == out_fn:
            for c in node.outputs:
                deps.extend(out_deps(node, c, ii))
        for target, tfn, jj in deps:
            if not jj:
                continue
            key = (target.uid, tfn)
            tj = key_of(jj)
            have = present.setdefault(key, set())
            if target.uid in visited and tj in have:
                continue
            have.add(tj)
            e = (target, tfn, jj)
            entries.append(e)
            stack.append(e)
    merged, order = {}, []
    for node, fn, ii in entries:
        if node.name is None:
            continue                                   # pseudo nodes carry no member
        key = (node.uid, fn)
        if key not in merged:
            merged[key] = (node, fn, set(), set())
            order.append(key)
        if id(ii) not in merged[key][3]:
            merged[key][3].add(id(ii))
            merged[key][2].update(ii)
    out = []
    for key in order:
        node, fn, ii, _ = merged[key]
        kind = fn if node.kind in ('conv', 'linear', 'convT') else ('out' if node.kind == 'dw' else node.kind)
        out.append(Member(node.name, kind, sorted(ii)))
    if aux is not None:
        aux.extend((n, len(ii)) for n, ii in sliced.values() if not isinstance(n.part[1], int))
    return out


def all_groups(graph, chan_fn, ignored=('conv_out',)):
    """dependency.py:498-527 as a lazy generator: yields (root_name, members) for full out-channel sets.
    `chan_fn()` returns a fresh ChannelView (channel counts change while the caller prunes between yields)."""
    visited = set()
    for node in graph.order:
        if node.kind not in ('conv', 'linear', 'dw', 'convT'):
            continue
        if node.name in ignored or node.name in visited:
            continue
        chan = chan_fn()
        n_out = chan.layer_out(node.name)
        members = coupled_members(graph, chan, node.name, list(range(n_out)))
        prunable = True
        for m in members:
            if m.kind in ('out', 'gn', 'ln', 'bn', 'inorm', 'prelu', 'embed'):  # pruned through an out-channel pruning function
                visited.add(m.name)
                if m.name in ignored:
                    prunable = False
        if prunable:
            yield node.name, members
