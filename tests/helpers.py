"""Shared test helpers (reference-free)."""
import importlib
import json
import os

import numpy as np
import torch

import golden_common as gc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def pkg(sub=None):
    name = 'diff-pruning_amd' + ('.' + sub if sub else '')
    return importlib.import_module(name)


def load_json(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def load_npz(name):
    return np.load(os.path.join(GOLD, name))


def oracle_params(cfg, seed, requires_grad=True):
    from oracle import unet_ref
    shapes = unet_ref.param_shapes(cfg)
    return {n: torch.from_numpy(gc.det_param(n, s, seed)).requires_grad_(requires_grad) for n, s in shapes.items()}


def make_model(cfg, seed, device='cuda'):
    unet = pkg('unet')
    m = unet.UNet2DModel(**cfg)
    gc.det_init_(m, seed)
    return m.to(device).eval()


def model_shapes(model):
    return {n: tuple(p.shape) for n, p in model.named_parameters()}


def relerr(a, ref):
    a = torch.as_tensor(a).double().cpu()
    ref = torch.as_tensor(ref).double().cpu()
    return float((a - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def oracle_prune_replay(P, G, cfg, ratio, graph_mod, record=None, graph=None, ignored=('conv_out',), gn_groups=None,
                        round_to=None, mode='sum_sq', channel_groups=None):
    """Run the oracle's score / select / slice arithmetic over the PRODUCT's group enumeration (host logic that is
    itself pinned against the reference's group tables).  P/G: {name: tensor} dicts, modified in place.
    Returns a list of dict(root, ch_groups, score, pruned, margin)."""
    from oracle import pruning_ref as R
    graph = graph if graph is not None else graph_mod.UNetGraph(cfg)
    init_out = {n[:-7]: P[n].shape[0] for n in P if n.endswith('.weight')}
    out = []

    def chan():
        return graph_mod.ChannelView({n: tuple(t.shape) for n, t in P.items()})

    for root, members in graph_mod.all_groups(graph, chan, ignored):
        mem = [(m.name, m.kind, list(m.idxs)) for m in members]
        mem_s = [m for m in mem if m[1] != 'ln']
        score = R.magnitude_score(P, mem_s) if mode == 'magnitude' else R.taylor_score(P, G, mem_s, mode)
        if score is None:
            continue
        ch_groups = (gn_groups or cfg['norm_num_groups']) if any(k == 'gn' for _, k, _ in mem) else 1
        if ch_groups == 1 and channel_groups:          # metapruner.py get_channel_groups: the first member listed in channel_groups
            ch_groups = next((channel_groups[n] for n, _, _ in mem if n in channel_groups), 1)      # (prune_ldm.py:78-82: heads)
        cur = P[root + '.weight'].shape[0]
        pruned = R.select_pruned(score, cur, init_out[root], ratio, ch_groups, round_to)
        if not pruned:
            continue
        margin = R.decision_margin(score, pruned, cur, ch_groups)
        out.append(dict(root=root, ch_groups=ch_groups, score=score.clone(), pruned=pruned, margin=margin, cur=cur))
        for m in graph_mod.coupled_members(graph, chan(), root, pruned):
            R.slice_member(P, G, m.name, 'gn' if m.kind == 'ln' else m.kind, m.idxs)
    return out


class _FakePrunerBase:
    """Stand-ins shaped like torch_pruning's pruner singletons: tp.function.prune_conv_out_channels etc. are BOUND METHODS
    `ConvPruner().prune_out_channels` (function.py:535-566); importance criteria compare handlers by identity / owner."""

    def prune_out_channels(self, layer, idxs):
        raise AssertionError('importance must not prune')

    def prune_in_channels(self, layer, idxs):
        raise AssertionError('importance must not prune')


def tp_like_groups(model, graph_mod):
    ConvPruner = type('ConvPruner', (_FakePrunerBase,), {})
    LinearPruner = type('LinearPruner', (_FakePrunerBase,), {})
    GroupNormPruner = type('GroupNormPruner', (_FakePrunerBase,), {})
    boxes = {torch.nn.Conv2d: ConvPruner(), torch.nn.Linear: LinearPruner(), torch.nn.GroupNorm: GroupNormPruner()}
    mods = dict(model.named_modules())
    from types import SimpleNamespace
    graph = graph_mod.UNetGraph(model.config)
    view = graph_mod.ChannelView({n: tuple(p.shape) for n, p in model.named_parameters()})
    out = []
    for root, members in graph_mod.all_groups(graph, lambda: view, ('conv_out',)):
        items = []
        for m in members:
            mod = mods[m.name]
            box = boxes[type(mod)]
            handler = box.prune_in_channels if m.kind == 'in' else box.prune_out_channels
            # exactly the attributes importance.py:378-411 reads: dep.target.module, dep.handler, and the idxs list
            items.append((SimpleNamespace(target=SimpleNamespace(module=mod, name=m.name), handler=handler), list(m.idxs)))
        out.append((root, items))
    return out


def fid_features(dims, n1, n2, seed):
    """The seeded feature matrices of tests/golden/fid.json (make_golden.py do_fid)."""
    r = np.random.default_rng(seed)
    mix = r.standard_normal((dims, dims)) / np.sqrt(dims)
    a = np.maximum(r.standard_normal((n1, dims)) @ mix + 0.3, 0).astype(np.float32)
    b = np.maximum(r.standard_normal((n2, dims)) @ (mix * 1.1) + 0.35, 0).astype(np.float32)
    return a, b


def inception_state_dict(seed):
    """Seeded weights for the FID Inception (torchvision key names): He-scaled convolutions, BatchNorm statistics away from
    the identity so that the folding is exercised."""
    import importlib
    m = importlib.import_module('diff-pruning_amd.metrics')
    sd = {}
    for k, v in m.FIDInception3().state_dict().items():
        r = np.random.default_rng([zlib_crc(k), seed])
        if k.endswith('num_batches_tracked'):
            sd[k] = torch.tensor(1)
        elif k.endswith('conv.weight') or k == 'fc.weight':
            fan = int(np.prod(v.shape[1:]))
            sd[k] = torch.from_numpy((r.standard_normal(tuple(v.shape)) * np.sqrt(2.0 / fan)).astype(np.float32))
        elif k.endswith('running_var'):
            sd[k] = torch.from_numpy((0.5 + r.random(tuple(v.shape))).astype(np.float32))
        elif k.endswith('bn.weight'):
            sd[k] = torch.from_numpy((0.8 + 0.4 * r.random(tuple(v.shape))).astype(np.float32))
        else:
            sd[k] = torch.from_numpy((0.1 * r.standard_normal(tuple(v.shape))).astype(np.float32))
    return sd


def zlib_crc(s):
    import zlib
    return zlib.crc32(s.encode())


def write_lmdb(path, items, psize=4096, max_leaf_keys=None):
    """Write `items` ({bytes key: bytes value}) as an LMDB 0.9 environment file `path`/data.mdb following the on-disk structure
    definitions of mdb.c (the same ones diff-pruning_amd/lmdb_reader.py restates): two meta pages, a B+tree of leaf / branch pages
    in ascending key order, values larger than a quarter page on overflow page runs (F_BIGDATA).  Test infrastructure: liblmdb is
    not available here, so this is a second, independent walk over the format (writer vs reader), not liblmdb's own output."""
    import os
    import struct
    os.makedirs(path, exist_ok=True)
    keys = sorted(items)
    pages = {}                                      # pgno -> bytes
    next_pg = [2]

    def alloc(n=1):
        p = next_pg[0]
        next_pg[0] += n
        return p

    def page(pgno, flags, nodes):
        """nodes: list of encoded node bytes; placed from the page end downwards, offsets table after the header."""
        buf = bytearray(psize)
        upper = psize
        offs = []
        for nd in nodes:
            nd = nd + b'\0' * (len(nd) & 1)          # nodes are 2-byte aligned
            upper -= len(nd)
            buf[upper:upper + len(nd)] = nd
            offs.append(upper)
        lower = 16 + 2 * len(nodes)
        assert lower <= upper, 'page overflow in the test writer'
        struct.pack_into('<QHHHH', buf, 0, pgno, 0, flags, lower, upper)
        for i, o in enumerate(offs):
            struct.pack_into('<H', buf, 16 + 2 * i, o)
        pages[pgno] = bytes(buf)

    n_overflow = 0
    leaf_nodes = []
    for k in keys:
        v = items[k]
        if len(v) > psize // 4:                      # an overflow run: header on its first page, payload contiguous behind it
            npg = (16 + len(v) + psize - 1) // psize
            pg = alloc(npg)
            buf = bytearray(npg * psize)
            struct.pack_into('<QHHI', buf, 0, pg, 0, 0x04, npg)
            buf[16:16 + len(v)] = v
            for i in range(npg):
                pages[pg + i] = bytes(buf[i * psize:(i + 1) * psize])
            n_overflow += npg
            nd = struct.pack('<HHHH', len(v) & 0xFFFF, len(v) >> 16, 0x01, len(k)) + k + struct.pack('<Q', pg)
        else:
            nd = struct.pack('<HHHH', len(v) & 0xFFFF, len(v) >> 16, 0, len(k)) + k + v
        leaf_nodes.append((k, nd))
    # leaves
    level, cur, used = [], [], 16
    for k, nd in leaf_nodes:
        need = len(nd) + (len(nd) & 1) + 2
        if cur and (used + need > psize or (max_leaf_keys and len(cur) >= max_leaf_keys)):
            level.append(cur)
            cur, used = [], 16
        cur.append((k, nd))
        used += need
    if cur:
        level.append(cur)
    n_leaf, n_branch, depth = len(level), 0, 1 if level else 0
    children = []
    for grp in level:
        pg = alloc()
        page(pg, 0x02, [nd for _, nd in grp])
        children.append((grp[0][0], pg))
    while len(children) > 1:                          # branch levels: first node of a page has an empty key
        depth += 1
        nxt, cur, used = [], [], 16
        groups = []
        for k, pg in children:
            kk = b'' if not cur else k
            nd = struct.pack('<HHHH', pg & 0xFFFF, (pg >> 16) & 0xFFFF, (pg >> 32) & 0xFFFF, len(kk)) + kk
            need = len(nd) + (len(nd) & 1) + 2
            if cur and (used + need > psize or (max_leaf_keys and len(cur) >= max_leaf_keys)):
                groups.append(cur)
                cur, used = [], 16
                nd = struct.pack('<HHHH', pg & 0xFFFF, (pg >> 16) & 0xFFFF, (pg >> 32) & 0xFFFF, 0)
                need = len(nd) + 2
            cur.append((k, nd))
            used += need
        groups.append(cur)
        for grp in groups:
            pg = alloc()
            page(pg, 0x01, [nd for _, nd in grp])
            nxt.append((grp[0][0], pg))
            n_branch += 1
        children = nxt
    root = children[0][1] if children else (1 << 64) - 1
    last_pg = next_pg[0] - 1

    def meta(pgno, txnid, root_pg):
        buf = bytearray(psize)
        struct.pack_into('<QHHHH', buf, 0, pgno, 0, 0x08, 0, 0)
        struct.pack_into('<IIQQ', buf, 16, 0xBEEFC0DE, 1, 0, 1 << 30)
        struct.pack_into('<IHHQQQQQ', buf, 16 + 24, psize, 0, 0, 0, 0, 0, 0, (1 << 64) - 1)                     # free DB: pad = page size
        struct.pack_into('<IHHQQQQQ', buf, 16 + 24 + 48, 0, 0, depth if root_pg != (1 << 64) - 1 else 0, n_branch, n_leaf,
                         n_overflow, len(keys) if root_pg != (1 << 64) - 1 else 0, root_pg)
        struct.pack_into('<QQ', buf, 16 + 24 + 96, last_pg, txnid)
        return bytes(buf)

    with open(os.path.join(path, 'data.mdb'), 'wb') as f:
        f.write(meta(0, 1, (1 << 64) - 1))          # the older transaction: an empty tree (the reader must pick the newer meta)
        f.write(meta(1, 2, root))
        for pg in range(2, next_pg[0]):
            f.write(pages.get(pg, b'\0' * psize))

