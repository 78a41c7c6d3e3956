"""ORACLE (test infrastructure only) -- CPU restatement of the vendored torch_pruning arithmetic.

Parity pinned against tests/golden/tiny_prune.json and cifar_c1.json (per-group score vectors and
pruned index lists recorded from the reference's vendored ddpm_exp/torch_pruning).  The 'sum_abs' and
'abs_sum' modes restate the *un-vendored* pip torch_pruning TaylorImportance(multivariable=False/True)
that ddpm_prune.py:60,66 calls; that package is not under /root/reference and is unpinned
(requirements.txt:7) -> for those two modes: PARITY UNPINNED (checked only against this restatement).

The sibling criteria selectable in ddpm_exp/prune.py:193-208 ('full1', 'full2', 'abs', 'fisher', 'magnitude') are
pinned against tests/golden/tiny_criteria.json (scores + masks of a whole sequential prune per criterion).

Reference lines followed (relative to /root/reference/ddpm_exp/torch_pruning):
  importance.py:375-434                    TaylorImportance.__call__ (vendored: sum of (w*g)^2; GN |w*g|)
  importance.py:482-548                    FullTaylorImportance (order 1: sum w*g, order 2: + sum (w*g)^2; |sum over members|)
  importance.py:611-670                    AbsTaylorImportance (sum |w*g|)
  importance.py:715-781                    FisherImportance (sum g^2; GN (w*g)^2)
  importance.py:59-126                     MagnitudeImportance (sum |w|^p, no GN term, mean over members, / mean)
  pruner/algorithms/metapruner.py:196-254  prune_local: target count, GroupNorm sub-group selection
  pruner/function.py:85-146,168-207,274-302  conv / linear / groupnorm slicing (weights AND grads)

A *group* here is plain data: a list of members [param_prefix, kind, idxs] with kind in
{'out','in','gn'} -- the same shape as the fixtures written by make_golden.py.
"""
import torch


def member_terms(w, g, kind, idxs, mode='sum_sq'):
    """One member's per-channel contribution (importance.py:378-418)."""
    idxs = sorted(idxs)
    if kind == 'out':
        wg = (w[idxs].flatten(1) * g[idxs].flatten(1))
    elif kind == 'in':
        wg = (w.transpose(0, 1).flatten(1)[idxs] * g.transpose(0, 1).flatten(1)[idxs])
    elif kind == 'gn':
        p = w[idxs] * g[idxs]
        if mode == 'full1':
            return p
        if mode == 'full2':
            return p + p.pow(2)
        if mode == 'fisher':
            return p.pow(2)
        return p.abs()
    else:
        raise ValueError(kind)
    if mode == 'full1':
        return wg.sum(1)
    if mode == 'full2':
        return wg.sum(1) + wg.pow(2).sum(1)
    if mode == 'abs':
        return wg.abs().sum(1)
    if mode == 'fisher':
        gg = g[idxs].flatten(1) if kind == 'out' else g.transpose(0, 1).flatten(1)[idxs]
        return gg.pow(2).sum(1)
    if mode == 'sum_sq':
        return wg.abs().pow(2).sum(1)
    if mode == 'sum_abs':
        return wg.abs().sum(1)
    if mode == 'abs_sum':
        return wg.sum(1).abs()
    raise ValueError(mode)


@torch.no_grad()
def taylor_score(P, G, members, mode='sum_sq'):
    """importance.py:375-434.  P / G: {name: weight tensor} / {name: grad tensor}."""
    terms = []
    for pre, kind, idxs in members:
        w = P[pre + '.weight']
        g = G[pre + '.weight']
        terms.append(member_terms(w, g, kind, idxs, mode))
    if not terms:
        return None
    n0 = len(terms[0])
    aligned = [t for t in terms if len(t) == n0]
    tot = torch.stack(aligned, dim=0).sum(0)
    return tot.abs() if mode in ('full1', 'full2') else tot


@torch.no_grad()
def magnitude_score(P, members, p=2):
    """importance.py:59-126 (group_reduction='mean', normalizer='mean'): GroupNorm members contribute nothing (the
    vendored class only matches prune_batchnorm_out_channels)."""
    terms = []
    for pre, kind, idxs in members:
        idxs = sorted(idxs)
        w = P[pre + '.weight']
        if kind == 'out':
            terms.append(w[idxs].flatten(1).abs().pow(p).sum(1))
        elif kind == 'in':
            terms.append(w.transpose(0, 1).flatten(1).abs().pow(p).sum(1)[idxs])
    if not terms:
        return None
    n0 = len(terms[0])
    imp = torch.stack([t for t in terms if len(t) == n0], dim=0).mean(0)
    return imp / imp.mean()


def select_pruned(score, cur_out, init_out, ratio, ch_groups, round_to=None):
    """metapruner.py:225-249.  Returns the *set* of pruned root out-channel indices (sorted list)."""
    n_pruned = cur_out - int(init_out * (1 - ratio))
    if round_to:
        n_pruned = n_pruned - (n_pruned % round_to)
    if n_pruned <= 0:
        return None
    if ch_groups > 1:
        gs = cur_out // ch_groups
        per = n_pruned // ch_groups
        out = []
        for c in range(ch_groups):
            sub = score[c * gs:(c + 1) * gs]
            out.append(torch.argsort(sub)[:per] + c * gs)
        idx = torch.cat(out, 0)
    else:
        idx = torch.argsort(score)[:n_pruned]
    return sorted(int(i) for i in idx.tolist())


def decision_margin(score, pruned, cur_out, ch_groups):
    """Relative gap between the largest pruned and the smallest kept score (SURVEY.md §7 'hard parts'),
    minimum over sub-groups.  Reported next to every mask comparison."""
    pruned = set(pruned)
    worst = float('inf')
    gs = cur_out // ch_groups if ch_groups > 1 else cur_out
    for c in range(ch_groups if ch_groups > 1 else 1):
        rng = range(c * gs, (c + 1) * gs)
        pv = [float(score[i]) for i in rng if i in pruned]
        kv = [float(score[i]) for i in rng if i not in pruned]
        if pv and kv:
            hi, lo = max(pv), min(kv)
            worst = min(worst, (lo - hi) / max(abs(lo), 1e-30))
    return worst


@torch.no_grad()
def slice_member(P, G, pre, kind, idxs):
    """function.py:88-146 / 171-201 / 275-294: drop `idxs`, keep grads aligned."""
    w = P[pre + '.weight']
    dim_len = w.shape[1] if kind == 'in' else w.shape[0]
    keep = sorted(set(range(dim_len)) - set(idxs))
    kt = torch.tensor(keep, dtype=torch.long)
    if kind == 'in':
        P[pre + '.weight'] = w[:, kt].contiguous()
        if G.get(pre + '.weight') is not None:
            G[pre + '.weight'] = G[pre + '.weight'][:, kt].contiguous()
        return
    for suffix in ('.weight', '.bias'):
        k = pre + suffix
        if k in P and P[k] is not None:
            P[k] = P[k][kt].contiguous()
            if G.get(k) is not None:
                G[k] = G[k][kt].contiguous()
