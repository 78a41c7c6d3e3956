import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
ops = importlib.import_module('diff-pruning_amd.ops')
import numpy as np
def rnd(*shape, seed=0, scale=1.0):
    g = np.random.default_rng(seed)
    t = torch.from_numpy((g.standard_normal(shape) * scale).astype(np.float32)).cuda()
    return t
ops.SPLITK_FOLD_MAX = 1 << 30
for (N, C1, C2, Cout, H, k) in [(12, 960, 0, 960, 8, 1), (6, 576, 0, 576, 16, 3)]:
    xa = rnd(N, C1, H, H, seed=1); xb = rnd(N, C2, H, H, seed=2) if C2 else None
    w = rnd(Cout, C1 + C2, k, k, seed=3, scale=0.02)
    b, tadd, res = rnd(Cout, seed=4), rnd(N, Cout, seed=6), rnd(N, Cout, H, H, seed=7)
    spec = ops.ConvSpec(k, 1, k // 2, 0)
    wp, ld = ops.pack_weight(w, 0); wd, ldd = ops.pack_weight(w, 1)
    dy = rnd(N, Cout, H, H, seed=5)
    seen = []
    real = ops._conv_ksplit
    ops._conv_ksplit = lambda p, d: (real(p, d), seen.append((p.ksplit, p.tile, bool(p.tile_counters))))[0]
    def run(kind):
        if kind == 0: return ops.conv_forward(xa, xb, wp, ld, Cout, spec).clone()
        if kind == 1: return ops.conv_forward(xa, xb, wp, ld, Cout, spec, bias=b).clone()
        if kind == 2: return ops.conv_forward(xa, xb, wp, ld, Cout, spec, bias=b, tadd=tadd, res=res, post_scale=0.7).clone()
        if kind == 3:
            acc = res.clone(); ops.conv_forward(xa, xb, wp, ld, Cout, spec, out=acc, accumulate=True); return acc
        return ops.conv_dgrad(dy, wd, ldd, C1 + C2, spec, (H, H), alpha=0.5).clone()
    for kind in (0, 4):
        ops.SPLITK_FOLD = False; r = run(kind)
        ops.SPLITK_FOLD = True
        tot = 0
        for rep in range(10):
            g = run(kind)
            bad = (g != r)
            nb = int(bad.sum()); tot += nb
            if nb and rep < 3:
                idx = bad.nonzero()[:40].tolist()
                HW = H * H
                print('   rep', rep, 'nbad', nb, [(n, m, h * H + w, 'pix', n * HW + h * H + w) for n, m, h, w in idx[::8]][:6],
                      'diffs', [float((g - r)[tuple(i)]) for i in idx[:3]], 'vals', [float(r[tuple(i)]) for i in idx[:3]])
        print(os.environ.get('DP_HIP_LIB', 'default')[-8:], (N, C1, C2, Cout, H, k), 'kind', kind, seen[-1], 'total bad over 10 reps', tot)
    ops._conv_ksplit = real
