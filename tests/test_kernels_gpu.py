"""GPU parity tests of every HIP kernel against a float64 PyTorch reference of the same op.

Tolerances are stated per test: fp32 kernels vs an fp64 reference, so the bound is a small multiple of
fp32 rounding on the accumulated magnitude (1e-5 relative to the output scale unless noted)."""
import importlib
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    dp = importlib.import_module('diff-pruning_amd')
    from importlib import import_module
    o = import_module('diff-pruning_amd.ops')
    o._lib()
    return o


DEV = 'cuda'


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def relerr(a, ref):
    ref = ref.double().cpu()
    a = a.double().cpu()
    return float((a - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def ref_conv(x, w, b, stride, pad, ups, asym):
    x = x.double().cpu()
    if ups:
        x = F.interpolate(x, scale_factor=2.0, mode='nearest')
    if asym:
        x = F.pad(x, (0, 1, 0, 1))
    return F.conv2d(x, w.double().cpu(), None if b is None else b.double().cpu(), stride=stride, padding=pad)


CONV_CASES = [
    # name, N, C1, C2, Cout, H, k, stride, pad, ups
    ('3x3_small', 2, 8, 0, 16, 8, 3, 1, 1, 0),
    ('3x3_odd', 3, 37, 0, 70, 8, 3, 1, 1, 0),
    ('3x3_cat', 2, 40, 29, 130, 16, 3, 1, 1, 0),
    ('3x3_wide', 2, 128, 0, 256, 16, 3, 1, 1, 0),
    ('3x3_in3', 2, 3, 0, 128, 32, 3, 1, 1, 0),
    ('3x3_out3', 2, 128, 0, 3, 32, 3, 1, 1, 0),
    ('1x1_cat', 2, 64, 31, 96, 8, 1, 1, 0, 0),
    ('1x1_4x4', 16, 179, 0, 192, 4, 1, 1, 0, 0),
    ('s2_asym', 2, 24, 0, 40, 16, 3, 2, 0, 0),
    ('s2_sym', 2, 24, 0, 40, 16, 3, 2, 1, 0),
    ('ups', 2, 24, 0, 40, 8, 3, 1, 1, 1),
    ('linear', 7, 100, 0, 90, 1, 1, 1, 0, 0),
    ('in4_s2_asym', 3, 4, 0, 40, 16, 3, 2, 0, 0),
    ('in3_ups', 2, 3, 0, 40, 8, 3, 1, 1, 1),
    ('out7_odd', 3, 50, 0, 7, 12, 3, 1, 1, 0),
    ('fast_sq', 3, 128, 0, 128, 8, 3, 1, 1, 0),
    ('fast_cat', 2, 64, 32, 144, 16, 3, 1, 1, 0),
    ('fast_1x1_cat', 5, 32, 48, 96, 12, 1, 1, 0, 0),
    ('fast_tail', 3, 80, 0, 200, 10, 3, 1, 1, 0),
    ('fast_cat128', 2, 128, 64, 96, 16, 3, 1, 1, 0),
    ('fast_4x4', 20, 64, 0, 80, 4, 3, 1, 1, 0),
    ('fast_w2', 9, 32, 0, 70, 2, 3, 1, 1, 0),
    ('t96_pruned', 3, 90, 0, 90, 16, 3, 1, 1, 0),
    ('t96_cat', 2, 180, 90, 180, 8, 3, 1, 1, 0),
    ('t96_1x1', 4, 359, 0, 180, 8, 1, 1, 0, 0),
    ('t96_cat_odd', 3, 181, 91, 180, 16, 3, 1, 1, 0),
    ('nt_straddle', 2, 130, 70, 256, 8, 3, 1, 1, 0),
]


@pytest.mark.parametrize('case', CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_forward_dgrad_wgrad(ops, report, case):
    name, N, C1, C2, Cout, H, k, stride, pad, ups = case
    Cin = C1 + C2
    xa = rnd(N, C1, H, H, seed=1)
    xb = rnd(N, C2, H, H, seed=2) if C2 else None
    w = rnd(Cout, Cin, k, k, seed=3, scale=1.0 / math.sqrt(Cin * k * k))
    b = rnd(Cout, seed=4)
    spec = ops.ConvSpec(k, stride, pad, ups)
    asym = (stride == 2 and pad == 0)
    xcat = xa if xb is None else torch.cat([xa, xb], 1)
    ref = ref_conv(xcat, w, b, stride, pad, ups, asym)
    wp, ld = ops.pack_weight(w, 0)
    y = ops.conv_forward(xa, xb, wp, ld, Cout, spec, bias=b)
    e_f = relerr(y, ref)
    # epilogue: tadd + res + post_scale
    tadd = rnd(N, Cout, seed=5)
    res = rnd(*y.shape, seed=6)
    y2 = ops.conv_forward(xa, xb, wp, ld, Cout, spec, bias=b, tadd=tadd, res=res, post_scale=0.5)
    ref2 = (ref + tadd.double().cpu()[:, :, None, None] + res.double().cpu()) * 0.5
    e_ep = relerr(y2, ref2)
    # accumulate into a channel slice of a bigger buffer (free image stride)
    big = rnd(N, Cout + 5, y.shape[2], y.shape[3], seed=7)
    big0 = big.clone()
    ops.conv_forward(xa, xb, wp, ld, Cout, spec, bias=b, out=big[:, 2:2 + Cout], accumulate=True)
    e_acc = relerr(big[:, 2:2 + Cout], ref + big0[:, 2:2 + Cout].double().cpu())
    assert torch.equal(big[:, :2], big0[:, :2]) and torch.equal(big[:, 2 + Cout:], big0[:, 2 + Cout:])

    # gradients via fp64 autograd
    xr = xcat.double().cpu().requires_grad_(True)
    wr = w.double().cpu().requires_grad_(True)
    xin = xr
    if ups:
        xin = F.interpolate(xin, scale_factor=2.0, mode='nearest')
    if asym:
        xin = F.pad(xin, (0, 1, 0, 1))
    yr = F.conv2d(xin, wr, None, stride=stride, padding=pad)
    dy = rnd(*yr.shape, seed=8)
    yr.backward(dy.double().cpu())
    wd, ldd = ops.pack_weight(w, 1)
    Hv = H << ups
    dxv = ops.conv_dgrad(dy, wd, ldd, Cin, spec, (Hv, Hv))
    if ups:
        dx = ops.downsum2x2(dxv)
    else:
        dx = dxv
    e_d = relerr(dx, xr.grad)
    gw = torch.zeros_like(w)
    ops.conv_wgrad(dy, xa, xb, gw, spec, accumulate=False)
    e_w = relerr(gw, wr.grad)
    gw2 = w.clone()
    ops.conv_wgrad(dy, xa, xb, gw2, spec, accumulate=True, alpha=0.5)
    e_w2 = relerr(gw2, w.double().cpu() + 0.5 * wr.grad)
    report['conv/' + name] = dict(fwd=e_f, epilogue=e_ep, acc=e_acc, dgrad=e_d, wgrad=e_w, wgrad_acc=e_w2)
    assert max(e_f, e_ep, e_acc, e_d, e_w, e_w2) < 2e-5, report['conv/' + name]


@pytest.mark.parametrize('shape', [(7, 100, 90), (256, 512, 256), (33, 37, 92), (5, 16, 3)], ids=str)
def test_linear_fwd_dgrad_wgrad(ops, report, shape):
    """nn.Linear on row-major [N, C]: NT forward (+bias), NN dgrad (+accumulate), TN wgrad (+accumulate, odd widths)."""
    N, Ci, Co = shape
    x, w, b, dy = rnd(N, Ci, seed=1), rnd(Co, Ci, seed=2, scale=Ci ** -0.5), rnd(Co, seed=3), rnd(N, Co, seed=4)
    xd, wd, bd, dyd = (t.double().cpu() for t in (x, w, b, dy))
    e_f = relerr(ops.linear_forward(x, w, b), xd @ wd.t() + bd)
    e_f0 = relerr(ops.linear_forward(x, w, None), xd @ wd.t())
    dx0 = rnd(N, Ci, seed=5)
    dx = dx0.clone()
    ops.linear_dgrad(dy, w, out=dx, accumulate=True)
    e_d = relerr(dx, dx0.double().cpu() + dyd @ wd)
    e_d0 = relerr(ops.linear_dgrad(dy, w), dyd @ wd)
    gw0 = rnd(Co, Ci, seed=6)
    gw = gw0.clone()
    ops.linear_wgrad(dy, x, gw, accumulate=True)
    e_w = relerr(gw, gw0.double().cpu() + dyd.t() @ xd)
    report['linear/%dx%dx%d' % shape] = dict(fwd=e_f, fwd_nobias=e_f0, dgrad_acc=e_d, dgrad=e_d0, wgrad_acc=e_w)
    assert max(e_f, e_f0, e_d, e_d0, e_w) < 2e-5


def test_conv_fwd_dgrad_splitk(ops, report, monkeypatch):
    """Forward / dgrad split-K (small pixel counts: 4x4 and 8x8 layers, small batches): engaged by the heuristic,
    run-to-run deterministic, and equal to the unsplit kernel up to summation order."""
    N, C1, C2, Cout, H = 4, 200, 56, 256, 8
    xa, xb = rnd(N, C1, H, H, seed=1), rnd(N, C2, H, H, seed=2)
    w = rnd(Cout, C1 + C2, 3, 3, seed=3, scale=0.02)
    b = rnd(Cout, seed=4)
    spec = ops.ConvSpec(3, 1, 1, 0)
    wp, ld = ops.pack_weight(w, 0)
    wd, ldd = ops.pack_weight(w, 1)
    dy = rnd(N, Cout, H, H, seed=5)
    seen = []
    real = ops._conv_ksplit
    monkeypatch.setattr(ops, '_conv_ksplit', lambda p, d: (real(p, d), seen.append(p.ksplit))[0])
    y1 = ops.conv_forward(xa, xb, wp, ld, Cout, spec, bias=b).clone()
    y2 = ops.conv_forward(xa, xb, wp, ld, Cout, spec, bias=b).clone()
    d1 = ops.conv_dgrad(dy, wd, ldd, C1 + C2, spec, (H, H)).clone()
    assert seen and min(seen) >= 2, seen
    assert torch.equal(y1, y2)
    monkeypatch.setattr(ops, 'CONV_SPLITK_BLOCKS', 0)
    y0 = ops.conv_forward(xa, xb, wp, ld, Cout, spec, bias=b)
    d0 = ops.conv_dgrad(dy, wd, ldd, C1 + C2, spec, (H, H))
    assert seen[-1] == 1
    ref = ref_conv(torch.cat([xa, xb], 1), w, b, 1, 1, 0, False)
    report['conv/splitk_fwd_dgrad'] = dict(fwd_vs_unsplit=relerr(y1, y0), dgrad_vs_unsplit=relerr(d1, d0),
                                           fwd_vs_fp64=relerr(y1, ref), ksplit=seen[:3])
    assert relerr(y1, y0) < 5e-6 and relerr(d1, d0) < 5e-6 and relerr(y1, ref) < 2e-5


def test_conv_wgrad_splitk_large(ops, report):
    """Split-K path (many pixels): B*HW = 16384 pixels, deterministic across runs."""
    N, Cin, Cout, H = 16, 64, 96, 32
    x = rnd(N, Cin, H, H, seed=11)
    dy = rnd(N, Cout, H, H, seed=12)
    spec = ops.ConvSpec(3, 1, 1, 0)
    gw = torch.zeros(Cout, Cin, 3, 3, device=DEV)
    ops.conv_wgrad(dy, x, None, gw, spec, accumulate=False)
    gw_b = torch.zeros_like(gw)
    ops.conv_wgrad(dy, x, None, gw_b, spec, accumulate=False)
    assert torch.equal(gw, gw_b), 'split-K wgrad must be run-to-run deterministic'
    xr = x.double().cpu()
    ref = torch.nn.grad.conv2d_weight(xr, (Cout, Cin, 3, 3), dy.double().cpu(), padding=1)
    e = relerr(gw, ref)
    report['conv/wgrad_splitk'] = e
    assert e < 2e-5


@pytest.mark.parametrize('Z,M,K,N', [(3, 64, 40, 64), (2, 256, 256, 256), (5, 16, 179, 16), (2, 256, 179, 256)])
def test_bmm_variants(ops, report, Z, M, K, N):
    a_km = rnd(Z, K, M, seed=1)
    b_kn = rnd(Z, K, N, seed=2)
    o = ops.bmm_tn(a_km, b_kn, alpha=0.25)
    e1 = relerr(o, 0.25 * torch.bmm(a_km.double().cpu().transpose(1, 2), b_kn.double().cpu()))
    a_mk = rnd(Z, M, K, seed=3)
    o = ops.bmm_nn(a_mk, b_kn)
    e2 = relerr(o, torch.bmm(a_mk.double().cpu(), b_kn.double().cpu()))
    b_nk = rnd(Z, N, K, seed=4)
    o = ops.bmm_nt(a_mk, b_nk)
    e3 = relerr(o, torch.bmm(a_mk.double().cpu(), b_nk.double().cpu().transpose(1, 2)))
    report['bmm/%d_%d_%d_%d' % (Z, M, K, N)] = dict(tn=e1, nn=e2, nt=e3)
    assert max(e1, e2, e3) < 1e-5


@pytest.mark.parametrize('N,C1,C2,H,G,silu', [(40, 256, 0, 8, 32, True), (33, 128, 128, 16, 32, False), (64, 512, 0, 4, 32, True),
                                              (2, 32, 0, 8, 8, True), (3, 128, 0, 32, 32, True), (2, 100, 92, 4, 32, True),
                                             (2, 64, 0, 16, 32, False), (1, 128, 0, 128, 32, True), (2, 32, 0, 3, 8, True),
                                             (2, 48, 80, 64, 16, True), (1, 20, 44, 32, 8, True), (2, 64, 0, 256, 32, False), (2, 128, 128, 32, 32, True)])
def test_groupnorm(ops, report, N, C1, C2, H, G, silu):
    xa = rnd(N, C1, H, H, seed=1) + 0.3
    xb = rnd(N, C2, H, H, seed=2) if C2 else None
    Cc = C1 + C2
    gamma = 1 + 0.2 * rnd(Cc, seed=3)
    beta = 0.1 * rnd(Cc, seed=4)
    eps = 1e-6
    y, stats = ops.groupnorm_fwd(xa, xb, gamma, beta, G, eps, silu)
    xr = (xa if xb is None else torch.cat([xa, xb], 1)).double().cpu().requires_grad_(True)
    gr = gamma.double().cpu().requires_grad_(True)
    br = beta.double().cpu().requires_grad_(True)
    yr = F.group_norm(xr, G, gr, br, eps)
    if silu:
        yr = F.silu(yr)
    e_f = relerr(y, yr.detach())
    dz = rnd(*y.shape, seed=5)
    yr.backward(dz.double().cpu())
    add1 = rnd(*y.shape, seed=6)
    dx, pws = ops.groupnorm_bwd(xa, xb, gamma, beta, stats, dz, G, silu, add1=add1)
    e_dx = relerr(dx, xr.grad + add1.double().cpu())
    dg = torch.zeros(Cc, device=DEV)
    db = torch.ones(Cc, device=DEV)
    ops.colsum_accum(pws, N, Cc, 2, 1, dg, accumulate=False)
    ops.colsum_accum(pws, N, Cc, 2, 0, db, accumulate=True)
    e_g = relerr(dg, gr.grad)
    e_b = relerr(db, br.grad + 1.0)
    report['gn/%d_%d_%d_%d_%d' % (N, C1, C2, H, int(silu))] = dict(fwd=e_f, dx=e_dx, dgamma=e_g, dbeta=e_b)
    assert max(e_f, e_dx, e_g, e_b) < 2e-5


ATTN_CASES = [
    # N, heads, d, dv, H (T = H * H), qkv-sliced, score scale-up
    (3, 1, 256, 256, 16, True, 1.0),       # CIFAR / bedroom attention at 16 x 16
    (2, 1, 512, 512, 16, True, 1.0),       # bedroom mid block
    (2, 1, 384, 384, 32, True, 1.0),       # LDM ds = 2: T = 1024
    (2, 1, 576, 576, 16, False, 1.0),      # LDM ds = 4: 18 channel tiles over 4 wavefronts (5 / 5 / 4 / 4)
    (2, 1, 179, 133, 8, False, 1.0),       # pruned widths: ragged channel tiles, value width != key width
    (2, 4, 8, 8, 8, False, 1.0),           # attention_head_dim 8: 4 heads of 8 channels
    (1, 6, 24, 24, 16, True, 1.0),
    (2, 1, 64, 64, 16, False, 40.0),       # peaked softmax: the running max keeps moving, rows dominated by a few keys
    (1, 1, 640, 640, 8, False, 1.0),       # the widest head the kernel takes
]


@pytest.mark.parametrize('N,heads,d,dv,H,sliced,gain', ATTN_CASES)
def test_fused_attention_matches_fp64_sdpa(ops, report, monkeypatch, N, heads, d, dv, H, sliced, gain):
    """dp_attention_fwd (QK^T -> online softmax -> P.V in one kernel, csrc/attention.hip) vs fp64 softmax attention and vs the
    three-launch path it replaces in sampling forwards; twice: bit-identical; both schedules."""
    T = H * H
    # the default ('auto') takes the kernel only where it measured faster (T <= 256, heads <= 512 wide): every supported shape here
    assert ops.FUSED_ATTN == 'auto' and ops.attention_fused_ok(T, d, dv) == (T <= 256 and d <= 512 and dv <= 512 and T % 32 == 0)
    monkeypatch.setattr(ops, 'FUSED_ATTN', True)
    if sliced and d == dv:                              # channel slices of one fused QKV activation (engine.attn_fwd)
        qkv = rnd(N, 3 * heads * d, H, H, seed=1)
        q, k, v = qkv[:, :heads * d], qkv[:, heads * d:2 * heads * d], qkv[:, 2 * heads * d:]
    else:
        q, k, v = rnd(N, heads * d, H, H, seed=1), rnd(N, heads * d, H, H, seed=2), rnd(N, heads * dv, H, H, seed=3)
    q = q * gain if not sliced else q
    scale = float(d) ** -0.5
    assert ops.attention_fused_ok(T, d, dv)
    o = ops.attention_fwd(q, k, v, heads, scale)
    o2 = ops.attention_fwd(q, k, v, heads, scale)
    assert torch.equal(o, o2)
    for variant in (1, 2):                              # plain / software-pipelined schedule of the same arithmetic
        assert relerr(ops.attention_fwd(q, k, v, heads, scale, variant=variant), o) < 1e-6, variant
    Z = N * heads
    qd, kd, vd = (t.double().cpu().reshape(Z, -1, T) for t in (q, k, v))
    pr = (scale * torch.bmm(qd.transpose(1, 2), kd)).softmax(-1)
    ref = torch.bmm(vd, pr.transpose(1, 2)).reshape(N, heads * dv, H, H)
    e = relerr(o, ref)
    s = ops.bmm_tn(q.contiguous().view(Z, d, T), k.contiguous().view(Z, d, T), alpha=scale)
    o3 = ops.bmm_nt(v.contiguous().view(Z, dv, T), ops.softmax_fwd(s, out=s)).view(N, heads * dv, H, H)
    e3 = relerr(o3, ref)
    report['attention_fused/%d_%d_%d_%d_%d' % (N, heads, d, dv, T)] = dict(fused=e, three_launch=e3)
    assert e < 1e-5 and e3 < 1e-5
    assert not ops.attention_fused_ok(48, 64, 64) and not ops.attention_fused_ok(64, 960, 960)


def test_fused_attention_in_the_sampling_forward(ops, report, monkeypatch):
    """UNetEngine / LdmEngine no-grad forwards with ops.FUSED_ATTN on vs the three-launch path (same weights, same inputs)."""
    import golden_common as gc
    from helpers import make_model, pkg
    model = make_model(gc.CIFAR_CFG, 0)
    x = rnd(4, 3, 32, 32, seed=5)
    t = torch.tensor([10, 400, 700, 999], device=DEV)
    monkeypatch.setattr(ops, 'FUSED_ATTN', False)
    with torch.no_grad():
        y0 = model(x, t).sample
        monkeypatch.setattr(ops, 'FUSED_ATTN', True)
        y1 = model(x, t).sample
        monkeypatch.setattr(ops, 'FUSED_ATTN', 'auto')                  # the default: T = 256, d = 256 -> the fused kernel
        y2 = model(x, t).sample
    assert torch.equal(y2, y1)
    e = relerr(y1, y0)
    ldm = pkg('ldm')
    m2 = ldm.UNetModel(**gc.LDM_TINY_CFG)
    gc.det_init_(m2, 9)
    m2 = m2.to(DEV).eval()
    xl, ctx = rnd(2, 3, 16, 16, seed=6), rnd(2, 1, 16, seed=7)
    tl = torch.tensor([3, 500], device=DEV)
    monkeypatch.setattr(ops, 'FUSED_ATTN', True)
    with torch.no_grad():
        z1 = m2(xl, tl, context=ctx)
        monkeypatch.setattr(ops, 'FUSED_ATTN', False)
        z0 = m2(xl, tl, context=ctx)
    e2 = relerr(z1, z0)
    report['attention_fused/engines'] = dict(cifar_unet=e, ldm_tiny=e2)
    assert 0 < e < 1e-5 and e2 < 1e-5                  # e > 0: the fused path was taken (different rounding), and it agrees


def test_softmax_silu_misc(ops, report):
    s = rnd(6, 256, 256, seed=1, scale=3.0)
    p = ops.softmax_fwd(s)
    pr = s.double().cpu().softmax(-1)
    e1 = relerr(p, pr)
    dp_ = rnd(6, 256, 256, seed=2)
    sr = s.double().cpu().requires_grad_(True)
    (sr * 0.125).softmax(-1).backward(dp_.double().cpu())
    p2 = ops.softmax_fwd(s * 0.125)
    ds = ops.softmax_bwd(p2, dp_, 0.125)
    e2 = relerr(ds, sr.grad)
    s16 = rnd(5, 16, 16, seed=3)
    e3 = relerr(ops.softmax_fwd(s16), s16.double().cpu().softmax(-1))
    s2k = rnd(3, 4, 1500, seed=4)
    e3b = relerr(ops.softmax_fwd(s2k), s2k.double().cpu().softmax(-1))
    x = rnd(1000, seed=5, scale=3.0)
    e4 = relerr(ops.silu_fwd(x), F.silu(x.double().cpu()))
    xr = x.double().cpu().requires_grad_(True)
    dy = rnd(1000, seed=6)
    F.silu(xr).backward(dy.double().cpu())
    e5 = relerr(ops.silu_bwd(x, dy), xr.grad)
    a = rnd(4, 6, 8, 8, seed=7)
    rows = ops.rowsum_nc(a[:, 1:5])
    e6 = relerr(rows, a[:, 1:5].double().cpu().sum((2, 3)))
    for shp, sl in (((3, 5, 64, 64), slice(1, 4)), ((2, 3, 67, 67), slice(0, 3)), ((2, 4, 5, 3), slice(1, 3))):
        b = rnd(*shp, seed=17)          # workgroup-per-plane / unaligned / tiny planes
        e6 = max(e6, relerr(ops.rowsum_nc(b[:, sl]), b[:, sl].double().cpu().sum((2, 3))))
    up = rnd(2, 3, 8, 8, seed=8)
    e7 = relerr(ops.downsum2x2(up), F.avg_pool2d(up.double().cpu(), 2) * 4)
    y = rnd(300, seed=9)
    y0 = y.clone()
    ops.axpby(x[:300].contiguous(), 2.0, y, -0.5)
    e8 = relerr(y, 2.0 * x[:300].double().cpu() - 0.5 * y0.double().cpu())
    dst = rnd(3, 10, 4, 4, seed=10)
    d0 = dst.clone()
    src = rnd(3, 4, 4, 4, seed=11)
    ops.copy_strided(src, dst[:, 3:7], accumulate=True)
    e9 = relerr(dst[:, 3:7], d0[:, 3:7].double().cpu() + src.double().cpu())
    assert torch.equal(dst[:, :3], d0[:, :3])
    report['misc'] = dict(softmax=e1, softmax_bwd=e2, softmax16=e3, softmax1500=e3b, silu=e4, silu_bwd=e5, rowsum=e6,
                          downsum=e7, axpby=e8, copy=e9)
    assert max(e1, e2, e3, e3b, e4, e5, e6, e7, e8, e9) < 1e-5


def test_schedule_kernels_vs_golden(ops, report):
    import golden_common as gc
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'schedule.npz'))
    acp = torch.from_numpy(g['alphas_cumprod']).to(DEV)
    x0 = torch.from_numpy(gc.det_clean((2, 3, 4, 4), 11)).to(DEV)
    eps = torch.from_numpy(gc.det_noise((2, 3, 4, 4), 12)).to(DEV)
    worst = 0.0
    for i, t in enumerate(g['ts']):
        tt = torch.tensor([int(t), int(t)], dtype=torch.long, device=DEV)
        out = ops.add_noise(x0, eps, acp, tt)
        worst = max(worst, float((out.cpu() - torch.from_numpy(g['add_noise'][i])).abs().max()))
    emb = ops.timestep_embedding(torch.tensor([0.0, 1.0, 999.0], device=DEV), 128, False, 1.0)
    e_t = float((emb.cpu() - torch.from_numpy(g['temb_128'])).abs().max())
    emb2 = ops.timestep_embedding(torch.tensor([0.0, 1.0, 999.0], device=DEV), 32, True, 0.0)
    e_t2 = float((emb2.cpu() - torch.from_numpy(g['temb_32_flip'])).abs().max())
    report['schedule'] = dict(add_noise_abs=worst, temb_abs=e_t, temb_flip_abs=e_t2)
    # tolerance: add_noise is 2 mults + 1 add (<= 2 ulp of |x| <= 4); sin/cos of arguments up to 999 rad: 2e-5 abs
    assert worst < 1e-6 and e_t < 5e-5 and e_t2 < 5e-5


def test_mse_wg_adam_ddim(ops, report):
    out = rnd(8, 3, 32, 32, seed=1)
    noise = rnd(8, 3, 32, 32, seed=2)
    n = out.numel()
    loss, dout = ops.mse_fwd_bwd(out, noise, 2.0 / n, 1.0 / n)
    ref = F.mse_loss(out.double().cpu(), noise.double().cpu())
    e1 = abs(float(loss) - float(ref)) / float(ref)
    e2 = relerr(dout, 2.0 / n * (out.double().cpu() - noise.double().cpu()))
    # importance reductions
    w = rnd(70, 37, 3, 3, seed=3)
    g = rnd(70, 37, 3, 3, seed=4)
    wg = (w.double().cpu() * g.double().cpu())
    res = {}
    for mode, f in ((0, lambda t: t.abs().pow(2)), (1, lambda t: t.abs())):
        o = torch.zeros(70, device=DEV)
        ops.wg_reduce(w, g, 0, mode, o, False)
        res['rows%d' % mode] = relerr(o, f(wg).flatten(1).sum(1))
        o = torch.ones(37, device=DEV)
        ops.wg_reduce(w, g, 1, mode, o, True)
        res['cols%d' % mode] = relerr(o, f(wg).transpose(0, 1).flatten(1).sum(1) + 1.0)
    o = torch.zeros(70, device=DEV)
    ops.wg_reduce(w, g, 0, 2, o, False)
    res['rows2'] = relerr(o, wg.flatten(1).sum(1).abs())
    o = torch.zeros(37, device=DEV)
    ops.wg_reduce(w, g, 1, 2, o, False)
    res['cols2'] = relerr(o, wg.transpose(0, 1).flatten(1).sum(1).abs())
    wl = rnd(50, 64, seed=5)
    gl = rnd(50, 64, seed=6)
    o = torch.zeros(64, device=DEV)
    ops.wg_reduce(wl, gl, 1, 0, o, False)
    res['lin_cols'] = relerr(o, (wl.double().cpu() * gl.double().cpu()).pow(2).sum(0))
    gn_w, gn_g = rnd(96, seed=7), rnd(96, seed=8)
    o = torch.zeros(96, device=DEV)
    ops.wg_reduce(gn_w, gn_g, 0, 3, o, False)
    res['gn'] = relerr(o, (gn_w.double().cpu() * gn_g.double().cpu()).abs())
    # adam + ema + clip
    n = 10000
    p = rnd(n, seed=9)
    gr = rnd(n, seed=10)
    m = 0.1 * rnd(n, seed=11)
    v = (0.1 * rnd(n, seed=12)).abs()
    ema = rnd(n, seed=13)
    pr, gd, mr, vr, er = [t.double().cpu().clone() for t in (p, gr, m, v, ema)]
    partial = ops.sumsq_partials(gr)
    nc = ops.clip_coef(partial, 1.0)
    ops.adam_ema(p, gr, m, v, ema, nc[1:2], 2e-4, 0.9, 0.999, 1e-8, 3, 0.9999)
    tot = gd.pow(2).sum().sqrt()
    coef = min(1.0, 1.0 / (float(tot) + 1e-6))
    gd = gd * coef
    mr = 0.9 * mr + 0.1 * gd
    vr = 0.999 * vr + 0.001 * gd * gd
    pr = pr - (2e-4 / (1 - 0.9 ** 3)) * mr / (vr.sqrt() / math.sqrt(1 - 0.999 ** 3) + 1e-8)
    er = 0.0001 * pr + 0.9999 * er
    res['adam_p'] = relerr(p, pr)
    res['adam_m'] = relerr(m, mr)
    res['adam_v'] = relerr(v, vr)
    res['ema'] = relerr(ema, er)
    res['gnorm'] = abs(float(nc[0]) - float(tot)) / float(tot)
    # ddim
    x = rnd(2, 3, 8, 8, seed=14)
    e = rnd(2, 3, 8, 8, seed=15)
    a_t, a_prev = 0.37, 0.52
    o = ops.ddim_step(x, e, a_t, a_prev)
    x0 = ((x.double().cpu() - (1 - a_t) ** 0.5 * e.double().cpu()) / a_t ** 0.5).clamp(-1, 1)
    res['ddim'] = relerr(o, a_prev ** 0.5 * x0 + (1 - a_prev) ** 0.5 * e.double().cpu())
    res['mse'] = e1
    res['mse_grad'] = e2
    report['scalar_kernels'] = res
    assert max(res.values()) < 1e-5, res


@pytest.mark.parametrize('N,C,H', [(2, 96, 8), (3, 384, 16), (2, 50, 3)])
def test_layernorm_geglu_rowvec(ops, report, N, C, H):
    x = rnd(N, C, H, H, seed=1) + 0.2
    gamma = 1 + 0.2 * rnd(C, seed=2)
    beta = 0.1 * rnd(C, seed=3)
    y, st = ops.layernorm_fwd(x, gamma, beta)
    xr = x.double().cpu().requires_grad_(True)
    gr = gamma.double().cpu().requires_grad_(True)
    br = beta.double().cpu().requires_grad_(True)
    yr = F.layer_norm(xr.permute(0, 2, 3, 1), (C,), gr, br, 1e-5).permute(0, 3, 1, 2)
    e_f = relerr(y, yr.detach())
    dy = rnd(N, C, H, H, seed=4)
    add = rnd(N, C, H, H, seed=5)
    yr.backward(dy.double().cpu())
    dx, pws = ops.layernorm_bwd(x, gamma, st, dy, add=add)
    e_dx = relerr(dx, xr.grad + add.double().cpu())
    dg = torch.zeros(C, device=DEV)
    db = torch.zeros(C, device=DEV)
    ops.colsum_accum(pws, N, C, 2, 1, dg, accumulate=False)
    ops.colsum_accum(pws, N, C, 2, 0, db, accumulate=False)
    e_g, e_b = relerr(dg, gr.grad), relerr(db, br.grad)
    # GEGLU
    z = rnd(N, 2 * C, H, H, seed=6)
    zr = z.double().cpu().requires_grad_(True)
    a, g = zr.chunk(2, dim=1)
    o_ref = a * F.gelu(g)
    o = ops.geglu_fwd(z)
    e_gf = relerr(o, o_ref.detach())
    do = rnd(N, C, H, H, seed=7)
    o_ref.backward(do.double().cpu())
    e_gb = relerr(ops.geglu_bwd(z, do), zr.grad)
    v = rnd(N, C, seed=8)
    e_rv = relerr(ops.add_rowvec(x, v), x.double().cpu() + v.double().cpu()[:, :, None, None])
    report['ln/%d_%d_%d' % (N, C, H)] = dict(fwd=e_f, dx=e_dx, dgamma=e_g, dbeta=e_b, geglu=e_gf, geglu_bwd=e_gb, rowvec=e_rv)
    assert max(e_f, e_dx, e_g, e_b, e_gf, e_gb, e_rv) < 1e-5


# ------------------------------------------------------------------------------------------------------------------
# round 2: Philox dropout masks (bit-exact vs the numpy restatement), dropout fused into GroupNorm, DDPM update
# ------------------------------------------------------------------------------------------------------------------
def test_dropout_masks_bit_exact_vs_philox_oracle(ops, report):
    """The device masks are the oracle's masks, element for element (integer parity): bare mask kernel over a window of
    the stream crossing the 2^32-element boundary of the counter, and the strided apply kernel on a channel-slice view."""
    from oracle import philox_ref as PH
    n_bad = 0
    for p, seed, site, step, idx0, n in ((0.1, 7, 'mid_block.resnets.0.dropout', 3, 0, 100003),
                                         (0.5, (1 << 63) + 12345, 'x.to_out.1', 4000000000, (1 << 34) - 777, 4099),
                                         (0.03, 0, 9, 0, 5, 1000)):
        d = ops.dropout_desc(p, seed, site, step)
        got = ops.dropout_mask(n, d, DEV, idx0).cpu().numpy()
        want = PH.dropout_multipliers(n, p, seed, site, step, idx0)
        n_bad += int((got != want).sum())
    x = rnd(3, 10, 6, 6, seed=1)
    view = x[:, 2:7]                                            # free image stride
    d = ops.dropout_desc(0.25, 11, 'a.dropout', 2, n_off=5)
    y = ops.dropout_apply(view, d)
    m = PH.dropout_multipliers(view.numel(), 0.25, 11, 'a.dropout', 2, 5 * view[0].numel()).reshape(view.shape)
    n_bad += int((y.cpu().numpy() != view.cpu().numpy() * m).sum())
    ops.dropout_apply(view, d, out=view)                        # in place on the strided view; untouched channels intact
    x2 = rnd(3, 10, 6, 6, seed=1)
    assert torch.equal(x[:, :2], x2[:, :2]) and torch.equal(x[:, 7:], x2[:, 7:]) and torch.equal(x[:, 2:7], y)
    report['dropout/mask_mismatches'] = n_bad
    assert n_bad == 0


@pytest.mark.parametrize('N,C1,C2,H,G', [(40, 256, 0, 8, 32), (33, 128, 128, 16, 32), (64, 512, 0, 4, 32), (128, 64, 0, 2, 8),
                                         (2, 32, 0, 8, 8), (3, 128, 0, 32, 32), (2, 100, 92, 4, 32), (2, 32, 0, 3, 8),
                                         (1, 128, 0, 128, 32), (2, 48, 80, 64, 16), (2, 128, 128, 32, 32)], ids=str)
def test_groupnorm_silu_dropout_fused(ops, report, N, C1, C2, H, G):
    """y = dropout(silu(gn(x))) in one kernel and its backward (mask regenerated, nothing stored) against fp64 autograd with
    the oracle's masks: vec4 / scalar / split (few large groups) variants, virtual concat, an image offset (rank shard)."""
    from oracle import philox_ref as PH
    xa = rnd(N, C1, H, H, seed=1) + 0.3
    xb = rnd(N, C2, H, H, seed=2) if C2 else None
    Cc = C1 + C2
    gamma, beta, eps = 1 + 0.2 * rnd(Cc, seed=3), 0.1 * rnd(Cc, seed=4), 1e-6
    n_off = 3
    d = ops.dropout_desc(0.1, 99, 'r.dropout', 5, n_off=n_off)
    y, stats = ops.groupnorm_fwd(xa, xb, gamma, beta, G, eps, True, drop=d)
    m = torch.from_numpy(PH.dropout_multipliers(N * Cc * H * H, 0.1, 99, 'r.dropout', 5, n_off * Cc * H * H)).view(N, Cc, H, H).double()
    xr = (xa if xb is None else torch.cat([xa, xb], 1)).double().cpu().requires_grad_(True)
    gr = gamma.double().cpu().requires_grad_(True)
    br = beta.double().cpu().requires_grad_(True)
    yr = F.silu(F.group_norm(xr, G, gr, br, eps)) * m
    zeros_match = bool(((y.cpu() == 0) == (m == 0)).all())
    e_f = relerr(y, yr.detach())
    dz = rnd(*y.shape, seed=5)
    yr.backward(dz.double().cpu())
    add1 = rnd(*y.shape, seed=6)
    dx, pws = ops.groupnorm_bwd(xa, xb, gamma, beta, stats, dz, G, True, add1=add1, drop=d)
    e_dx = relerr(dx, xr.grad + add1.double().cpu())
    dg, db = torch.zeros(Cc, device=DEV), torch.zeros(Cc, device=DEV)
    ops.colsum_accum(pws, N, Cc, 2, 1, dg, accumulate=False)
    ops.colsum_accum(pws, N, Cc, 2, 0, db, accumulate=False)
    e_g, e_b = relerr(dg, gr.grad), relerr(db, br.grad)
    report['gn_dropout/%d_%d_%d_%d' % (N, C1, C2, H)] = dict(fwd=e_f, dx=e_dx, dgamma=e_g, dbeta=e_b, zeros_match=zeros_match)
    assert zeros_match and max(e_f, e_dx, e_g, e_b) < 2e-5


def test_ddpm_step_kernel(ops, report):
    x, e, vn = rnd(2, 3, 8, 8, seed=14), rnd(2, 3, 8, 8, seed=15), rnd(2, 3, 8, 8, seed=16)
    sa, sb, c0, c1, sig = 0.61, 0.79, 0.013, 0.985, 0.07
    o = ops.ddpm_step(x, e, sa, sb, c0, c1, sig, vn, clip=True, clip_range=0.8)
    xd, ed, vd = x.double().cpu(), e.double().cpu(), vn.double().cpu()
    ref = c0 * ((xd - sb * ed) / sa).clamp(-0.8, 0.8) + c1 * xd + sig * vd
    o2 = ops.ddpm_step(x, e, sa, sb, c0, c1, clip=False)
    ref2 = c0 * ((xd - sb * ed) / sa) + c1 * xd
    o3 = ops.ddim_step(x, e, 0.37, 0.52, clip=True, clip_range=0.5)
    x0 = ((xd - (1 - 0.37) ** 0.5 * ed) / 0.37 ** 0.5).clamp(-0.5, 0.5)
    ref3 = 0.52 ** 0.5 * x0 + (1 - 0.52) ** 0.5 * ed
    report['ddpm_step'] = dict(noise_clip=relerr(o, ref), plain=relerr(o2, ref2), ddim_clip_range=relerr(o3, ref3))
    assert max(relerr(o, ref), relerr(o2, ref2), relerr(o3, ref3)) < 1e-5


def test_colsum_batch_equals_immediate(ops, report):
    """Queued column sums (one launch per 80 items) are bit-identical to the immediate launches, accumulate included."""
    q = ops.ColsumQueue()
    want, got = [], []
    for i, (N, C, ws, wo) in enumerate([(256, 256, 2, 1), (256, 256, 2, 0), (7, 90, 1, 0), (128, 513, 1, 0), (3, 64, 2, 1)] * 40):
        src = rnd(N, C, ws, seed=100 + i)
        a, b = rnd(C, seed=300 + i), None
        b = a.clone()
        ops.colsum_accum(src, N, C, ws, wo, a, True)
        q.add(src, N, C, ws, wo, b, True)
        want.append(a)
        got.append(b)
    q.flush()
    assert not q.items
    bad = sum(0 if torch.equal(a, b) else 1 for a, b in zip(want, got))
    report['colsum_batch/items'] = dict(n=len(want), mismatching=bad)
    assert bad == 0


def test_wgrad_xcd_order_is_a_pure_relabelling(ops, report, monkeypatch):
    """The XCD-aware workgroup order of the fast weight-gradient kernel only changes WHICH workgroup computes a (tile, tap,
    split): results are bit-identical to the plain order, for grids whose size is / is not a multiple of 8 and two sources."""
    spec = ops.ConvSpec(3, 1, 1, 0)
    bad = 0
    for (N, C1, C2, Cout, H) in ((32, 256, 0, 256, 16), (16, 128, 128, 128, 32), (8, 90, 180, 180, 16), (24, 256, 0, 384, 8)):
        x = rnd(N, C1, H, H, seed=1)
        x2 = rnd(N, C2, H, H, seed=2) if C2 else None
        dy = rnd(N, Cout, H, H, seed=3)
        outs = []
        for flag in ('1', None):
            if flag:
                monkeypatch.setenv('DP_NO_XCD', flag)
            else:
                monkeypatch.delenv('DP_NO_XCD')
            gw = torch.zeros(Cout, C1 + C2, 3, 3, device=DEV)
            ops.conv_wgrad(dy, x, x2, gw, spec, accumulate=False)
            outs.append(gw)
        bad += 0 if torch.equal(outs[0], outs[1]) else 1
        ref = torch.nn.grad.conv2d_weight((x if x2 is None else torch.cat([x, x2], 1)).double().cpu(), (Cout, C1 + C2, 3, 3),
                                          dy.double().cpu(), padding=1)
        assert relerr(outs[1], ref) < 2e-5
    report['wgrad_xcd/mismatching_shapes'] = bad
    assert bad == 0


@pytest.mark.parametrize('shape', [(2, 128, 3, 32), (3, 37, 4, 20), (1, 16, 1, 5), (5, 50, 2, 16)], ids=str)
def test_conv_few_output_channels_direct_kernel(ops, report, monkeypatch, shape):
    """conv_out-shaped layers (<= 4 output channels, 3x3 'same'): the direct stencil kernel against fp64 F.conv2d and against
    the matrix path it replaces (ragged tiles, channel tails, ReLU epilogue)."""
    N, C, M, H = shape
    x = rnd(N, C, H, H, seed=1)
    w = rnd(M, C, 3, 3, seed=2, scale=1.0 / math.sqrt(9 * C))
    b = rnd(M, seed=3)
    spec = ops.ConvSpec(3, 1, 1, 0)
    wp, ld = ops.pack_weight(w, 0)
    ref = ref_conv(x, w, b, 1, 1, 0, False)
    y = ops.conv_forward(x, None, wp, ld, M, spec, bias=b)
    yr = ops.conv_forward(x, None, wp, ld, M, spec, bias=b, relu=True)
    monkeypatch.setenv('DP_NO_FEW_OUT', '1')
    ym = ops.conv_forward(x, None, wp, ld, M, spec, bias=b)
    e, em, er = relerr(y, ref), relerr(y, ym.double().cpu()), relerr(yr, ref.clamp_min(0))
    report['conv_few_out/%s' % (shape,)] = dict(fwd=e, vs_matrix_path=em, relu=er)
    assert max(e, em, er) < 2e-5


def test_conv_fast_x4_loads_equal_dword_loads(ops, report, monkeypatch):
    """The 16-byte B-tile loads of the fast convolution kernel (4 pixels per lane, border element zeroed in LDS) put exactly the
    values of the 4-byte path into LDS: forward, dgrad and the attention products are bit-identical with and without them."""
    bad = []

    def guarded(t):                 # the 16-byte loads need one readable float in front of the tensor (x_guard)
        g = ops.empty_act(tuple(t.shape), t.device)
        g.copy_(t)
        assert g.storage_offset() >= 1
        return g

    cases = [(4, 128, 0, 128, 32, 3), (3, 256, 0, 256, 16, 3), (5, 96, 160, 256, 8, 3), (9, 64, 0, 80, 4, 3), (2, 90, 180, 180, 16, 3),
             (3, 128, 64, 96, 12, 1), (6, 256, 0, 256, 2, 1), (2, 37, 0, 70, 8, 3)]
    for (N, C1, C2, Cout, H, k) in cases:
        xa = guarded(rnd(N, C1, H, H, seed=1))
        xb = guarded(rnd(N, C2, H, H, seed=2)) if C2 else None
        w = rnd(Cout, C1 + C2, k, k, seed=3, scale=0.05)
        b = rnd(Cout, seed=4)
        dy = guarded(rnd(N, Cout, H, H, seed=5))
        spec = ops.ConvSpec(k, 1, k // 2, 0)
        wp, ld = ops.pack_weight(w, 0)
        wd, ldd = ops.pack_weight(w, 1)
        outs = []
        for flag in ('1', None):
            if flag:
                monkeypatch.setenv('DP_NO_X4', flag)
            else:
                monkeypatch.delenv('DP_NO_X4')
            y = ops.conv_forward(xa, xb, wp, ld, Cout, spec, bias=b)
            d = ops.conv_dgrad(dy, wd, ldd, C1 + C2, spec, (H, H))
            outs.append((y, d))
        if not (torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])):
            bad.append((N, C1, C2, Cout, H, k))
        ref = ref_conv(xa if xb is None else torch.cat([xa, xb], 1), w, b, 1, k // 2, 0, False)
        assert relerr(outs[1][0], ref) < 2e-5
    q, kk = rnd(4, 256, 256, seed=7), rnd(4, 256, 256, seed=8)
    monkeypatch.setenv('DP_NO_X4', '1')
    s0 = ops.bmm_tn(q, kk, alpha=0.0625)
    monkeypatch.delenv('DP_NO_X4')
    s1 = ops.bmm_tn(q, kk, alpha=0.0625)
    if not torch.equal(s0, s1):
        bad.append('bmm_tn')
    report['conv_x4/mismatching'] = bad
    assert not bad, bad


def test_input_pipeline_kernel_bit_exact(ops, report):
    """Row f4: uint8 -> fp32 NCHW with ToTensor / RandomHorizontalFlip / Normalize (or data_transform) in one kernel, bit-exact
    against the oracle restatement (same flip decisions, same fp32 operation order), both source layouts, a rank offset."""
    import importlib
    from oracle import data_ref
    data = importlib.import_module('diff-pruning_amd.data')
    rng = np.random.default_rng(3)
    bad = []
    for hwc, shape in ((True, (9, 32, 32, 3)), (False, (5, 3, 32, 32)), (True, (2, 17, 23, 3))):
        u8 = rng.integers(0, 256, shape, dtype=np.uint8)
        for mode, flip, dq, n_off in ((1, 0.5, False, 0), (2, 0.5, True, 11), (0, 0.0, False, 3), (1, 0.25, False, 1 << 33)):
            got = data.to_device_batch(u8, hwc, torch.device(DEV), mode, flip, seed=77, epoch=4, n_off=n_off, dequant=dq)
            want = data_ref.transform_batch(u8, hwc, mode, flip, 77, 4, n_off, dq)
            if not torch.equal(got.cpu(), want):
                bad.append((hwc, shape, mode, flip, dq, float((got.cpu() - want).abs().max())))
    report['data_pipeline/mismatching'] = bad
    assert not bad, bad
    # loader end to end on the device: two ranks == one global batch
    ds = data.ArrayDataset(rng.integers(0, 256, (20, 3, 32, 32), dtype=np.uint8), hwc=False)
    one = list(data.DeviceLoader(ds, 8, DEV, seed=9))
    r0 = list(data.DeviceLoader(ds, 4, DEV, seed=9, rank=0, world=2))
    r1 = list(data.DeviceLoader(ds, 4, DEV, seed=9, rank=1, world=2))
    assert [b.shape[0] for b in one] == [8, 8, 4] and len(r0) == 3 and len(r1) == 2       # the last global batch holds 4 samples
    for a, b0, b1 in zip(one, r0, r1):
        assert torch.equal(a, torch.cat([b0, b1]))
    assert torch.equal(one[2], r0[2])


def test_pool_resize_ssim_kernels(ops, report):
    """Row f3 glue kernels against PyTorch: pooling with TensorFlow-style averages, ATen-exact bilinear resize, SSIM, MSE."""
    import importlib
    from oracle import metrics_ref as M
    metrics = importlib.import_module('diff-pruning_amd.metrics')
    x = rnd(3, 20, 35, 35, seed=1)
    xd = x.double().cpu()
    res = {}
    res['avg_3_1_1'] = relerr(metrics.pool2d(x, 3, 1, 1, 'avg'), F.avg_pool2d(xd, 3, 1, 1, count_include_pad=False))
    res['max_3_1_1'] = relerr(metrics.pool2d(x, 3, 1, 1, 'max'), F.max_pool2d(xd, 3, 1, 1))
    res['max_3_2_0'] = relerr(metrics.pool2d(x, 3, 2, 0, 'max'), F.max_pool2d(xd, 3, 2))
    view = rnd(2, 30, 17, 17, seed=2)[:, 4:20]
    res['max_view'] = relerr(metrics.pool2d(view, 3, 2, 0, 'max'), F.max_pool2d(view.double().cpu(), 3, 2))
    img = torch.rand(4, 3, 32, 32, device=DEV)
    got = metrics.resize_bilinear(img, (299, 299), 2.0, -1.0)
    want = 2 * F.interpolate(img.cpu(), size=(299, 299), mode='bilinear', align_corners=False) - 1
    res['resize_299'] = float((got.cpu() - want).abs().max())
    img2 = torch.rand(2, 3, 300, 280, device=DEV)
    res['resize_down'] = float((metrics.resize_bilinear(img2, (299, 299)).cpu()
                                - F.interpolate(img2.cpu(), size=(299, 299), mode='bilinear', align_corners=False)).abs().max())
    a = torch.rand(5, 3, 32, 32, device=DEV)
    b = (a + 0.1 * torch.randn_like(a)).clamp(0, 1)
    res['ssim_32'] = relerr(metrics.ssim(a, b), M.ssim(a.cpu(), b.cpu()))
    a2 = torch.rand(2, 3, 70, 45, device=DEV)
    b2 = (a2 + 0.2 * torch.randn_like(a2)).clamp(0, 1)
    res['ssim_70x45'] = relerr(metrics.ssim(a2, b2), M.ssim(a2.cpu(), b2.cpu()))
    res['ssim_self'] = float((metrics.ssim(a, a) - 1).abs().max())
    res['mse'] = relerr(metrics.mse_per_image(a, b), F.mse_loss(a.double().cpu(), b.double().cpu(), reduction='none').mean(dim=(1, 2, 3)))
    report['metrics_kernels'] = res
    assert res['resize_299'] < 2e-6 and res['resize_down'] < 2e-6 and res['ssim_self'] < 1e-5
    assert max(res['avg_3_1_1'], res['max_3_1_1'], res['max_3_2_0'], res['max_view'], res['ssim_32'], res['ssim_70x45'], res['mse']) < 2e-5


@pytest.mark.parametrize('shape', [(5, 3, 1, 7, 1, 0, 3), (4, 24, 7, 1, 1, 3, 0), (3, 16, 5, 5, 1, 2, 2), (3, 8, 3, 3, 2, 0, 0),
                                   (2, 32, 1, 3, 1, 0, 1), (2, 48, 1, 1, 1, 0, 0)], ids=str)
def test_general_conv_forward_relu(ops, report, shape):
    """Non-square / valid / strided forward convolutions with the bias + ReLU epilogue (the FID Inception's BasicConv2d)."""
    N, Cin, kh, kw, stride, ph, pw = shape
    Cout, H, W = 40, 17, 17
    x = rnd(N, Cin, H, W, seed=1)
    w = rnd(Cout, Cin, kh, kw, seed=2, scale=1.0 / math.sqrt(Cin * kh * kw))
    b = rnd(Cout, seed=3)
    spec = ops.ConvSpec.general(kh, kw, stride, ph, pw)
    wp, ld = ops.pack_weight(w, 0)
    y = ops.conv_forward(x, None, wp, ld, Cout, spec, bias=b, relu=True)
    ref = F.relu(F.conv2d(x.double().cpu(), w.double().cpu(), b.double().cpu(), stride=stride, padding=(ph, pw)))
    assert y.shape == ref.shape
    e = relerr(y, ref)
    big = torch.zeros(N, Cout + 6, y.shape[2], y.shape[3], device=DEV)
    ops.conv_forward(x, None, wp, ld, Cout, spec, bias=b, relu=True, out=big[:, 3:3 + Cout])
    e2 = relerr(big[:, 3:3 + Cout], ref)
    report['conv_general/%s' % (shape,)] = dict(fwd=e, slice=e2)
    assert max(e, e2) < 2e-5 and float(big[:, :3].abs().max()) == 0


@pytest.mark.parametrize('N,C1,C2,H,G', [(40, 128, 0, 32, 32), (2, 100, 92, 4, 32), (2, 32, 0, 3, 8), (4, 256, 0, 16, 32), (3, 128, 0, 32, 32),
                                         (40, 256, 0, 8, 32), (33, 128, 128, 16, 32), (64, 512, 0, 4, 32), (2, 128, 128, 32, 32)], ids=str)
def test_groupnorm_bwd_row_sums(ops, report, N, C1, C2, H, G):
    """The GroupNorm backward kernels also emit rows[n, c] = sum_hw dx (the next layer's bias / time-embedding-projection
    gradient rows): equal to a separate row-sum pass over dx up to summation order, and dx itself is unchanged."""
    xa = rnd(N, C1, H, H, seed=1) + 0.3
    xb = rnd(N, C2, H, H, seed=2) if C2 else None
    Cc = C1 + C2
    gamma, beta = 1 + 0.2 * rnd(Cc, seed=3), 0.1 * rnd(Cc, seed=4)
    y, stats = ops.groupnorm_fwd(xa, xb, gamma, beta, G, 1e-6, True)
    dz, add1 = rnd(*y.shape, seed=5), rnd(*y.shape, seed=6)
    dx0, pws0 = ops.groupnorm_bwd(xa, xb, gamma, beta, stats, dz, G, True, add1=add1)
    dx, pws, rows = ops.groupnorm_bwd(xa, xb, gamma, beta, stats, dz, G, True, add1=add1, want_rows=True)
    assert torch.equal(dx, dx0) and torch.equal(pws, pws0)
    if rows is None:                      # few large groups: the split kernels ran (no fused sums; the engine falls back)
        assert N * G < ops.GN_SPLIT_GROUPS and H * H >= 1024
        return
    assert rows.shape == (N, Cc)
    e = relerr(rows, dx.double().cpu().sum((2, 3)))
    report['gn_rows/%d_%d_%d_%d' % (N, C1, C2, H)] = e
    assert e < 1e-5


@pytest.mark.parametrize('shape', [(3, 5, 4, 4), (2, 7, 3, 5), (1, 2, 1, 1), (4, 16, 16, 16)], ids=str)
def test_upsample2x_equals_nearest_interpolate(ops, report, shape):
    """dp_upsample2x (vector path for even widths, scalar otherwise, channel-slice sources) is F.interpolate(nearest, x2) bit for
    bit; the materialised-upsample convolution equals the gather-form (ups = 1) convolution it replaces."""
    N, C, H, W = shape
    x = rnd(N, C, H, W, seed=1)
    y = ops.upsample2x(x)
    assert torch.equal(y.cpu(), F.interpolate(x.cpu(), scale_factor=2.0, mode='nearest'))
    big = rnd(N, C + 3, H, W, seed=2)
    ys = ops.upsample2x(big[:, 1:1 + C])
    assert torch.equal(ys.cpu(), F.interpolate(big[:, 1:1 + C].cpu(), scale_factor=2.0, mode='nearest'))
    w = rnd(6, C, 3, 3, seed=3, scale=0.2)
    b = rnd(6, seed=4)
    wp, ld = ops.pack_weight(w, 0)
    a = ops.conv_forward(y, None, wp, ld, 6, ops.ConvSpec(3, 1, 1, 0), bias=b)
    g = ops.conv_forward(x, None, wp, ld, 6, ops.ConvSpec(3, 1, 1, 1), bias=b)
    e = relerr(a, g.double().cpu())
    report['upsample2x/%s' % (shape,)] = dict(conv_vs_gather_form=e)
    assert e < 2e-6


@pytest.mark.parametrize('case', [(2, 24, 40, 8, 0), (3, 37, 19, 6, 1), (2, 128, 128, 16, 0), (1, 5, 7, 2, 1), (2, 64, 96, 3, 0)], ids=str)
def test_stride2_dgrad_by_parity_classes(ops, report, case):
    """Input gradient of the stride-2 3x3 downsample convolution (asymmetric (0,1,0,1) pad of Diffusers, symmetric pad 1 of the
    LDM UNet) as four stride-1 convolutions + dp_interleave2x2 (with the fused skip-gradient add, also from a channel slice):
    against fp64 autograd and against the zero-inserted single-launch form; odd widths take the scalar interleave path."""
    N, Cin, Cout, Ho, pad = case
    H = 2 * Ho
    x = rnd(N, Cin, H, H, seed=1)
    w = rnd(Cout, Cin, 3, 3, seed=2, scale=1.0 / math.sqrt(9 * Cin))
    dy = rnd(N, Cout, Ho, Ho, seed=3)
    spec = ops.ConvSpec(3, 2, pad, 0)
    xr = x.double().cpu().requires_grad_(True)
    xin = F.pad(xr, (0, 1, 0, 1)) if pad == 0 else xr
    y = F.conv2d(xin, w.double().cpu(), None, stride=2, padding=pad)
    assert y.shape[2] == Ho
    y.backward(dy.double().cpu())
    packs = [ops.pack_weight_s2(w, ph, pw, pad) for ph in (0, 1) for pw in (0, 1)]
    dx = ops.conv_dgrad_s2(dy, packs, Cin, spec, (H, H))
    wd, ldd = ops.pack_weight(w, 1)
    dz = ops.conv_dgrad(dy, wd, ldd, Cin, spec, (H, H))
    big = rnd(N, Cin + 5, H, H, seed=4)
    dxa = ops.conv_dgrad_s2(dy, packs, Cin, spec, (H, H), add=big[:, 2:2 + Cin])
    e, ez, ea = relerr(dx, xr.grad), relerr(dx, dz.double().cpu()), relerr(dxa, xr.grad + big[:, 2:2 + Cin].double().cpu())
    report['dgrad_s2/%s' % (case,)] = dict(vs_fp64=e, vs_zero_inserted=ez, with_add=ea)
    assert max(e, ez, ea) < 2e-5


@pytest.mark.parametrize('shape', [(2, 8, 12, 4, 4), (3, 37, 50, 5, 3), (2, 128, 96, 8, 8), (1, 3, 5, 1, 1)], ids=str)
def test_upsample_conv_subpixel_primitives_and_composite(ops, report, shape):
    """Upsample2D as four 2x2 convolutions at low resolution: dp_ups_weff / dp_ups_wfold against their definition,
    dp_deinterleave2x2 as the inverse of dp_interleave2x2, the class convolutions (ConvSpec.same: 2x2 taps, top / left padding
    1 or 0) in forward, input-gradient and weight-gradient form, and the composite against fp64 autograd of
    interpolate(nearest x2) + conv2d(3x3, pad 1) and against the single-launch gather form it replaces."""
    N, Ci, Co, H, W = shape
    x = rnd(N, Ci, H, W, seed=1)
    w = rnd(Co, Ci, 3, 3, seed=2, scale=1.0 / math.sqrt(9 * Ci))
    b = rnd(Co, seed=3)
    dy = rnd(N, Co, 2 * H, 2 * W, seed=4)
    xr, wr = x.double().cpu().requires_grad_(True), w.double().cpu().requires_grad_(True)
    yr = F.conv2d(F.interpolate(xr, scale_factor=2.0, mode='nearest'), wr, b.double().cpu(), padding=1)
    yr.backward(dy.double().cpu())

    def t(parity, k):
        return (0 if k == 0 else 1) if parity == 0 else (1 if k == 2 else 0)

    weff = ops.ups_weff(w)
    want = torch.zeros(4, Co, Ci, 2, 2, dtype=torch.float64)
    for ph in (0, 1):
        for pw in (0, 1):
            for ky in range(3):
                for kx in range(3):
                    want[2 * ph + pw, :, :, t(ph, ky), t(pw, kx)] += w.double().cpu()[:, :, ky, kx]
    e_weff = relerr(weff, want)
    q = ops.deinterleave2x2(dy)
    assert all(torch.equal(q[2 * ph + pw].cpu(), dy.cpu()[:, :, ph::2, pw::2]) for ph in (0, 1) for pw in (0, 1))
    assert torch.equal(ops.interleave2x2(q).cpu(), dy.cpu())
    # composite
    qy = ops.empty_act((4, N, Co, H, W), x.device)
    dx = None
    gweff = torch.empty(4, Co, Ci, 2, 2, device=x.device)
    for c, spec in enumerate(ops.UPS_CLASS_SPECS):
        wp, ld = ops.pack_weight(weff[c], 0)
        ops.conv_forward(x, None, wp, ld, Co, spec, bias=b, out=qy[c])
        wd, ldd = ops.pack_weight(weff[c], 1)
        dx = ops.conv_dgrad(q[c], wd, ldd, Ci, spec, (H, W), out=dx, accumulate=c > 0)
        ops.conv_wgrad(q[c], x, None, gweff[c], spec, accumulate=False)
    y = ops.interleave2x2(qy)
    gw0 = rnd(Co, Ci, 3, 3, seed=5)
    gw = ops.ups_wfold(gweff, gw0.clone(), accumulate=True)
    wp9, ld9 = ops.pack_weight(w, 0)
    y_gather = ops.conv_forward(x, None, wp9, ld9, Co, ops.ConvSpec(3, 1, 1, 1), bias=b)
    e = dict(weff=e_weff, fwd=relerr(y, yr.detach()), fwd_vs_gather_form=relerr(y, y_gather.double().cpu()),
             dgrad=relerr(dx, xr.grad), wgrad=relerr(gw, gw0.double().cpu() + wr.grad))
    report['ups_subpixel/%s' % (shape,)] = e
    assert max(e.values()) < 2e-5, e


def test_randn_philox_matches_oracle_and_is_shard_invariant(ops, report):
    """dp_randn_philox (x_T and the loss noise of the LDM importance pass) against oracle/philox_ref.randn: the same Philox
    words and Box-Muller pairing (device logf / cosf / sinf vs numpy: a few ulp of the draw), a window that starts at an odd
    offset beyond 2^32 elements, and any shard of a draw BIT-equal to that slice of the whole draw."""
    from oracle import philox_ref as PH
    worst, stats = 0.0, None
    for seed, sid, step, idx0, n in ((21, 0x7854, 0, 0, 1 << 20), ((1 << 62) + 5, 3, 999, (1 << 34) + 3, 4099),
                                     (0, 0, 0, 2, 7)):
        got = ops.randn_philox((n,), seed, sid, step, idx0=idx0, device=DEV).cpu().numpy()
        want = PH.randn(n, seed, sid, step, idx0)
        worst = max(worst, float(np.abs(got - want).max()))
        if stats is None:
            stats = dict(mean=float(got.mean()), std=float(got.std()), kurt=float((got ** 4).mean()), absmax=float(np.abs(got).max()))
    whole = ops.randn_philox((6, 3, 64, 64), 21, 1, 5, device=DEV)
    per = 3 * 64 * 64
    n_bad = 0
    for lo, hi in ((0, 2), (2, 4), (4, 5), (5, 6), (1, 6)):
        part = ops.randn_philox((hi - lo, 3, 64, 64), 21, 1, 5, idx0=lo * per, device=DEV)
        n_bad += int((part != whole[lo:hi]).sum())
    odd = ops.randn_philox((1001,), 21, 1, 5, idx0=4099, device=DEV)             # unaligned start and end
    n_bad += int((odd != whole.reshape(-1)[4099:5100]).sum())
    report['randn_philox'] = dict(abs_worst=worst, shard_mismatches=n_bad, **stats)
    assert worst < 5e-6 and n_bad == 0                     # |draw| <= 6.8; 5e-6 absolute ~ a few fp32 ulp of the largest draws
    assert abs(stats['mean']) < 5e-3 and abs(stats['std'] - 1) < 5e-3 and abs(stats['kurt'] - 3) < 0.05


def test_early_exit_ratio_state_machine(ops):
    """dp_early_exit_update_ratio = prune_ldm.py:104,124-129 (`max_loss = -1; if loss > max_loss: max_loss = loss; if loss /
    max_loss < thres: break`) on a loss sequence, against the same lines evaluated on fp32 0-d tensors on the host."""
    seq = [0.5, 0.8, 0.3, 0.081, 0.0799999, 0.5, 0.01]
    for thr in (0.1, -1.0):
        state = torch.tensor([-1.0, 0.0, 0.0], device=DEV)
        rec = torch.zeros(16, device=DEV)
        for l in seq:
            ops.early_exit_update_ratio(torch.tensor([l], device=DEV), thr, state, rec)
        mx, want = torch.tensor(-1.0), []
        for l in seq:
            lt = torch.tensor(l, dtype=torch.float32)
            want.append(float(lt))
            if lt > mx:
                mx = lt
            if thr >= 0 and bool(lt / mx < thr):
                break
        st = state.cpu().tolist()
        assert int(st[2]) == len(want) and rec[:len(want)].cpu().tolist() == want and float(mx) == st[0]
        assert st[1] == (1.0 if len(want) < len(seq) else 0.0)


@pytest.mark.parametrize('N,C1,C2,Cout,H,k', [(4, 200, 56, 256, 8, 3), (12, 960, 0, 960, 8, 1), (6, 576, 0, 576, 16, 3), (4, 192, 0, 192, 8, 3),
                                               (3, 96, 0, 96, 4, 3), (2, 64, 64, 100, 4, 3)], ids=str)
def test_conv_splitk_fold_equals_reduction_launch(ops, report, monkeypatch, N, C1, C2, Cout, H, k):
    """Split-K partials reduced by the last-arriving workgroup (dp_conv_gemm_params.tile_counters) against the separate
    reduction launch: same ascending-split summation whoever arrives last -> BIT-identical outputs, over 25 repetitions each
    (the arrival order differs from run to run), with every epilogue operand (bias, per-image addend, residual, scale,
    accumulate), for forward and input-gradient launches on the 128- and 96-row fast tiles and the general kernel."""
    xa = rnd(N, C1, H, H, seed=1)
    xb = rnd(N, C2, H, H, seed=2) if C2 else None
    w = rnd(Cout, C1 + C2, k, k, seed=3, scale=0.02)
    b, tadd, res = rnd(Cout, seed=4), rnd(N, Cout, seed=6), rnd(N, Cout, H, H, seed=7)
    spec = ops.ConvSpec(k, 1, k // 2, 0)
    wp, ld = ops.pack_weight(w, 0)
    wd, ldd = ops.pack_weight(w, 1)
    dy = rnd(N, Cout, H, H, seed=5)
    seen = []
    real = ops._conv_ksplit
    monkeypatch.setattr(ops, '_conv_ksplit', lambda p, d: (real(p, d), seen.append((p.ksplit, bool(p.tile_counters))))[0])

    def run():
        y = ops.conv_forward(xa, xb, wp, ld, Cout, spec, bias=b, tadd=tadd, res=res, post_scale=0.7)
        acc = res.clone()
        ops.conv_forward(xa, xb, wp, ld, Cout, spec, out=acc, accumulate=True)
        d = ops.conv_dgrad(dy, wd, ldd, C1 + C2, spec, (H, H), alpha=0.5)
        return y.clone(), acc, d.clone()

    monkeypatch.setattr(ops, 'SPLITK_FOLD', False)
    ref = run()
    assert seen and all(s[0] >= 2 and not s[1] for s in seen), seen
    del seen[:]
    monkeypatch.setattr(ops, 'SPLITK_FOLD', True)
    monkeypatch.setattr(ops, 'SPLITK_FOLD_MAX', 1 << 30)          # every split count, also the ones the default leaves to the launch
    n_bad = 0
    for rep in range(25):
        got = run()
        n_bad += sum(0 if torch.equal(a, r) else 1 for a, r in zip(got, ref))
    assert all(s[0] >= 2 and s[1] for s in seen), seen[:3]
    report['conv/splitk_fold/%d_%d_%d_%d' % (C1 + C2, Cout, H, k)] = dict(ksplit=seen[0][0], mismatching_outputs_of_75=n_bad)
    assert n_bad == 0


def test_conv_splitk_fold_under_concurrent_streams(ops, report, monkeypatch):
    """Advisor finding of round 4: the in-kernel split-K fold orders its slab stores before the ticket with write-through (sc1)
    stores + `s_waitcnt vmcnt(0)` + a relaxed agent-scope atomic and reads the slabs back with sc1 loads -- no fence.  Its one
    bit-identity test ran on an otherwise idle GPU.  Here three HIP streams issue fold launches AT THE SAME TIME (the main stream,
    the weight-gradient side stream and a second timestep pipeline's stream are what a sweep runs), each over more than eight
    XCD-spanning tiles, with big Winograd / weight-gradient launches of the other streams filling the L2s in between; 60 rounds x 3
    streams x 3 launches, every output compared bit for bit with the reduction-launch form (DP_SPLITK_FOLD=0: the safety switch)."""
    spec = ops.ConvSpec(3, 1, 1, 0)
    shapes = [(12, 384, 384, 32), (16, 256, 256, 16), (6, 576, 576, 16)]      # (N, Cin, Cout, H): 288 / 64 / 60 tiles of 128 x 128
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(2)]
    work = []
    for i, (N, Ci, Co, H) in enumerate(shapes):
        x, w = rnd(N, Ci, H, H, seed=10 + i), rnd(Co, Ci, 3, 3, seed=20 + i, scale=0.02)
        dy, res, b = rnd(N, Co, H, H, seed=30 + i), rnd(N, Co, H, H, seed=40 + i), rnd(Co, seed=50 + i)
        work.append(dict(x=x, w=w, dy=dy, res=res, b=b, fw=ops.pack_weight(w, 0), bw=ops.pack_weight(w, 1), Ci=Ci, Co=Co, H=H))
    big_x, big_w = rnd(64, 128, 32, 32, seed=7), rnd(128, 128, 3, 3, seed=8, scale=0.02)
    big_u = ops.pack_weight_wino(big_w, 0)
    big_p = ops.pack_weight(big_w, 0)
    big_g = torch.zeros_like(big_w)
    seen = []
    real = ops._conv_ksplit
    monkeypatch.setattr(ops, '_conv_ksplit', lambda p, d: (real(p, d), seen.append((p.ksplit, bool(p.tile_counters))))[0])
    monkeypatch.setattr(ops, 'WINO_MIN_TILES', 1 << 30)            # the three test convolutions stay on the direct (split-K) kernel
    monkeypatch.setattr(ops, 'SPLITK_FOLD_MAX', 8)

    def launches(wk):
        y = ops.conv_forward(wk['x'], None, *wk['fw'], wk['Co'], spec, bias=wk['b'], res=wk['res'], post_scale=0.7)
        acc = wk['res'].clone()
        ops.conv_forward(wk['x'], None, *wk['fw'], wk['Co'], spec, out=acc, accumulate=True)
        d = ops.conv_dgrad(wk['dy'], *wk['bw'], wk['Ci'], spec, (wk['H'], wk['H']), alpha=0.5)
        return y, acc, d

    monkeypatch.setattr(ops, 'SPLITK_FOLD', False)
    ref = [tuple(t.clone() for t in launches(wk)) for wk in work]
    torch.cuda.synchronize()
    assert all(k >= 2 and not f for k, f in seen), seen
    del seen[:]
    monkeypatch.setattr(ops, 'SPLITK_FOLD', True)
    n_bad, n_cmp = 0, 0
    for rnd_i in range(60):
        outs = []
        for si, st in enumerate(streams):
            st.wait_stream(streams[0])
            with torch.cuda.stream(st):
                wk = work[(si + rnd_i) % 3]                       # every stream sees every shape over the rounds
                if rnd_i % 2 == si % 2:                           # L2 / CU contention from the kernels a sweep runs beside the folds
                    monkeypatch.setattr(ops, 'WINO_MIN_TILES', 0)
                    ops.conv_forward(big_x, None, *big_p, 128, spec, wino=big_u)
                    monkeypatch.setattr(ops, 'WINO_MIN_TILES', 1 << 30)
                else:
                    ops.conv_wgrad(big_x, big_x, None, big_g, spec, accumulate=False)
                outs.append(((si + rnd_i) % 3, launches(wk)))
        for st in streams[1:]:
            streams[0].wait_stream(st)
        torch.cuda.synchronize()
        for wi, got in outs:
            for a, r in zip(got, ref[wi]):
                n_cmp += 1
                n_bad += 0 if torch.equal(a, r) else 1
    folded = [k for k, f in seen if f]
    report['conv/splitk_fold_concurrent'] = dict(compared=n_cmp, mismatching=n_bad, fold_launches=len(folded), ksplits=sorted(set(folded)))
    assert folded and all(f for k, f in seen if 2 <= k <= 8), seen[:6]
    assert n_bad == 0


@pytest.mark.parametrize('taps,mc,splits,acc', [(9, 192 * 192, 28, True), (9, 96 * 179, 3, False), (4, 256 * 256, 9, True), (9, 37, 1, True),
                                                (4, 1000, 64, False)], ids=str)
def test_splitk_reduction_kernels_round5_equal_scalar_forms(ops, report, monkeypatch, taps, mc, splits, acc):
    """Round 5 rewrote the split-K reduction launches for memory-level parallelism (one thread per (m, c) with all taps and 36
    loads in flight; four pixels per thread in the convolution epilogue): per element they perform the additions of the scalar
    forms in the same order, so the outputs must be BIT-identical (DP_NO_REDUCE_MC / DP_NO_EPI4 select the scalar forms)."""
    import ctypes as C
    lib = ops._lib()
    ws = rnd(splits, taps, mc, seed=3)
    base = rnd(mc * taps, seed=4)
    outs = []
    for env in (None, '1'):
        if env is None:
            monkeypatch.delenv('DP_NO_REDUCE_MC', raising=False)
        else:
            monkeypatch.setenv('DP_NO_REDUCE_MC', env)
        out = base.clone()
        assert lib.dp_splitk_reduce_taps(C.c_void_p(ws.data_ptr()), taps * mc, splits, C.c_void_p(out.data_ptr()), mc, taps,
                                         1 if acc else 0, ops._stream()) == 0
        outs.append(out)
    torch.cuda.synchronize()
    ref = ws.double().sum(0).view(taps, mc).t().reshape(-1) + (base.double() if acc else 0)
    assert torch.equal(outs[0], outs[1])
    assert relerr(outs[0], ref) < 1e-5


@pytest.mark.parametrize('N,C,Cout,H', [(12, 960, 960, 8), (6, 576, 384, 16), (4, 192, 200, 8)], ids=str)
def test_conv_splitk_epilogue4_equals_scalar_epilogue(ops, monkeypatch, N, C, Cout, H):
    """dp_conv_splitk_epilogue with four pixels per thread against the one-element form: bit-identical with every epilogue operand."""
    import os
    spec = ops.ConvSpec(3, 1, 1, 0)
    x, w = rnd(N, C, H, H, seed=1), rnd(Cout, C, 3, 3, seed=2, scale=0.02)
    b, tadd, res = rnd(Cout, seed=4), rnd(N, Cout, seed=6), rnd(N, Cout, H, H, seed=7)
    wp, ld = ops.pack_weight(w, 0)
    monkeypatch.setattr(ops, 'SPLITK_FOLD', False)                 # the reduction launch, not the in-kernel fold
    monkeypatch.setattr(ops, 'WINO_MIN_TILES', 1 << 30)
    got = []
    for env in (None, '1'):
        if env is None:
            monkeypatch.delenv('DP_NO_EPI4', raising=False)
        else:
            monkeypatch.setenv('DP_NO_EPI4', env)
        y = ops.conv_forward(x, None, wp, ld, Cout, spec, bias=b, tadd=tadd, res=res, post_scale=0.7, relu=True)
        acc = res.clone()
        ops.conv_forward(x, None, wp, ld, Cout, spec, out=acc, accumulate=True)
        got.append((y.clone(), acc))
    torch.cuda.synchronize()
    assert torch.equal(got[0][0], got[1][0]) and torch.equal(got[0][1], got[1][1])


def test_pack_weight_batch_equals_per_layer_pack(ops):
    """dp_pack_weight_batch (every layer of a finetune step in a few launches) against dp_pack_weight, element for element:
    3x3 / 1x1 convolutions and a Linear, both operand layouts, widths that need the ld padding (90 -> 92)."""
    ws = [rnd(90, 45, 3, 3, seed=1), rnd(128, 256, 1, 1, seed=2), rnd(33, 70, seed=3), rnd(192, 192, 3, 3, seed=4), rnd(3, 96, 3, 3, seed=5)]
    items = [(w, m) for w in ws for m in (0, 1)] * 9                       # > 64 items: more than one launch
    items += [(w, ('wino', m)) for w in ws if w.dim() == 4 and w.shape[2] == 3 for m in (0, 1)]     # Winograd F(2, 3) operands
    items += [(w, ('wino2d', m)) for w in ws if w.dim() == 4 and w.shape[2] == 3 for m in (0, 1)]   # F(2x2, 3x3) operands
    got = ops.pack_weight_batch(items)
    for (w, m), (buf, ld) in zip(items, got):
        if isinstance(m, tuple):
            ref, ld0 = (ops.pack_weight_wino if m[0] == 'wino' else ops.pack_weight_wino2d)(w, m[1])
        else:
            ref, ld0 = ops.pack_weight(w, m)
        assert ld == ld0 and torch.equal(buf, ref), (tuple(w.shape), m)


@pytest.mark.parametrize('N,C1,C2,Cout,H', [(8, 128, 0, 128, 32), (4, 256, 128, 128, 32), (16, 256, 0, 256, 16), (64, 256, 0, 256, 8),
                                             (256, 96, 0, 192, 4), (3, 24, 8, 40, 16), (5, 64, 0, 70, 8), (7, 16, 0, 16, 64), (1, 32, 0, 48, 256),
                                             (12, 384, 0, 384, 32), (6, 192, 192, 100, 16)], ids=str)
def test_conv_winograd_f43_matches_fp64(ops, report, monkeypatch, N, C1, C2, Cout, H):
    """dp_conv_wino43 (3x3 / stride 1 / pad 1 as a one-dimensional Winograd F(4, 3) implicit GEMM: half the multiplies; the no-grad
    forwards only) against the fp64 convolution: two concat sources, bias, per-image addend, residual, scale, ReLU-free and
    accumulate forms, output-channel tails, 8-channel K chunks, W = 4 ... 256, tiles spanning several images, split-K -- next to
    F(2, 3)'s and the direct kernel's error on the same inputs.  Bar: 1e-5 of the output scale, run-to-run bit-identical; the
    verdict's accuracy gate (5e-6) is met by the <= 256-channel shapes and missed by the 384-channel ones (5.2e-6 measured) -- one
    of the reasons the kernel is opt-in (DP_WINO43=1), see ops.WINO43."""
    monkeypatch.setattr(ops, 'WINO_MIN_TILES', 0)
    monkeypatch.setattr(ops, 'WINO43', True)
    monkeypatch.setattr(ops, 'WINO43_MIN_TILES', 0)
    xa, xb = rnd(N, C1, H, H, seed=1), (rnd(N, C2, H, H, seed=2) if C2 else None)
    w = rnd(Cout, C1 + C2, 3, 3, seed=3, scale=0.05)
    b, tadd, res = rnd(Cout, seed=4), rnd(N, Cout, seed=6), rnd(N, Cout, H, H, seed=7)
    spec = ops.ConvSpec(3, 1, 1, 0)
    wp, ld = ops.pack_weight(w, 0)
    U2, U4 = ops.pack_weight_wino(w, 0), ops.pack_weight_wino43(w)
    x = torch.cat([xa, xb], 1) if C2 else xa
    ref = torch.nn.functional.conv2d(x.double().cpu(), w.double().cpu(), b.double().cpu(), padding=1)
    ref = (ref + tadd.double().cpu()[:, :, None, None] + res.double().cpu()) * 0.7
    launched = []
    real = ops._conv_wino43
    monkeypatch.setattr(ops, '_conv_wino43', lambda *a: (lambda r: (launched.append(bool(r)), r)[1])(real(*a)))
    y4 = ops.conv_forward(xa, xb, wp, ld, Cout, spec, bias=b, tadd=tadd, res=res, post_scale=0.7, wino43=U4)
    y4b = ops.conv_forward(xa, xb, wp, ld, Cout, spec, bias=b, tadd=tadd, res=res, post_scale=0.7, wino43=U4)
    y2 = ops.conv_forward(xa, xb, wp, ld, Cout, spec, bias=b, tadd=tadd, res=res, post_scale=0.7, wino=U2)
    yd = ops.conv_forward(xa, xb, wp, ld, Cout, spec, bias=b, tadd=tadd, res=res, post_scale=0.7)
    acc = res.clone()
    ops.conv_forward(xa, xb, wp, ld, Cout, spec, out=acc, accumulate=True, alpha=0.5, wino43=U4)
    ref_acc = res.double().cpu() + 0.5 * torch.nn.functional.conv2d(x.double().cpu(), w.double().cpu(), None, padding=1)
    assert launched == [True, True, True], launched            # the kernel took every launch (incl. the split-K ones of small grids)
    e4, e2, ed, ea = relerr(y4, ref), relerr(y2, ref), relerr(yd, ref), relerr(acc, ref_acc)
    report['conv/winograd_f43/%d+%d_%d_%d_%d' % (C1, C2, Cout, H, N)] = dict(f43=e4, f23=e2, direct=ed, accumulate=ea)
    assert torch.equal(y4, y4b)
    assert e4 < 1e-5 and ea < 1e-5, (e4, ea)


@pytest.mark.parametrize('N,C1,C2,Cout,H', [(8, 128, 0, 128, 32), (4, 256, 128, 128, 32), (16, 256, 0, 256, 16), (64, 256, 0, 256, 8),
                                             (256, 96, 0, 192, 4), (3, 24, 8, 40, 16), (5, 64, 0, 70, 8), (7, 16, 0, 16, 64), (1, 32, 0, 48, 256)], ids=str)
def test_conv_winograd_f23_matches_fp64(ops, report, monkeypatch, N, C1, C2, Cout, H):
    """dp_conv_wino (3x3 / stride 1 / pad 1 as a one-dimensional Winograd F(2, 3) implicit GEMM) against the fp64 convolution,
    forward (two concat sources, bias, per-image addend, residual, scale; accumulate) and input gradient, next to the direct
    kernel's error on the same inputs; K chunks of 16 and of 8 channels; output-channel tails; tiles that span several images."""
    monkeypatch.setattr(ops, 'WINO_MIN_TILES', 0)
    xa, xb = rnd(N, C1, H, H, seed=1), (rnd(N, C2, H, H, seed=2) if C2 else None)
    w = rnd(Cout, C1 + C2, 3, 3, seed=3, scale=0.05)
    b, tadd, res = rnd(Cout, seed=4), rnd(N, Cout, seed=6), rnd(N, Cout, H, H, seed=7)
    dy = rnd(N, Cout, H, H, seed=5)
    spec = ops.ConvSpec(3, 1, 1, 0)
    wp, ld = ops.pack_weight(w, 0)
    wd, ldd = ops.pack_weight(w, 1)
    U0, U1 = ops.pack_weight_wino(w, 0), ops.pack_weight_wino(w, 1)
    x = torch.cat([xa, xb], 1) if C2 else xa
    ref = torch.nn.functional.conv2d(x.double().cpu(), w.double().cpu(), b.double().cpu(), padding=1)
    ref = (ref + tadd.double().cpu()[:, :, None, None] + res.double().cpu()) * 0.7
    ref_d = 0.5 * torch.nn.functional.conv_transpose2d(dy.double().cpu(), w.double().cpu(), padding=1)
    launched = []
    real = ops._conv_wino
    monkeypatch.setattr(ops, '_conv_wino', lambda *a: (launched.append(real(*a)), launched[-1])[1])
    y_w = ops.conv_forward(xa, xb, wp, ld, Cout, spec, bias=b, tadd=tadd, res=res, post_scale=0.7, wino=U0)
    y_d = ops.conv_forward(xa, xb, wp, ld, Cout, spec, bias=b, tadd=tadd, res=res, post_scale=0.7)
    acc = res.clone()
    ops.conv_forward(xa, xb, wp, ld, Cout, spec, out=acc, accumulate=True, wino=U0)
    d_w = ops.conv_dgrad(dy, wd, ldd, C1 + C2, spec, (H, H), alpha=0.5, wino=U1)
    d_d = ops.conv_dgrad(dy, wd, ldd, C1 + C2, spec, (H, H), alpha=0.5)
    assert launched == [True, True, Cout % 8 == 0], launched          # dgrad contracts over Cout: 70 channels keep the direct form
    ref_acc = res.double().cpu() + torch.nn.functional.conv2d(x.double().cpu(), w.double().cpu(), None, padding=1)
    e = dict(fwd=relerr(y_w, ref), fwd_direct=relerr(y_d, ref), acc=relerr(acc, ref_acc), dgrad=relerr(d_w, ref_d),
             dgrad_direct=relerr(d_d, ref_d))
    y_w2 = ops.conv_forward(xa, xb, wp, ld, Cout, spec, bias=b, tadd=tadd, res=res, post_scale=0.7, wino=U0)
    # split-K form (small grids): the (channel chunk, kernel row) loop over blockIdx.z + the reduction launch
    seen = []
    real_sup = ops._lib().dp_conv_wino
    monkeypatch.setattr(ops, 'WINO_MIN_TILES', 4 * (-(-Cout // 64)) * (-(-(N * H * H) // 128)))      # wants 4 slices, gets min(4, K tiles / 8)
    if (C1 + C2) >= 96:                                  # >= 16 K tiles: at least two slices of 8
        import ctypes
        y_s = ops.conv_forward(xa, xb, wp, ld, Cout, spec, bias=b, tadd=tadd, res=res, post_scale=0.7, wino=U0)
        acc_s = res.clone()
        ops.conv_forward(xa, xb, wp, ld, Cout, spec, out=acc_s, accumulate=True, wino=U0)
        y_s2 = ops.conv_forward(xa, xb, wp, ld, Cout, spec, bias=b, tadd=tadd, res=res, post_scale=0.7, wino=U0)
        e['fwd_splitk'], e['acc_splitk'] = relerr(y_s, ref), relerr(acc_s, ref_acc)
        assert launched[-3:] == [True, True, True] and torch.equal(y_s, y_s2)
        assert e['fwd_splitk'] < 3e-6 and e['acc_splitk'] < 3e-6, e
    report['conv/winograd_f23/%d_%d_%d_%d' % (N, C1 + C2, Cout, H)] = dict(e, run_to_run_equal=bool(torch.equal(y_w, y_w2)))
    assert torch.equal(y_w, y_w2)
    assert e['fwd'] < 3e-6 and e['acc'] < 3e-6 and e['dgrad'] < 3e-6, e


@pytest.mark.parametrize('N,C1,C2,Cout,H', [(8, 128, 0, 128, 32), (4, 256, 128, 128, 32), (16, 256, 0, 256, 16), (64, 256, 0, 256, 8),
                                             (256, 96, 0, 192, 4), (3, 24, 8, 40, 16), (5, 64, 0, 70, 8), (7, 16, 0, 16, 64), (3, 128, 0, 64, 8),
                                             (1, 8, 0, 16, 4), (2, 256, 0, 128, (6, 16)), (1, 32, 0, 48, 256), (2, 16, 8, 64, 128), (3, 8, 0, 16, (4, 128)),
                                             (1, 128, 0, 128, 128), (64, 96, 0, 96, 32), (8, 96, 0, 96, 32), (4, 64, 0, 160, 16), (1, 32, 0, 96, 128),
                                             (70, 64, 0, 80, 32), (128, 32, 0, 24, 32), (40, 96, 96, 96, 32)],
                         ids=str)
def test_conv_winograd_f2x2_3x3_matches_fp64(ops, report, monkeypatch, N, C1, C2, Cout, H):
    """dp_conv_wino2d (3x3 / stride 1 / pad 1 as a TWO-dimensional Winograd F(2x2, 3x3) implicit GEMM, csrc/winograd2d.hip) against the
    fp64 convolution: forward (two concat sources, bias, per-image addend, residual, scale; accumulate) and input gradient, next to
    the direct and the F(2, 3) kernels' errors on the same inputs; output-channel tails (40, 70 rows in 64-row tiles); pixel blocks
    that span several images (8 x 8, 4 x 4), partial last blocks (3 x 64 pixels, 1 x 16), non-square images; images wider than 64 pixels (2 x 64-pixel segments with a halo tile: 128 and
    256 columns, segment borders inside the image and on it); split-K; run-to-run bits."""
    monkeypatch.setattr(ops, 'WINO_MIN_TILES', 0)
    monkeypatch.setattr(ops, 'WINO2D_MIN_TILES', 0)
    Hh, Ww = H if isinstance(H, tuple) else (H, H)
    xa, xb = rnd(N, C1, Hh, Ww, seed=1), (rnd(N, C2, Hh, Ww, seed=2) if C2 else None)
    w = rnd(Cout, C1 + C2, 3, 3, seed=3, scale=0.05)
    b, tadd, res = rnd(Cout, seed=4), rnd(N, Cout, seed=6), rnd(N, Cout, Hh, Ww, seed=7)
    dy = rnd(N, Cout, Hh, Ww, seed=5)
    spec = ops.ConvSpec(3, 1, 1, 0)
    wp, ld = ops.pack_weight(w, 0)
    wd, ldd = ops.pack_weight(w, 1)
    U0, U1 = ops.pack_weight_wino(w, 0), ops.pack_weight_wino(w, 1)
    V0, V1 = ('2d',) + tuple(ops.pack_weight_wino2d(w, 0)), ('2d',) + tuple(ops.pack_weight_wino2d(w, 1))
    x = torch.cat([xa, xb], 1) if C2 else xa
    ref = torch.nn.functional.conv2d(x.double().cpu(), w.double().cpu(), b.double().cpu(), padding=1)
    ref = (ref + tadd.double().cpu()[:, :, None, None] + res.double().cpu()) * 0.7
    ref_d = 0.5 * torch.nn.functional.conv_transpose2d(dy.double().cpu(), w.double().cpu(), padding=1)
    launched = []
    real = ops._conv_wino2d
    monkeypatch.setattr(ops, '_conv_wino2d', lambda *a: (launched.append(real(*a)), launched[-1])[1])
    y_w = ops.conv_forward(xa, xb, wp, ld, Cout, spec, bias=b, tadd=tadd, res=res, post_scale=0.7, wino=V0)
    y_1 = ops.conv_forward(xa, xb, wp, ld, Cout, spec, bias=b, tadd=tadd, res=res, post_scale=0.7, wino=U0)
    y_d = ops.conv_forward(xa, xb, wp, ld, Cout, spec, bias=b, tadd=tadd, res=res, post_scale=0.7)
    acc = res.clone()
    ops.conv_forward(xa, xb, wp, ld, Cout, spec, out=acc, accumulate=True, wino=V0)
    d_w = ops.conv_dgrad(dy, wd, ldd, C1 + C2, spec, (Hh, Ww), alpha=0.5, wino=V1)
    d_d = ops.conv_dgrad(dy, wd, ldd, C1 + C2, spec, (Hh, Ww), alpha=0.5)
    assert launched == [True, True, Cout % 8 == 0], launched          # dgrad contracts over Cout: 70 channels keep the direct form
    ref_acc = res.double().cpu() + torch.nn.functional.conv2d(x.double().cpu(), w.double().cpu(), None, padding=1)
    e = dict(fwd=relerr(y_w, ref), fwd_f23=relerr(y_1, ref), fwd_direct=relerr(y_d, ref), acc=relerr(acc, ref_acc),
             dgrad=relerr(d_w, ref_d), dgrad_direct=relerr(d_d, ref_d))
    y_w2 = ops.conv_forward(xa, xb, wp, ld, Cout, spec, bias=b, tadd=tadd, res=res, post_scale=0.7, wino=V0)
    # split-K form (small grids): the channel-chunk loop over blockIdx.z + the reduction launch
    monkeypatch.setattr(ops, 'WINO2D_MIN_TILES', 4 * (-(-Cout // 64)) * (-(-(N * Hh * Ww) // 128)))   # wants 4 slices, gets min(4, K tiles / 8)
    if (C1 + C2) >= 128:                                 # >= 16 K tiles of 8 channels: at least two slices of 8
        y_s = ops.conv_forward(xa, xb, wp, ld, Cout, spec, bias=b, tadd=tadd, res=res, post_scale=0.7, wino=V0)
        acc_s = res.clone()
        ops.conv_forward(xa, xb, wp, ld, Cout, spec, out=acc_s, accumulate=True, wino=V0)
        y_s2 = ops.conv_forward(xa, xb, wp, ld, Cout, spec, bias=b, tadd=tadd, res=res, post_scale=0.7, wino=V0)
        e['fwd_splitk'], e['acc_splitk'] = relerr(y_s, ref), relerr(acc_s, ref_acc)
        assert launched[-3:] == [True, True, True] and torch.equal(y_s, y_s2)
        assert e['fwd_splitk'] < 3e-6 and e['acc_splitk'] < 3e-6, e
    report['conv/winograd_f2x2_3x3/%d_%d_%d_%s' % (N, C1 + C2, Cout, H)] = dict(e, run_to_run_equal=bool(torch.equal(y_w, y_w2)))
    assert torch.equal(y_w, y_w2)
    assert e['fwd'] < 3e-6 and e['acc'] < 3e-6 and e['dgrad'] < 3e-6, e


@pytest.mark.parametrize('N,C1,C2,Cout,H', [(8, 128, 0, 128, 32), (4, 256, 128, 128, 32), (16, 256, 0, 256, 16), (64, 256, 0, 192, 8),
                                             (2, 64, 0, 96, 64), (8, 40, 0, 70, 16), (1, 32, 0, 48, 256), (8, 96, 0, 96, 32), (4, 192, 96, 96, 16),
                                             (4, 179, 0, 90, 16)], ids=str)
def test_conv_wgrad_winograd_matches_fp64(ops, report, monkeypatch, N, C1, C2, Cout, H):
    """dp_wgrad_wino (3x3 / stride 1 / pad 1 weight gradient by the transposed Winograd F(2, 3) algorithm) against the fp64 weight
    gradient, next to the direct kernel's error: two concat sources, row / column tails (96, 70, 40 channels), images of 8 .. 256
    pixels width, accumulation into an existing gradient, several split counts; run-to-run bit-identical."""
    monkeypatch.setattr(ops, 'WGRAD_WINO_MIN_WORK', 0)
    monkeypatch.setattr(ops, 'WGRAD_WINO_MIN_FILL', 0.0)
    monkeypatch.setattr(ops, 'WGRAD_WINO2D', False)                    # (the two-dimensional kernel has its own test below)
    xa, xb = rnd(N, C1, H, H, seed=1), (rnd(N, C2, H, H, seed=2) if C2 else None)
    dy = rnd(N, Cout, H, H, seed=5)
    spec = ops.ConvSpec(3, 1, 1, 0)
    x = torch.cat([xa, xb], 1) if C2 else xa
    xd = x.double().cpu().requires_grad_(False)
    wz = torch.zeros(Cout, C1 + C2, 3, 3, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv2d(xd, wz, None, padding=1).backward(dy.double().cpu())
    ref = 0.5 * wz.grad
    used = []
    real = ops._conv_wgrad_wino
    monkeypatch.setattr(ops, '_conv_wgrad_wino', lambda *a: (lambda r: (used.append(r is not None), r)[1])(real(*a)))
    g0 = rnd(Cout, C1 + C2, 3, 3, seed=9)
    gw = g0.clone()
    ops.conv_wgrad(dy, xa, xb, gw, spec, alpha=0.5, accumulate=True)
    gw2 = g0.clone()
    ops.conv_wgrad(dy, xa, xb, gw2, spec, alpha=0.5, accumulate=True)
    assert used == [True, True], used
    monkeypatch.setattr(ops, 'WGRAD_WINO', False)
    gd = g0.clone()
    ops.conv_wgrad(dy, xa, xb, gd, spec, alpha=0.5, accumulate=True)
    e_w, e_d = relerr(gw - g0, ref), relerr(gd - g0, ref)
    report['wgrad/winograd_f23/%d_%d_%d_%d' % (N, C1 + C2, Cout, H)] = dict(wino=e_w, direct=e_d, run_to_run_equal=bool(torch.equal(gw, gw2)))
    assert torch.equal(gw, gw2)
    assert e_w < 5e-6, (e_w, e_d)


@pytest.mark.parametrize('N,C1,C2,Cout,H', [(8, 128, 0, 128, 32), (4, 256, 128, 128, 32), (16, 256, 0, 256, 16), (64, 256, 0, 192, 8),
                                             (8, 40, 0, 70, 16), (8, 96, 0, 96, 32), (4, 192, 96, 96, 16), (4, 179, 0, 90, 16), (1, 8, 0, 16, 8),
                                             (3, 32, 32, 24, (16, 32)), (2, 64, 0, 64, (32, 8)), (8, 64, 0, 160, 8)], ids=str)
def test_conv_wgrad_winograd_f3x3_2x2_matches_fp64(ops, report, monkeypatch, N, C1, C2, Cout, H):
    """dp_wgrad_wino2d (3x3 / stride 1 / pad 1 weight gradient by the TWO-dimensional transposed Winograd algorithm F(3x3, 2x2),
    csrc/wgrad2d.hip) against the fp64 weight gradient, next to the F(3, 2) and the direct kernels' errors: two concat sources (boundary
    on a multiple of 32), row / column tails (96, 70, 90, 40, 179 channels in 64 x 32 tiles), images of 8, 16 and 32 pixels width incl.
    non-square ones, a single 64-pixel K tile, accumulation into an existing gradient, several split counts; run-to-run bits."""
    monkeypatch.setattr(ops, 'WGRAD_WINO_MIN_WORK', 0)
    monkeypatch.setattr(ops, 'WGRAD_WINO_MIN_FILL', 0.0)
    monkeypatch.setattr(ops, 'WGRAD_WINO2D_MIN_FILL', 0.0)
    Hh, Ww = H if isinstance(H, tuple) else (H, H)
    xa, xb = rnd(N, C1, Hh, Ww, seed=1), (rnd(N, C2, Hh, Ww, seed=2) if C2 else None)
    dy = rnd(N, Cout, Hh, Ww, seed=5)
    spec = ops.ConvSpec(3, 1, 1, 0)
    x = torch.cat([xa, xb], 1) if C2 else xa
    wz = torch.zeros(Cout, C1 + C2, 3, 3, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv2d(x.double().cpu(), wz, None, padding=1).backward(dy.double().cpu())
    ref = 0.5 * wz.grad
    used = []
    real = ops._conv_wgrad_wino2d
    monkeypatch.setattr(ops, '_conv_wgrad_wino2d', lambda *a: (lambda r: (used.append(r is not None), r)[1])(real(*a)))
    g0 = rnd(Cout, C1 + C2, 3, 3, seed=9)
    gw = g0.clone()
    ops.conv_wgrad(dy, xa, xb, gw, spec, alpha=0.5, accumulate=True)
    gw2 = g0.clone()
    ops.conv_wgrad(dy, xa, xb, gw2, spec, alpha=0.5, accumulate=True)
    assert used == [True, True], used
    # other split counts (the default aims at ~1024 workgroups): one slice, and as many as there are K tiles
    e_split = {}
    for blocks in (1, 1 << 20):
        monkeypatch.setattr(ops, 'WGRAD_BLOCKS', blocks)
        gs = g0.clone()
        ops.conv_wgrad(dy, xa, xb, gs, spec, alpha=0.5, accumulate=True)
        e_split[blocks] = relerr(gs - g0, ref)
    monkeypatch.setattr(ops, 'WGRAD_BLOCKS', 1024)
    monkeypatch.setattr(ops, 'WGRAD_WINO2D', False)
    g1 = g0.clone()
    ops.conv_wgrad(dy, xa, xb, g1, spec, alpha=0.5, accumulate=True)
    monkeypatch.setattr(ops, 'WGRAD_WINO', False)
    gd = g0.clone()
    ops.conv_wgrad(dy, xa, xb, gd, spec, alpha=0.5, accumulate=True)
    e_w, e_1, e_d = relerr(gw - g0, ref), relerr(g1 - g0, ref), relerr(gd - g0, ref)
    report['wgrad/winograd_f3x3_2x2/%d_%d_%d_%s' % (N, C1 + C2, Cout, H)] = dict(wino2d=e_w, wino1d=e_1, direct=e_d, splits=e_split,
                                                                             run_to_run_equal=bool(torch.equal(gw, gw2)))
    assert torch.equal(gw, gw2)
    assert e_w < 5e-6 and max(e_split.values()) < 5e-6, (e_w, e_split, e_1, e_d)
