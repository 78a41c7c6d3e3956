"""End-to-end GPU parity: HIP engine vs the oracle and vs the reference's golden vectors.

Bars: prune masks (integer channel indices) bit-exact; fp32 tensors within the tolerance written at each assert
(fp32 kernels with a different summation order than ATen-CPU: 1e-5 relative per tensor for single passes,
2e-5 for accumulated gradients)."""
import os

import numpy as np
import pytest
import torch

import golden_common as gc
from conftest import isolated, run_isolated
from helpers import load_json, load_npz, make_model, oracle_params, pkg, relerr, oracle_prune_replay

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _inputs(B, H, s1=1, s2=2):
    return (torch.from_numpy(gc.det_clean((B, 3, H, H), s1)), torch.from_numpy(gc.det_noise((B, 3, H, H), s2)))


class _direct_kernels:
    """`with _direct_kernels():` -- every convolution on the direct implicit-GEMM kernels (what DP_WINO=0 DP_WGRAD_WINO=0 select):
    the full-size tests compare the masks of the default (Winograd F(2, 3)) dispatch against this run."""

    def __enter__(self):
        ops = pkg('ops')
        self.saved = (ops.WINO, ops.WGRAD_WINO)
        ops.WINO = ops.WGRAD_WINO = False

    def __exit__(self, *exc):
        ops = pkg('ops')
        ops.WINO, ops.WGRAD_WINO = self.saved
        return False


def test_tiny_forward_matches_reference_and_oracle(report):
    from oracle import diffusion_ref as D, unet_ref as U
    cfg = gc.TINY_CFG
    g = load_npz('tiny_unet.npz')
    model = make_model(cfg, 5)
    sched = pkg('diffusion').DDPMScheduler()
    clean, noise = _inputs(2, 16)
    t = torch.tensor([3, 500])
    noisy = sched.add_noise(clean.to(DEV), noise.to(DEV), t.to(DEV))
    with torch.no_grad():
        y = model(noisy, t.to(DEV)).sample
    e_ref = float((y.cpu() - torch.from_numpy(g['fwd_out'])).abs().max())
    P = oracle_params(cfg, 5, requires_grad=False)
    yo = U.unet_forward(P, cfg, D.add_noise(D.alphas_cumprod(), clean, noise, t), t)
    e_or = float((y.cpu() - yo).abs().max())
    report['e2e/tiny_fwd'] = dict(vs_reference_abs=e_ref, vs_oracle_abs=e_or)
    assert e_ref < 1e-5 and e_or < 1e-5          # |y| ~ 1: 1e-5 absolute


def _run_sweep(model, clean, noise, steps, thr=None):
    sweep = pkg('sweep')
    sched = pkg('diffusion').DDPMScheduler()
    return sweep.taylor_sweep(model, sched, clean.to(DEV), noise.to(DEV), num_steps=steps, thr=thr)


def test_tiny_sweep_gradients_match_reference(report):
    cfg = gc.TINY_CFG
    g = load_npz('tiny_unet.npz')
    fx = load_json('tiny_prune.json')
    model = make_model(cfg, 5)
    clean, noise = _inputs(2, 16)
    res = _run_sweep(model, clean, noise, 4)
    assert res['steps'] == 4
    e_loss = max(abs(a - b) / b for a, b in zip(res['losses'], g['losses']))
    worst = 0.0
    P = dict(model.named_parameters())
    for k in g.files:
        if k.startswith('grad::'):
            e = relerr(P[k[6:]].grad, g[k])
            worst = max(worst, e)
            assert e < 2e-5, (k, e)
    bad = []
    for n, (s, a, q) in fx['grad_stats'].items():
        gr = P[n].grad.double()
        if abs(float(gr.abs().sum()) - a) > 5e-5 * a + 1e-8 * gr.numel():
            bad.append((n, float(gr.abs().sum()), a))
    report['e2e/tiny_sweep'] = dict(loss_rel=e_loss, grad_rel_worst=worst, n_bad_stats=len(bad))
    assert e_loss < 1e-5 and not bad, bad[:5]


def _compare_prune(model, fx, report, key):
    sweep = pkg('sweep')
    pr = sweep.prune_model(model, 0.3)
    assert len(pr.records) == len(fx['prune'])
    worst_score, min_margin, mism = 0.0, 1e9, []
    from oracle import pruning_ref as R
    for (root, chg, score, pruned), ref in zip(pr.records, fx['prune']):
        assert root == ref['root'] and chg == ref['ch_groups']
        rs = torch.from_numpy(gc.b64_to_f32(ref['score']))
        worst_score = max(worst_score, relerr(score, rs))
        min_margin = min(min_margin, R.decision_margin(rs, ref['pruned'], ref['cur'], ref['ch_groups']))
        if pruned != ref['pruned']:
            mism.append(root)
    report[key] = dict(groups=len(pr.records), score_rel_worst=worst_score, min_decision_margin=min_margin,
                       mask_mismatches=mism)
    assert not mism, mism                                  # bit-exact integer masks, every group
    assert worst_score < 1e-4
    assert {n: list(p.shape) for n, p in model.named_parameters()} == fx['shapes_after']
    assert sum(p.numel() for p in model.parameters()) == fx['params_after']


def test_tiny_prune_masks_bit_exact_and_post_prune_forward(report):
    cfg = gc.TINY_CFG
    fx = load_json('tiny_prune.json')
    model = make_model(cfg, 5)
    clean, noise = _inputs(2, 16)
    _run_sweep(model, clean, noise, 4)
    _compare_prune(model, fx, report, 'e2e/tiny_prune')
    sched = pkg('diffusion').DDPMScheduler()
    t = torch.tensor([3, 500], device=DEV)
    with torch.no_grad():
        y2 = model(sched.add_noise(clean.to(DEV), noise.to(DEV), t), t).sample
    e = float((y2.cpu() - torch.from_numpy(gc.b64_to_f32(fx['fwd_after']))).abs().max())
    report['e2e/tiny_prune']['fwd_after_abs'] = e
    assert e < 1e-5


@pytest.mark.parametrize('crit', ['full1', 'full2', 'abs', 'fisher', 'magnitude'])
def test_sibling_criteria_masks_bit_exact(report, crit):
    """FullTaylor(order 1/2) / AbsTaylor / Fisher / Magnitude importance (ddpm_exp/prune.py:193-208) on the HIP reductions:
    every group's mask identical to the vendored reference classes, scores within fp32 tolerance."""
    pruning, sweep = pkg('pruning'), pkg('sweep')
    fx = load_json('tiny_criteria.json')[crit]
    model = make_model(gc.TINY_CFG, 5)
    clean, noise = _inputs(2, 16)
    _run_sweep(model, clean, noise, 4)
    imp = dict(full1=lambda: pruning.FullTaylorImportance(order=1), full2=lambda: pruning.FullTaylorImportance(order=2),
               abs=pruning.AbsTaylorImportance, fisher=pruning.FisherImportance, magnitude=pruning.MagnitudeImportance)[crit]()
    pr = sweep.prune_model(model, 0.3, importance=imp)
    assert len(pr.records) == len(fx['groups'])
    worst, mism = 0.0, []
    for (root, chg, score, pruned), ref in zip(pr.records, fx['groups']):
        assert root == ref['root'] and chg == ref['ch_groups']
        worst = max(worst, relerr(score, torch.from_numpy(gc.b64_to_f32(ref['score']))))
        if pruned != [i for a, b in ref['pruned'] for i in range(a, b)]:
            mism.append(root)
    report['e2e/criterion_' + crit] = dict(groups=len(pr.records), score_rel_worst=worst, mask_mismatches=mism)
    assert not mism, mism
    assert worst < 1e-4
    assert sum(p.numel() for p in model.parameters()) == fx['params_after']


def test_ddpm_exp_sweep_flavour_break_before_backward(report):
    """ddpm_exp/prune.py:236-258 on the HIP engine: loss summed over C,H,W, threshold test before the backward (the breaking
    step contributes no gradient), original-DDPM checkpoint layout -- against the reference twin's recorded run."""
    ckpt, sweep, unet = pkg('checkpoint'), pkg('sweep'), pkg('unet')
    fx = load_json('ddpm_original.json')
    sw, c = fx['sweep'], fx['cfg']
    cfg = ckpt.unet2d_config_from_ddpm_original(c['ch'], c['ch_mult'], c['num_res_blocks'], c['attn_resolutions'], c['image_size'])
    orig = {n: torch.from_numpy(gc.det_param(n, tuple(s), fx['seed'])) for n, s in fx['shapes'].items()}
    kmap = ckpt.ddpm_original_key_map(orig.keys())
    model = unet.UNet2DModel(**cfg)
    model.load_state_dict(ckpt.convert_ddpm_original(orig))
    model = model.to(DEV).eval()
    clean = torch.from_numpy(gc.det_clean((2, 3, 16, 16), sw['clean_seed'])).to(DEV)
    noise = torch.from_numpy(gc.det_noise((2, 3, 16, 16), sw['noise_seed'])).to(DEV)
    res = sweep.taylor_sweep(model, pkg('diffusion').DDPMScheduler(), clean, noise, num_steps=1000, thr=sw['thr'],
                             loss_kind='sum', accumulate_breaking_step=False)
    e_loss = max(abs(a - b) / b for a, b in zip(res['losses'], sw['losses']))
    P = dict(model.named_parameters())
    bad = []
    for on, (s, a) in sw['grad_stats'].items():
        g = P[kmap[on]].grad.double()
        if abs(float(g.abs().sum()) - a) > 5e-5 * a + 1e-5 * g.numel():
            bad.append(on)
    report['e2e/ddpm_exp_sweep'] = dict(steps=res['steps'], ref_steps=len(sw['losses']), loss_rel=e_loss, bad=bad)
    assert res['steps'] == len(sw['losses']) and e_loss < 2e-5 and not bad, bad[:5]


def test_diff_pruning_early_exit_step(report):
    cfg = gc.TINY_CFG
    fx = load_json('tiny_prune.json')['early_exit']
    model = make_model(cfg, 5)
    clean, noise = _inputs(2, 16)
    res = _run_sweep(model, clean, noise, 1000, thr=fx['thr'])
    report['e2e/early_exit'] = dict(steps=res['steps'], ref_steps=fx['steps'])
    assert res['steps'] == fx['steps']                     # stops at exactly the reference's timestep
    assert np.allclose(res['losses'], fx['losses'], rtol=1e-5)


def test_ddim_sampling_matches_reference(report):
    g = load_npz('ddim.npz')
    diffusion = pkg('diffusion')
    model = make_model(gc.TINY_CFG, 5)
    sch = diffusion.DDIMScheduler()
    sch.set_timesteps(100)
    x = torch.from_numpy(gc.det_noise((2, 3, 16, 16), 21)).to(DEV)
    errs = []
    with torch.no_grad():
        for i, t in enumerate(sch.timesteps[:5]):
            x = sch.step(model(x, t).sample, t, x, eta=0.0).prev_sample
            errs.append(float((x.cpu() - torch.from_numpy(g['x_steps'][i])).abs().max()))
    # full pipeline call with a seeded CPU generator: same x_T as the reference's randn_tensor path
    pipe = diffusion.DDIMPipeline(model, diffusion.DDIMScheduler())
    sch2 = pipe.scheduler
    sch2.set_timesteps(10)
    x = torch.from_numpy(gc.det_noise((2, 3, 16, 16), 22)).to(DEV)
    with torch.no_grad():
        for t in sch2.timesteps:
            x = sch2.step(model(x, t).sample, t, x).prev_sample
    img = (x / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).cpu()
    e_img = float((img - torch.from_numpy(g['chain10_image'])).abs().max())
    report['e2e/ddim'] = dict(step_abs=errs, chain10_image_abs=e_img)
    assert max(errs) < 5e-5 and e_img < 2e-4               # same tolerances as the oracle-vs-reference test
    out = pipe(batch_size=2, generator=torch.Generator().manual_seed(0), num_inference_steps=4, output_type='numpy')
    assert out.images.shape == (2, 16, 16, 3) and np.isfinite(out.images).all()


def test_cifar_c1_masks_bit_exact(report):
    """Config C1 of BASELINE.json: CIFAR-10 UNet, B=4, 8 timesteps, Taylor ratio 0.3 -> every pruned index list equals
    the reference's; 35.75 M -> 19 851 157 parameters."""
    cfg = gc.CIFAR_CFG
    fx = load_json('cifar_c1.json')
    model = make_model(cfg, 0)
    clean, noise = _inputs(4, 32)
    res = _run_sweep(model, clean, noise, 8)
    e_loss = max(abs(a - b) / b for a, b in zip(res['losses'], fx['losses']))
    P = dict(model.named_parameters())
    bad = []
    for n, (s, a, q) in fx['grad_stats'].items():
        gr = P[n].grad.double()
        if abs(float(gr.abs().sum()) - a) > 5e-5 * a + 1e-8 * gr.numel():
            bad.append((n, float(gr.abs().sum()), a))
    report['e2e/c1'] = dict(loss_rel=e_loss, n_bad_grad_stats=len(bad))
    assert e_loss < 1e-5 and not bad, bad[:5]
    _compare_prune(model, fx, report, 'e2e/c1_prune')
    assert sum(p.numel() for p in model.parameters()) == 19851157


def test_full_size_determinism_and_shard_linearity(report):
    """BASELINE.json configs[1] at its full size (CIFAR-10 UNet, batch 256), through size-independent properties:
    (a) the sweep is run-to-run bit-identical (fixed-order reductions, also with the weight-gradient stream),
    (b) gradients are linear in the batch shards: two 128-image shards, each scaled for the global batch exactly as the
        data-parallel path scales them, sum to the full-batch gradients (the property the one all-reduce relies on),
    (c) the prune masks from the summed shard gradients equal the full-batch masks, group by group."""
    sweep, diffusion = pkg('sweep'), pkg('diffusion')
    cfg, B, steps = gc.CIFAR_CFG, 256, 2
    clean, noise = _inputs(B, 32, 11, 12)
    clean, noise = clean.to(DEV), noise.to(DEV)
    sched = diffusion.DDPMScheduler()

    def run(lo, hi, n_steps=steps):
        model = make_model(cfg, 0)
        flat = sweep.flatten_grads(model)
        step = sweep.HipSweepStep(model, sched, clean[lo:hi], noise[lo:hi], B * clean[0].numel(), 'mse', B)
        res = sweep.taylor_sweep(model, sched, clean[lo:hi], noise[lo:hi], num_steps=n_steps, step_fn=step, flat_grads=flat)
        torch.cuda.synchronize()
        return model, flat, res['losses']

    m_full, g_full, l_full = run(0, B)
    m_again, g_again, l_again = run(0, B)
    assert torch.equal(g_full, g_again) and l_full == l_again                 # (a)
    del m_again, g_again
    m1, g1, l1 = run(0, B // 2)
    m2, g2, l2 = run(B // 2, B)
    e_loss = max(abs((a + b) - c) / c for a, b, c in zip(l1, l2, l_full))
    e_grad = relerr(g1 + g2, g_full)
    g1.add_(g2)                                                               # what the all-reduce leaves on every rank
    pr_full = sweep.prune_model(m_full, 0.3)
    pr_sum = sweep.prune_model(m1, 0.3)
    mism = [a[0] for a, b in zip(pr_full.records, pr_sum.records) if a[3] != b[3]]
    report['e2e/full_size'] = dict(loss_rel=e_loss, shard_grad_rel=e_grad, groups=len(pr_full.records), mask_mismatches=mism)
    assert e_loss < 1e-5 and e_grad < 2e-5                                    # (b): fp32 re-association only
    assert len(pr_full.records) == len(pr_sum.records) == 50 and not mism     # (c)
    assert sum(p.numel() for p in m_full.parameters()) == sum(p.numel() for p in m1.parameters())


def test_autograd_bridge_and_finetune_step(report):
    """`loss.backward()` through the one-node autograd bridge equals the sweep engine; one finetune step equals the
    oracle's clip+Adam+EMA arithmetic."""
    from oracle import diffusion_ref as D
    cfg = gc.TINY_CFG
    diffusion, train = pkg('diffusion'), pkg('train')
    model = make_model(cfg, 5)
    model.train()
    sched = diffusion.DDPMScheduler()
    clean, noise = _inputs(4, 16, 3, 4)
    t = torch.tensor([1, 250, 500, 998])
    # autograd bridge
    noisy = sched.add_noise(clean.to(DEV), noise.to(DEV), t.to(DEV))
    out = model(noisy, t.to(DEV)).sample
    loss = (noise.to(DEV) - out).square().sum(dim=(1, 2, 3)).mean(dim=0)
    loss.backward()
    P = oracle_params(cfg, 5)
    lo = D.finetune_loss(P, cfg, clean, noise, t)
    lo.backward()
    worst = 0.0
    for n, p in model.named_parameters():
        if P[n].grad.abs().max() > 1e-6:
            worst = max(worst, relerr(p.grad, P[n].grad))
    e_l = abs(float(loss.detach()) - float(lo.detach())) / float(lo.detach())
    # finetune engine, 2 steps
    model2 = make_model(cfg, 5)
    eng = train.FinetuneEngine(model2, sched, lr=2e-4, ema_decay=0.9999)
    P2 = oracle_params(cfg, 5)
    names = list(P2)
    m = [torch.zeros_like(P2[n]) for n in names]
    v = [torch.zeros_like(P2[n]) for n in names]
    ema = [P2[n].detach().clone() for n in names]
    worst_g = 0.0
    for step in (1, 2):
        l_gpu = eng.step(clean.to(DEV), noise.to(DEV), t.to(DEV))
        for n in names:
            P2[n].grad = None
        l_cpu = D.finetune_loss(P2, cfg, clean, noise, t)
        l_cpu.backward()
        gpu_g = {n: p.grad.detach().cpu().clone() for n, p in model2.named_parameters()}   # raw (un-clipped) grads
        for n in names:
            if float(P2[n].grad.abs().max()) > 1e-6:
                worst_g = max(worst_g, relerr(gpu_g[n], P2[n].grad))
        # The optimizer arithmetic is checked on IDENTICAL gradients (Adam's g/(sqrt(v)+eps) turns the fp32 rounding
        # noise of near-zero gradient elements into +-lr steps, so parameters are not comparable across two
        # independently rounded backward passes; gradients are -- see worst_g).
        with torch.no_grad():
            D.adam_ema_step([P2[n] for n in names], [gpu_g[n] for n in names], m, v, ema, step)
    pm = dict(model2.named_parameters())
    e_p = max(relerr(pm[n], P2[n].detach()) for n in names)
    es = eng.ema_state()
    e_e = max(relerr(es[n], e) for n, e in zip(names, ema))
    assert worst_g < 5e-5, worst_g
    report['e2e/finetune'] = dict(bridge_grad_rel=worst, bridge_loss_rel=e_l, param_rel_after2=e_p, ema_rel_after2=e_e,
                                  loss2_rel=abs(float(l_gpu) - float(l_cpu)) / float(l_cpu))
    assert worst < 5e-5 and e_l < 1e-5
    # Adam normalises the gradient: parameters move by ~lr per step, so 1e-5 relative on parameters of size ~0.1-1
    assert e_p < 1e-5 and e_e < 1e-5


def test_pruned_model_sweep_matches_oracle(report):
    """After pruning, channel counts are no longer multiples of 16 and the concat boundaries fall inside K-chunks /
    N-tiles: the straddling variants of the contraction kernels are exercised in forward, dgrad and wgrad."""
    from oracle import diffusion_ref as D
    cfg = gc.TINY_CFG
    model = make_model(cfg, 5)
    clean, noise = _inputs(2, 16)
    _run_sweep(model, clean, noise, 4)
    pkg('sweep').prune_model(model, 0.3)
    P = {n: p.detach().cpu().clone().requires_grad_(True) for n, p in model.named_parameters()}
    res = _run_sweep(model, clean, noise, 2)
    ref = D.taylor_sweep(P, cfg, clean, noise, 2)
    worst = 0.0
    for n, p in model.named_parameters():
        if float(P[n].grad.abs().max()) > 1e-7:
            worst = max(worst, relerr(p.grad, P[n].grad))
    report['e2e/pruned_sweep'] = dict(loss_rel=max(abs(a - b) / b for a, b in zip(res['losses'], ref)), grad_rel_worst=worst,
                                      widths=sorted({int(p.shape[0]) for p in model.parameters()}))
    assert report['e2e/pruned_sweep']['loss_rel'] < 1e-5 and worst < 2e-5


def test_bedroom_topology_sweep_matches_oracle(report):
    """6-level bedroom/church topology (attention in the 5th down / 2nd up block, 18 skips) at reduced width, 64x64."""
    from oracle import diffusion_ref as D
    cfg = dict(gc.BEDROOM_CFG, block_out_channels=[32, 32, 64, 64, 128, 128], sample_size=64)
    model = make_model(cfg, 7)
    clean, noise = _inputs(2, 64, 11, 12)
    res = _run_sweep(model, clean, noise, 2)
    P = oracle_params(cfg, 7)
    ref = D.taylor_sweep(P, cfg, clean, noise, 2)
    worst = 0.0
    for n, p in model.named_parameters():
        if float(P[n].grad.abs().max()) > 1e-7:
            worst = max(worst, relerr(p.grad, P[n].grad))
    report['e2e/bedroom_topology'] = dict(loss_rel=max(abs(a - b) / b for a, b in zip(res['losses'], ref)), grad_rel_worst=worst)
    assert report['e2e/bedroom_topology']['loss_rel'] < 1e-5 and worst < 2e-5


def test_ldm_unet_forward_backward_matches_reference(report):
    """Row a17: the LDM (CompVis) UNet on the HIP engine vs the reference's own UNetModel (golden fixtures)."""
    ldm = pkg('ldm')
    ops = pkg('ops')
    cfg = gc.LDM_TINY_CFG
    g = load_npz('ldm_unet.npz')
    fx = load_json('ldm_unet_stats.json')
    model = ldm.UNetModel(**cfg)
    gc.det_init_(model, 9)
    model = model.to(DEV).eval()
    x = torch.from_numpy(gc.det_noise((2, 3, 16, 16), 31)).to(DEV)
    ctx = torch.from_numpy(gc.det_noise((2, 1, 16), 32)).to(DEV)
    noise = torch.from_numpy(gc.det_noise((2, 3, 16, 16), 33)).to(DEV)
    t = torch.tensor([7, 640], device=DEV)
    eng = model.engine()
    grads = {n: torch.zeros_like(p) for n, p in model.named_parameters()}
    eng.bind(eng.P, grads)
    y = eng.forward(x, t, ctx, save=True)
    e_f = float((y.cpu() - torch.from_numpy(g['fwd_out'])).abs().max())
    per = y[0].numel()
    loss, dout = ops.mse_fwd_bwd(y, noise, 2.0 / (per * 2), 1.0 / (per * 2))        # mean_B(mean_CHW) for B = 2
    eng.backward(dout)
    e_l = abs(float(loss) - float(g['loss'])) / float(g['loss'])
    worst = 0.0
    for k in g.files:
        if k.startswith('grad::') and float(np.abs(g[k]).max()) > 0:
            worst = max(worst, relerr(grads[k[6:]], g[k]))
    bad = []
    for n, (s, a) in fx['grad_stats'].items():
        got = float(grads[n].double().abs().sum())
        if abs(got - a) > 5e-5 * a + 1e-8 * grads[n].numel():
            bad.append((n, got, a))
    report['e2e/ldm_unet'] = dict(fwd_abs=e_f, loss_rel=e_l, grad_rel_worst=worst, n_bad_stats=len(bad))
    assert e_f < 1e-5 and e_l < 1e-5 and worst < 2e-5 and not bad, bad[:5]
    # module-level forward (sampling path) agrees with the engine forward
    # (a no-grad forward keeps nothing for a backward and takes the fused attention kernel where it is the faster one -- round 4
    # default --, the forward above materialises the scores: same function, different summation order inside the attention)
    with torch.no_grad():
        y2 = model(x, t, context=ctx)
    assert float((y2 - y).abs().max()) < 1e-5
    monkeypatch_fa = ops.FUSED_ATTN
    ops.FUSED_ATTN = False
    try:
        with torch.no_grad():
            y3 = model(x, t, context=ctx)
    finally:
        ops.FUSED_ATTN = monkeypatch_fa
    assert torch.equal(y3, y)


def _ldm_model_with_grads():
    ldm, ops = pkg('ldm'), pkg('ops')
    sweep = pkg('sweep')
    cfg = gc.LDM_TINY_CFG
    model = ldm.UNetModel(**cfg)
    gc.det_init_(model, 9)
    model = model.to(DEV).eval()
    sweep.flatten_grads(model)
    x = torch.from_numpy(gc.det_noise((2, 3, 16, 16), 31)).to(DEV)
    ctx = torch.from_numpy(gc.det_noise((2, 1, 16), 32)).to(DEV)
    noise = torch.from_numpy(gc.det_noise((2, 3, 16, 16), 33)).to(DEV)
    t = torch.tensor([7, 640], device=DEV)
    eng = model.engine()
    eng.bind(eng.P, {n: p.grad for n, p in model.named_parameters()})
    y = eng.forward(x, t, ctx, save=True)
    n = y.numel()
    loss, dout = ops.mse_fwd_bwd(y, noise, 2.0 / n, 1.0 / n)
    eng.backward(dout)
    return model, (x, ctx, noise, t)


def test_ldm_prune_masks_bit_exact(report):
    """Row a17: scores + masks of the LDM UNet (109 groups, round_to=2, head channel groups) vs the reference."""
    ldm, pruning = pkg('ldm'), pkg('pruning')
    from oracle import pruning_ref as R
    fx = load_json('ldm_prune.json')
    model, (x, ctx, noise, t) = _ldm_model_with_grads()
    channel_groups = {}
    for m in model.modules():
        if isinstance(m, ldm.CrossAttention):
            channel_groups[m.to_q] = channel_groups[m.to_k] = channel_groups[m.to_v] = m.heads
    pr = pruning.MagnitudePruner(model, None, importance=pruning.TaylorImportance(), iterative_steps=1,
                                 channel_groups=channel_groups, ch_sparsity=0.3, ignored_layers=[model.out], round_to=2)
    for g in pr.step(interactive=True):
        g.prune()
    model._engine.packs.clear()
    assert len(pr.records) == len(fx['prune'])
    mism, worst, margin = [], 0.0, 1e9
    for (root, chg, score, pruned), ref in zip(pr.records, fx['prune']):
        assert root == ref['root'] and chg == ref['ch_groups']
        rs = torch.from_numpy(gc.b64_to_f32(ref['score']))
        worst = max(worst, relerr(score, rs))
        if ref['pruned']:
            margin = min(margin, R.decision_margin(rs, ref['pruned'], ref['cur'], ref['ch_groups']))
        if pruned != ref['pruned']:
            mism.append(root)
    report['e2e/ldm_prune'] = dict(groups=len(pr.records), score_rel_worst=worst, min_decision_margin=margin, mask_mismatches=mism)
    assert not mism and worst < 1e-4
    assert {n: list(p.shape) for n, p in model.named_parameters()} == fx['shapes_after']
    assert sum(p.numel() for p in model.parameters()) == fx['params_after']
    with torch.no_grad():
        y2 = model(x, t, context=ctx)
    e = float((y2.cpu() - torch.from_numpy(gc.b64_to_f32(fx['fwd_after']))).abs().max())
    report['e2e/ldm_prune']['fwd_after_abs'] = e
    assert e < 1e-5


def test_ldm_importance_sweep_matches_oracle(report):
    """prune_ldm.py:101-131 on the GPU (CFG DDIM sampling -> loss at t -> backward) vs the oracle restatement (which is pinned
    against the script's own loop and the reference's LatentDiffusion / DDIMSampler: tests/golden/ldm_driver.json,
    ldm_loss_at_t.npz, ldm_sampler.npz -- tests/test_cpu.py)."""
    from oracle import ldm_ref as L
    ldm, ldm_sweep = pkg('ldm'), pkg('ldm_sweep')
    cfg = gc.LDM_TINY_CFG
    emb_w = torch.from_numpy(gc.det_noise((1001, 16), 77))
    rng = np.random.default_rng(5)
    draws = [(torch.tensor(rng.choice(1000, size=2, replace=False)), torch.from_numpy(gc.det_noise((2, 3, 16, 16), 300 + t)),
              torch.from_numpy(gc.det_noise((2, 3, 16, 16), 400 + t))) for t in range(3)]
    model = ldm.UNetModel(**cfg)
    gc.det_init_(model, 9)
    model = model.to(DEV).eval()
    embedder = ldm_sweep.ClassEmbedder(16, 1001)
    with torch.no_grad():
        embedder.embedding.weight.copy_(emb_w)
    embedder = embedder.to(DEV)
    res = ldm_sweep.ldm_importance_sweep(model, embedder, num_steps=3, thr=None, n_samples=2, ddim_steps=4,
                                         latent_shape=(3, 16, 16), draws=lambda t: draws[t])
    P = {k: torch.from_numpy(gc.det_param(k, s, 9)).requires_grad_(True) for k, s in L.ldm_param_shapes(cfg).items()}
    acp = L.ldm_alphas_cumprod()
    uc = emb_w[torch.tensor([1000, 1000])][:, None, :]
    losses = []
    for t, (xc, x_T, noise) in enumerate(draws):
        c = emb_w[xc][:, None, :]
        x0 = L.ddim_sample_cfg({k: v.detach() for k, v in P.items()}, cfg, acp, x_T, c, uc, S=4, scale=3.0)
        loss = L.ldm_loss_at_t(P, cfg, acp, x0, torch.full((2,), t, dtype=torch.long), c, noise)
        losses.append(float(loss.detach()))
        loss.backward()
    worst = 0.0
    for k, p in model.named_parameters():
        if float(P[k].grad.abs().max()) > 1e-7:
            worst = max(worst, relerr(p.grad, P[k].grad))
    e_l = max(abs(a - b) / abs(b) for a, b in zip(res['losses'], losses))
    report['e2e/ldm_sweep'] = dict(loss_rel=e_l, grad_rel_worst=worst, steps=res['steps'])
    # 4 CFG-DDIM steps (guidance scale 3 amplifies rounding) feed the loss: 1e-4 on losses / gradients
    assert res['steps'] == 3 and e_l < 1e-4 and worst < 2e-4


def test_ldm_two_importance_steps_in_flight(report):
    """ldm_importance_sweep(pipelines=2): odd steps on a second engine / stream / gradient buffer.  Losses and the stop step are
    those of the sequential loop bit for bit (the loss test is ordered across the pipelines); the gradient is the same sum
    re-associated (even steps + odd steps).  Case 2 breaks in the middle (threshold 0.995 on a loss sequence that is not
    monotone): the breaking step and the one enqueued behind it add nothing, the steps before it are kept."""
    ldm, ldm_sweep = pkg('ldm'), pkg('ldm_sweep')
    cfg = gc.LDM_TINY_CFG
    emb_w = torch.from_numpy(gc.det_noise((1001, 16), 77))
    out = {}
    for thr in (None, 0.995):
        for pipelines in (1, 2):
            model = ldm.UNetModel(**cfg)
            gc.det_init_(model, 9)
            model = model.to(DEV).eval()
            embedder = ldm_sweep.ClassEmbedder(16, 1001)
            with torch.no_grad():
                embedder.embedding.weight.copy_(emb_w)
            embedder = embedder.to(DEV)
            res = ldm_sweep.ldm_importance_sweep(model, embedder, num_steps=7, thr=thr, n_samples=2, ddim_steps=4, seed=3,
                                                 latent_shape=(3, 16, 16), pipelines=pipelines)
            torch.cuda.synchronize()
            out[thr, pipelines] = (res['losses'], res['steps'], res['accumulated'], res['flat_grads'].clone())
        a, b = out[thr, 1], out[thr, 2]
        assert a[0] == b[0] and a[1:3] == b[1:3], (thr, a[:3], b[:3])
        assert relerr(b[3], a[3]) < 1e-5, thr
    assert out[None, 1][1] == 7
    assert 1 < out[0.995, 1][1] < 7, out[0.995, 1][:3]            # the break is in the middle, on either pipeline
    report['e2e/ldm_pipelines'] = dict(steps_full=out[None, 2][1], steps_thr=out[0.995, 2][1],
                                       grad_rel=relerr(out[None, 2][3], out[None, 1][3]))


def test_hipgraph_sweep_matches_eager(report):
    """One captured timestep replayed per t gives bit-identical losses and gradients to the eager launch sequence."""
    import time
    cfg = gc.CIFAR_CFG
    clean, noise = _inputs(4, 32)
    sweep = pkg('sweep')
    sched = pkg('diffusion').DDPMScheduler()
    out = {}
    for mode in (False, True):
        model = make_model(cfg, 0)
        sweep.taylor_sweep(model, sched, clean.to(DEV), noise.to(DEV), num_steps=1, use_graph=mode)     # warm
        model = make_model(cfg, 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = sweep.taylor_sweep(model, sched, clean.to(DEV), noise.to(DEV), num_steps=8, use_graph=mode)
        torch.cuda.synchronize()
        out[mode] = (res['losses'], torch.cat([p.grad.reshape(-1) for p in model.parameters()]).clone(), time.perf_counter() - t0)
    assert out[False][0] == out[True][0]
    assert torch.equal(out[False][1], out[True][1])
    report['e2e/hipgraph'] = dict(eager_s=out[False][2], graph_s=out[True][2], config='C1: CIFAR UNet B=4, 8 timesteps (incl. capture)')


# ------------------------------------------------------------------------------------------------------------------
# round 2
# ------------------------------------------------------------------------------------------------------------------
def _mask_report(pr, refs, expand_ranges=False):
    """(mismatching roots, worst score error, smallest decision margin) of a prune vs reference records."""
    from oracle import pruning_ref as R
    mism, worst, margin = [], 0.0, 1e9
    assert len(pr.records) == len(refs)
    for (root, chg, score, pruned), ref in zip(pr.records, refs):
        assert root == ref['root'] and chg == ref['ch_groups']
        rs = ref['score'] if torch.is_tensor(ref['score']) else torch.from_numpy(gc.b64_to_f32(ref['score']))
        worst = max(worst, relerr(score, rs))
        want = gc.expand(ref['pruned']) if expand_ranges else ref['pruned']
        margin = min(margin, R.decision_margin(rs, want, ref['cur'], ref['ch_groups']))
        if pruned != want:
            mism.append(root)
    return mism, worst, margin


def test_long_sweep_1000_steps_matches_reference(report):
    """The accumulation length of config C2: 1000 backward passes accumulated into the gradients (SURVEY §7 names error growth
    over exactly this as the risk to bit-exact masks), tiny UNet, vs a 1000-step run of the reference itself."""
    fx = load_json('tiny_long_sweep.json')
    model = make_model(gc.TINY_CFG, 5)
    clean, noise = _inputs(2, 16)
    res = _run_sweep(model, clean, noise, 1000)
    assert res['steps'] == 1000
    e_loss = max(abs(a - b) / b for a, b in zip(res['losses'], fx['losses']))
    P = dict(model.named_parameters())
    bad, worst_stat = [], 0.0
    for n, (s, a, q) in fx['grad_stats'].items():
        if n.endswith('to_k.bias'):          # exactly zero in exact arithmetic (softmax is shift-invariant along the keys)
            continue
        got = float(P[n].grad.double().abs().sum())
        worst_stat = max(worst_stat, abs(got - a) / max(a, 1e-30))
        if abs(got - a) > 5e-5 * a + 1e-8 * P[n].grad.numel():
            bad.append((n, got, a))
    pr = pkg('sweep').prune_model(model, 0.3)
    mism, worst, margin = _mask_report(pr, fx['prune'])
    report['e2e/long_sweep_1000'] = dict(loss_rel=e_loss, grad_abs_sum_rel_worst=worst_stat, n_bad_stats=len(bad),
                                         score_rel_worst=worst, min_decision_margin=margin, mask_mismatches=mism,
                                         groups=len(pr.records))
    assert e_loss < 1e-5 and not bad, bad[:5]
    assert not mism, mism                                   # bit-exact masks after 1000 accumulated steps
    assert worst < 1e-4
    assert sum(p.numel() for p in model.parameters()) == fx['params_after']


@isolated()
def test_c1_size_1000_step_sweep_masks_match_reference(report):
    """Config C2's accumulation length on the C1-size model: CIFAR-10 UNet (35.7 M parameters), B=4, 1000 accumulated
    timesteps, against a 1000-step run of the reference itself (tests/golden/cifar_long_sweep.json): losses, gradient
    statistics, the scores and masks of all 50 groups; the smallest decision margin is reported next to the result."""
    fx = load_json('cifar_long_sweep.json')
    model = make_model(gc.CIFAR_CFG, 0)
    clean, noise = _inputs(4, 32)
    res = _run_sweep(model, clean, noise, 1000)
    assert res['steps'] == 1000 and len(fx['losses']) == 1000
    e_loss = max(abs(a - b) / b for a, b in zip(res['losses'], fx['losses']))
    P = dict(model.named_parameters())
    bad, worst_stat = [], 0.0
    for n, (s, a, q) in fx['grad_stats'].items():
        if n.endswith('to_k.bias'):          # exactly zero in exact arithmetic (softmax is shift-invariant along the keys)
            continue
        got = float(P[n].grad.double().abs().sum())
        worst_stat = max(worst_stat, abs(got - a) / max(a, 1e-30))
        if abs(got - a) > 5e-5 * a + 1e-8 * P[n].grad.numel():
            bad.append((n, got, a))
    pr = pkg('sweep').prune_model(model, 0.3)
    mism, worst, margin = _mask_report(pr, fx['prune'])
    report['e2e/c1_size_1000_steps'] = dict(loss_rel=e_loss, grad_abs_sum_rel_worst=worst_stat, n_bad_stats=len(bad),
                                            score_rel_worst=worst, min_decision_margin=margin, mask_mismatches=mism,
                                            groups=len(pr.records))
    assert e_loss < 1e-5 and not bad, bad[:5]
    assert not mism, mism                                   # bit-exact masks after 1000 accumulated steps
    assert worst < 1e-4
    assert sum(p.numel() for p in model.parameters()) == fx['params_after']


@isolated()
def test_c3_bedroom256_full_size(report):
    """BASELINE.json configs[2] at its real size: google/ddpm-ema-bedroom-256 topology (113.7 M parameters), 256x256 images,
    4 images per GPU (batch 32 over 8 GPUs), Diff-Pruning threshold 0.05:
      (a) run-to-run bit-identity of the sweep (split GroupNorm, workgroup-per-plane row sums, wgrad stream),
      (b) two 2-image shards scaled for the global batch sum to the 4-image gradients; (c) same masks from the summed shards,
      (d) micro-batched (2 + 2) == un-batched on the GPU, same early-exit step,
      (e) one-image steps against the oracle on the host cores, including the early-exit decision."""
    import copy
    from oracle import diffusion_ref as D
    sweep, diffusion = pkg('sweep'), pkg('diffusion')
    cfg = gc.BEDROOM_CFG
    B, H, steps = 4, 256, 2
    model = make_model(cfg, 0)
    assert sum(p.numel() for p in model.parameters()) == 113673219
    clean, noise = _inputs(B, H, 11, 12)
    clean, noise = clean.to(DEV), noise.to(DEV)
    sched = diffusion.DDPMScheduler()
    per = clean[0].numel()

    def run(lo, hi, thr=0.05, micro=None, n_steps=steps, global_b=B):
        flat = sweep.flatten_grads(model)
        step = sweep.HipSweepStep(model, sched, clean[lo:hi], noise[lo:hi], global_b * per, 'mse', global_b)
        step.micro = micro
        res = sweep.taylor_sweep(model, sched, clean[lo:hi], noise[lo:hi], num_steps=n_steps, thr=thr, step_fn=step,
                                 flat_grads=flat)
        torch.cuda.synchronize()
        return flat, res

    g_full, r_full = run(0, B)
    g_again, r_again = run(0, B)
    assert r_full['steps'] == steps                                                    # 0.05 does not trigger in 2 steps
    assert torch.equal(g_full, g_again) and r_full['losses'] == r_again['losses']      # (a)
    del g_again
    with _direct_kernels():                                                            # (f) the same sweep on the direct kernels only
        model._engine.packs.clear()
        g_direct, r_direct = run(0, B)
    model._engine.packs.clear()
    e_wino_grad = relerr(g_full, g_direct)
    e_wino_loss = max(abs(a - b) / b for a, b in zip(r_full['losses'], r_direct['losses']))
    g_micro, r_micro = run(0, B, micro=2)                                              # (d)
    e_micro = relerr(g_micro, g_full)
    assert r_micro['steps'] == r_full['steps'] and np.allclose(r_micro['losses'], r_full['losses'], rtol=1e-6)
    del g_micro
    g1, r1 = run(0, B // 2)
    g2, r2 = run(B // 2, B)                                                            # model.grad now views g2
    e_loss = max(abs((a + b) - c) / c for a, b, c in zip(r1['losses'], r2['losses'], r_full['losses']))
    e_shard = relerr(g1 + g2, g_full)
    g2.add_(g1)                                                                        # what the all-reduce leaves
    del g1
    from oracle import pruning_ref as R
    m_full, m_direct = copy.deepcopy(model), copy.deepcopy(model)
    flat_f = sweep.flatten_grads(m_full)
    flat_f.copy_(g_full)
    sweep.flatten_grads(m_direct).copy_(g_direct)
    pr_sum = sweep.prune_model(model, 0.3)
    pr_full = sweep.prune_model(m_full, 0.3)
    pr_direct = sweep.prune_model(m_direct, 0.3)
    mism = [a[0] for a, b in zip(pr_full.records, pr_sum.records) if a[3] != b[3]]
    mism_direct = [a[0] for a, b in zip(pr_full.records, pr_direct.records) if a[3] != b[3]]
    margin = min(R.decision_margin(sc, pruned, len(sc), chg) for _, chg, sc, pruned in pr_full.records)
    e_score = max(relerr(b[2], a[2]) for a, b in zip(pr_full.records, pr_sum.records))
    e_wino_score = max(relerr(b[2], a[2]) for a, b in zip(pr_full.records, pr_direct.records))
    params_after = sum(p.numel() for p in model.parameters())
    del m_full, m_direct, flat_f, g_full, g_direct, g2, pr_sum, pr_full, pr_direct
    torch.cuda.empty_cache()
    # (e) one image against the oracle, with a threshold that can trigger within the 3 steps run
    model1 = make_model(cfg, 0)
    c1, n1 = _inputs(1, H, 11, 12)
    thr_e = 0.9995
    res1 = sweep.taylor_sweep(model1, sched, c1.to(DEV), n1.to(DEV), num_steps=3, thr=thr_e)
    P = oracle_params(cfg, 0)
    ref = D.taylor_sweep(P, cfg, c1, n1, 3, thr=thr_e)
    e_l1 = max(abs(a - b) / b for a, b in zip(res1['losses'], ref))
    worst = 0.0
    for n, p in model1.named_parameters():
        if float(P[n].grad.abs().max()) > 1e-7:
            worst = max(worst, relerr(p.grad, P[n].grad))
    report['e2e/c3_bedroom256'] = dict(micro_vs_full_grad_rel=e_micro, shard_loss_rel=e_loss, shard_grad_rel=e_shard,
                                       mask_mismatches=mism, groups=71, params_after=params_after,
                                       b1_steps=res1['steps'], b1_ref_steps=len(ref), b1_loss_rel=e_l1, b1_grad_rel_worst=worst,
                                       losses=r_full['losses'], min_decision_margin=margin, shard_score_rel_worst=e_score,
                                       wino_vs_direct_score_rel=e_wino_score, wino_vs_direct_grad_rel=e_wino_grad,
                                       wino_vs_direct_loss_rel=e_wino_loss, wino_vs_direct_mask_mismatches=mism_direct)
    assert e_micro < 1e-5 and e_loss < 1e-5 and e_shard < 2e-5                         # fp32 re-association only
    assert not mism                                                                    # (c)
    # (f) Winograd F(2, 3) vs the direct kernels at full size: same masks for all 71 groups, and the thinnest decision of the run is
    # an order of magnitude outside both the shard re-association and the Winograd-vs-direct movement of the scores
    assert not mism_direct and e_wino_loss < 1e-5 and e_wino_grad < 2e-5
    assert margin > 10 * max(e_score, e_wino_score)
    assert res1['steps'] == len(ref) and e_l1 < 1e-5 and worst < 2e-5                  # (e)


def test_multi_head_unet_matches_reference(report):
    """attention_head_dim 8 (4 / 6 / 8 heads: head_to_batch_dim, attention_processor.py:283-305) on the HIP engine against the
    reference UNet2DModel: forward, sweep losses, gradient statistics; then Taylor masks with head-grouped q/k/v selection
    (ldm_prune.py:73-79) against the oracle's prune of the same gradients."""
    from oracle import diffusion_ref as D
    graph, pruning, unet = pkg('graph'), pkg('pruning'), pkg('unet')
    cfg = load_json('groups_more.json')['heads8_4lvl']['cfg']
    fx, g = load_json('tiny_heads.json'), load_npz('tiny_heads.npz')
    model = make_model(cfg, 4)
    assert model._multi_head
    sched = pkg('diffusion').DDPMScheduler()
    clean, noise = _inputs(2, 16, 81, 82)
    t = torch.tensor([5, 700], device=DEV)
    with torch.no_grad():
        y = model(sched.add_noise(clean.to(DEV), noise.to(DEV), t), t).sample
    e_f = float((y.cpu() - torch.from_numpy(g['fwd_out'])).abs().max())
    res = _run_sweep(model, clean, noise, 2)
    e_loss = max(abs(a - b) / b for a, b in zip(res['losses'], g['losses']))
    Pm = dict(model.named_parameters())
    bad = []
    for n, (s, a, q) in fx['grad_stats'].items():
        got = float(Pm[n].grad.double().abs().sum())
        if abs(got - a) > 5e-5 * a + 1e-8 * Pm[n].grad.numel():
            bad.append((n, got, a))
    # oracle gradients + its prune replay (vendored Taylor criterion, per-head channel groups on q/k/v)
    P = oracle_params(cfg, 4)
    D.taylor_sweep(P, cfg, clean, noise, 2)
    worst = max(relerr(p.grad, P[n].grad) for n, p in model.named_parameters() if float(P[n].grad.abs().max()) > 1e-7)
    channel_groups = {}
    for m in model.modules():
        if isinstance(m, unet.Attention):
            channel_groups[m.to_q] = channel_groups[m.to_k] = channel_groups[m.to_v] = m.heads
    pr = pruning.MagnitudePruner(model, None, importance=pruning.TaylorImportance(), iterative_steps=1,
                                 channel_groups=channel_groups, ch_sparsity=0.3, ignored_layers=[model.conv_out])
    for grp in pr.step(interactive=True):
        grp.prune()
    pruning.fix_static_attributes(model)
    n_head_grouped = sum(1 for r in pr.records if r[1] not in (1, cfg['norm_num_groups']))
    # the pruned multi-head model still runs (inner width no longer equals heads * attention_head_dim)
    with torch.no_grad():
        y2 = model(sched.add_noise(clean.to(DEV), noise.to(DEV), t), t).sample
    from oracle import unet_ref as U
    Pp = {n: p.detach().cpu() for n, p in model.named_parameters()}
    with torch.no_grad():
        yo = U.unet_forward(Pp, cfg, D.add_noise(D.alphas_cumprod(), clean, noise, t.cpu()), t.cpu())
    e_after = float((y2.cpu() - yo).abs().max())
    report['e2e/multi_head'] = dict(fwd_abs=e_f, loss_rel=e_loss, n_bad_stats=len(bad), grad_rel_worst=worst,
                                    groups=len(pr.records), head_grouped_groups=n_head_grouped, fwd_after_prune_abs=e_after)
    assert e_f < 1e-5 and e_loss < 1e-5 and not bad and worst < 2e-5, bad[:3]
    assert n_head_grouped > 0 and e_after < 1e-5


def test_checkpoint_roundtrip_runs_on_the_hip_path(report, tmp_path):
    """SURVEY §8(f) rank 1 on the GPU: a Diffusers directory written by the reference loads and its HIP forward equals the
    recorded reference output; a pruned model saved with save_pruned / torch.save(model) and loaded back gives the identical
    HIP output (ddpm_prune.py:131-135 -> ddpm_train.py:289-300 / ddpm_sample.py:25-33)."""
    import os
    ckpt, diffusion, sweep = pkg('checkpoint'), pkg('diffusion'), pkg('sweep')
    from helpers import GOLD
    src = os.path.join(GOLD, 'pretrained_micro')
    pipe = diffusion.DDPMPipeline.from_pretrained(src).to(DEV)
    x = torch.from_numpy(gc.det_noise((1, 3, 8, 8), 72)).to(DEV)
    with torch.no_grad():
        y = pipe.unet(x, torch.tensor([10], device=DEV)).sample
    want = np.load(os.path.join(src, 'expected.npz'))['fwd_out']
    e_load = float((y.cpu() - torch.from_numpy(want)).abs().max())
    # prune the tiny UNet, save three ways, load, compare HIP outputs bit for bit
    model = make_model(gc.TINY_CFG, 5)
    clean, noise = _inputs(2, 16)
    _run_sweep(model, clean, noise, 2)
    pr = sweep.prune_model(model, 0.3)
    t = torch.tensor([3, 500], device=DEV)
    xin = clean.to(DEV)
    with torch.no_grad():
        y0 = model(xin, t).sample
    d = str(tmp_path / 'pruned')
    ckpt.save_pruned(model, d, pr.pruning_history())
    back = ckpt.load_pruned(d).to(DEV).eval()
    with torch.no_grad():
        y1 = back(xin, t).sample
    torch.save(model, str(tmp_path / 'unet_pruned.pth'))
    whole = torch.load(str(tmp_path / 'unet_pruned.pth'), weights_only=False).to(DEV).eval()
    with torch.no_grad():
        y2 = whole(xin, t).sample
    report['e2e/checkpoint_gpu'] = dict(pretrained_fwd_abs=e_load, pruned_params=sum(p.numel() for p in back.parameters()))
    assert e_load < 1e-5
    assert torch.equal(y0, y1) and torch.equal(y0, y2)


def test_in_place_weight_swap_is_seen_by_the_engine(report):
    """EMAModel.copy_to / restore write weights through `param.data.copy_` (training_utils.py:231,286; used at
    ddpm_train.py:387-401), which bumps no version counter: the packed conv operands must not survive it."""
    from oracle import unet_ref as U
    cfg = gc.TINY_CFG
    model = make_model(cfg, 5)
    x = torch.from_numpy(gc.det_noise((2, 3, 16, 16), 3)).to(DEV)
    t = torch.tensor([7, 300], device=DEV)
    with torch.no_grad():
        y_a = model(x, t).sample.clone()
    new = {n: torch.from_numpy(gc.det_param(n, tuple(p.shape), 6)) for n, p in model.named_parameters()}
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.data.copy_(new[n].to(DEV))                    # the EMAModel.copy_to write pattern
        y_b = model(x, t).sample
        yo = U.unet_forward(new, cfg, x.cpu(), t.cpu())
    e = float((y_b.cpu() - yo).abs().max())
    # pinned: the caller promises frozen weights, the packs are kept (and a write inside the block is the caller's bug)
    with model.pin_weights():
        with torch.no_grad():
            y_c = model(x, t).sample
    report['e2e/ema_swap'] = dict(fwd_abs_after_swap=e, changed=float((y_b - y_a).abs().max()))
    assert e < 1e-5 and torch.equal(y_b, y_c) and float((y_b - y_a).abs().max()) > 1e-3


def test_ddpm_sampling_matches_reference(report):
    """DDPMScheduler.step + DDPMPipeline (scheduling_ddpm.py:312-406, pipeline_ddpm.py:24-105) on the HIP engine against
    sequences recorded from the reference (variance noise from the same seeded CPU generator)."""
    g = load_npz('ddpm.npz')
    diffusion = pkg('diffusion')
    model = make_model(gc.TINY_CFG, 5)
    errs = {}
    for tag, n_inf, vt in (('full', 1000, 'fixed_small'), ('s50', 50, 'fixed_small'), ('large', 50, 'fixed_large')):
        sch = diffusion.DDPMScheduler(variance_type=vt)
        sch.set_timesteps(n_inf)
        gen = torch.Generator().manual_seed(123)
        x = torch.from_numpy(gc.det_noise((2, 3, 16, 16), 23)).to(DEV)
        worst = 0.0
        with torch.no_grad():
            for i, t in enumerate(sch.timesteps[:4]):
                x = sch.step(model(x, t).sample, t, x, generator=gen).prev_sample
                worst = max(worst, float((x.cpu() - torch.from_numpy(g['x_' + tag][i])).abs().max()))
        errs[tag] = worst
    pipe = diffusion.DDPMPipeline(model, diffusion.DDPMScheduler())
    img = pipe(batch_size=2, generator=torch.Generator().manual_seed(9), num_inference_steps=6, output_type='numpy').images
    errs['pipe6_image'] = float(np.abs(img - g['pipe6']).max())
    report['e2e/ddpm'] = errs
    assert max(errs['full'], errs['s50'], errs['large']) < 5e-5 and errs['pipe6_image'] < 2e-4


def test_dropout_finetune_forward_backward_matches_reference(report):
    """Training-mode loss and gradients with dropout 0.1 on every nn.Dropout (utils.set_dropout) against the reference
    UNet2DModel run with the same (Philox) masks -- tests/golden/tiny_dropout.json -- through FinetuneEngine's step with
    lr 0 (weights untouched), and through the one-node autograd bridge (`loss.backward()`)."""
    train, diffusion = pkg('train'), pkg('diffusion')
    fx = load_json('tiny_dropout.json')
    cfg = gc.TINY_CFG
    model = make_model(cfg, 5)
    sched = diffusion.DDPMScheduler()
    ft = train.FinetuneEngine(model, sched, lr=0.0, dropout=fx['p'], dropout_seed=fx['seed'], use_ema=False)
    assert len(model.dropout_table()) == fx['sites']
    ft.step_count = fx['step'] - 1
    clean, noise = _inputs(4, 16, 3, 4)
    t = torch.tensor([1, 250, 500, 998])
    loss = ft.step(clean.to(DEV), noise.to(DEV), t)
    e_l = abs(float(loss) - fx['loss']) / fx['loss']
    bad, worst = [], 0.0
    for n, p in model.named_parameters():
        if n.endswith('to_k.bias'):          # exactly zero in exact arithmetic (softmax is shift-invariant along the keys):
            continue                         # both sides hold rounding noise only
        s, a, q = fx['grad_stats'][n]
        got = float(p.grad.double().abs().sum())
        worst = max(worst, abs(got - a) / max(a, 1e-30))
        if abs(got - a) > 5e-5 * a + 1e-8 * p.grad.numel():
            bad.append((n, got, a))
    # autograd bridge in train mode: same masks when seed / step agree
    model2 = make_model(cfg, 5)
    train.set_dropout(model2, fx['p'])
    model2.train()
    model2.dropout_seed, model2._dropout_step = fx['seed'], fx['step'] - 1
    noisy = sched.add_noise(clean.to(DEV), noise.to(DEV), t.to(DEV))
    out = model2(noisy, t.to(DEV)).sample
    e_out = float((out.detach().cpu() - torch.from_numpy(gc.b64_to_f32(fx['fwd_out']))).abs().max())
    l2 = (noise.to(DEV) - out).square().sum(dim=(1, 2, 3)).mean(dim=0)
    l2.backward()
    e_bridge = max(relerr(p2.grad, p.grad) for (n, p), (_, p2) in zip(model.named_parameters(), model2.named_parameters())
                   if float(p.grad.abs().max()) > 1e-6)
    report['e2e/dropout_tiny'] = dict(loss_rel=e_l, grad_abs_sum_rel_worst=worst, n_bad=len(bad), fwd_abs=e_out,
                                      bridge_vs_engine_grad_rel=e_bridge)
    assert e_l < 1e-5 and not bad and e_out < 1e-4 and e_bridge < 1e-6, bad[:3]


@isolated()
def test_c4_pruned_cifar_finetune_with_dropout(report):
    """BASELINE.json configs[3] as the reference runs it (scripts/finetune_ddpm_cifar10.sh): the ratio-0.3 PRUNED CIFAR-10 UNet
    (19 851 157 parameters), batch 128 per GPU, dropout 0.1, lr 2e-4, EMA 0.9999, two optimizer steps: loss and raw
    gradients against the oracle with the same masks (5e-5), clip + Adam + EMA on identical gradients (1e-5)."""
    from oracle import diffusion_ref as D, philox_ref as PH
    train, diffusion, sweep = pkg('train'), pkg('diffusion'), pkg('sweep')
    cfg, B = gc.CIFAR_CFG, 128
    model = make_model(cfg, 0)
    clean, noise = _inputs(4, 32)
    _run_sweep(model, clean, noise, 8)
    sweep.prune_model(model, 0.3)
    assert sum(p.numel() for p in model.parameters()) == 19851157
    for p in model.parameters():
        p.grad = None
    sched = diffusion.DDPMScheduler()
    P = {n: p.detach().cpu().clone().requires_grad_(True) for n, p in model.named_parameters()}
    names = list(P)
    ft = train.FinetuneEngine(model, sched, dropout=0.1, dropout_seed=31, ema_decay=0.9999,
                              lr_scheduler=train.get_scheduler('constant', 2e-4))
    table = model.dropout_table()
    m = [torch.zeros_like(P[n]) for n in names]
    v = [torch.zeros_like(P[n]) for n in names]
    ema = [P[n].detach().clone() for n in names]
    gen = torch.Generator().manual_seed(17)
    worst_g, e_loss = 0.0, 0.0
    for step in (1, 2):
        fc = torch.from_numpy(gc.det_clean((B, 3, 32, 32), 50 + step))
        fn = torch.from_numpy(gc.det_noise((B, 3, 32, 32), 60 + step))
        t = train.antithetic_timesteps(B, 1000, gen)
        l_gpu = ft.step(fc.to(DEV), fn.to(DEV), t)
        for n in names:
            P[n].grad = None
        l_cpu = D.finetune_loss(P, cfg, fc, fn, t, PH.DropSpec(table, 31, step, 0))
        l_cpu.backward()
        e_loss = max(e_loss, abs(float(l_gpu) - float(l_cpu.detach())) / float(l_cpu.detach()))
        gpu_g = {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters()}
        for n in names:
            if float(P[n].grad.abs().max()) > 1e-6:
                worst_g = max(worst_g, relerr(gpu_g[n], P[n].grad))
        with torch.no_grad():       # optimizer arithmetic on IDENTICAL gradients (see test_autograd_bridge_and_finetune_step)
            D.adam_ema_step([P[n] for n in names], [gpu_g[n] for n in names], m, v, ema, step)
    pm = dict(model.named_parameters())
    e_p = max(relerr(pm[n], P[n].detach()) for n in names)
    es = ft.ema_state()
    e_e = max(relerr(es[n], e) for n, e in zip(names, ema))
    report['e2e/c4_finetune_dropout'] = dict(loss_rel=e_loss, grad_rel_worst=worst_g, param_rel_after2=e_p, ema_rel_after2=e_e,
                                             sites=len(table), lr=ft.last_lr)
    assert e_loss < 1e-5 and worst_g < 5e-5
    assert e_p < 1e-5 and e_e < 1e-5


def test_importance_accepts_torch_pruning_shaped_groups_on_device(report):
    """Boundary B1 on the GPU: groups shaped like a real torch_pruning DependencyGraph's (bound-method handlers of the tp pruner
    singletons, live nn layers) give bit-identical scores to the product's own groups through the fused |w*g| kernels."""
    from helpers import tp_like_groups
    pruning, graph_mod = pkg('pruning'), pkg('graph')
    model = make_model(gc.TINY_CFG, 5)
    clean, noise = _inputs(2, 16)
    _run_sweep(model, clean, noise, 2)
    imp = pruning.TaylorImportance()
    own = pruning.MagnitudePruner(model, None, importance=imp, iterative_steps=1, ch_sparsity=0.3, ignored_layers=[model.conv_out])
    own_groups = {g[0][0].target.name: g for g in own.DG.get_all_groups(ignored_layers=own.ignored_layers)}
    n = 0
    for root, items in tp_like_groups(model, graph_mod):
        s_tp, s_own = imp(items, ch_groups=1), imp(own_groups[root], ch_groups=1)
        assert s_tp is not None and s_tp.is_cuda and torch.equal(s_tp, s_own), root
        n += 1
    report['e2e/tp_shaped_groups'] = n
    assert n == 50


def test_device_side_early_exit_equals_host_loop(report):
    """Diff-Pruning early exit with the state on the device (no host read of the loss per step, stop flag polled every 8
    steps): stops at the reference's step, and the overshoot timesteps enqueued past the stop are exact no-ops -- gradients
    bit-identical to the host-synchronised loop -- for both flavours of the loop (ddpm_prune.py:102-106 accumulates the
    breaking step, ddpm_exp/prune.py:249-256 does not)."""
    cfg = gc.TINY_CFG
    fx = load_json('tiny_prune.json')['early_exit']
    sweep, sched = pkg('sweep'), pkg('diffusion').DDPMScheduler()
    clean, noise = _inputs(2, 16)
    out = {}
    for flavour in (True, False):
        for dev_exit in (True, False):
            model = make_model(cfg, 5)
            res = sweep.taylor_sweep(model, sched, clean.to(DEV), noise.to(DEV), num_steps=1000, thr=fx['thr'],
                                     accumulate_breaking_step=flavour, device_exit=dev_exit, poll_every=8)
            out[(flavour, dev_exit)] = (res['steps'], res['losses'], torch.cat([p.grad.reshape(-1) for p in model.parameters()]))
    assert out[(True, True)][0] == out[(True, False)][0] == fx['steps'] and fx['steps'] % 8 != 0     # overshoot steps ran
    assert np.allclose(out[(True, True)][1], fx['losses'], rtol=1e-5)
    for flavour in (True, False):
        a, b = out[(flavour, True)], out[(flavour, False)]
        assert a[0] == b[0] and a[1] == b[1] and torch.equal(a[2], b[2]), flavour
    assert not torch.equal(out[(True, True)][2], out[(False, True)][2])          # the breaking step's gradient differs
    report['e2e/device_early_exit'] = dict(steps=out[(True, True)][0], ref_steps=fx['steps'])


def test_two_half_batch_pipelines_match_single_pipeline(report):
    """Two half-batch pipelines on two HIP streams (second flat gradient buffer, folded in once per sweep): run-to-run
    bit-identical, equal to the single pipeline up to fp32 re-association, same prune masks; Diff-Pruning stops at the same step."""
    cfg = gc.TINY_CFG
    sweep, sched = pkg('sweep'), pkg('diffusion').DDPMScheduler()
    B = 32
    clean, noise = _inputs(B, 16, 7, 8)
    clean, noise = clean.to(DEV), noise.to(DEV)

    def run(halves, thr=None, steps=3):
        model = make_model(cfg, 5)
        flat = sweep.flatten_grads(model)
        step = sweep.HipSweepStep(model, sched, clean, noise, B * clean[0].numel(), 'mse', B, halves=halves)
        assert (step._half is not None) == (halves == 2)
        res = sweep.taylor_sweep(model, sched, clean, noise, num_steps=steps, thr=thr, step_fn=step, flat_grads=flat)
        torch.cuda.synchronize()
        return model, flat, res

    m2, g2, r2 = run(2)
    _, g2b, r2b = run(2)
    m1, g1, r1 = run(1)
    assert torch.equal(g2, g2b) and r2['losses'] == r2b['losses']
    e_g = relerr(g2, g1)
    e_l = max(abs(a - b) / b for a, b in zip(r2['losses'], r1['losses']))
    pr2, pr1 = sweep.prune_model(m2, 0.3), sweep.prune_model(m1, 0.3)
    mism = [a[0] for a, b in zip(pr2.records, pr1.records) if a[3] != b[3]]
    _, _, e2 = run(2, thr=0.999, steps=40)
    _, _, e1 = run(1, thr=0.999, steps=40)
    report['e2e/two_half_pipelines'] = dict(grad_rel=e_g, loss_rel=e_l, mask_mismatches=mism, exit_steps=(e2['steps'], e1['steps']))
    assert e_g < 2e-5 and e_l < 1e-6 and not mism
    assert e2['steps'] == e1['steps'] and np.allclose(e2['losses'], e1['losses'], rtol=1e-6)


def test_fid_inception_forward_and_statistics(report):
    """Row f3: the FID Inception network on the HIP kernels (BatchNorm folded, one conv + bias + ReLU launch per BasicConv2d,
    branches written into channel slices) against the oracle's PyTorch restatement with seeded weights (torchvision is absent:
    network parity unpinned), every output block; then the streaming mean / covariance and the Frechet distance against numpy
    on the device features and against the values fid_score.py itself produced for seeded feature matrices."""
    from oracle import metrics_ref as M
    from helpers import inception_state_dict, fid_features
    metrics = pkg('metrics')
    sd = inception_state_dict(3)
    net = metrics.InceptionV3([0, 1, 2, 3], state_dict=sd).to(DEV)
    img = torch.rand(3, 3, 32, 32, generator=torch.Generator().manual_seed(1))
    outs = net(img.to(DEV))
    refs = M.inception_forward(sd, img, (0, 1, 2, 3))
    errs = [relerr(o, r) for o, r in zip(outs, refs)]
    assert [tuple(o.shape) for o in outs] == [(3, 64, 73, 73), (3, 192, 35, 35), (3, 768, 17, 17), (3, 2048, 1, 1)]
    # no resize (299 x 299 input straight in) and the default single-block call
    net3 = metrics.InceptionV3(resize_input=False, state_dict=sd).to(DEV)
    big = torch.rand(2, 3, 299, 299, generator=torch.Generator().manual_seed(2))
    e_big = relerr(net3(big.to(DEV))[0], M.inception_forward(sd, big, (3,), resize_input=False)[0])
    # streaming statistics on the device vs numpy on the same rows, three uneven batches
    fx = load_json('fid.json')[1]
    a, b = fid_features(fx['dims'], fx['n1'], fx['n2'], fx['seed'])
    mus, sigmas = [], []
    for feats in (a, b):
        st = metrics.FeatureStats(fx['dims'], torch.device(DEV))
        for lo, hi in ((0, 100), (100, 101), (101, len(feats))):
            st.update(torch.from_numpy(feats[lo:hi]).to(DEV))
        mu, sg = st.finalize()
        rm, rs = M.activation_statistics(feats)
        mus.append(mu)
        sigmas.append(sg)
        assert np.abs(mu - rm).max() < 1e-6 and np.abs(sg - rs).max() < 2e-6 * np.abs(rs).max()
    fid = metrics.calculate_frechet_distance(mus[0], sigmas[0], mus[1], sigmas[1])
    report['e2e/fid'] = dict(block_rel_errs=errs, no_resize_rel=e_big, fid=fid, fid_reference=fx['fid'])
    assert max(errs) < 5e-5 and e_big < 5e-5                      # ~95 fp32 layers deep
    assert abs(fid - fx['fid']) < 1e-4 * fx['fid']


def test_compare_directories_ssim_and_fid_paths(report, tmp_path):
    """compute_ssim.py / fid_score.py entry points over directories of PNG samples (PIL decode on the host, ToTensor on the
    device): mean SSIM / MSE vs the oracle, FID of a directory against itself = 0 and against saved statistics (--save-stats)."""
    from PIL import Image
    from oracle import metrics_ref as M
    from helpers import inception_state_dict
    metrics = pkg('metrics')
    rng = np.random.default_rng(5)
    d1, d2 = tmp_path / 'a', tmp_path / 'b'
    d1.mkdir(); d2.mkdir()
    A = rng.integers(0, 256, (12, 32, 32, 3), dtype=np.uint8)
    B = np.clip(A.astype(np.int32) + rng.integers(-30, 31, A.shape), 0, 255).astype(np.uint8)
    for i in range(12):
        Image.fromarray(A[i]).save(str(d1 / ('%03d.png' % i)))
        Image.fromarray(B[i]).save(str(d2 / ('%03d.png' % i)))
    s, m = metrics.compare_directories(str(d1), str(d2), DEV, batch_size=5)
    ta = torch.from_numpy(A).permute(0, 3, 1, 2).float() / 255
    tb = torch.from_numpy(B).permute(0, 3, 1, 2).float() / 255
    s_ref = float(M.ssim(ta, tb).mean())
    m_ref = float(((ta - tb) ** 2).mean())
    net = metrics.InceptionV3([0], state_dict=inception_state_dict(3)).to(DEV)          # 64-dim features: 12 samples suffice
    f_self = metrics.calculate_fid_given_paths([str(d1), str(d1)], 5, DEV, 64, model=net)
    f_ab = metrics.calculate_fid_given_paths([str(d1), str(d2)], 5, DEV, 64, model=net)
    metrics.save_fid_stats([str(d1), str(tmp_path / 'a_stats.npz')], 5, DEV, 64, model=net)
    f_npz = metrics.calculate_fid_given_paths([str(tmp_path / 'a_stats.npz'), str(d2)], 5, DEV, 64, model=net)
    report['e2e/ssim_fid_paths'] = dict(ssim=s, ssim_ref=s_ref, mse=m, mse_ref=m_ref, fid_self=f_self, fid_ab=f_ab, fid_npz=f_npz)
    assert abs(s - s_ref) < 1e-5 and abs(m - m_ref) < 1e-6 * max(m_ref, 1e-9) + 1e-9
    assert abs(f_self) < 1e-3 and f_ab > 0 and abs(f_ab - f_npz) < 1e-6 * max(f_ab, 1.0)


@pytest.mark.parametrize('name', ['res_cat', 'plain_cnn', 'token_mixer', 'generator'])
def test_traced_model_taylor_prune_on_device(report, name):
    """Row f2: the device criterion on a network that is NOT one of the two UNet families.  A plain-PyTorch toy network
    (tests/golden/toy_nets.py) lives on the GPU, its gradients come from torch autograd, the groups from the autograd
    tracer (trace.py) and the scores from the HIP kernels (dp_wg_reduce / dp_gather_add): every group's score equals the
    vendored criterion's formula (importance.py:383-428: sum over same-sized members of sum((w*g)^2) resp. |w*g| for
    GroupNorm; BatchNorm / LayerNorm members add nothing) and the pruned network still runs."""
    import toy_nets
    pruning = pkg('pruning')
    model, inputs, ignored = toy_nets.build(name)
    model = model.cuda()
    inputs = tuple(t.cuda() for t in inputs)
    out = model(*inputs)
    out.square().mean().backward()
    pr = pruning.MetaPruner(model, inputs, importance=pruning.TaylorImportance(), iterative_steps=1, ch_sparsity=0.3,
                            ignored_layers=ignored)
    worst, n_groups = 0.0, 0
    for g in pr.step(interactive=True):
        root, ch_groups, score, idxs = pr.records[-1]
        all_g = next(gg for gg in pr.DG.get_all_groups(pr.ignored_layers) if gg[0][0].target.name == root)
        n0 = len(all_g[0][1])
        want = torch.zeros(n0, dtype=torch.float64)
        for dep, ix in all_g:
            m = dep.target.module
            if len(ix) != n0 or dep.kind in ('ln', 'bn', 'inorm', 'prelu', 'embed'):
                continue
            wg = (m.weight.data.double() * m.weight.grad.data.double()).cpu()
            if getattr(m, 'transposed', False):              # ConvTranspose: [Cin, Cout, k, k] (importance.py:390-392)
                wg = wg.transpose(0, 1)
            if dep.kind == 'out':
                want += wg[ix].flatten(1).square().sum(1)
            elif dep.kind == 'in':
                want += wg.transpose(0, 1).flatten(1)[ix].square().sum(1)
            elif dep.kind == 'gn':
                want += wg[ix].abs()
        err = float((score.double() - want).abs().max() / want.abs().max().clamp_min(1e-30))
        worst = max(worst, err)
        assert err < 1e-5, (root, err)
        g.prune()
        n_groups += 1
    with torch.no_grad():
        y = model(*inputs)
    assert torch.isfinite(y).all() and n_groups >= 2
    report['traced/%s' % name] = dict(groups=n_groups, score_rel_err=worst, params_after=sum(p.numel() for p in model.parameters()))


# ------------------------------------------------------------------------------------------------------------------
# round 3: config C5 at its real size + its data-parallel form, config C2 run as written
# ------------------------------------------------------------------------------------------------------------------
def _ldm_masks(model):
    ldm, pruning = pkg('ldm'), pkg('pruning')
    channel_groups = {}
    for m in model.modules():
        if isinstance(m, ldm.CrossAttention):
            channel_groups[m.to_q] = channel_groups[m.to_k] = channel_groups[m.to_v] = m.heads
    pr = pruning.MagnitudePruner(model, None, importance=pruning.TaylorImportance(), iterative_steps=1,
                                 channel_groups=channel_groups, ch_sparsity=0.3, ignored_layers=[model.out], round_to=2)
    for g in pr.step(interactive=True):
        g.prune()
    return pr


@isolated()
def test_c5_ldm_cin256_full_size(report):
    """BASELINE.json configs[4] at its real size: the cin256-v2 UNet (400 920 579 parameters), 6 latents of 3 x 64 x 64, one
    512-wide class token, 20 CFG-DDIM steps per importance step (prune_ldm.py:103-131) -- through size-independent properties:
      (a) the pass is run-to-run bit-identical (default device Philox draws, on-device early-exit state),
      (b) the per-rank shares of a 2-rank (3 + 3) and of a 4-rank (2 + 2 + 1 + 1, config C5's "4 x MI355X") job, each computed
          with `shard=(rank, world)` exactly as a rank computes it, sum to the one-process losses and gradients,
      (c) the 109 prune masks from the summed 4-rank shares equal the one-process masks,
      (d) one latent against the oracle on the host cores: a 2-step CFG-DDIM sampling, the loss at t and the gradients."""
    import copy
    import random
    from oracle import ldm_ref as L
    from oracle import pruning_ref as R
    ldm, ldm_sweep, sweep = pkg('ldm'), pkg('ldm_sweep'), pkg('sweep')
    cfg = gc.LDM_CIN256_CFG
    model = ldm.UNetModel(**cfg)
    gc.det_init_(model, 9)
    model = model.to(DEV).eval()
    assert sum(p.numel() for p in model.parameters()) == 400920579
    emb_w = torch.from_numpy(gc.det_param('embedding.weight', (1001, 512), 61))
    embedder = ldm_sweep.ClassEmbedder(512, 1001)
    with torch.no_grad():
        embedder.embedding.weight.copy_(emb_w)
    embedder = embedder.to(DEV)

    def run(shard=None, steps=2):
        res = ldm_sweep.ldm_importance_sweep(model, embedder, num_steps=steps, thr=None, n_samples=6, ddim_steps=20,
                                             latent_shape=(3, 64, 64), class_rng=random.Random(4), seed=21, shard=shard)
        torch.cuda.synchronize()
        return res['flat_grads'], res

    g_full, r_full = run()
    g_again, r_again = run()
    assert r_full['steps'] == 2 and r_full['accumulated'] == 2
    assert torch.equal(g_full, g_again) and r_full['losses'] == r_again['losses']      # (a)
    del g_again
    # (e) the whole pass (CFG sampler + scored forward / backward) on the direct kernels only: what DP_WINO=0 DP_WGRAD_WINO=0 run
    with _direct_kernels():
        model._engine.packs.clear()
        g_direct, r_direct = run()
        g_direct = g_direct.clone()
    model._engine.packs.clear()
    e_wino_grad = relerr(g_full, g_direct)
    e_wino_loss = max(abs(a - b) / b for a, b in zip(r_full['losses'], r_direct['losses']))
    out = {}
    for world in (2, 4):                                                               # (b)
        acc, lsum, sizes = None, [0.0, 0.0], []
        for r in range(world):
            g, res = run((r, world))
            sizes.append(res['shard'][1] - res['shard'][0])
            lsum = [a + b for a, b in zip(lsum, res['losses'])]
            acc = g.clone() if acc is None else acc.add_(g)
        out[world] = dict(sizes=sizes, loss_rel=max(abs(a - b) / b for a, b in zip(lsum, r_full['losses'])),
                          grad_rel=relerr(acc, g_full))
        if world == 2:
            del acc
    assert out[2]['sizes'] == [3, 3] and out[4]['sizes'] == [2, 2, 1, 1]
    m_full, m_direct = copy.deepcopy(model), copy.deepcopy(model)                      # (c)
    sweep.flatten_grads(m_full).copy_(g_full)
    sweep.flatten_grads(m_direct).copy_(g_direct)
    sweep.flatten_grads(model).copy_(acc)
    del acc, g_full, g_direct
    pr_full, pr_sum, pr_direct = _ldm_masks(m_full), _ldm_masks(model), _ldm_masks(m_direct)
    mism = [a[0] for a, b in zip(pr_full.records, pr_sum.records) if a[3] != b[3]]
    mism_direct = [a[0] for a, b in zip(pr_full.records, pr_direct.records) if a[3] != b[3]]
    margin = min(R.decision_margin(sc, pruned, len(sc), chg) for _, chg, sc, pruned in pr_full.records)
    e_score = max(relerr(b[2], a[2]) for a, b in zip(pr_full.records, pr_sum.records))   # one process vs the summed 4-rank shares
    e_wino_score = max(relerr(b[2], a[2]) for a, b in zip(pr_full.records, pr_direct.records))   # Winograd vs direct kernels
    params_after = sum(p.numel() for p in model.parameters())
    del m_full, m_direct, pr_full, pr_sum, pr_direct
    torch.cuda.empty_cache()
    # (d) one latent vs the oracle on the host
    model1 = ldm.UNetModel(**cfg)
    gc.det_init_(model1, 9)
    model1 = model1.to(DEV).eval()
    xc = torch.tensor([417])
    x_T = torch.from_numpy(gc.det_noise((1, 3, 64, 64), 301))
    noise = torch.from_numpy(gc.det_noise((1, 3, 64, 64), 302))
    t_loss = 250
    sched = ldm_sweep.LdmSchedule()
    c_dev, uc_dev = embedder(xc.to(DEV)), embedder(torch.tensor([1000], device=DEV))
    x0_dev = ldm_sweep.ddim_sample_cfg(model1, sched, x_T.to(DEV), c_dev, uc_dev, S=2, scale=3.0)
    P = {k: torch.from_numpy(gc.det_param(k, s_, 9)).requires_grad_(True) for k, s_ in L.ldm_param_shapes(cfg).items()}
    acp = L.ldm_alphas_cumprod()
    c, uc = emb_w[xc][:, None, :], emb_w[torch.tensor([1000])][:, None, :]
    with torch.no_grad():
        x0 = L.ddim_sample_cfg({k: v.detach() for k, v in P.items()}, cfg, acp, x_T, c, uc, S=2, scale=3.0)
    e_x0 = relerr(x0_dev, x0)
    # the loss / gradient comparison starts from the SAME x_start (the oracle's), so the sampler's error does not leak into it
    sweep.flatten_grads(model1)
    step = ldm_sweep.LdmSweepStep(model1, sched)
    with model1.pin_weights():
        loss = step.loss(x0.to(DEV), torch.full((1,), t_loss, dtype=torch.long, device=DEV), c_dev, noise.to(DEV))
        step.backward()
    ref = L.ldm_loss_at_t(P, cfg, acp, x0, torch.full((1,), t_loss, dtype=torch.long), c, noise)
    ref.backward()
    e_l = abs(float(loss) - float(ref.detach())) / float(ref.detach())
    worst, worst_name = 0.0, None
    for k, p in model1.named_parameters():
        if float(P[k].grad.abs().max()) > 1e-7:
            e = relerr(p.grad, P[k].grad)
            if e > worst:
                worst, worst_name = e, k
    report['e2e/c5_ldm_cin256'] = dict(losses=r_full['losses'], shards2=out[2], shards4=out[4], mask_mismatches=mism, groups=109,
                                       min_decision_margin=margin, shard_score_rel_worst=e_score, params_after=params_after,
                                       sample_rel=e_x0, b1_loss_rel=e_l, b1_grad_rel_worst=worst, b1_grad_worst_name=worst_name,
                                       wino_vs_direct_score_rel=e_wino_score, wino_vs_direct_grad_rel=e_wino_grad,
                                       wino_vs_direct_loss_rel=e_wino_loss, wino_vs_direct_mask_mismatches=mism_direct)
    assert out[2]['loss_rel'] < 1e-5 and out[4]['loss_rel'] < 1e-5 and out[2]['grad_rel'] < 2e-5 and out[4]['grad_rel'] < 2e-5
    assert not mism                                                                    # (c)
    assert not mism_direct                # (e) all 109 masks of the default (Winograd) dispatch == the direct kernels' masks
    # the thinnest decision of the 109 groups is an order of magnitude outside the re-association noise AND outside the movement of
    # the scores between the Winograd and the direct kernels (20 guided sampling steps + the scored pass, both changed)
    assert margin > 10 * max(e_score, e_wino_score)
    assert e_x0 < 1e-4 and e_l < 1e-5 and worst < 5e-5                                 # (d)


C2_LEGS = ('full_wino', 'full_direct', 'shard0', 'shard1')


def _c2_store():
    """Where the legs of config C2 leave their results for the comparison test: a directory named in the environment, so the child
    interpreters of `isolated` see the one their parent made (the parent makes it when this module is imported)."""
    import tempfile
    d = os.environ.get('DP_C2_STORE')
    if not d or not os.path.isdir(d):
        d = tempfile.mkdtemp(prefix='dp_c2_')
        os.environ['DP_C2_STORE'] = d
    return d


if os.environ.get('DP_TEST_CHILD') != '1' and torch.cuda.is_available():
    _c2_store()


@pytest.mark.parametrize('leg', C2_LEGS)
@isolated(params=('leg',))
def test_c2_cifar_batch256_1000_steps_leg(leg, report):
    """One 1000-timestep sweep of BASELINE.json configs[1] (CIFAR-10 UNet, batch 256; ddpm_prune.py:94-109) per leg, each in its own
    interpreter (<= 75 s): the full batch on the default (Winograd) dispatch, the full batch on the direct kernels only, and the two
    128-image shares of a 2-rank job scaled for the global batch as the data-parallel path scales them.  Each leg leaves its losses,
    its flat gradient buffer and (full-batch legs) its prune records for test_c2_cifar_batch256_1000_steps_as_written."""
    sweep, diffusion = pkg('sweep'), pkg('diffusion')
    from oracle import pruning_ref as R
    cfg, B, steps = gc.CIFAR_CFG, 256, 1000
    clean, noise = _inputs(B, 32, 11, 12)
    clean, noise = clean.to(DEV), noise.to(DEV)
    sched = diffusion.DDPMScheduler()
    lo, hi = {'full_wino': (0, B), 'full_direct': (0, B), 'shard0': (0, B // 2), 'shard1': (B // 2, B)}[leg]

    def run(n_steps, direct):
        model = make_model(cfg, 0)
        flat = sweep.flatten_grads(model)
        step = sweep.HipSweepStep(model, sched, clean[lo:hi], noise[lo:hi], B * clean[0].numel(), 'mse', B)
        if direct:
            with _direct_kernels():
                res = sweep.taylor_sweep(model, sched, clean[lo:hi], noise[lo:hi], num_steps=n_steps, step_fn=step, flat_grads=flat)
                torch.cuda.synchronize()
        else:
            res = sweep.taylor_sweep(model, sched, clean[lo:hi], noise[lo:hi], num_steps=n_steps, step_fn=step, flat_grads=flat)
            torch.cuda.synchronize()
        return model, flat, res

    def records(model, direct=False):
        if direct:
            with _direct_kernels():
                pr = sweep.prune_model(model, 0.3)
        else:
            pr = sweep.prune_model(model, 0.3)
        return [(root, chg, torch.as_tensor(sc).cpu(), [int(i) for i in pruned]) for root, chg, sc, pruned in pr.records]

    out = {}
    torch.cuda.reset_peak_memory_stats()
    model, flat, res = run(steps, leg == 'full_direct')
    assert res['steps'] == steps and len(res['losses']) == steps
    # Regression test of the round-5 abort (DESIGN.md section 5 "Round 6"): with Tensor.record_stream on the weight-gradient stream's
    # operands the allocator RESERVED ~104 GB in this very sweep for a 10 GB working set (+ ~55 GB for every further sweep of the
    # process, each on fresh streams) until the device had 0 bytes free.  Now: a small multiple of what is allocated, no retries.
    st = torch.cuda.memory_stats()
    mem = dict(reserved_peak_gb=st['reserved_bytes.all.peak'] / 2**30, allocated_peak_gb=st['allocated_bytes.all.peak'] / 2**30,
               alloc_retries=st['num_alloc_retries'], free_gb=torch.cuda.mem_get_info()[0] / 2**30)
    report['e2e/c2_memory_' + leg] = mem
    assert mem['reserved_peak_gb'] < 64 and mem['alloc_retries'] == 0 and mem['free_gb'] > 200, mem
    out['losses'] = res['losses']
    if leg != 'full_direct':
        out['flat'] = flat.cpu()
    if leg.startswith('full'):
        out['records'] = records(model, leg == 'full_direct')
        out['params_after'] = sum(p.numel() for p in model.parameters())
    if leg == 'shard0':
        # Winograd F(2, 3) vs the direct kernels on a 24-timestep sweep each (the relative movement of an accumulated score does not
        # grow with the number of accumulated timesteps; the full-length comparison is legs full_wino / full_direct)
        lo, hi = 0, B
        del model, flat
        for key in ('wino', 'direct'):
            m_s, _, _ = run(24, key == 'direct')
            out['short_' + key] = records(m_s, key == 'direct')
            del m_s
    torch.save(out, os.path.join(_c2_store(), leg + '.pt'))


def test_c2_cifar_batch256_1000_steps_as_written(report, tmp_path):
    """BASELINE.json configs[1] run as written: CIFAR-10 UNet, batch 256, the full 1000-timestep Taylor sweep + prune
    (ddpm_prune.py:94-109).  One full-batch run and the two 128-image shares of a 2-rank job: the masks from the summed shares equal
    the full-batch masks for all 50 groups, and the smallest decision margin of the 1000 x 256 run is reported next to the observed
    score difference.  The same 1000 x 256 sweep on the direct kernels only (no Winograd): the same 50 masks, score movement an order
    of magnitude inside the margin.  The four sweeps are the legs above (one interpreter each); a leg that has not run yet is run."""
    from oracle import pruning_ref as R
    sweep = pkg('sweep')
    store, legs = _c2_store(), {}
    for leg in C2_LEGS:
        path = os.path.join(store, leg + '.pt')
        if not os.path.exists(path):
            rc, tail, _ = run_isolated('tests/test_e2e_gpu.py::test_c2_cifar_batch256_1000_steps_leg[%s]' % leg, tmp_path)
            assert rc == 0, (leg, rc, tail)
        legs[leg] = torch.load(path, weights_only=False)
    cfg, B, steps = gc.CIFAR_CFG, 256, 1000
    full, direct, s0, s1 = (legs[k] for k in C2_LEGS)
    e_loss = max(abs((a + b) - c) / c for a, b, c in zip(s0['losses'], s1['losses'], full['losses']))
    g_sum = s0['flat'].to(DEV) + s1['flat'].to(DEV)                           # what the all-reduce leaves on every rank
    e_grad = relerr(g_sum, full['flat'])
    m_sum = make_model(cfg, 0)
    sweep.flatten_grads(m_sum).copy_(g_sum)
    rec_sum = sweep.prune_model(m_sum, 0.3).records
    rec_full, rec_d = full['records'], direct['records']
    mism = [a[0] for a, b in zip(rec_full, rec_sum) if list(a[3]) != list(b[3])]
    margin = min(R.decision_margin(sc, pruned, len(sc), chg) for _, chg, sc, pruned in rec_full)
    e_score = max(relerr(b[2], a[2]) for a, b in zip(rec_full, rec_sum))
    mism_direct_full = [a[0] for a, b in zip(rec_full, rec_d) if list(a[3]) != list(b[3])]
    e_wino_score_full = max(relerr(a[2], b[2]) for a, b in zip(rec_full, rec_d))
    e_wino_loss_full = max(abs(a - b) / b for a, b in zip(full['losses'], direct['losses']))
    short = {k: s0['short_' + k] for k in ('wino', 'direct')}
    mism_direct = [a[0] for a, b in zip(short['wino'], short['direct']) if list(a[3]) != list(b[3])]
    e_wino_score = max(relerr(a[2], b[2]) for a, b in zip(short['wino'], short['direct']))
    margin_short = min(R.decision_margin(sc, pruned, len(sc), chg) for _, chg, sc, pruned in short['direct'])
    params_sum = sum(p.numel() for p in m_sum.parameters())
    report['e2e/c2_as_written'] = dict(steps=steps, batch=B, loss_rel=e_loss, shard_grad_rel=e_grad, groups=len(rec_full),
                                       wino_vs_direct_score_rel_1000_steps=e_wino_score_full, wino_vs_direct_loss_rel_1000_steps=e_wino_loss_full,
                                       wino_vs_direct_mask_mismatches_1000_steps=mism_direct_full,
                                       mask_mismatches=mism, min_decision_margin=margin, shard_score_rel_worst=e_score,
                                       params_after=full['params_after'],
                                       wino_vs_direct_score_rel_24_steps=e_wino_score, wino_vs_direct_mask_mismatches_24_steps=mism_direct,
                                       min_decision_margin_24_steps=margin_short)
    assert e_loss < 1e-5 and e_grad < 5e-5                                    # fp32 re-association over 1000 accumulations
    assert len(rec_full) == len(rec_sum) == 50 and not mism
    assert not mism_direct and margin_short > 10 * e_wino_score
    # the decisions are an order of magnitude outside the shard re-association and the Winograd-vs-direct movement of the scores
    assert margin > 10 * max(e_score, e_wino_score, e_wino_score_full)
    assert len(rec_d) == 50 and not mism_direct_full and e_wino_loss_full < 1e-5
    assert full['params_after'] == direct['params_after'] == params_sum == 19851157
    import shutil
    shutil.rmtree(store, ignore_errors=True)
    os.environ.pop('DP_C2_STORE', None)


def test_batched_prune_tail_equals_member_by_member(report, monkeypatch):
    """The prune tail's batched kernels (dp_group_score: all member reductions of a group in one launch + one fold;
    dp_slice_batch: every weight / bias / gradient of the group in one launch) against the member-by-member path they replace
    (dp_wg_reduce + dp_axpby / dp_gather_add per member, index_select per tensor): scores, masks, sliced weights AND sliced
    gradients bit-identical over the whole sequential prune of the CIFAR UNet (50 groups) and of the LDM UNet (109 groups)."""
    import time
    pruning, sweep, ldm = pkg('pruning'), pkg('sweep'), pkg('ldm')
    out = {}

    def cifar():
        m = make_model(gc.CIFAR_CFG, 0)
        clean, noise = _inputs(4, 32)
        _run_sweep(m, clean, noise, 3)
        return m, (lambda mm: sweep.prune_model(mm, 0.3))

    def ldm_tiny():
        m, _ = _ldm_model_with_grads()
        return m, _ldm_masks

    for name, build in (('cifar', cifar), ('ldm', ldm_tiny)):
        res = {}
        for batch in (True, False, True):
            monkeypatch.setattr(pruning, 'PRUNE_BATCH', batch)
            m, prune = build()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pr = prune(m)
            torch.cuda.synchronize()
            res[batch] = (pr, m, (time.perf_counter() - t0) * 1e3)
        (pa, ma, ta), (pb, mb, tb) = res[True], res[False]
        assert len(pa.records) == len(pb.records) == (50 if name == 'cifar' else 109)
        for ra, rb in zip(pa.records, pb.records):
            assert ra[0] == rb[0] and ra[3] == rb[3] and torch.equal(ra[2], rb[2]), ra[0]      # root, mask, score bits
        for (n1, p1), (n2, p2) in zip(ma.named_parameters(), mb.named_parameters()):
            assert n1 == n2 and torch.equal(p1, p2) and torch.equal(p1.grad, p2.grad), n1
        out[name] = dict(batched_ms=ta, member_by_member_ms=tb, groups=len(pa.records))
    report['e2e/prune_tail_batched'] = out


def test_rank_sharded_sampling_to_dir_and_fid_features(report, tmp_path):
    """ddpm_sample.py:55-74 on the device: rank 1 of 2 samples total // (batch * world) batches into `process_1/` from the
    generator seeded seed + 1 -- the PNGs are the pipeline's own images for that generator, and the FID features accumulated on
    the fly (Inception on the HIP kernels, seeded weights) equal the features of the PNGs read back through the directory path."""
    from PIL import Image
    from helpers import inception_state_dict
    diffusion, metrics = pkg('diffusion'), pkg('metrics')
    model = make_model(gc.TINY_CFG, 5)
    pipe = diffusion.DDIMPipeline(model, diffusion.DDIMScheduler())
    net = metrics.InceptionV3([3], state_dict=inception_state_dict(3)).to(DEV)
    stats = metrics.FeatureStats(2048, torch.device(DEV))
    n = metrics.sample_to_dir(pipe, str(tmp_path), total_samples=8, batch_size=2, seed=3, rank=1, world=2, num_inference_steps=4,
                              stats=stats, inception=net)
    assert n == 4
    files = sorted((tmp_path / 'process_1').iterdir(), key=lambda f: int(f.stem))
    assert [f.name for f in files] == ['0.png', '1.png', '2.png', '3.png'] and not (tmp_path / 'process_0').exists()
    gen = torch.Generator().manual_seed(3 + 1)
    want = np.concatenate([pipe(batch_size=2, generator=gen, num_inference_steps=4, output_type='numpy').images for _ in range(2)])
    got = np.stack([np.asarray(Image.open(f), dtype=np.uint8) for f in files])
    assert np.array_equal(got, (want * 255).round().astype('uint8'))
    mu, sigma = stats.finalize()
    mu2, sigma2 = metrics.compute_statistics_of_path(str(tmp_path / 'process_1'), net, 2, 2048, DEV)
    e_mu = float(np.abs(mu - mu2).max() / max(np.abs(mu2).max(), 1e-30))
    e_sg = float(np.abs(sigma - sigma2).max() / max(np.abs(sigma2).max(), 1e-30))
    report['e2e/sample_to_dir'] = dict(images=n, mu_rel=e_mu, sigma_rel=e_sg)
    assert e_mu < 1e-5 and e_sg < 1e-4


def test_two_timesteps_in_flight_match_single_pipeline(report):
    """Plain Taylor with two TIMESTEPS in flight (odd-position timesteps on a second engine / stream / gradient buffer at the full
    batch, folded in once per sweep): run-to-run bit-identical, equal to the single pipeline up to fp32 re-association of the
    per-timestep gradient sum, same prune masks; with a threshold (sequential early exit) the second pipeline is not used."""
    cfg = gc.TINY_CFG
    sweep, sched = pkg('sweep'), pkg('diffusion').DDPMScheduler()
    B = 8
    clean, noise = _inputs(B, 16, 7, 8)
    clean, noise = clean.to(DEV), noise.to(DEV)

    def run(pipes, thr=None, steps=7, device_exit=True):
        model = make_model(cfg, 5)
        flat = sweep.flatten_grads(model)
        step = sweep.HipSweepStep(model, sched, clean, noise, B * clean[0].numel(), 'mse', B, timestep_pipelines=pipes)
        res = sweep.taylor_sweep(model, sched, clean, noise, num_steps=steps, thr=thr, step_fn=step, flat_grads=flat, use_graph=False,
                                 device_exit=device_exit)
        torch.cuda.synchronize()
        assert bool(step._tp) == (pipes == 2 and thr is None)       # created lazily, and never for a sequential (threshold) sweep
        return model, flat, res

    m2, g2, r2 = run(2)
    _, g2b, r2b = run(2)
    m1, g1, r1 = run(1)
    assert torch.equal(g2, g2b) and r2['losses'] == r2b['losses']
    e_g = relerr(g2, g1)
    assert r2['losses'] == r1['losses']                       # per-timestep losses: same kernels on the same inputs
    pr2, pr1 = sweep.prune_model(m2, 0.3), sweep.prune_model(m1, 0.3)
    mism = [a[0] for a, b in zip(pr2.records, pr1.records) if a[3] != b[3]]
    _, ge2, e2 = run(2, thr=0.999, steps=40)
    _, ge1, e1 = run(1, thr=0.999, steps=40)
    report['e2e/two_timestep_pipelines'] = dict(grad_rel=e_g, mask_mismatches=mism, exit_steps=(e2['steps'], e1['steps']))
    assert e_g < 2e-5 and not mism
    assert e2['steps'] == e1['steps'] and e2['losses'] == e1['losses'] and torch.equal(ge2, ge1)
    # advisor finding of round 3: the host-synchronised loop (device_exit=False) with two pipelines requested read the loss of
    # odd timesteps from a stream it had not waited for; a threshold now pins the sweep to one pipeline
    _, ge3, e3 = run(2, thr=0.999, steps=40, device_exit=False)
    assert e3['steps'] == e1['steps'] and e3['losses'] == e1['losses'] and torch.equal(ge3, ge1)


@pytest.mark.parametrize('which', ['tiny_forward', 'tiny_sweep', 'tiny_prune', 'cifar_c1', 'c1_size_1000', 'ddim', 'pruned_sweep', 'ldm_fwd_bwd',
                                   'finetune', 'ldm_prune', 'ldm_sweep', 'multi_head', 'bedroom_topology', 'ddpm', 'criteria'])
@pytest.mark.parametrize('flavour', ['f23', 'f2x2_3x3'])
def test_reference_fixtures_with_winograd_on_every_supported_layer(which, flavour, report, monkeypatch):
    """Round 6: flavour f2x2_3x3 = the two-dimensional F(2x2, 3x3) kernel (csrc/winograd2d.hip) on every layer it supports (grid and
    row-fill rules dropped), the one-dimensional kernel on the rest; flavour f23 = the one-dimensional kernel everywhere (WINO2D off).
    Round 4: the 3x3 / stride-1 convolutions of big launches run as a Winograd F(2, 3) implicit GEMM (csrc/winograd.hip; the
    default leaves grids of < 512 tiles -- every fixture-sized model -- on the direct kernel).  Here the threshold is dropped, so
    EVERY supported layer of the fixture models takes the Winograd kernel, forward and input gradient, and the reference's recorded
    outputs / gradients / prune masks must still come out: masks bit-exact, tensors within the tolerances of the original tests."""
    ops = pkg('ops')
    monkeypatch.setattr(ops, 'WINO', True)                          # (also under DP_WINO=0 / DP_WGRAD_WINO=0 in the environment)
    monkeypatch.setattr(ops, 'WGRAD_WINO', True)
    monkeypatch.setattr(ops, 'WINO_MIN_TILES', 0)
    monkeypatch.setattr(ops, 'WGRAD_WINO_MIN_WORK', 0)              # ... and every supported weight gradient its Winograd kernel
    monkeypatch.setattr(ops, 'WGRAD_WINO_MIN_FILL', 0.0)
    monkeypatch.setattr(ops, 'WINO2D', flavour == 'f2x2_3x3')
    monkeypatch.setattr(ops, 'WINO2D_MIN_TILES', 0)
    monkeypatch.setattr(ops, 'WINO2D_MIN_FILL', 0.0)
    monkeypatch.setattr(ops, 'WGRAD_WINO2D', flavour == 'f2x2_3x3')   # ... and the two-dimensional weight-gradient kernel (csrc/wgrad2d.hip)
    monkeypatch.setattr(ops, 'WGRAD_WINO2D_MIN_FILL', 0.0)
    nw = [0]
    real_w = ops._conv_wgrad_wino
    monkeypatch.setattr(ops, '_conv_wgrad_wino', lambda *a: (lambda r: (nw.__setitem__(0, nw[0] + (r is not None)), r)[1])(real_w(*a)))
    nw2 = [0]
    real_w2 = ops._conv_wgrad_wino2d
    monkeypatch.setattr(ops, '_conv_wgrad_wino2d', lambda *a: (lambda r: (nw2.__setitem__(0, nw2[0] + (r is not None)), r)[1])(real_w2(*a)))
    n = [0]
    real = ops._conv_wino
    monkeypatch.setattr(ops, '_conv_wino', lambda *a: (lambda r: (n.__setitem__(0, n[0] + bool(r)), r)[1])(real(*a)))
    n2 = [0]
    real2 = ops._conv_wino2d
    monkeypatch.setattr(ops, '_conv_wino2d', lambda *a: (lambda r: (n2.__setitem__(0, n2[0] + bool(r)), r)[1])(real2(*a)))
    sub = {}
    {'tiny_forward': test_tiny_forward_matches_reference_and_oracle, 'tiny_sweep': test_tiny_sweep_gradients_match_reference,
     'tiny_prune': test_tiny_prune_masks_bit_exact_and_post_prune_forward, 'cifar_c1': test_cifar_c1_masks_bit_exact,
     'c1_size_1000': test_c1_size_1000_step_sweep_masks_match_reference, 'ddim': test_ddim_sampling_matches_reference,
     'pruned_sweep': test_pruned_model_sweep_matches_oracle, 'ldm_fwd_bwd': test_ldm_unet_forward_backward_matches_reference,
     'finetune': test_autograd_bridge_and_finetune_step,
     # round 5 (verdict item 1a): the fixtures whose decision margins are the thinnest -- the LDM masks (margin 6e-5), the LDM
     # importance-pass driver, the multi-head and 6-level topologies, ancestral sampling and the five sibling criteria
     'ldm_prune': test_ldm_prune_masks_bit_exact, 'ldm_sweep': test_ldm_importance_sweep_matches_oracle,
     'multi_head': test_multi_head_unet_matches_reference, 'bedroom_topology': test_bedroom_topology_sweep_matches_oracle,
     'ddpm': test_ddpm_sampling_matches_reference,
     'criteria': lambda rep: [test_sibling_criteria_masks_bit_exact(rep, c) for c in ('full1', 'full2', 'abs', 'fisher', 'magnitude')],
     }[which](sub)
    report['wino_forced/%s/%s' % (flavour, which)] = dict(sub, winograd_launches=n[0], winograd_2d_launches=n2[0], winograd_wgrad_launches=nw[0],
                                                          winograd_2d_wgrad_launches=nw2[0])
    no_backward = which in ('tiny_forward', 'ddim', 'ddpm')
    assert n[0] > 0 and (nw[0] + nw2[0] > 0 or no_backward)
    assert (n2[0] > 0) == (flavour == 'f2x2_3x3'), (flavour, n2[0])
    assert (nw2[0] > 0) == (flavour == 'f2x2_3x3' and not no_backward), (flavour, nw2[0])


@pytest.mark.parametrize('overlap', [False, True], ids=['one_stream', 'wgrad_side_stream'])
def test_finetune_step_replayed_natively_equals_eager(report, monkeypatch, overlap):
    """Round 5 (verdict item 2a): FinetuneEngine.step captured once and re-issued from the library's C loop -- the per-step scalars
    (lr of a warm-up schedule, Adam bias corrections, the dropout masks' step) live on the device, inputs / timesteps in static
    buffers, the weight re-packing inside the captured step.  Five optimizer steps with dropout 0.1 and a cosine warm-up schedule:
    losses, gradient norms, parameters, Adam moments and EMA weights BIT-identical to the eager engine's, with and without the
    weight-gradient side stream inside the capture."""
    train, diffusion = pkg('train'), pkg('diffusion')
    cfg = gc.TINY_CFG
    B = 8
    gen = torch.Generator().manual_seed(3)
    batches = [(_inputs(B, 16, 20 + k, 30 + k), train.antithetic_timesteps(B, 1000, gen)) for k in range(5)]

    def run(replay):
        model = make_model(cfg, 5)
        sched = train.get_scheduler('cosine', 2e-4, num_warmup_steps=2, num_training_steps=10)
        ft = train.FinetuneEngine(model, diffusion.DDPMScheduler(), lr=2e-4, dropout=0.1, dropout_seed=7, lr_scheduler=sched, replay=replay)
        ft.REPLAY_OVERLAP = overlap
        model.engine().overlap_wgrad = overlap
        out = []
        for (c, n), t in batches:
            loss = ft.step(c.to(DEV), n.to(DEV), t.to(DEV))
            out.append((float(loss), float(ft.last_grad_norm), ft.last_lr))
            model._engine.overlap_wgrad = overlap
        torch.cuda.synchronize()
        return ft, out

    fe, oe = run(False)
    fr, orr = run(True)
    assert fe._cap is None and fr._cap is not None and fr._cap['call'].replay is not None
    info = fr._cap['call'].info
    report['e2e/finetune_replay_%s' % ('two_streams' if overlap else 'one_stream')] = dict(
        steps=len(oe), losses=[a[0] for a in orr], replay=info)
    assert oe == orr                                             # losses, gradient norms, learning rates
    assert torch.equal(fe.flat_p, fr.flat_p) and torch.equal(fe.ema, fr.ema) and torch.equal(fe.m, fr.m) and torch.equal(fe.v, fr.v)
    assert (info['side_nodes'] > 0) == overlap
    # a different batch shape re-captures; an evaluation forward between steps does not disturb the captured step
    with torch.no_grad():
        fr.model.eval()
        fr.model(batches[0][0][0].to(DEV), torch.tensor([5], device=DEV))
    l2 = fr.step(batches[0][0][0].to(DEV), batches[0][0][1].to(DEV), batches[0][1].to(DEV))
    l2e = fe.step(batches[0][0][0].to(DEV), batches[0][0][1].to(DEV), batches[0][1].to(DEV))
    assert float(l2) == float(l2e) and torch.equal(fe.flat_p, fr.flat_p)


def test_sampling_forward_replayed_natively_equals_eager(report, monkeypatch):
    """Round 5 (verdict item 2a): the UNet forward of a DDIM / DDPM sampling loop captured once and re-issued natively
    (UNet2DModel.sampling_forward; automatic from 32 steps on, forced here with DP_SAMPLE_REPLAY=1): images BIT-identical to the eager loop."""
    diffusion, unet = pkg('diffusion'), pkg('unet')
    model = make_model(gc.TINY_CFG, 5)
    out = {}
    for name, Pipe, kw in (('ddim', diffusion.DDIMPipeline, dict(num_inference_steps=12, eta=0.0)),
                           ('ddim_eta', diffusion.DDIMPipeline, dict(num_inference_steps=9, eta=0.5)),
                           ('ddpm', diffusion.DDPMPipeline, dict(num_inference_steps=10))):
        sched = diffusion.DDIMScheduler() if 'ddim' in name else diffusion.DDPMScheduler()
        pipe = Pipe(model, sched)
        imgs = {}
        for mode in ('1', '0'):
            monkeypatch.setenv('DP_SAMPLE_REPLAY', mode)
            made = []
            real = unet.UNet2DModel.sampling_forward
            monkeypatch.setattr(unet.UNet2DModel, 'sampling_forward',
                                lambda self, *a, **k: (lambda f: (made.append(type(f).__name__), f)[1])(real(self, *a, **k)))
            imgs[mode] = pipe(batch_size=3, generator=torch.Generator().manual_seed(11), output_type='numpy', **kw).images
            monkeypatch.setattr(unet.UNet2DModel, 'sampling_forward', real)
            assert made == ['_CapturedForward' if mode == '1' else '_EagerForward'], made
        assert np.array_equal(imgs['1'], imgs['0']), name
        out[name] = float(np.abs(imgs['1']).mean())
    report['e2e/sampling_replay'] = out


def test_replay_at_the_benchmarked_sizes(report, monkeypatch):
    """The replayed paths at the sizes `bench.py --config c4_finetune / ddim` times them (BASELINE.json configs[3] and the sampling of
    its model): the ratio-0.3 pruned CIFAR UNet (19 851 157 parameters), (a) four optimizer steps at batch 128 with dropout 0.1 --
    the first eager, three replayed natively with the engine's own stream rule -- against the eager engine: losses, gradient norms,
    parameters, Adam moments and EMA weights BIT-identical; (b) a 40-step DDIM loop at batch 256, where the automatic rule (>= 32
    calls) takes the captured forward, against DP_SAMPLE_REPLAY=0: images BIT-identical."""
    train, diffusion, sweep, unet = pkg('train'), pkg('diffusion'), pkg('sweep'), pkg('unet')

    def pruned():
        model = make_model(gc.CIFAR_CFG, 0)
        clean, noise = _inputs(4, 32)
        _run_sweep(model, clean, noise, 8)
        sweep.prune_model(model, 0.3)
        for p in model.parameters():
            p.grad = None
        return model

    B = 128
    gen = torch.Generator().manual_seed(23)
    batches = [(torch.from_numpy(gc.det_clean((B, 3, 32, 32), 70 + k)).to(DEV), torch.from_numpy(gc.det_noise((B, 3, 32, 32), 80 + k)).to(DEV),
                train.antithetic_timesteps(B, 1000, gen).to(DEV)) for k in range(4)]
    res = {}
    for replay in (False, True):
        model = pruned()
        assert sum(p.numel() for p in model.parameters()) == 19851157
        ft = train.FinetuneEngine(model, diffusion.DDPMScheduler(), lr=2e-4, dropout=0.1, dropout_seed=31, replay=replay)
        out = []
        for c, n, t in batches:
            loss = ft.step(c, n, t)
            out.append((float(loss), float(ft.last_grad_norm)))
        torch.cuda.synchronize()
        res[replay] = (ft, out, model)
    (fe, oe, _), (fr, orr, m_r) = res[False], res[True]
    assert fe._cap is None and fr._cap is not None and fr._cap['call'].replay is not None
    assert oe == orr, (oe, orr)
    assert torch.equal(fe.flat_p, fr.flat_p) and torch.equal(fe.ema, fr.ema) and torch.equal(fe.m, fr.m) and torch.equal(fe.v, fr.v)
    info = fr._cap['call'].info
    # (b) sampling with the finetuned weights of (a)
    m_r.eval()
    pipe = diffusion.DDIMPipeline(m_r, diffusion.DDIMScheduler())
    imgs, kinds = {}, {}
    real = unet.UNet2DModel.sampling_forward
    for mode in (None, '0'):
        if mode is None:
            monkeypatch.delenv('DP_SAMPLE_REPLAY', raising=False)
        else:
            monkeypatch.setenv('DP_SAMPLE_REPLAY', mode)
        made = []
        monkeypatch.setattr(unet.UNet2DModel, 'sampling_forward',
                            lambda self, *a, **k: (lambda f: (made.append(type(f).__name__), f)[1])(real(self, *a, **k)))
        imgs[mode] = pipe(batch_size=256, generator=torch.Generator().manual_seed(5), num_inference_steps=40, eta=0.0,
                          output_type='numpy').images
        monkeypatch.setattr(unet.UNet2DModel, 'sampling_forward', real)
        kinds[mode] = made
    assert kinds[None] == ['_CapturedForward'] and kinds['0'] == ['_EagerForward'], kinds
    assert np.array_equal(imgs[None], imgs['0'])
    report['e2e/replay_at_bench_sizes'] = dict(finetune_losses=[a[0] for a in orr], finetune_replay=info,
                                               ddim_steps=40, ddim_batch=256, image_mean=float(np.abs(imgs[None]).mean()))


@pytest.mark.parametrize('L_ctx', [3, 5, 77])
def test_ldm_general_cross_attention_matches_oracle(report, L_ctx):
    """Round 5: cross-attention over L > 1 context tokens on the HIP kernels (ldm/modules/attention.py:152-193, general form; 77 = a
    CLIP text context; 3 and 5: odd key counts on the general contraction kernels) against the oracle on the host: forward, loss,
    every parameter gradient -- incl. norm2 / attn2.to_q / attn2.to_k, which are exactly zero for one token -- and the CFG pair +
    context cache of a sampling loop against two plain forwards."""
    from oracle import ldm_ref as L
    ldm, ops = pkg('ldm'), pkg('ops')
    cfg = gc.LDM_TINY_CFG
    model = ldm.UNetModel(**cfg)
    gc.det_init_(model, 9)
    model = model.to(DEV).eval()
    x = torch.from_numpy(gc.det_noise((2, 3, 16, 16), 31))
    ctx = torch.from_numpy(gc.det_noise((2, L_ctx, 16), 123))
    noise = torch.from_numpy(gc.det_noise((2, 3, 16, 16), 33))
    t = torch.tensor([7, 640])
    eng = model.engine()
    grads = {n: torch.zeros_like(p) for n, p in model.named_parameters()}
    eng.bind(eng.P, grads)
    y = eng.forward(x.to(DEV), t.to(DEV), ctx.to(DEV), save=True)
    n = y.numel()
    loss, dout = ops.mse_fwd_bwd(y, noise.to(DEV), 2.0 / n, 1.0 / n)
    eng.backward(dout)
    P = {k: torch.from_numpy(gc.det_param(k, s_, 9)).requires_grad_(True) for k, s_ in L.ldm_param_shapes(cfg).items()}
    yo = L.ldm_unet_forward(P, cfg, x, t, ctx)
    lo = (yo - noise).square().mean(dim=(1, 2, 3)).mean()
    lo.backward()
    e_f = float((y.cpu() - yo.detach()).abs().max())
    e_l = abs(float(loss) - float(lo.detach())) / float(lo.detach())
    worst, nonzero = 0.0, 0
    for k in P:
        ref = P[k].grad
        if float(ref.abs().max()) > 1e-7:
            worst = max(worst, relerr(grads[k], ref))
            nonzero += ('attn2.to_q' in k or 'attn2.to_k' in k or '.norm2.' in k)
        else:
            assert float(grads[k].abs().max()) < 1e-6, k
    ctx2 = torch.cat([torch.from_numpy(gc.det_noise(tuple(ctx.shape), 124)), ctx]).to(DEV)
    x2, t2 = x.to(DEV), t.to(DEV)
    with torch.no_grad(), model.pin_weights() as pinned:
        plain = model(torch.cat([x2, x2]), torch.cat([t2, t2]), context=ctx2)
        with pinned._engine.context_cache(ctx2):
            pa = model.forward_cfg_pair(x2, t2, ctx2)
            pb = model.forward_cfg_pair(x2, t2, ctx2)
    e_pair = float((pa - plain).abs().max())
    report['e2e/ldm_cross_attention_L%d' % L_ctx] = dict(fwd_abs=e_f, loss_rel=e_l, grad_rel_worst=worst, cfg_pair_abs=e_pair,
                                                       attn2_q_k_norm2_grads_nonzero=nonzero)
    assert e_f < 1e-5 and e_l < 1e-5 and worst < 2e-5 and nonzero > 0
    assert e_pair < 1e-5 and torch.equal(pa, pb)


@pytest.mark.parametrize('tag', ['h2d2', 'hc16_L3'])
def test_ldm_multi_head_and_depth_matches_reference(report, tag):
    """Round 6 (verdict "missing" item 4): LDM UNets with `num_heads` > 1 / `num_head_channels` and `transformer_depth` > 1
    (ldm/modules/attention.py:152-258, openaimodel.py:542-559) on the HIP engine against the reference's own UNetModel
    (tests/golden/ldm_heads.*, make_golden_ldm.py heads): forward, loss, gradients (full tensors + per-parameter sums), then the
    Taylor prune with the head channel groups of prune_ldm.py:78-82 -- every group's mask bit for bit -- and the pruned model's
    forward (head widths no longer equal to the construction-time dim_head, query / value widths different).  h2d2: 2 heads x 2
    blocks over the class token; hc16_L3: 4 / 6 / 10 heads of 16 channels over 3 context tokens.  Also the sampling path: CFG
    pair + context cache against the plain forward."""
    ldm, ops, pruning, sweep = pkg('ldm'), pkg('ops'), pkg('pruning'), pkg('sweep')
    from oracle import pruning_ref as R
    rec = load_json('ldm_heads.json')[tag]
    g = load_npz('ldm_heads.npz')
    cfg = rec['cfg']
    model = ldm.UNetModel(**cfg)
    gc.det_init_(model, 9)
    assert {n: list(p.shape) for n, p in model.named_parameters()} == rec['shapes']
    model = model.to(DEV).eval()
    sweep.flatten_grads(model)
    x = torch.from_numpy(gc.det_noise((2, 3, 16, 16), 31)).to(DEV)
    ctx = torch.from_numpy(gc.det_noise((2, rec['context_tokens'], cfg['context_dim']), 32)).to(DEV)
    noise = torch.from_numpy(gc.det_noise((2, 3, 16, 16), 33)).to(DEV)
    t = torch.tensor([7, 640], device=DEV)
    eng = model.engine()
    grads = {n: p.grad for n, p in model.named_parameters()}
    eng.bind(eng.P, grads)
    y = eng.forward(x, t, ctx, save=True)
    e_f = float((y.cpu() - torch.from_numpy(g[tag + '::fwd_out'])).abs().max())
    n = y.numel()
    loss, dout = ops.mse_fwd_bwd(y, noise, 2.0 / n, 1.0 / n)
    eng.backward(dout)
    e_l = abs(float(loss) - float(g[tag + '::loss'])) / float(g[tag + '::loss'])
    worst = 0.0
    for k in g.files:
        if k.startswith(tag + '::grad::') and float(np.abs(g[k]).max()) > 0:
            worst = max(worst, relerr(grads[k.split('::grad::')[1]], g[k]))
    bad = [(k, float(grads[k].double().abs().sum()), a) for k, (s_, a) in rec['grad_stats'].items()
           if abs(float(grads[k].double().abs().sum()) - a) > 5e-5 * a + 1e-8 * grads[k].numel()]
    with torch.no_grad(), model.pin_weights() as pinned:
        y2 = model(x, t, context=ctx)
        ctx2 = torch.cat([torch.from_numpy(gc.det_noise(tuple(ctx.shape), 124)).to(DEV), ctx])
        plain = model(torch.cat([x, x]), torch.cat([t, t]), context=ctx2)
        with pinned._engine.context_cache(ctx2):
            pa = model.forward_cfg_pair(x, t, ctx2)
            pb = model.forward_cfg_pair(x, t, ctx2)
    e_pair = float((pa - plain).abs().max())
    out = dict(fwd_abs=e_f, loss_rel=e_l, grad_rel_worst=worst, n_bad_stats=len(bad), nograd_fwd_abs=float((y2 - y).abs().max()),
               cfg_pair_abs=e_pair)
    report['e2e/ldm_heads_' + tag] = out
    assert e_f < 1e-5 and e_l < 1e-5 and worst < 2e-5 and not bad, (out, bad[:5])
    assert out['nograd_fwd_abs'] < 1e-5 and e_pair < 1e-5 and torch.equal(pa, pb)
    # the prune, with the head channel groups
    channel_groups = {}
    for m in model.modules():
        if isinstance(m, ldm.CrossAttention):
            channel_groups[m.to_q] = channel_groups[m.to_k] = channel_groups[m.to_v] = m.heads
    pr = pruning.MagnitudePruner(model, None, importance=pruning.TaylorImportance(), iterative_steps=1,
                                 channel_groups=channel_groups, ch_sparsity=0.3, ignored_layers=[model.out], round_to=2)
    for grp in pr.step(interactive=True):
        grp.prune()
    model._engine.packs.clear()
    assert len(pr.records) == len(rec['prune'])
    mism, worst_s, margin, head_groups = [], 0.0, 1e9, 0
    for (root, chg, score, pruned), ref in zip(pr.records, rec['prune']):
        assert root == ref['root'] and chg == ref['ch_groups']
        head_groups += chg not in (1, 32)
        rs = torch.from_numpy(gc.b64_to_f32(ref['score']))
        worst_s = max(worst_s, relerr(score, rs))
        if ref['pruned']:
            margin = min(margin, R.decision_margin(rs, ref['pruned'], ref['cur'], ref['ch_groups']))
        if pruned != ref['pruned']:
            mism.append(root)
    out.update(groups=len(pr.records), head_groups=head_groups, score_rel_worst=worst_s, min_decision_margin=margin, mask_mismatches=mism)
    assert not mism and worst_s < 1e-4 and head_groups > 0, out
    assert {k: list(p.shape) for k, p in model.named_parameters()} == rec['shapes_after']
    assert sum(p.numel() for p in model.parameters()) == rec['params_after']
    with torch.no_grad():
        y3 = model(x, t, context=ctx)
    out['fwd_after_abs'] = float((y3.cpu() - torch.from_numpy(g[tag + '::fwd_after'])).abs().max())
    assert out['fwd_after_abs'] < 1e-5, out


@pytest.mark.parametrize('which', ['tiny_forward', 'ddim', 'ddpm', 'ldm_sweep', 'sampling_replay'])
def test_no_grad_forwards_with_winograd_f43_on_every_supported_layer(which, report, monkeypatch):
    """Round 5 (verdict item 5): the no-grad forwards -- sampling loops, the LDM importance pass's CFG sampler -- take the Winograd
    F(4, 3) convolution (half the multiplies, ~1e-6 fp32 error) wherever the kernel takes the shape (thresholds dropped here, so the
    fixture-sized models use it everywhere); scored forwards never do.  The reference's recorded sampling outputs must still come
    out within the tolerances of the original tests, and the gradients / losses of the LDM pass (whose sampler feeds the scored
    step) within theirs."""
    ops = pkg('ops')
    monkeypatch.setattr(ops, 'WINO', True)
    monkeypatch.setattr(ops, 'WINO43', True)
    monkeypatch.setattr(ops, 'WINO43_MIN_TILES', 0)
    n = [0]
    real = ops._conv_wino43
    monkeypatch.setattr(ops, '_conv_wino43', lambda *a: (lambda r: (n.__setitem__(0, n[0] + bool(r)), r)[1])(real(*a)))
    sub = {}
    if which == 'sampling_replay':
        test_sampling_forward_replayed_natively_equals_eager(sub, monkeypatch)
    else:
        {'tiny_forward': test_tiny_forward_matches_reference_and_oracle, 'ddim': test_ddim_sampling_matches_reference,
         'ddpm': test_ddpm_sampling_matches_reference, 'ldm_sweep': test_ldm_importance_sweep_matches_oracle}[which](sub)
    report['wino43_forced/' + which] = dict(sub, f43_launches=n[0])
    assert n[0] > 0



def test_ieee_sigmoid_build_gives_the_same_masks(report, tmp_path):
    """The default build evaluates sigmoid / SiLU as v_exp_f32 + v_rcp_f32 (csrc/dp_common.h, DP_FAST_SIGMOID) in every kernel of
    the SCORED forward and backward; `-DDP_IEEE_SIGMOID` builds expf + the IEEE division (__graft_entry__.build() compiles both).
    The fixtures with the thinnest decision margins -- written by the reference itself -- are run on the exact build too, in a child
    interpreter with DP_HIP_LIB pointing at it: bit-exact masks on BOTH builds (this suite runs them on the default one)."""
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'diff-pruning_amd', 'libdp_hip_ieee.so')
    assert os.path.exists(lib), 'libdp_hip_ieee.so is built by __graft_entry__.build()'
    nodes = ['tests/test_e2e_gpu.py::' + n for n in (
        'test_tiny_prune_masks_bit_exact_and_post_prune_forward', 'test_cifar_c1_masks_bit_exact', 'test_ldm_prune_masks_bit_exact',
        'test_long_sweep_1000_steps_matches_reference', 'test_sibling_criteria_masks_bit_exact', 'test_ddim_sampling_matches_reference')]
    rc, tail, rep = run_isolated(nodes, tmp_path, extra_env={'DP_HIP_LIB': lib})
    report['e2e/ieee_sigmoid_build'] = dict(rc=rc, nodes=len(nodes), masks={k: v.get('mask_mismatches') for k, v in rep.items()
                                                                          if isinstance(v, dict) and 'mask_mismatches' in v})
    assert rc == 0, tail
    assert report['e2e/ieee_sigmoid_build']['masks'] and not any(report['e2e/ieee_sigmoid_build']['masks'].values())
