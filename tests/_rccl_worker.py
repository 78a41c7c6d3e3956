"""Worker for tests/test_rccl_gpu.py: ONE process, ONE MI355X, a 1-rank `nccl` (= RCCL) process group, DP_FORCE_DIST=1.

Every exchange step of the data-parallel path then really goes through RCCL on the device -- the stream-ordered scalar-loss
all-reduce inside the Diff-Pruning poll loop, the flat-gradient all-reduce after a two-pipeline sweep, the three async gradient
buckets of the finetune step, the LDM importance pass, the FID statistics -- and is compared with the same call outside the
process group.  A one-rank sum is the identity, so results must be bit-identical wherever the two runs execute the same
kernels (the finetune step switches the batched time-embedding backward off under a process group: tolerance there).
Writes a JSON report; exits non-zero on any mismatch."""
import importlib
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE, os.path.join(HERE, 'golden')):
    if p not in sys.path:
        sys.path.insert(0, p)
import golden_common as gc   # noqa: E402

DEV = 'cuda'


def pkg(sub):
    return importlib.import_module('diff-pruning_amd.' + sub)


def make_model(cfg, seed):
    m = pkg('unet').UNet2DModel(**cfg)
    gc.det_init_(m, seed)
    return m.to(DEV).eval()


def inputs(B, H, s1, s2):
    return (torch.from_numpy(gc.det_clean((B, 3, H, H), s1)).to(DEV), torch.from_numpy(gc.det_noise((B, 3, H, H), s2)).to(DEV))


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


def run_all(forced):
    """Every data-parallel entry point once; `forced`: inside the 1-rank RCCL group with DP_FORCE_DIST=1."""
    os.environ['DP_FORCE_DIST'] = '1' if forced else '0'
    sweep, diffusion, train, ldm, ldm_sweep, metrics = (pkg(n) for n in ('sweep', 'diffusion', 'train', 'ldm', 'ldm_sweep', 'metrics'))
    assert sweep.dist_active() == forced
    out = {}
    sched = diffusion.DDPMScheduler()
    cfg = gc.TINY_CFG
    clean, noise = inputs(8, 16, 7, 8)
    # (1) Diff-Pruning: per-step scalar-loss all-reduce on the stream, early-exit state on the device, polled every 8 steps
    model = make_model(cfg, 5)
    flat = sweep.flatten_grads(model)
    res = sweep.taylor_sweep(model, sched, clean, noise, num_steps=40, thr=0.999, flat_grads=flat, use_graph=False)
    torch.cuda.synchronize()
    out['dp'] = dict(steps=res['steps'], losses=res['losses'], grads=flat.clone())
    # (1b) the host-synchronised form of the same loop (all-reduce, then float(loss) every step)
    model = make_model(cfg, 5)
    flat = sweep.flatten_grads(model)
    res = sweep.taylor_sweep(model, sched, clean, noise, num_steps=40, thr=0.999, flat_grads=flat, use_graph=False, device_exit=False)
    out['dp_host'] = dict(steps=res['steps'], losses=res['losses'], grads=flat.clone())
    # (2) plain Taylor with two timesteps in flight, then ONE flat-gradient all-reduce
    model = make_model(cfg, 5)
    flat = sweep.flatten_grads(model)
    step = sweep.HipSweepStep(model, sched, clean, noise, 8 * clean[0].numel(), 'mse', 8, timestep_pipelines=2)
    res = sweep.taylor_sweep(model, sched, clean, noise, num_steps=7, step_fn=step, flat_grads=flat, use_graph=False)
    torch.cuda.synchronize()
    assert step._tp
    pr = sweep.prune_model(model, 0.3)
    out['taylor2'] = dict(losses=res['losses'], grads=flat.clone(), masks=[r[3] for r in pr.records])
    # (3) finetune: three async gradient buckets enqueued at the backward milestones + the rest
    model = make_model(cfg, 5)
    eng = train.FinetuneEngine(model, sched, lr=2e-4, ema_decay=0.9999)
    t = torch.tensor([1, 250, 500, 998, 3, 7, 600, 999], device=DEV)
    losses = [float(eng.step(clean, noise, t)) for _ in range(2)]
    torch.cuda.synchronize()
    out['finetune'] = dict(losses=losses, norm=float(eng.last_grad_norm), params=eng.flat_p.clone(), ema=eng.ema.clone())
    # (4) LDM importance pass: latents sharded over the (one) rank, stream-ordered loss all-reduce, ratio-form early exit
    lcfg = gc.LDM_TINY_CFG
    lm = ldm.UNetModel(**lcfg)
    gc.det_init_(lm, 9)
    lm = lm.to(DEV).eval()
    emb = ldm_sweep.ClassEmbedder(16, 1001)
    with torch.no_grad():
        emb.embedding.weight.copy_(torch.from_numpy(gc.det_noise((1001, 16), 77)))
    emb = emb.to(DEV)
    import random
    res = ldm_sweep.ldm_importance_sweep(lm, emb, num_steps=4, thr=0.5, n_samples=3, ddim_steps=4, latent_shape=(3, 16, 16),
                                         class_rng=random.Random(3), seed=11)
    torch.cuda.synchronize()
    out['ldm'] = dict(steps=res['steps'], accumulated=res['accumulated'], losses=res['losses'], grads=res['flat_grads'].clone())
    if forced:
        # (4b) round 5: the sampler sharded by CFG rows (what 4 ranks x 6 latents run: 3 rows each) -- here the one rank owns all
        # 2 n rows, so every DDIM step's eps exchange goes through RCCL and the result must equal the latent-split pass closely
        # (row-batched forwards instead of the shared-stem CFG pair: fp32 re-association only)
        lm2 = ldm.UNetModel(**lcfg)
        gc.det_init_(lm2, 9)
        lm2 = lm2.to(DEV).eval()
        res2 = ldm_sweep.ldm_importance_sweep(lm2, emb, num_steps=4, thr=0.5, n_samples=3, ddim_steps=4, latent_shape=(3, 16, 16),
                                              class_rng=random.Random(3), seed=11, sampler_shard='rows')
        torch.cuda.synchronize()
        assert tuple(res2['sampler_rows']) == (0, 6)
        out['ldm_rows'] = dict(steps=res2['steps'], accumulated=res2['accumulated'], losses=res2['losses'],
                               grads=res2['flat_grads'].clone())
    # (5) FID statistics: all_gather of the has-data flags, broadcast of the shift, ONE all-reduce of (n, s1, s2)
    st = metrics.FeatureStats(24, torch.device(DEV))
    feats = torch.from_numpy(np.random.default_rng(4).standard_normal((40, 24)).astype(np.float32)).to(DEV)
    st.update(feats[:16])
    st.update(feats[16:])
    st.all_reduce()
    mu, sigma = st.finalize()
    out['fid'] = dict(mu=mu, sigma=sigma, reduced=getattr(st, '_reduced', None) is not None)
    return out


def main():
    report_path = sys.argv[1]
    port = sys.argv[2]
    torch.cuda.set_device(0)
    plain = run_all(False)
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%s' % port, rank=0, world_size=1, device_id=torch.device('cuda:0'))
    try:
        assert dist.get_backend() == 'nccl'
        forced = run_all(True)
        warm = torch.ones(4, device=DEV)
        dist.all_reduce(warm)
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    rep = {}
    ok = True

    def same(key, a, b, exact=True, tol=0.0):
        nonlocal ok
        if torch.is_tensor(a):
            good = torch.equal(a, b) if exact else relerr(a, b) <= tol
            rep[key] = dict(equal=bool(torch.equal(a, b)), rel=relerr(a, b))
        elif isinstance(a, np.ndarray):
            good = bool(np.array_equal(a, b)) if exact else bool(np.allclose(a, b, rtol=tol, atol=tol))
            rep[key] = dict(equal=bool(np.array_equal(a, b)))
        else:
            good = a == b if exact else all(abs(x - y) <= tol * abs(y) for x, y in zip(a, b)) if isinstance(a, list) else abs(a - b) <= tol * abs(b)
            rep[key] = dict(equal=bool(a == b))
        if not good:
            ok = False
            rep[key]['FAILED'] = True

    for k in ('dp', 'dp_host'):
        assert 1 < plain[k]['steps'] < 40, plain[k]['steps']          # the threshold fires inside the sweep
        same(k + '/steps', forced[k]['steps'], plain[k]['steps'])
        same(k + '/losses', forced[k]['losses'], plain[k]['losses'])
        same(k + '/grads', forced[k]['grads'], plain[k]['grads'])
    same('dp_vs_host/steps', plain['dp']['steps'], plain['dp_host']['steps'])
    same('taylor2/losses', forced['taylor2']['losses'], plain['taylor2']['losses'])
    same('taylor2/grads', forced['taylor2']['grads'], plain['taylor2']['grads'])
    same('taylor2/masks', forced['taylor2']['masks'], plain['taylor2']['masks'])
    # finetune: inside a process group the time-embedding projections' gradients take the per-block path (final at their
    # segment's milestone) instead of the two batched GEMMs at the end of the backward pass: same sums, re-associated
    same('finetune/losses', forced['finetune']['losses'], plain['finetune']['losses'], exact=False, tol=1e-6)
    same('finetune/norm', forced['finetune']['norm'], plain['finetune']['norm'], exact=False, tol=1e-5)
    same('finetune/params', forced['finetune']['params'], plain['finetune']['params'], exact=False, tol=1e-5)
    same('finetune/ema', forced['finetune']['ema'], plain['finetune']['ema'], exact=False, tol=1e-6)
    for k in ('steps', 'accumulated', 'losses', 'grads'):
        same('ldm/' + k, forced['ldm'][k], plain['ldm'][k])
    same('ldm_rows/steps', forced['ldm_rows']['steps'], plain['ldm']['steps'])
    same('ldm_rows/accumulated', forced['ldm_rows']['accumulated'], plain['ldm']['accumulated'])
    same('ldm_rows/losses', torch.tensor(forced['ldm_rows']['losses']), torch.tensor(plain['ldm']['losses']), exact=False, tol=1e-5)
    same('ldm_rows/grads', forced['ldm_rows']['grads'], plain['ldm']['grads'], exact=False, tol=5e-5)
    assert forced['fid']['reduced'] and not plain['fid']['reduced']
    same('fid/mu', forced['fid']['mu'], plain['fid']['mu'], exact=False, tol=1e-9)
    same('fid/sigma', forced['fid']['sigma'], plain['fid']['sigma'], exact=False, tol=1e-9)
    rep['ok'] = ok
    rep['ldm_steps'] = [plain['ldm']['steps'], plain['ldm']['accumulated']]
    rep['dp_steps'] = plain['dp']['steps']
    with open(report_path, 'w') as f:
        json.dump(rep, f, indent=1, sort_keys=True)
    print(json.dumps(rep, sort_keys=True))
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
