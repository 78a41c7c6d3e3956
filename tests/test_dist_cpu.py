"""world_size-2 (gloo, CPU) test of the data-parallel sweep and of the data-parallel finetune step: batch shards + scalar-loss all-reduce for the early exit +
one gradient all-reduce give the same stop step, the same gradients and the same prune masks as a single process."""
import os
import socket
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, outdir):
    port = str(_free_port())
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, '_dist_worker.py'), str(r), str(world), port, outdir])
             for r in range(world)]
    for p in procs:
        assert p.wait(timeout=600) == 0


def test_two_rank_sweep_equals_single_process(tmp_path):
    out = str(tmp_path)
    _run(1, out)
    _run(2, out)
    one = torch.load(os.path.join(out, 'r0_w1.pt'))
    r0 = torch.load(os.path.join(out, 'r0_w2.pt'))
    r1 = torch.load(os.path.join(out, 'r1_w2.pt'))
    assert r0['global_batch'] == r1['global_batch'] == one['global_batch'] == 4
    assert r0['steps'] == r1['steps'] == one['steps'] and 1 < one['steps'] < 50     # early exit taken, same step
    for a, b, c in zip(one['losses'], r0['losses'], r1['losses']):
        assert abs(a - b) <= 1e-5 * abs(a) and b == c
    for n, g in one['grads'].items():
        assert torch.equal(r0['grads'][n], r1['grads'][n])                  # all-reduced: identical on both ranks
        scale = float(g.abs().max())
        if scale > 1e-7:
            assert float((r0['grads'][n] - g).abs().max()) <= 2e-5 * scale, n
    assert r0['masks'] == r1['masks'] == one['masks']                       # identical prune masks
    # finetune steps (C4): same loss, same pre-clip gradient norm, same parameters after two Adam + EMA steps
    for a, b, c in zip(one['ft_losses'], r0['ft_losses'], r1['ft_losses']):
        assert abs(a - b) <= 1e-5 * abs(a) and b == c
    assert abs(one['ft_norm'] - r0['ft_norm']) <= 1e-4 * one['ft_norm'] and r0['ft_norm'] == r1['ft_norm']
    for n, p in one['ft_params'].items():
        assert torch.equal(r0['ft_params'][n], r1['ft_params'][n])
        assert float((r0['ft_params'][n] - p).abs().max()) <= 2e-4 * float(p.abs().max()) + 1e-7, n


def _run_ldm(world, outdir, thr, pipelines=None):
    port = str(_free_port())
    env = dict(os.environ, **({'DP_LDM_PIPELINES': str(pipelines)} if pipelines else {}))
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, '_dist_worker_ldm.py'), str(r), str(world), port, outdir,
                               str(thr)], env=env) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=900) == 0


def test_two_rank_ldm_importance_pass_equals_single_process(tmp_path):
    """Config C5's data-parallel form: the latents of a step sharded over two ranks -- 1 + 1 on the reference
    script's own recorded run (tests/golden/ldm_driver.json: losses, gradients after K backward passes, the break before the
    backward at t = 2), 2 + 1 on a 3-latent run with the default Philox draws -- give the losses, the stop step, the
    accumulated gradients and the 109 prune masks of one process."""
    import json
    out = str(tmp_path)
    _run_ldm(1, out, 0.97)
    _run_ldm(2, out, 0.97)
    one = torch.load(os.path.join(out, 'ldm_r0_w1.pt'))
    r0 = torch.load(os.path.join(out, 'ldm_r0_w2.pt'))
    r1 = torch.load(os.path.join(out, 'ldm_r1_w2.pt'))
    fx = json.load(open(os.path.join(HERE, 'golden', 'ldm_driver.json')))
    # (1) against the script's recorded run
    assert tuple(r0['driver']['shard']) == (0, 1) and tuple(r1['driver']['shard']) == (1, 2)
    for r in (one, r0, r1):
        d = r['driver']
        assert d['steps'] == d['accumulated'] == fx['K']
        assert all(abs(a - b) <= 2e-5 * abs(b) for a, b in zip(d['losses'], fx['losses'][:fx['K']]))
        for name, want in fx['grad_abs_sum_after_K'].items():
            assert abs(d['grad_abs_sum'][name] - want) < 1e-3 * want + 1e-6, name
        b, bc = r['break'], fx['break_case']
        assert b['steps'] == 3 and b['accumulated'] == 2
        assert all(abs(a - w) <= 2e-5 * abs(w) for a, w in zip(b['losses'], bc['printed_losses'] + [bc['breaking_loss']]))
        for name, got in b['grad_abs_sum'].items():
            want = bc['grad_abs_sum'].get(name, 0.0)
            assert abs(got - want) < 1e-3 * want + 1e-6, name
    assert r0['driver']['losses'] == r1['driver']['losses']
    # (2) uneven shards, default draws, early exit
    u1, a, b = one['uneven'], r0['uneven'], r1['uneven']
    assert tuple(a['shard']) == (0, 2) and tuple(b['shard']) == (2, 3) and tuple(u1['shard']) == (0, 3)
    # round 5: 3 latents on 2 ranks = 6 CFG forward rows; the latent split (2 + 1) would leave rank 0 four rows per DDIM step, so the
    # SAMPLER is sharded by rows (3 + 3: rank 0 the three unconditional forwards, rank 1 the three conditional ones, eps exchanged
    # by one all-reduce per DDIM step) while the scored forward / backward stays on the latent shards -- config C5's 12 rows on 4 GPUs
    assert tuple(a['sampler_rows']) == (0, 3) and tuple(b['sampler_rows']) == (3, 6) and u1['sampler_rows'] is None
    assert a['steps'] == b['steps'] == u1['steps'] == 3 and a['accumulated'] == b['accumulated'] == u1['accumulated'] == 2
    for x, y, z in zip(u1['losses'], a['losses'], b['losses']):
        assert abs(x - y) <= 1e-5 * abs(x) and y == z
    for n, g in u1['grads'].items():
        assert torch.equal(a['grads'][n], b['grads'][n])                    # all-reduced: identical on both ranks
        scale = float(g.abs().max())
        if scale > 1e-7:
            assert float((a['grads'][n] - g).abs().max()) <= 5e-5 * scale, n
    assert len(u1['masks']) == 109 and a['masks'] == b['masks'] == u1['masks']
    # (3) the same two-rank job with two importance steps in flight per rank (odd steps on a second engine and gradient buffer,
    #     the loss all-reduce + state update ordered across the pipelines): same losses, stop step, masks
    out2 = os.path.join(out, 'pipes2')
    os.makedirs(out2)
    _run_ldm(2, out2, 0.97, pipelines=2)
    p0, p1 = torch.load(os.path.join(out2, 'ldm_r0_w2.pt')), torch.load(os.path.join(out2, 'ldm_r1_w2.pt'))
    assert p0['uneven']['sampler_rows'] is None            # two steps in flight keep the latent split (no per-DDIM-step exchange)
    for key in ('driver', 'break', 'uneven'):
        if key == 'uneven':          # the one-pipeline run sharded its sampler by rows: same losses up to fp32 re-association
            assert all(abs(x - y) <= 1e-5 * abs(y) for x, y in zip(p0[key]['losses'], r0[key]['losses']))
        else:
            assert p0[key]['losses'] == r0[key]['losses']
        assert p0[key]['steps'] == r0[key]['steps']
        assert p0[key]['accumulated'] == r0[key]['accumulated'] and p1[key]['losses'] == p0[key]['losses']
    for n, g in a['grads'].items():
        scale = float(g.abs().max())
        assert torch.equal(p0['uneven']['grads'][n], p1['uneven']['grads'][n])
        if scale > 1e-7:
            assert float((p0['uneven']['grads'][n] - g).abs().max()) <= 1e-5 * scale, n
    assert p0['uneven']['masks'] == a['masks']


def test_rank_sharded_sampling_feeds_one_allreduce_fid_statistics(tmp_path):
    """ddpm_sample.py:55-74 on two ranks: per-rank folder `process_{rank}`, generator seed + rank, total // (batch * world)
    batches each; the FID feature statistics of both ranks' samples come out of ONE all-reduce of (n, sum, outer-product sum)
    and equal numpy's mean / cov over the union of the images the two folders hold."""
    import numpy as np
    from PIL import Image
    out = str(tmp_path)
    port = str(_free_port())
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, '_dist_worker_fid.py'), str(r), '2', port, out]) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    r0, r1 = np.load(os.path.join(out, 'fid_r0_w2.npz')), np.load(os.path.join(out, 'fid_r1_w2.npz'))
    assert int(r0['n']) == int(r1['n']) == 24
    assert np.array_equal(r0['mu'], r1['mu']) and np.array_equal(r0['sigma'], r1['sigma'])
    proj = np.random.default_rng(0).standard_normal((3 * 8 * 8, 12)).astype(np.float32)
    feats = []
    for r in range(2):
        d = os.path.join(out, 'process_%d' % r)
        for i in range(12):
            a = np.asarray(Image.open(os.path.join(d, '%d.png' % i)), dtype=np.float32) / 255.0          # [8, 8, 3]
            feats.append(a.transpose(2, 0, 1).reshape(-1) @ proj)
    feats = np.stack(feats).astype(np.float64)
    assert np.allclose(r0['mu'], feats.mean(0), rtol=1e-5, atol=1e-6)
    assert np.allclose(r0['sigma'], np.cov(feats, rowvar=False), rtol=1e-4, atol=1e-5)
    # the two ranks drew from different generators (seed + rank)
    a0 = np.asarray(Image.open(os.path.join(out, 'process_0', '0.png')))
    a1 = np.asarray(Image.open(os.path.join(out, 'process_1', '0.png')))
    assert not np.array_equal(a0, a1)
