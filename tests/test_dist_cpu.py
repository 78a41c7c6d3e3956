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
