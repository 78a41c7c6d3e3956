"""The data-parallel code paths against RCCL itself (round-3 verdict: "no line of this repo has ever run against RCCL").

The GPU test box has ONE MI355X, so the process group has one rank; DP_FORCE_DIST=1 makes the product take its distributed
branches anyway (sweep.dist_active).  What this proves: `async_op=True` work handles on slices of the flat gradient buffer, the
eager `device_id=` init, the ordering between the engine's side streams / second timestep pipeline and RCCL's stream, and
`dist.all_reduce` inside the poll loop all behave under the `nccl` backend as they do under gloo -- results equal the
non-distributed run (a one-rank sum is the identity).  What it cannot prove: xGMI bandwidth or multi-rank scaling."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_every_exchange_step_of_the_path_runs_through_a_one_rank_rccl_group(tmp_path, report):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / 'rccl.json')
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    p = subprocess.run([sys.executable, os.path.join(HERE, '_rccl_worker.py'), out, str(port)], env=env, timeout=600,
                       capture_output=True, text=True)
    assert p.returncode == 0, (p.stdout[-3000:], p.stderr[-3000:])
    rep = json.load(open(out))
    report['e2e/rccl_one_rank'] = rep
    assert rep['ok'] is True
    for k in ('dp/grads', 'dp_host/grads', 'taylor2/grads', 'taylor2/masks', 'ldm/grads', 'ldm/losses'):
        assert rep[k]['equal'] is True, k
