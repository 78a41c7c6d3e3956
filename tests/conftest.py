import importlib
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests', 'golden')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # The oracle legs are fp32 PyTorch on the HOST: 32 x 32-pixel convolutions at batch <= 128 are oversubscribed on the GPU box's 256
    # cores (bench.py's cpu_baseline sweep, profiles/round5_bench_line.json: 52.6 images/s on 16 threads, 28.7 on 32, 6.3 on 128).
    # DP_TEST_THREADS overrides; machines with fewer cores keep what they have.
    import torch
    torch.set_num_threads(min(int(os.environ.get('DP_TEST_THREADS', '16')), torch.get_num_threads()))


@pytest.fixture(scope='session')
def dp():
    """The product package (directory name contains a hyphen, so it is imported by string)."""
    return importlib.import_module('diff-pruning_amd')


_REPORT = {}


@pytest.fixture(scope='session')
def report():
    """Numeric evidence collected by the GPU tests; dumped to gpurun_out/ so a run can be inspected afterwards."""
    yield _REPORT
    out = os.path.join(ROOT, 'gpurun_out')
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, 'test_report.json'), 'w') as f:
            json.dump(_REPORT, f, indent=1, sort_keys=True)
    except OSError:
        pass
