import importlib
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests', 'golden')):
    if p not in sys.path:
        sys.path.insert(0, p)

OUT = os.path.join(ROOT, 'gpurun_out')
# A test that runs in a child interpreter (see `isolated` below) is told so through the environment.
CHILD = os.environ.get('DP_TEST_CHILD') == '1'
_T0 = time.time()
_CONFIG = None


def pytest_configure(config):
    global _CONFIG
    _CONFIG = config
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # The oracle legs are fp32 PyTorch on the HOST: 32 x 32-pixel convolutions at batch <= 128 are oversubscribed on the GPU box's 256
    # cores (bench.py's cpu_baseline sweep, profiles/round5_bench_line.json: 52.6 images/s on 16 threads, 28.7 on 32, 6.3 on 128).
    # DP_TEST_THREADS overrides; machines with fewer cores keep what they have.
    import torch
    torch.set_num_threads(min(int(os.environ.get('DP_TEST_THREADS', '16')), torch.get_num_threads()))
    if not CHILD:
        try:
            os.makedirs(OUT, exist_ok=True)
            with open(os.path.join(OUT, 'last_test.txt'), 'w') as f:
                f.write('# node ids in start order; the last line is the test that was running when the process ended\n')
        except OSError:
            pass


def _breadcrumb(text):
    """One line on the REAL stdout (pytest's terminal writer, never captured) and appended + fsync'ed to gpurun_out/last_test.txt:
    whatever kills the interpreter (SIGABRT from the HIP runtime, glibc, the OOM killer) leaves the culprit's node id behind
    (round-5 driver run: rc 134 with no test named)."""
    line = '[dp-test %7.1fs pid %d] %s' % (time.time() - _T0, os.getpid(), text)
    try:
        tw = _CONFIG.get_terminal_writer()
        tw.line()
        tw.line(line)
        tw.flush()
    except Exception:
        pass
    try:
        with open(os.path.join(OUT, 'last_test.txt'), 'a') as f:
            f.write(line + '\n')
            f.flush()
            os.fsync(f.fileno())
    except OSError:
        pass


def pytest_runtest_logstart(nodeid, location):
    _breadcrumb('START ' + nodeid)


def pytest_runtest_logfinish(nodeid, location):
    if CHILD:
        _breadcrumb('END   ' + nodeid)


@pytest.fixture(scope='session')
def dp():
    """The product package (directory name contains a hyphen, so it is imported by string)."""
    return importlib.import_module('diff-pruning_amd')


_REPORT = {}


@pytest.fixture(scope='session')
def report():
    """Numeric evidence collected by the GPU tests; dumped to gpurun_out/ so a run can be inspected afterwards.  A child
    interpreter (`isolated`) dumps to the file its parent names and the parent merges it."""
    yield _REPORT
    path = os.environ.get('DP_TEST_REPORT') if CHILD else os.path.join(OUT, 'test_report.json')
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, 'w') as f:
            json.dump(_REPORT, f, indent=1, sort_keys=True)
    except OSError:
        pass


_SIGNALLED = []


def pytest_terminal_summary(terminalreporter):
    for node, rc, tail in _SIGNALLED:
        terminalreporter.write_line('[dp-test] CHILD KILLED BY SIGNAL %d (re-run once): %s' % (-rc, node))
        terminalreporter.write_line(tail)


def run_isolated(nodeid, tmp_dir, timeout=900, extra_env=None):
    """Run ONE test node in a child interpreter (the driver's own pytest flags) and return (returncode, tail of its output,
    its report dict).  A signal in the child (SIGABRT = -6, SIGSEGV = -11) is one red test carrying the child's stderr --
    "Memory access fault by GPU node ..." or the glibc message -- instead of a dead suite with zero recorded passes."""
    rep_path = os.path.join(str(tmp_dir), 'child_report.json')
    log_path = os.path.join(str(tmp_dir), 'child.log')
    env = dict(os.environ, DP_TEST_CHILD='1', DP_TEST_REPORT=rep_path, PYTHONFAULTHANDLER='1')
    env.setdefault('AMD_LOG_LEVEL', '1')                     # HIP runtime errors only; they land in the child's log
    env.update(extra_env or {})
    nodeids = [nodeid] if isinstance(nodeid, str) else list(nodeid)
    nodeid = nodeids[0] if len(nodeids) == 1 else '%s (+%d)' % (nodeids[0], len(nodeids) - 1)
    cmd = [sys.executable, '-m', 'pytest'] + nodeids + ['-x', '-q', '-m', 'gpu', '-p', 'no:cacheprovider', '--capture=sys', '--durations=0']
    t0 = time.time()
    with open(log_path, 'wb') as log:
        try:
            rc = subprocess.run(cmd, cwd=ROOT, env=env, stdout=log, stderr=subprocess.STDOUT, timeout=timeout).returncode
        except subprocess.TimeoutExpired:
            rc = 'timeout after %d s' % timeout
    with open(log_path, 'rb') as f:
        tail = f.read()[-6000:].decode('utf-8', 'replace')
    rep = {}
    if os.path.exists(rep_path):
        with open(rep_path) as f:
            rep = json.load(f)
    _breadcrumb('child %s rc=%s %.1fs' % (nodeid, rc, time.time() - t0))
    return rc, tail, rep


def isolated(timeout=900, params=()):
    """Decorator for the full-size tests (tens of seconds each, the C1 ... C5 configurations): in the driver's process the test
    body is replaced by a child run of the same node id; in the child (DP_TEST_CHILD=1) the body itself runs.  `params`: the
    test's parametrize argument names (they select the node id; fixtures are only needed in the child)."""
    def deco(fn):
        if CHILD:
            return fn
        import functools
        import inspect

        @functools.wraps(fn)
        def wrapper(*args, **kwargs):
            if 'request' not in kwargs:                      # called as a plain function by another test: run the body here
                return fn(*args, **kwargs)
            request = kwargs.pop('request')
            tmp = request.getfixturevalue('tmp_path')
            rc, tail, rep = run_isolated(request.node.nodeid, tmp, timeout)
            if isinstance(rc, int) and rc < 0:
                # killed by a signal (SIGABRT -6, SIGSEGV -11, SIGKILL -9): keep the evidence -- log line, report entry, terminal
                # summary -- and run the node ONCE more, so that a crash that depends on the lease's timing is recorded as such
                # (first attempt died, second passed / died again) instead of ending the suite with the other rows unrun
                _SIGNALLED.append((request.node.nodeid, rc, tail[-1500:]))
                _REPORT.setdefault('harness/child_signals', []).append(dict(node=request.node.nodeid, rc=rc, tail=tail[-1500:]))
                rc, tail, rep = run_isolated(request.node.nodeid, tmp, timeout)
            _REPORT.update(rep)
            assert rc == 0, 'child interpreter for %s ended with %s\n%s' % (request.node.nodeid, rc, tail)

        # pytest resolves fixtures from the signature: keep only `request` (and parametrize arguments, which select the node id)
        sig = inspect.signature(fn)
        keep = [p for n, p in sig.parameters.items() if n in params]
        wrapper.__signature__ = sig.replace(parameters=keep + [inspect.Parameter('request', inspect.Parameter.POSITIONAL_OR_KEYWORD)])
        return wrapper
    return deco
