import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def pytest_sessionstart(session):
    """Two-rank dry run of the data-parallel training bench on a 1-GPU box: both ranks on the one
    GPU, gloo instead of RCCL (RCCL refuses two ranks on one device).  It is started HERE, before this process has
    touched the GPU - a child must not be exec'ed from a process that has initialised it - and only when the GPU tests
    are selected on a box that has a GPU; tests/test_configs_gpu.py asserts on the recorded result."""
    session.config._gd4d_dp2_dryrun = session.config._gd4d_dp2_overlap = session.config._gd4d_dp2_infer = None
    mark = session.config.getoption('-m') or ''
    if 'gpu' not in mark or 'not gpu' in mark or os.environ.get('GD4D_SKIP_DP2_DRYRUN'):
        return
    if not os.path.exists('/dev/kfd'):                      # no GPU on this box; decided WITHOUT touching the HIP runtime:
        return                                              # a child must not be started from a process that initialised it
    env = dict(os.environ, GD4D_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')

    def two_ranks(*bench_args):
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
               '127.0.0.1', '--master-port', str(_free_port()), 'bench.py', '--gpus', '2', *bench_args]
        try:
            r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
            return dict(rc=r.returncode, out=r.stdout, err=r.stderr[-6000:])
        except Exception as e:                               # reported by the test that reads it
            return dict(rc=-1, out='', err=f'{type(e).__name__}: {e}')
    session.config._gd4d_dp2_dryrun = two_ranks('--mode', 'train', '--levels', 'vov', '--frames', '1', '--layers', '2', '--steps', '2',
                                                '--warmup', '1')
    # the same step with the per-layer gradient buckets all-reduced from autograd hooks underneath the backward (--overlap-comm)
    session.config._gd4d_dp2_overlap = two_ranks('--mode', 'train', '--levels', 'vov', '--frames', '1', '--layers', '2', '--steps', '2',
                                                 '--warmup', '1', '--overlap-comm')
    # the N > 1 INFERENCE line (replicas, two requests in flight per rank): what the driver's scaling runs launch
    session.config._gd4d_dp2_infer = two_ranks('--inflight', '2', '--frames', '1', '--steps', '3', '--warmup', '1', '--no-roofline', '--min-seconds', '0.05',
                                               '--no-cpu-baseline')


@pytest.fixture(scope='session')
def repo_root():
    return ROOT
