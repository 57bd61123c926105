"""The driver's multi-GPU command line, rehearsed on the 1-GPU box: `python -m torch.distributed.run --nproc-per-node 2
bench.py --gpus 2` with both ranks on the one card over gloo (bench.py's CLAMD_BENCH_BACKEND switch; RCCL refuses two
ranks per device).  Checks the whole distributed code path of bench.py — rendezvous, parameter broadcast, GradSync,
barriers, max-over-ranks timing, ONE JSON line from rank 0 with the whole-job aggregate."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_on_one_gpu():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, CLAMD_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
           '--size', '128', '--batch', '4']
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]                  # rank 0 only
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 3 and d['warmup'] == 1 and d['scaling'] == 'weak'
    assert d['config']['global_batch'] == 8 and d['config']['parallelism'] == 'dp2'
    assert d['value'] == pytest.approx(8 * 3 / (d['ms_per_step'] * 3 / 1e3), rel=1e-3)      # all ranks' images / max-over-ranks time
    assert 'cpu_baseline' not in d and 'also' not in d                                       # N > 1: headline only
    assert d['config']['final_loss'] == d['config']['final_loss']                            # finite
