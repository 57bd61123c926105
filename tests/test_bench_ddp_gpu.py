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


@pytest.mark.parametrize('launcher', ['torch.distributed.run', 'bare'])
def test_bench_two_ranks_on_one_gpu(launcher):
    """launcher='bare': `python bench.py --gpus 2` with no launcher at all -- bench.py starts the two rank processes itself
    (fresh processes, before any GPU call of the parent), rank 0 prints the one JSON line, a dead rank gives a non-zero exit."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, CLAMD_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    tail = [os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--size', '128', '--batch', '4']
    if launcher == 'bare':
        cmd = [sys.executable] + tail
    else:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
               '--master-port', str(port)] + tail
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]                  # rank 0 only
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 3 and d['warmup'] == 1 and d['scaling'] == 'weak'
    assert d['config']['global_batch'] == 8 and d['config']['parallelism'] == 'dp2'
    assert d['value'] == pytest.approx(8 * 3 / (d['ms_per_step'] * 3 / 1e3), rel=1e-3, abs=0.006)   # all ranks' images / max-over-ranks time (value is printed with 2 decimals)
    assert 'cpu_baseline' not in d and 'also' not in d                                       # N > 1: headline only
    assert d['config']['final_loss'] == d['config']['final_loss']                            # finite
    c = d['comm']
    assert c['rccl_ranks'] == 2 and c['backend'] == 'gloo' and c['collectives_per_step'] >= 1
    assert c['gradient_bytes_per_step'] == 4 * 31_044_821 and c['exposed_comm_ms_per_step'] >= 0.0
    assert c['cu_reserve'] == 0                                                               # gloo holds no CUs
    # per-bucket record: every gradient byte is exchanged exactly once per step, launches ordered in time
    assert c['exchange_dtype'] == 'fp32' and sum(b['bytes'] for b in c['buckets']) == c['gradient_bytes_per_step']
    offs = [b['launch_offset_ms'] for b in c['buckets']]
    assert offs == sorted(offs) and offs[0] >= 0.0


def test_bench_bare_form_reports_a_dead_rank():
    """A rank that fails (here: an impossible image size) must end the whole bare-form job with a non-zero exit code."""
    env = dict(os.environ, CLAMD_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0', '--size', '40',
                          '--batch', '1'], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith('{')]


def test_bench_rccl_code_path_with_one_rank():
    """The N > 1 code path of bench.py with the REAL backend -- ddp.init_rccl (channel cap in the environment), RCCL
    communicator, per-stage all-reduces on the side stream, exposed-communication timing, the `comm` record -- rehearsed
    with the one rank a one-GPU box allows (CLAMD_BENCH_FORCE_DIST)."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, CLAMD_BENCH_FORCE_DIST='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('NCCL_MAX_NCHANNELS', None); env.pop('NCCL_MIN_NCHANNELS', None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1',
           '--size', '128', '--batch', '4']
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][0])
    c = d['comm']
    assert c['backend'] == 'nccl' and c['rccl_ranks'] == 1 and c['NCCL_MAX_NCHANNELS'] == '8' and c['NCCL_MIN_NCHANNELS'] == '4'
    assert c['collectives_per_step'] >= 1 and c['exposed_comm_ms_per_step'] >= 0.0 and c['cu_reserve'] == 0
    assert 'roofline' in d and d['n_gpus'] == 1


@pytest.mark.parametrize('dtype', ['bf16', 'fp32'])
def test_step_survives_stolen_cus(dtype):
    """SURVEY.md §8e ('cap RCCL channels/CUs'), rehearsed on one GPU: a dummy kernel holds 8 CUs for the whole measurement,
    as 8 RCCL channel workgroups would during a collective.  With the weight-gradient kernels on their second stream
    (unet.WGRAD_STREAM) the workgroups of two kernels share whatever is free, and the default grids lose 1.19x (bf16) /
    1.40x (fp32) while the CUs are held (one stream: 1.36x / 1.69x); the one-workgroup-per-tile Winograd grid loses 1.21x
    (fp32) but costs 6 % when nothing is held.  DESIGN.md §5 has the break-even; here: that nothing queues behind the
    holder (a step that waited for it would take > 10x)."""
    # its own process with GPU_MAX_HW_QUEUES=8, as every multi-rank process runs (bench.py, ddp.init_rccl): a rank has the default
    # stream, the engine's second and third streams, GradSync's stream and RCCL's -- here the holder's -- and with the default 4
    # hardware queues two of them share one (measured: the step then queues behind the holder, 118 instead of 21 ms)
    env = dict(os.environ, GPU_MAX_HW_QUEUES='8')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'cu_steal.py'), dtype, '8', '5', 'basic'], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    print(r)
    # the measured ratios (1.2x / 1.4x) live in profiles/ and DESIGN.md section 5; a correctness suite on a shared or throttled
    # GPU only asserts that the step does not queue behind the holder
    assert r['stolen_over_base'] < 5 and r['reserved_over_base'] < 5 and r['pertile_over_base'] < 5, r


def test_hardware_queue_count_is_measured():
    """ddp.hw_queues: the number of hardware queues the runtime multiplexes streams onto, measured with spin kernels on eight streams (the
    engine's third stream and GradSync's warning depend on it; GPU_MAX_HW_QUEUES is only read when the runtime starts).  Two processes, the
    variable set to 2 and to 8 before HIP starts: the measurement follows."""
    code = ("import torch, continual_learning_amd as C; from continual_learning_amd import ddp; "
            "print('Q', ddp.hw_queues(torch.device('cuda', 0)), ddp._HW_PROBE)")
    got = {}
    for q in ('2', '8'):
        env = dict(os.environ, GPU_MAX_HW_QUEUES=q)
        out = subprocess.run([sys.executable, '-c', code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        line = [l for l in out.stdout.splitlines() if l.startswith('Q ')][-1]
        print(q, line)
        got[q] = int(line.split()[1])
    print(got)
    assert got['2'] <= 2 < got['8'], got


def test_bench_single_gpu_line_carries_calibration_and_traffic_fields():
    """The one-GPU driver line: `roofline` with the sustained-MFMA calibration measured in the same process, the traffic ratio fields (null for a
    workload no PMC profile was committed for), an `also` entry per further dtype -- on a small workload so that the test stays short."""
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'CLAMD_BENCH_BACKEND', 'CLAMD_BENCH_FORCE_DIST'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '2', '--warmup', '1', '--size', '128', '--batch', '4',
                          '--no-cpu-baseline', '--also', 'bf16'], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['n_gpus'] == 1 and d['dtype'] == 'f32' and d['unit'] == 'images/sec' and d['vs_baseline'] is None
    r = d['roofline']
    assert r['bound'] == 'mfma' and r['peak'] == 157.3 and 0 < r['frac'] < 1
    assert 100 < r['sustained_mfma_tflops_measured'] < 160 and abs(r['frac_of_sustained'] - r['achieved'] / r['sustained_mfma_tflops_measured']) < 2e-3
    # 128 x 128 bs4 has no committed PMC profile: the traffic fields are there and empty, nothing is invented
    assert 'hbm_kernels' in d and d['l2_fabric_bytes_per_step'] is None and d['algorithmic_bytes_per_step'] is None and d['fabric_over_algorithmic'] is None
    (a,) = d['also']
    assert a['dtype'] == 'bf16' and 1000 < a['mfma_sustained_tflops_measured'] < 2600 and 0 < a['conv3x3_igemm_frac_of_sustained'] < 1.2
