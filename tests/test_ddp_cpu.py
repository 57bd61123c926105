"""world_size-2 gloo test of the data-parallel gradient exchange (ddp.GradSync): stage buckets are contiguous slices
of the flat gradient buffer, every element is all-reduced exactly once, the 1/world factor lands in the optimiser.
Runs on CPU tensors (the engine's buffers are plain torch tensors; no kernel is launched)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import continual_learning_amd as C
        from continual_learning_amd.unet import _Engine
        torch.manual_seed(rank)                          # different initial weights per rank on purpose
        m = C.UNet(5, 3, 4)
        C.ddp.broadcast_parameters(m, src=0)
        w0 = m.state_dict()['dec3.block.0.weight'].clone()
        gathered = [torch.zeros_like(w0) for _ in range(world)]
        dist.all_gather(gathered, w0)
        same = all(torch.equal(gathered[0], t) for t in gathered)
        opt = C.FusedAdam(m.parameters(), lr=1e-3, betas=[0.5, 0.99])
        sync = C.ddp.GradSync(m, opt, min_bucket_bytes=64 << 10)
        eng = _Engine(m, 2, 32, 32, torch.device('cpu'))
        n = eng.gflat.numel()
        eng.gflat.copy_(torch.arange(n, dtype=torch.float32) * (rank + 1))
        launches = []
        orig = sync._launch
        sync._launch = lambda flat: (launches.append((flat.data_ptr(), flat.numel())), orig(flat))[1]
        for st in reversed(eng.stages):                  # the order the backward produces gradients
            sync.stage_done(eng, st)
        sync.wait()
        expect = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))
        ok = torch.equal(eng.gflat, expect)
        base = eng.gflat.data_ptr()
        spans = sorted(((p - base) // 4, (p - base) // 4 + k) for p, k in launches)
        contiguous = spans[0][0] == 0 and spans[-1][1] == n and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        q.put((rank, same, ok, contiguous, len(launches), opt.grad_scale))
    finally:
        dist.destroy_process_group()


def test_gradsync_gloo_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for rank, same, ok, contiguous, nlaunch, gscale in res:
        assert same, 'broadcast_parameters did not equalise the replicas'
        assert ok, 'flat gradient buffer is not the sum over ranks'
        assert contiguous, 'buckets must tile the flat buffer exactly once'
        assert 2 <= nlaunch <= 9
        assert gscale == 0.5


def _worker_bf16(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import continual_learning_amd as C
        from continual_learning_amd.unet import _Engine
        torch.manual_seed(0)
        m = C.UNet(5, 3, 4, compute_dtype='bf16')
        opt = C.FusedAdam(m.parameters(), lr=1e-3, betas=[0.5, 0.99])
        sync = C.ddp.GradSync(m, opt, min_bucket_bytes=64 << 10)          # grad_dtype follows the model: bf16
        eng = _Engine(m, 2, 32, 32, torch.device('cpu'))
        n = eng.gflat.numel()
        g = torch.Generator().manual_seed(100 + rank)
        local = torch.randn(n, generator=g)
        eng.gflat.copy_(local)
        for st in reversed(eng.stages):
            sync.stage_done(eng, st)
        sync.wait()
        others = [torch.randn(n, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
        exact = sum(others)
        rel = float((eng.gflat - exact).norm() / exact.norm())
        q.put((rank, sync.grad_dtype, rel, sum(b['bytes'] for b in sync.bucket_report()), 4 * n, eng.gflat.clone().numpy()))
    finally:
        dist.destroy_process_group()


def test_gradsync_bf16_exchange_gloo_world2():
    """GradSync(grad_dtype='bf16') (BASELINE.json configs[2]: "bf16 DDP"): every bucket is rounded to bf16, summed and widened
    back -- half the bytes on the wire, the summed gradient within bf16 rounding of the fp32 exchange (north_star bounds the
    bf16 path at 1e-2-level agreement; here rel L2 < 1e-2), identical on every rank."""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker_bf16, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted((q.get(timeout=180) for _ in range(world)), key=lambda r: r[0])
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    for rank, gd, rel, wire, full, flat in res:
        assert gd == 'bf16'
        assert wire * 2 == full, (wire, full)                  # 2 bytes per gradient element on the wire
        assert 0 < rel < 1e-2, rel                             # measured 3.9e-3: two rne roundings of N(0,1) values
    assert (res[0][5] == res[1][5]).all(), 'ranks must end with the same summed gradient'
