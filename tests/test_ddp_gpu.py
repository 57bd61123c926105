"""Data-parallel train step on the GPU with real kernels: two ranks share the ONE card of the test box
and exchange gradients over gloo (RCCL refuses two ranks on one device; the driver's multi-GPU bench is the RCCL run).
What is checked is everything around the collective — side-stream ordering of the per-stage all-reduces against the
weight-gradient kernels before them and the Adam kernel after them, bucket coverage, the 1/world factor:

* identical shards on both ranks: the summed gradient is exactly 2x the local one and (g + g) * 0.5 == g, so the DDP
  run must reproduce a single-process run BIT FOR BIT (first-step gradients, the loss sequence and the final weights):
  every reduction of the step is a fixed-order sum (no float atomics), whichever way the two processes share the card;
* different shards: replicas stay BIT-identical to each other (same summed gradient, element-wise Adam) and the loss of
  every rank is finite.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

CFG = dict(num_classes=5, conv_dim=8, size=64, batch=2, steps=3)


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _train(model_dtype, shard, ddp, per_tile=False):
    """K train steps (trainer.py:172-176 order) on shard `shard`; returns (losses, flat parameter vector).  per_tile: the
    Winograd grid GradSync(wino_per_tile=True) selects when there is more than one rank (one workgroup per tile: other
    partial rows of the BatchNorm statistics than the persistent grid -- each bit-reproducible, not bit-identical to each other)."""
    import continual_learning_amd as C
    dev = torch.device('cuda', 0)
    torch.manual_seed(7)
    model = C.UNet(CFG['num_classes'], 3, CFG['conv_dim'], compute_dtype=model_dtype).to(dev).train()
    opt = C.FusedAdam(model.parameters(), lr=1e-3, betas=[0.5, 0.99])
    crit = C.CrossEntropyLoss()
    if per_tile:
        model.tuning.wino_persist = 0
    if ddp:
        C.ddp.broadcast_parameters(model)
        # small buckets: several collectives per backward; bit-level claims need the fp32 exchange (a bf16 model defaults to bf16)
        C.ddp.GradSync(model, opt, min_bucket_bytes=16 << 10, wino_per_tile=per_tile, grad_dtype=os.environ.get('TEST_GRAD_DTYPE', 'fp32'))
        assert model.tuning.wino_persist == (0 if per_tile and dist.get_world_size() > 1 else 1)
    b, s = CFG['batch'], CFG['size']
    x = torch.from_numpy(C.synth.images(99, b, 3, s, s, first_image=shard * b)).to(dev)
    y = torch.from_numpy(C.synth.labels(99, b, s, s, CFG['num_classes'], first_image=shard * b)).to(dev)
    losses, grad0 = [], None
    for i in range(CFG['steps']):
        out = model(x)
        opt.zero_grad()
        loss = crit(out, y)
        loss.backward()
        if i == 0:
            if ddp:
                model.grad_sync.wait()                             # the optimiser's pre-step hook; idempotent
            grad0 = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).cpu()
        opt.step()
        losses.append(float(loss.detach()))
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu()
    return losses, flat, grad0


def _worker(rank, world, port, dtype, same_shard, q, per_tile=False):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        losses, flat, grad0 = _train(dtype, 0 if same_shard else rank, ddp=True, per_tile=per_tile)
        q.put((rank, losses, flat.numpy(), grad0.numpy()))
    finally:
        dist.destroy_process_group()


def _run_world2(dtype, same_shard, per_tile=False):
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, dtype, same_shard, q, per_tile)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda r: r[0])
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize('dtype,per_tile', [('fp32', False), ('bf16', False), ('bf16x3', False), ('fp32', True)])
def test_ddp_identical_shards_reproduce_single_process(dtype, per_tile):
    # per_tile: the opt-in one-workgroup-per-tile Winograd grid (GradSync(wino_per_tile=True)); the reference uses the same grid
    ref_losses, ref_flat, ref_grad = _train(dtype, 0, ddp=False, per_tile=per_tile)
    for rank, losses, flat, grad0 in _run_world2(dtype, same_shard=True, per_tile=per_tile):
        g = torch.from_numpy(grad0)
        assert torch.equal(g, 2 * ref_grad), f'rank {rank}: summed gradient is not exactly 2x the local gradient ' \
                                             f'(rel {float((g - 2 * ref_grad).norm() / (2 * ref_grad).norm()):.2e})'
        assert losses == ref_losses, f'rank {rank}: {losses} vs {ref_losses}'
        assert torch.equal(torch.from_numpy(flat), ref_flat), f'rank {rank}: weights after {CFG["steps"]} steps differ'


def test_ddp_replicas_stay_identical():
    (r0, l0, f0, g0), (r1, l1, f1, g1) = _run_world2('fp32', same_shard=False)
    assert (g0 == g1).all(), 'summed gradients differ between ranks'
    assert (f0 == f1).all(), 'replicas diverged'
    assert all(map(lambda v: v == v and abs(v) < 1e3, l0 + l1))
    assert l0 != l1                                             # different shards, different local losses


def test_rccl_single_rank_process_group():
    """The RCCL ("nccl") code path itself — communicator creation on the device, all-reduce launches on the side stream
    between our kernels — with the one rank a 1-GPU box allows: a world-1 sum is the identity, so the run must match the
    plain single-process run bit for bit."""
    ref_losses, ref_flat, ref_grad = _train('fp32', 0, ddp=False)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), RANK='0', WORLD_SIZE='1')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        losses, flat, grad0 = _train('fp32', 0, ddp=True)
    finally:
        dist.destroy_process_group()
    assert torch.equal(grad0, ref_grad), f'rel {float((grad0 - ref_grad).norm() / ref_grad.norm()):.2e}'
    assert losses == ref_losses
    assert torch.equal(flat, ref_flat)


def test_ddp_bf16_gradient_exchange_tracks_the_fp32_exchange():
    """GradSync(grad_dtype='bf16') on the card (conversion kernels on the side stream around the collective, buckets that
    start at arbitrary elements of the flat buffer): two ranks on different shards stay bit-identical to each other, and
    the summed gradient of the first step is within 1e-2 of the run that exchanges fp32 -- what rounding the summed gradient to bf16
    costs (north_star configs[2]); the 3-step weight update stays the same optimisation (see the bound below)."""
    ref = _run_world2('bf16', same_shard=False)
    os.environ['TEST_GRAD_DTYPE'] = 'bf16'
    try:
        got = _run_world2('bf16', same_shard=False)
    finally:
        del os.environ['TEST_GRAD_DTYPE']
    (_, l0, f0, g0), (_, l1, f1, g1) = got
    assert (f0 == f1).all() and (g0 == g1).all(), 'replicas diverged under the bf16 exchange'
    gr, fr = torch.from_numpy(ref[0][3]), torch.from_numpy(ref[0][2])
    rel_g = float((torch.from_numpy(g0) - gr).norm() / gr.norm())
    print(f'bf16 exchange: summed gradient rel {rel_g:.2e}')
    assert 0 < rel_g < 1e-2, rel_g                   # two bf16 roundings of the summed gradient
    import continual_learning_amd as C
    torch.manual_seed(7)
    w_init = torch.cat([p.detach().reshape(-1) for p in C.UNet(CFG['num_classes'], 3, CFG['conv_dim'], compute_dtype='bf16').parameters()])
    upd_ref, upd = fr - w_init, torch.from_numpy(f0) - w_init
    rel_u = float((upd - upd_ref).norm() / upd_ref.norm())
    print(f'bf16 exchange: update after {CFG["steps"]} steps rel {rel_u:.2e}')
    # Adam's sign-like first steps amplify rounding-level gradient differences into different trajectories: after ONE update the
    # second-step gradients of stock torch fp32 and stock torch fp64 are already 10-33 % apart on this network (tests/diag/fold_two_step.py);
    # 0.24-0.27 measured here.  The bound only says the bf16 exchange is not a different optimisation (a sign error would give ~1.4).
    assert rel_u < 0.5, rel_u
