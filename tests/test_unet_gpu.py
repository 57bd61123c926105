"""GPU: whole-model parity of the drop-in UNet / CrossEntropyLoss / FusedAdam train step against golden vectors
captured from the real reference (tests/golden, oracle/gen_golden.py) and against the CPU oracles.

north_star tolerance: logits within 1e-3 relative (rel L2 and max-abs/max-abs) for the fp32 path; the bf16 path is
reported against the tolerance it can reach (bf16 storage of 23 stacked conv layers) and must keep
mIoU within +-0.1 of the reference.
"""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import rel_l2

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from oracle import np_unet as O
from oracle import torch_cpu as TC

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def C():
    import continual_learning_amd as C
    C._lib.load()
    return C


def load_closed_form(C, model, seed=0):
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()
              if not k.endswith(('running_mean', 'running_var', 'num_batches_tracked'))}
    st = C.synth.closed_form_state(shapes, seed)
    sd = model.state_dict()
    sd.update({k: torch.from_numpy(v) for k, v in st.items()})
    model.load_state_dict(sd, strict=True)
    return st


def maxrel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


@pytest.mark.parametrize('fixture,nc,cd,size', [('unet_cd4_c2_32.npz', 2, 4, 32), ('unet_cd8_c21_64.npz', 21, 8, 64)])
@pytest.mark.parametrize('dtype', ['fp32', 'fp32-direct', 'bf16', 'bf16x3'])
def test_train_steps_vs_reference_golden(C, golden, fixture, nc, cd, size, dtype, monkeypatch):
    # 'fp32' runs the Winograd kernels where the engine uses them, 'fp32-direct' the direct implicit-GEMM kernels
    from continual_learning_amd import unet as U
    direct = dtype == 'fp32-direct'
    if direct:
        monkeypatch.setattr(U, 'WINOGRAD', False)
        dtype = 'fp32'
    g = golden(fixture)
    model = C.UNet(nc, 3, cd, compute_dtype=dtype)
    assert [n for n, _ in model.named_parameters()] == list(g['grad_names'])      # names AND order (Adam state is positional)
    assert len(model.state_dict()) == 136
    load_closed_form(C, model)
    model = model.cuda().train()
    x = torch.from_numpy(C.synth.images(1234, 2, 3, size, size)).cuda()
    y = torch.from_numpy(C.synth.labels(1234, 2, size, size, nc)).cuda()
    opt = C.FusedAdam(model.parameters(), lr=float(g['lr']), betas=[0.5, 0.99])
    crit = C.CrossEntropyLoss()
    fp32 = dtype != 'bf16'          # 'bf16x3' (split-bf16, fp32 storage) is held to the fp32 path's bounds
    losses = []
    for s in range(3):
        out = model(x)                 # trainer.py:172-176 order
        opt.zero_grad()
        loss = crit(out, y)
        loss.backward()
        if s == 0:
            lg = out.detach().cpu().numpy()
            assert lg.shape == (2, nc, size, size)
            # bf16: measured 2.35e-2 / 1.98e-2 rel L2 and 3.9e-2 / 2.8e-2 max rel on the two fixtures (round 4, -s prints them): bounds <= 2x
            tol, tolm = (1e-3, 1e-3) if fp32 else (4.5e-2, 7.5e-2)
            print(f'measured[{fixture} {dtype}]: logits rel L2 {rel_l2(lg, g["logits"]):.3e}, max rel {maxrel(lg, g["logits"]):.3e}')
            assert rel_l2(lg, g['logits']) < tol and maxrel(lg, g['logits']) < tolm, (rel_l2(lg, g['logits']), maxrel(lg, g['logits']))
            gn = np.array([float(p.grad.double().norm()) for p in model.parameters()])
            # These toy models normalise over as few as 8 samples at the deepest level (2x2 spatial, batch 2), which
            # amplifies rounding differences by ~10^2-10^3; the full-size test below holds every tensor to 5e-3.
            if dtype == 'fp32':
                np.testing.assert_allclose(gn, g['grad_norms'], rtol=2e-3, atol=1e-6)
            elif dtype == 'bf16x3':
                np.testing.assert_allclose(gn, g['grad_norms'], rtol=3e-2, atol=1e-5)
            elif cd >= 8:     # bf16 on the 2x2-bottleneck toy (cd4 @32x32) is chaotic: BatchNorm over 8 samples
                big = g['grad_norms'] > 0.2 * g['grad_norms'].max()
                assert np.median(np.abs(gn[big] / g['grad_norms'][big] - 1)) < 0.5
            if 'g0/enc1.0.weight' in g.files:
                for n, p in model.named_parameters():
                    ref = g['g0/' + n]
                    if fp32 and np.linalg.norm(ref) > 1e-6:      # conv biases in front of BatchNorm have ~0 gradient
                        assert rel_l2(p.grad.cpu().numpy(), ref) < (5e-3 if dtype == 'fp32' else 5e-2), n
            stats = np.concatenate([v.cpu().numpy().reshape(-1) for k, v in model.state_dict().items()
                                    if k.endswith(('running_mean', 'running_var'))])
            assert rel_l2(stats, g['stats1']) < (1e-4 if dtype == 'fp32' else 2e-2)
            pred = out.detach().argmax(1)
            if fp32:
                assert np.array_equal(np.bincount(pred.cpu().numpy().reshape(-1), minlength=nc), g['pred_hist'])
            m = C.eval_metrics(y, out.detach(), nc)
            np.testing.assert_allclose([float(v) for v in m], g['metrics'], rtol=1e-5 if fp32 else 0.2, atol=0 if fp32 else 0.1)
        opt.step()
        losses.append(float(loss))
    np.testing.assert_allclose(losses, g['losses'], rtol=2e-4 if dtype == 'fp32' else (2e-3 if fp32 else 6e-2))
    # Weights after three Adam steps, element by element: only for the direct kernels.  These toy models (BatchNorm over
    # as few as 8 values, Adam steps of +-lr whatever the gradient's size) amplify rounding-level differences into sign
    # flips, so the check is tied to one summation order; the Winograd path is held to the losses, logits, gradient norms,
    # statistics and metrics above, to the per-kernel tests and to the full-size golden test.
    if dtype == 'fp32' and direct and 'w3/enc1.0.weight' in g.files:
        sd = model.state_dict()
        for k in g.files:
            if k.startswith('w3/'):
                got, ref = sd[k[3:]].cpu().numpy(), g[k]
                # Adam's sign-like first steps amplify rounding-level gradients (|update| = lr whatever |g|)
                bad = int((np.abs(got - ref) > 0.5 * float(g['lr'])).sum())
                is_stat = k.endswith(('running_mean', 'running_var'))
                assert rel_l2(got, ref) < 1e-2 and (is_stat or bad <= max(4, 0.05 * got.size)), k
    assert int(model.state_dict()['enc1.2.num_batches_tracked']) == 3


@pytest.mark.parametrize('dtype', ['fp32', 'bf16', 'bf16x3'])
def test_config1_exact_vs_reference_golden(C, golden, dtype):
    """BASELINE.json configs[0] exactly -- UNet(num_classes=2, conv_dim=64) at 64x64, batch 2 (the reference's CPU-runnable
    plumbing configuration) -- 3 train steps against the capture from the reference: full logits, loss sequence, per-tensor
    gradient norms, arg-max histogram, metrics."""
    g = golden('unet_cd64_c2_64.npz')
    assert (int(g['num_classes']), int(g['conv_dim']), int(g['batch']), int(g['size'])) == (2, 64, 2, 64)
    fp32 = dtype != 'bf16'
    model = C.UNet(2, 3, 64, compute_dtype=dtype)
    load_closed_form(C, model)
    model = model.cuda().train()
    x = torch.from_numpy(C.synth.images(1234, 2, 3, 64, 64)).cuda()
    y = torch.from_numpy(C.synth.labels(1234, 2, 64, 64, 2)).cuda()
    opt = C.FusedAdam(model.parameters(), lr=float(g['lr']), betas=[0.5, 0.99])
    crit = C.CrossEntropyLoss()
    losses = []
    for s in range(3):
        out = model(x); opt.zero_grad(); loss = crit(out, y); loss.backward()
        if s == 0:
            lg = out.detach().cpu().numpy()
            # bf16: measured 2.27e-2 rel L2, 2.84e-2 max rel (round 4): bounds <= 2x
            tol, tolm = (1e-3, 1e-3) if fp32 else (4.5e-2, 5.5e-2)
            print(f'measured[config1 {dtype}]: logits rel L2 {rel_l2(lg, g["logits"]):.3e}, max rel {maxrel(lg, g["logits"]):.3e}')
            assert rel_l2(lg, g['logits']) < tol and maxrel(lg, g['logits']) < tolm, (rel_l2(lg, g['logits']), maxrel(lg, g['logits']))
            gn = np.array([float(p.grad.double().norm()) for p in model.parameters()])
            big = g['grad_norms'] > 1e-4 * g['grad_norms'].max()
            np.testing.assert_allclose(gn[big], g['grad_norms'][big], rtol={'fp32': 1e-2, 'bf16x3': 3e-2, 'bf16': 0.4}[dtype])
            m = C.eval_metrics(y, out.detach(), 2)
            np.testing.assert_allclose([float(v) for v in m], g['metrics'], rtol=1e-5 if fp32 else 0.2, atol=0 if fp32 else 0.1)
            if fp32:
                assert np.array_equal(np.bincount(out.detach().argmax(1).cpu().numpy().reshape(-1), minlength=2), g['pred_hist'])
        opt.step()
        losses.append(float(loss.detach()))
    # the third loss sits behind two of Adam's sign-like first updates (|update| = lr whatever |g|): rounding-level gradient differences
    # are amplified there -- bf16x3 measures 3.7e-4 with and 5.3e-4 without the normalised tensors of enc1 / enc2 / dec4 / last (the
    # algebraic BatchNorm fold, tools/fold_parity.py; fp32: 2.6e-4 and 2.3e-5), the first two losses 2e-7 and 4e-5
    np.testing.assert_allclose(losses, g['losses'], rtol={'fp32': 5e-4, 'bf16x3': 2e-3, 'bf16': 5e-2}[dtype])
    np.testing.assert_allclose(losses[:2], g['losses'][:2], rtol={'fp32': 5e-5, 'bf16x3': 2e-4, 'bf16': 5e-2}[dtype])


@pytest.mark.parametrize('dtype', ['fp32', 'bf16', 'bf16x3'])
def test_full_size_config2_vs_reference_golden(C, golden, dtype):
    """BASELINE.json configs[1]/[2] shape: UNet(21,3,64), 256x256, bs16 -- logits subsample, loss, per-tensor gradient
    norms, arg-max histogram and mIoU captured from the reference's CPU path."""
    g = golden('unet_cd64_c21_256.npz')
    model = C.UNet(21, 3, 64, compute_dtype=dtype)
    load_closed_form(C, model)
    model = model.cuda().train()
    x = torch.from_numpy(C.synth.images(1234, 16, 3, 256, 256)).cuda()
    y = torch.from_numpy(C.synth.labels(1234, 16, 256, 256, 21)).cuda()
    opt = C.FusedAdam(model.parameters(), lr=float(g['lr']), betas=[0.5, 0.99])
    crit = C.CrossEntropyLoss()
    fp32 = dtype != 'bf16'          # 'bf16x3' (split-bf16, fp32 storage) is held to the fp32 path's bounds
    losses = []
    for s in range(2):
        out = model(x)
        opt.zero_grad()
        loss = crit(out, y)
        loss.backward()
        if s == 0:
            sub = out.detach().reshape(-1)[::int(g['logits_flat_stride'])].cpu().numpy()
            # bf16: storage rounding of 23 stacked convolutions; measured 1.5e-2 (the reference itself with bf16-rounded operands
            # gives the same) -- bound at < 2x the measurement so that a regression shows
            tol = 1e-3 if fp32 else 2.5e-2
            assert rel_l2(sub, g['logits']) < tol, f'logits rel L2 {rel_l2(sub, g["logits"]):.3e} (bound {tol})'
            assert maxrel(sub, g['logits']) < tol * (1 if fp32 else 2.5), f'logits max-abs ratio {maxrel(sub, g["logits"]):.3e}'
            gn = np.array([float(p.grad.double().norm()) for p in model.parameters()])
            big = g['grad_norms'] > 1e-4 * g['grad_norms'].max()
            np.testing.assert_allclose(gn[big], g['grad_norms'][big], rtol={'fp32': 5e-3, 'bf16x3': 2e-2, 'bf16': 0.3}[dtype])
            m = C.eval_metrics(y, out.detach(), 21)
            # mIoU "identical" (fp32) / within +-0.1 (bf16)
            assert abs(float(m[2]) - float(g['metrics'][2])) < (1e-5 if fp32 else 0.1)
            hist = np.bincount(out.detach().argmax(1).cpu().numpy().reshape(-1), minlength=21)
            assert np.abs(hist - g['pred_hist']).sum() <= (2e-4 if fp32 else 0.1) * hist.sum()
        opt.step()
        losses.append(float(loss))
    np.testing.assert_allclose(losses, g['losses'], rtol=2e-4 if fp32 else 3e-2)


@pytest.mark.parametrize('dtype', ['fp32', 'bf16', 'bf16x3'])
def test_full_size_config5_vs_reference_golden(C, golden, dtype):
    """BASELINE.json configs[4] (512x512, bs32 per GPU, stated in bf16) against captures from the reference's CPU path:
    (a) the full 2-step train-step capture at 512x512 with bs8 (the reference's bs32 step needs ~60 GB of host memory the
    build container does not have): logits subsample, losses, per-tensor gradient norms, arg-max histogram, mIoU;
    (b) the train-mode forward + loss at the real bs32: logits subsample, loss, histogram, mIoU, BatchNorm running stats."""
    fp32 = dtype != 'bf16'          # 'bf16x3' is held to the fp32 path's bounds
    g = golden('unet_cd64_c21_512_b8.npz')
    model = C.UNet(21, 3, 64, compute_dtype=dtype)
    load_closed_form(C, model)
    model = model.cuda().train()
    B, size = int(g['batch']), int(g['size'])
    assert (B, size) == (8, 512)
    x = torch.from_numpy(C.synth.images(1234, B, 3, size, size)).cuda()
    y = torch.from_numpy(C.synth.labels(1234, B, size, size, 21)).cuda()
    opt = C.FusedAdam(model.parameters(), lr=float(g['lr']), betas=[0.5, 0.99])
    crit = C.CrossEntropyLoss()
    losses = []
    for s in range(2):
        out = model(x)
        opt.zero_grad()
        loss = crit(out, y)
        loss.backward()
        if s == 0:
            sub = out.detach().reshape(-1)[::int(g['logits_flat_stride'])].cpu().numpy()
            # bf16: storage rounding of 23 stacked convolutions; measured 1.5e-2 (the reference itself with bf16-rounded operands
            # gives the same) -- bound at < 2x the measurement so that a regression shows
            tol = 1e-3 if fp32 else 2.5e-2
            assert rel_l2(sub, g['logits']) < tol, f'logits rel L2 {rel_l2(sub, g["logits"]):.3e} (bound {tol})'
            assert maxrel(sub, g['logits']) < tol * (1 if fp32 else 2.5), f'logits max-abs ratio {maxrel(sub, g["logits"]):.3e}'
            gn = np.array([float(p.grad.double().norm()) for p in model.parameters()])
            big = g['grad_norms'] > 1e-4 * g['grad_norms'].max()
            np.testing.assert_allclose(gn[big], g['grad_norms'][big], rtol={'fp32': 5e-3, 'bf16x3': 2e-2, 'bf16': 0.3}[dtype])
            m = C.eval_metrics(y, out.detach(), 21)
            assert abs(float(m[2]) - float(g['metrics'][2])) < (1e-5 if fp32 else 0.1)
            hist = np.bincount(out.detach().argmax(1).cpu().numpy().reshape(-1), minlength=21)
            assert np.abs(hist - g['pred_hist']).sum() <= (2e-4 if fp32 else 0.1) * hist.sum()
        opt.step()
        losses.append(float(loss.detach()))
    np.testing.assert_allclose(losses, g['losses'], rtol=2e-4 if fp32 else 3e-2)
    del model, opt, out, loss
    torch.cuda.empty_cache()
    # ---- (b) the real batch: bs32 ----
    f = golden('unet_cd64_c21_512_b32_fwd.npz')
    assert (int(f['batch']), int(f['size'])) == (32, 512)
    model = C.UNet(21, 3, 64, compute_dtype=dtype)
    load_closed_form(C, model)
    model = model.cuda().train()
    x = torch.from_numpy(C.synth.images(1234, 32, 3, 512, 512)).cuda()
    y = torch.from_numpy(C.synth.labels(1234, 32, 512, 512, 21)).cuda()
    with torch.no_grad():
        out = model(x)
        loss = crit(out, y)
    sub = out.reshape(-1)[::int(f['logits_flat_stride'])].cpu().numpy()
    tol = 1e-3 if fp32 else 2.5e-2
    assert rel_l2(sub, f['logits']) < tol, f'logits rel L2 {rel_l2(sub, f["logits"]):.3e} (bound {tol})'
    assert float(loss) == pytest.approx(float(f['loss']), rel=2e-4 if fp32 else 3e-2)
    m = C.eval_metrics(y, out, 21)
    assert abs(float(m[2]) - float(f['metrics'][2])) < (1e-5 if fp32 else 0.1)
    hist = np.bincount(out.argmax(1).cpu().numpy().reshape(-1), minlength=21)
    assert np.abs(hist - f['pred_hist']).sum() <= (2e-4 if fp32 else 0.1) * hist.sum()
    stats = np.concatenate([v.cpu().numpy().reshape(-1) for k, v in model.state_dict().items() if k.endswith(('running_mean', 'running_var'))])
    assert rel_l2(stats, f['stats1']) < (1e-4 if dtype == 'fp32' else 1e-3 if fp32 else 2e-2)


@pytest.mark.parametrize('dtype', ['bf16', 'bf16x3'])
def test_full_size_config5_bs32_full_train_step(C, golden, dtype):
    """BASELINE.json configs[4] at its REAL size -- 512x512, bs32 per GPU -- the whole train step (forward, CE, backward,
    Adam), not only the forward: the step-0 loss equals the reference's capture (the reference's own bs32 backward needs more
    host memory than the build container has, so loss + logits pin the forward and the properties below pin the rest), two
    runs of two steps are bit-identical (every reduction at this size is a fixed-order sum), every parameter tensor moved,
    and the second loss is lower than the first."""
    f = golden('unet_cd64_c21_512_b32_fwd.npz')
    x = torch.from_numpy(C.synth.images(1234, 32, 3, 512, 512)).cuda()
    y = torch.from_numpy(C.synth.labels(1234, 32, 512, 512, 21)).cuda()
    crit = C.CrossEntropyLoss()

    def two_steps():
        model = C.UNet(21, 3, 64, compute_dtype=dtype)
        load_closed_form(C, model)
        model = model.cuda().train()
        w0 = [p.detach().clone() for p in model.parameters()]
        opt = C.FusedAdam(model.parameters(), lr=1e-4, betas=[0.5, 0.99])
        losses = []
        for _ in range(2):
            out = model(x); opt.zero_grad(); loss = crit(out, y); loss.backward(); opt.step()
            losses.append(loss.detach().clone())
        torch.cuda.synchronize()
        flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).clone()
        moved = [bool((p.detach() != w).any()) for p, w in zip(model.parameters(), w0)]
        gn = float(torch.cat([p.grad.reshape(-1) for p in model.parameters()]).double().norm())
        del model, opt, out, loss
        torch.cuda.empty_cache()
        return losses, flat, moved, gn

    l1, f1, moved, gn = two_steps()
    tol = 3e-2 if dtype == 'bf16' else 2e-4
    assert float(l1[0]) == pytest.approx(float(f['loss']), rel=tol), f'step-0 loss {float(l1[0]):.6f} vs reference {float(f["loss"]):.6f}'
    assert all(moved), 'Adam must move every parameter tensor'
    assert float(l1[1]) < float(l1[0]) and gn == gn and gn > 0
    l2, f2, _, _ = two_steps()
    assert torch.equal(l1[0], l2[0]) and torch.equal(l1[1], l2[1]), 'losses of two identical runs differ'
    assert torch.equal(f1, f2), f'{int((f1 != f2).sum())} weights differ between two identical runs'


@pytest.mark.parametrize('dtype', ['fp32', 'bf16', 'bf16x3'])
def test_miou_after_training_on_fixed_64_image_set(C, golden, dtype):
    """north_star: 'mIoU within +-0.1 of reference on a fixed synthetic 21-class set' (64 images).  The
    reference itself (models/unet.py imported by oracle/gen_golden.py) was trained for 4 epochs of 4 x bs16 steps at
    256x256 on the fixed set (images carry their labels) and evaluated on it; this path repeats the run: per-step loss curve, the training-time mIoU of
    the last epoch (trainer.py:183-188) and the eval-mode mIoU (trainer.py:270-284) over the accumulated confusion matrix."""
    g = golden('train_cd64_c21_256_n64.npz')
    nc, B, size, nimg, epochs = int(g['num_classes']), int(g['batch']), int(g['size']), int(g['nimg']), int(g['epochs'])
    fp32 = dtype != 'bf16'
    model = C.UNet(nc, 3, int(g['conv_dim']), compute_dtype=dtype)
    load_closed_form(C, model)
    model = model.cuda().train()
    opt = C.FusedAdam(model.parameters(), lr=float(g['lr']), betas=[0.5, 0.99])
    crit = C.CrossEntropyLoss()
    data = []
    for i in range(nimg // B):      # images that carry their labels (synth.images_with_signal), as in the capture
        lab = C.synth.labels(1234, B, size, size, nc, first_image=i * B)
        data.append((torch.from_numpy(C.synth.images_with_signal(1234, lab, nc, first_image=i * B)).cuda(), torch.from_numpy(lab).cuda()))
    losses, train_conf = [], None
    for ep in range(epochs):
        for x, y in data:
            out = model(x); opt.zero_grad(); loss = crit(out, y); loss.backward(); opt.step()
            losses.append(float(loss.detach()))
            if ep == epochs - 1:
                c, _ = C.metrics.argmax_confusion(out.detach(), y, nc)
                train_conf = c if train_conf is None else train_conf + c
    model.eval()
    eval_conf = None
    with torch.no_grad():
        for x, y in data:
            c, _ = C.metrics.argmax_confusion(model(x), y, nc)
            eval_conf = c if eval_conf is None else eval_conf + c
    train_miou = float(C.metrics.metrics_from_confusion(train_conf)[2])
    eval_miou = float(C.metrics.metrics_from_confusion(eval_conf)[2])
    print(f'{dtype}: loss {losses[0]:.4f} -> {losses[-1]:.4f} (reference {g["losses"][0]:.4f} -> {g["losses"][-1]:.4f}); '
          f'train mIoU {train_miou:.4f} (reference {float(g["train_miou"]):.4f}), eval mIoU {eval_miou:.4f} (reference {float(g["eval_miou"]):.4f})')
    assert g['losses'][-1] < 0.9 * g['losses'][0]                       # the reference does learn on this set
    assert losses[:4] == pytest.approx(list(g['losses'][:4]), rel=1e-3 if fp32 else 3e-2)
    assert losses == pytest.approx(list(g['losses']), rel=2e-2 if fp32 else 8e-2)
    assert abs(train_miou - float(g['train_miou'])) < (0.02 if fp32 else 0.1)
    # the eval-mode number is a poorly conditioned one: 16 momentum-0.1 updates into the running statistics, mIoU 0.09 in the reference itself
    # (bf16x3 measures 0.110 with the algebraic BatchNorm fold; the north_star bound is 0.1 either side)
    assert abs(eval_miou - float(g['eval_miou'])) < {'fp32': 0.02, 'bf16x3': 0.05, 'bf16': 0.1}[dtype]
    assert int(eval_conf.sum()) == nimg * size * size


def test_eval_mode_and_state_dict_roundtrip(C):
    """.eval() uses running statistics (trainer.py:271); state_dict loads into the stock-torch counterpart and back."""
    torch.manual_seed(0)
    model = C.UNet(5, 3, 8).cuda()
    ref = TC.build_unet(5, 3, 8)
    ref.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()}, strict=True)
    x = torch.randn(2, 3, 32, 48)
    model.train(); ref.train()
    for _ in range(2):                       # move the running stats away from (0, 1)
        with torch.no_grad():
            model(x.cuda()); ref(x)
    model.eval(); ref.eval()
    with torch.no_grad():
        a, b = model(x.cuda()).cpu().numpy(), ref(x).numpy()
    assert rel_l2(a, b) < 1e-4
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    for k, v in ref.state_dict().items():
        assert rel_l2(sd[k].double().numpy(), v.double().numpy()) < 1e-4 or k.endswith('num_batches_tracked'), k
    assert int(sd['last.5.num_batches_tracked']) == 2


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_predict_fuses_argmax_into_the_head(C, dtype):
    """UNet.predict == torch.max(model(x), 1)[1] (trainer.py:279) with the arg-max in the head kernel's
    epilogue (no logits in HBM), in eval and in train mode, odd class counts, ties to the lower class; Trainer.test() =
    the reference's pixel accuracy (trainer.py:270-284)."""
    for nc, cd, B, H, W in ((21, 8, 2, 64, 96), (2, 4, 3, 32, 32), (40, 8, 1, 48, 64)):
        torch.manual_seed(nc)
        m = C.UNet(nc, 3, cd, compute_dtype=dtype).cuda()
        x = torch.from_numpy(C.synth.images(5, B, 3, H, W)).cuda()
        for mode in (True, False):
            m.train(mode)
            with torch.no_grad():
                want = m(x).argmax(1)
            if mode:                     # a train-mode forward moves the running statistics; predict() must see the same ones
                bufs = {k: v.clone() for k, v in m.state_dict().items()}
            got = m.predict(x)
            assert got.dtype == torch.int64 and tuple(got.shape) == (B, H, W)
            if mode:
                assert torch.equal(got, want)      # batch statistics of the same batch
                m.load_state_dict(bufs)
            else:
                assert torch.equal(got, want)
    # ties: a head with all-zero weights and equal biases -> every class equal -> class 0
    with torch.no_grad():
        m.last[6].weight.zero_(); m.last[6].bias.fill_(0.25)
    assert int(m.eval().predict(x).abs().max()) == 0
    cfg = C.default_config(n_iters=2, lr=1e-3, num_classes=5, conv_dim=4, compute_dtype=dtype, stats_every=1)
    data = [(torch.from_numpy(C.synth.images(3, 2, 3, 32, 32, first_image=2 * i)), torch.from_numpy(C.synth.labels(3, 2, 32, 32, 5, first_image=2 * i)))
            for i in range(2)]
    tr = C.Trainer(data, cfg)
    tr.train_val(epochs=1)
    acc = tr.test(data)
    tr.model.eval()
    with torch.no_grad():
        ref = 100.0 * sum(int((tr.model(a.cuda()).argmax(1) == b.cuda()).sum()) for a, b in data) / sum(b.numel() for _, b in data)
    assert acc == pytest.approx(ref, abs=1e-9) and tr.model.training is False


@pytest.mark.parametrize('dtype', ['fp32', 'bf16x3', 'bf16'])
def test_blocks_run_on_their_own(C, dtype):
    """The reference's blocks (models/unet.py:8-38, :50-55, :66-72) as stand-alone custom ops (blocks.py): every child of UNet called on
    its own -- enc1 (plain), enc2 (DownBlock: pool first), dec1 (UpBlock: ConvTranspose last), last (1x1 head last), and the inner
    ``.block`` sequence -- against the stock torch modules with the same parameters: output, input gradient, every parameter gradient,
    running statistics; train and eval mode.  One autograd Function over libclamd kernels per block, no torch operator."""
    import torch.nn as nn
    from oracle import torch_cpu as TC
    dev = torch.device('cuda', 0)
    torch.manual_seed(11)
    m = C.UNet(5, 3, 8, compute_dtype=dtype).to(dev).train()
    ref = TC.build_unet(5, 3, 8).to(dev).train()
    ref.load_state_dict(m.state_dict())
    tol = {'fp32': 2e-5, 'bf16x3': 2e-4, 'bf16': 3e-2}[dtype]
    # gradients: a random 8 / 16-channel block sits on its ReLU ties -- a rounding-level change of a pre-activation flips one and moves the
    # gradient by a visible amount; fp32 is held per tensor, the rounded dtypes on the block's whole gradient
    gtol = {'fp32': 2e-3, 'bf16x3': 3e-2, 'bf16': 0.3}[dtype]
    cases = [('enc1', (2, 3, 32, 48)), ('enc2', (2, 8, 32, 48)), ('dec1', (2, 64, 4, 6)), ('dec4', (2, 32, 16, 24)), ('last', (2, 16, 32, 48))]
    for name, shape in cases:
        ours, theirs = getattr(m, name), getattr(ref, name)
        x = torch.randn(*shape, device=dev)
        xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        oa, ob = ours(xa), theirs(xb)
        assert oa.shape == ob.shape and oa.dtype == torch.float32
        assert rel_l2(oa.detach().cpu().numpy(), ob.detach().cpu().numpy()) < tol, name
        g = torch.randn_like(ob)
        oa.backward(g); ob.backward(g)
        assert rel_l2(xa.grad.cpu().numpy(), xb.grad.cpu().numpy()) < gtol, name
        pa, pb = dict(ours.named_parameters()), dict(theirs.named_parameters())
        assert list(pa) == list(pb)
        if dtype == 'fp32':
            for k in pa:
                r = pb[k].grad
                assert float((pa[k].grad - r).norm()) <= gtol * float(r.norm()) + 1e-6 * r.numel() ** 0.5, (name, k)
        ga, gb = (torch.cat([d[k].grad.reshape(-1) for k in pa]) for d in (pa, pb))
        assert float((ga - gb).norm() / gb.norm()) < gtol, name
        ba, bb = dict(ours.named_buffers()), dict(theirs.named_buffers())
        for k in ba:
            assert torch.allclose(ba[k].float(), bb[k].float(), rtol=1e-4 if dtype != 'bf16' else 2e-2, atol=1e-5 if dtype != 'bf16' else 1e-3), (name, k)
        m.zero_grad(); ref.zero_grad()
    # the inner sequence is the same op; eval mode normalises with the running statistics
    x = torch.randn(2, 8, 32, 48, device=dev)
    assert m.enc2.block(x).shape == (2, 16, 16, 24)
    m.eval(); ref.eval()
    ref.load_state_dict(m.state_dict())
    with torch.no_grad():
        assert rel_l2(m.enc2(x).cpu().numpy(), ref.enc2(x).cpu().numpy()) < tol
        assert rel_l2(m.enc2.block(x).cpu().numpy(), ref.enc2(x).cpu().numpy()) < tol
    assert isinstance(m.enc2.block[1], nn.Conv2d)


@pytest.mark.parametrize('dtype', ['fp32', 'bf16x3', 'bf16'])
def test_loss_hands_d_logits_to_the_backward_pass(C, dtype):
    """loss.CrossEntropyLoss writes d logits a second time in the head data gradient's layout when the logits come from this package's UNet
    (unet.dlogits_sink); the engine takes that copy only when autograd hands back the very tensor the loss wrote.  (a) plain backward: the
    conversion kernel is skipped and the gradients equal those of a run through the conversion (a stock-torch loss on the same logits);
    (b) a scaled loss: exactly twice the gradients; (c) a second consumer of the logits: autograd sums two gradients, the engine converts."""
    import torch.nn.functional as F
    dev = torch.device('cuda', 0)
    x = torch.from_numpy(C.synth.images(8, 2, 3, 32, 32)).to(dev)
    y = torch.from_numpy(C.synth.labels(8, 2, 32, 32, 5)).to(dev)
    crit = C.CrossEntropyLoss()

    def run(make_loss):
        torch.manual_seed(9)
        m = C.UNet(5, 3, 8, compute_dtype=dtype).to(dev).train()
        out = m(x)
        eng = next(iter(m._engines.values()))
        loss = make_loss(out)
        taken = eng.dl_src is not None
        loss.backward()
        torch.cuda.synchronize()
        return torch.cat([p.grad.flatten() for p in m.parameters()]).clone(), taken, float(loss)

    ga, taken_a, la = run(lambda out: crit(out, y))
    gt, taken_t, lt = run(lambda out: F.cross_entropy(out, y))                 # torch's loss: gradient arrives as a plain NCHW tensor -> converted
    assert taken_a and not taken_t and abs(la - lt) < 1e-5 * abs(lt)
    assert float((ga - gt).norm() / gt.norm()) < {'fp32': 1e-5, 'bf16x3': 1e-4, 'bf16': 2e-2}[dtype]      # torch's softmax differs in the last bits; measured 1.4e-5 in bf16x3
    gb, taken_b, _ = run(lambda out: 2.0 * crit(out, y))
    # a power of two: exact -- except that the bf16x3 copy is re-split after the multiplication (hi + lo rounded to fp32 once more)
    assert taken_b and (torch.equal(gb, 2.0 * ga) if dtype != 'bf16x3' else float((gb - 2.0 * ga).norm() / ga.norm()) < 1e-3)
    gc, _, _ = run(lambda out: crit(out, y) + 0.0 * out.sum())                 # two gradients summed by autograd: a new tensor -> converted
    assert float((gc - ga).norm() / ga.norm()) < (1e-6 if dtype != 'bf16' else 1e-2)
    # (d) the package loss evaluated for logging and DROPPED, then another loss backpropagated through the same logits: its gradient tensor
    # has the shape and dtype of the one the dropped loss wrote and would get that tensor's address back from the caching allocator if the
    # engine did not hold it -- the stale NHWC copy (gradient w.r.t. y) must not pass for the gradient w.r.t. y2
    y2 = torch.from_numpy(C.synth.labels(77, 2, 32, 32, 5)).to(dev)

    def dropped_then_other(out):
        logged = crit(out, y)
        del logged
        return F.cross_entropy(out, y2)

    gd, _, _ = run(dropped_then_other)
    g2, _, _ = run(lambda out: F.cross_entropy(out, y2))
    assert float((gd - g2).norm() / g2.norm()) < (1e-6 if dtype != 'bf16' else 1e-2)
    assert float((gd - ga).norm() / ga.norm()) > 1e-2                          # and y2 is a different target
    # a misaligned view of the logits (odd storage offset) takes the scalar loss kernel instead of failing
    torch.manual_seed(9)
    m = C.UNet(5, 3, 8, compute_dtype=dtype).to(dev).train()
    base = torch.zeros(2 * 5 * 32 * 32 + 1, device=dev)
    lg = base[1:].view(2, 5, 32, 32)
    lg.copy_(m(x).detach())
    lg.requires_grad_(False)
    la_odd = float(crit(lg, y))
    assert abs(la_odd - la) < 1e-5 * abs(la)
    # a later forward invalidates the hand-over of an earlier one
    torch.manual_seed(9)
    m = C.UNet(5, 3, 8, compute_dtype=dtype).to(dev).train()
    out1 = m(x)
    with torch.no_grad():
        m(x)
    from continual_learning_amd import unet as U
    assert U.dlogits_sink(out1, 2, 5, 32, 32) is None


def test_misuse_of_shapes_and_stale_activations(C):
    model = C.UNet(3, 3, 4)
    with pytest.raises(RuntimeError):
        model(torch.zeros(1, 3, 32, 32))                         # CPU tensor: no fallback
    model = model.cuda()
    with pytest.raises(AssertionError):
        model(torch.zeros(1, 3, 40, 32, device='cuda'))          # H % 16 != 0 (models/unet.py:88-91 asserts)
    with pytest.raises(ValueError):
        model(torch.zeros(1, 4, 32, 32, device='cuda'))
    a = model(torch.zeros(1, 3, 32, 32, device='cuda'))
    model(torch.zeros(1, 3, 32, 32, device='cuda'))
    with pytest.raises(RuntimeError):
        a.sum().backward()                                       # activations of the first forward were overwritten


def test_trainer_loop_matches_torch_counterpart(C):
    """Trainer (scheduler.step() first, trainer.py:147) vs the stock-torch counterpart for 2 epochs x 3 batches."""
    cfg = C.default_config(n_iters=5, lr=1e-3, num_classes=4, conv_dim=4, compute_dtype='fp32', stats_every=1)
    data = [(torch.from_numpy(C.synth.images(7, 2, 3, 32, 32, first_image=2 * i)),
             torch.from_numpy(C.synth.labels(7, 2, 32, 32, 4, first_image=2 * i))) for i in range(3)]
    tr = C.Trainer(data, cfg)
    ref = TC.build_unet(4, 3, 4)
    ref.load_state_dict({k: v.cpu() for k, v in tr.model.state_dict().items()})
    ropt = TC.make_optimizer(ref, lr=1e-3)
    rsch = TC.make_scheduler(ropt, 5, 0.9)
    crit = torch.nn.CrossEntropyLoss()
    import warnings
    for ep in range(2):
        stats = tr.train_val(epochs=1)[0]
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            rsch.step()
        ls = []
        for xb, yb in data:
            _, l = TC.train_step(ref, ropt, crit, xb, yb)
            ls.append(float(l))
        assert abs(stats['lr'] - ropt.param_groups[0]['lr']) < 1e-12
        assert abs(stats['loss'] - np.mean(ls)) < 2e-3 * np.mean(ls)
        assert 0.0 <= stats['mean_iu'] <= 1.0


def _confusion_miou(C, model, data, nc):
    conf = None
    with torch.no_grad():
        for x, y in data:
            c, _ = C.metrics.argmax_confusion(model(x), y, nc)
            conf = c if conf is None else conf + c
    return float(C.metrics.metrics_from_confusion(conf)[2])


def _torch_miou(model, data, nc):
    conf = np.zeros((nc, nc))
    with torch.no_grad():
        for x, y in data:
            conf += O.confusion(y.cpu().numpy(), model(x).argmax(1).cpu().numpy(), nc)
    return float(O.mean_iu2(conf.astype(np.float32)))


@pytest.mark.parametrize('nc,cd,B,size,n1,n2,dtype', [(21, 8, 4, 64, 6, 6, 'fp32'), (21, 64, 16, 256, 3, 3, 'fp32'),
                                                      (21, 64, 16, 256, 3, 3, 'bf16')])
def test_continual_two_task_split(C, nc, cd, B, size, n1, n2, dtype):
    """BASELINE.json configs[3] as SURVEY.md §8d specifies it: task 1 on classes 0-10 (other pixels -> 0), snapshot, task 2 on
    classes 11-20 with the distillation term towards the frozen task-1 model AND the L2-to-old-weights term, mIoU reported
    for both tasks.  Build-defined procedure (the reference has no continual-learning code: parity unpinned), checked
    against the same procedure composed from stock torch ops (oracle/torch_cpu.continual_two_task) on the same device."""
    dev = torch.device('cuda', 0)
    lam_d, T, lam_2, lr, c_old = 0.5, 2.0, 0.01, 1e-3, 11
    mk = lambda lo, hi, n: [(torch.from_numpy(C.synth.images(9, B, 3, size, size, first_image=i * B)).to(dev),
                             torch.from_numpy(C.synth.labels(9, B, size, size, nc, first_image=i * B, class_lo=lo, class_hi=hi)).to(dev))
                            for i in range(n)]
    task1, task2 = mk(0, 11, n1), mk(11, 21, n2)
    assert all(int(y.max()) < 11 for _, y in task1) and all(set(np.unique(y.cpu().numpy())) <= {0, *range(11, 21)} for _, y in task2)
    torch.manual_seed(5)
    ref = TC.build_unet(nc, 3, cd).to(dev)
    cfg = C.default_config(n_iters=100, lr=lr, num_classes=nc, conv_dim=cd, compute_dtype=dtype, stats_every=1)
    tr = C.Trainer(task1, cfg)
    tr.model.load_state_dict(ref.state_dict())
    R = TC.continual_two_task(ref, task1, task2, c_old, lam_d, T, lam_2, lr)
    fp32 = dtype == 'fp32'
    tol_l = 2e-3 if fp32 else 5e-2
    # ---- task 1: the plain hot loop ----
    losses1 = [float(tr.train_step(x, y)[1].detach()) for x, y in task1]
    assert losses1 == pytest.approx(R['losses1'], rel=tol_l)
    # ---- snapshot: a frozen eval-mode copy with its own buffers ----
    tr.begin_task2(c_old=c_old, distill_lambda=lam_d, temperature=T, l2_lambda=lam_2)
    assert not tr.old_model.training and tr.old_model is not tr.model and not tr.old_model._engines
    assert all(not p.requires_grad for p in tr.old_model.parameters())
    for k, v in tr.old_model.state_dict().items():
        assert torch.equal(v, tr.model.state_dict()[k]) and v.data_ptr() != tr.model.state_dict()[k].data_ptr(), k
    old_flat = torch.cat([p.detach().reshape(-1) for p in tr.old_model.parameters()]).double()
    # ---- task 2: CE + distillation (inside the loss kernel) + L2-to-old-weights (inside the Adam kernel) ----
    names = [n for n, _ in tr.model.named_parameters()]
    for i, (x, y) in enumerate(task2):
        before = [p.detach().clone() for p in tr.model.parameters()]
        st0 = [(tr.optim.state[p]['exp_avg'].clone(), tr.optim.state[p]['exp_avg_sq'].clone()) for p in tr.model.parameters()]
        step0 = float(tr.optim.state[next(iter(tr.model.parameters()))]['step'])
        out, loss = tr.train_step(x, y)
        tot, ce, kd, l2 = R['losses2'][i]
        assert float(loss.detach()) == pytest.approx(tot, rel=tol_l), (i, float(loss.detach()), tot)
        # the penalty the Adam kernel accumulated = lam * sum ||theta - theta_old||^2 over the weights BEFORE this update
        pen = float(tr.optim.l2_penalty())
        want = lam_2 * float(((torch.cat([b.reshape(-1) for b in before]).double() - old_flat) ** 2).sum())
        assert pen == pytest.approx(want, rel=1e-4, abs=1e-12), (i, pen, want)
        # ... and stays near the torch composition's own trajectory (Adam's sign-like steps: different rounding, other weights)
        assert pen == pytest.approx(l2, rel=(0.2 if fp32 else 0.3), abs=1e-9)
        if i == n2 - 1:
            # the L2-augmented Adam update, element by element: p' = adam(p, g + 2 lam (p - p_old)) with the kernel's own raw g
            for n_, p, b, (m0, v0), o in zip(names, tr.model.parameters(), before, st0, tr.old_model.parameters()):
                g = p.grad.cpu().numpy() + 2 * lam_2 * (b.cpu().numpy() - o.detach().cpu().numpy())
                want_p, _, _ = O.adam_step(b.cpu().numpy(), g, m0.cpu().numpy(), v0.cpu().numpy(), int(step0) + 1, lr)
                upd, upd_ref = (p.detach() - b).cpu().numpy(), want_p - b.cpu().numpy()
                assert rel_l2(upd, upd_ref) < 2e-3, n_
    # ---- mIoU on both tasks after the sequence, against the torch composition ----
    tr.model.eval(); ref.eval()
    for name, data in (('task1', task1), ('task2', task2)):
        a, b = _confusion_miou(C, tr.model, data, nc), _torch_miou(ref, data, nc)
        print(f'continual {dtype} cd{cd} {size}x{size}: {name} mIoU ours {a:.4f} torch {b:.4f}')
        assert abs(a - b) < (0.02 if fp32 else 0.1), (name, a, b)


def test_adjoint_identities_full_size(C):
    """Size-independent property at BASELINE size (64ch, 256x256, bs16 is ~1 s of GPU): for a linear conv,
    <conv(x,w), g> == <x, dgrad(g,w)> == <w, wgrad(x,g)> (fp32 path, fp64 inner products)."""
    lib, ptr = C._lib, C._lib.ptr
    B, Cc, H, W = 16, 64, 256, 256
    gen = torch.Generator(device='cuda').manual_seed(0)
    x = torch.randn(B, H, W, Cc, device='cuda', generator=gen)
    g = torch.randn(B, H, W, Cc, device='cuda', generator=gen)
    w = torch.randn(Cc, Cc, 3, 3, device='cuda', generator=gen) / 24.0
    wf = torch.zeros(9 * Cc * Cc, device='cuda'); wd = torch.zeros(9 * Cc * Cc, device='cuda')
    tab = C.ops.PackTable(0); tab.conv3x3(w, wf, wd, [(Cc, Cc)], Cc); tab.finalize('cuda').run(0)
    s = lib.stream_ptr()
    y = torch.empty_like(x); gx = torch.empty_like(x); gw = torch.empty_like(w)
    lib.call('clamd_conv3x3', ptr(x), Cc, ptr(wf), None, ptr(y), Cc, None, None, None, 0, B, H, W, Cc, Cc, 0, 0, 0, None, s)
    lib.call('clamd_conv3x3', ptr(g), Cc, ptr(wd), None, ptr(gx), Cc, None, None, None, 0, B, H, W, Cc, Cc, 0, 0, 0, None, s)
    wsb = lib.load().clamd_wgrad_workspace_bytes(0, B, H, W, Cc, Cc, 0)
    ws = torch.empty(wsb // 4 + 4, device='cuda')
    lib.call('clamd_wgrad', 0, ptr(g), Cc, ptr(x), Cc, ptr(ws), wsb, ptr(gw), B, H, W, Cc, Cc, Cc, Cc, Cc, Cc, Cc, Cc, 0, None, s)
    torch.cuda.synchronize()
    a = float((y.double() * g.double()).sum()); b = float((x.double() * gx.double()).sum()); c = float((w.double() * gw.double()).sum())
    assert abs(a - b) < 1e-5 * abs(a) and abs(a - c) < 1e-5 * abs(a), (a, b, c)


def _one_step(C, dtype, nc, cd, B, size, seed=0, hook=None, steps=1):
    """`steps` train steps from seeded weights on seeded data; returns (loss of the last step, flat gradient of the last
    step, flat weights after it) as GPU tensors."""
    x = torch.from_numpy(C.synth.images(3, B, 3, size, size)).cuda()
    y = torch.from_numpy(C.synth.labels(3, B, size, size, nc)).cuda()
    torch.manual_seed(seed)
    m = C.UNet(nc, 3, cd, compute_dtype=dtype).cuda().train()
    opt = C.FusedAdam(m.parameters(), lr=1e-3, betas=[0.5, 0.99])
    if hook is not None:
        hook(m, opt)
    crit = C.CrossEntropyLoss()
    for _ in range(steps):
        out = m(x); opt.zero_grad(); loss = crit(out, y); loss.backward()
        if m.grad_sync is not None:
            m.grad_sync.wait()
        grads = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()
        opt.step()
    torch.cuda.synchronize()
    return loss.detach().clone(), grads, torch.cat([p.detach().reshape(-1) for p in m.parameters()]).clone(), m


@pytest.mark.parametrize('dtype,nc,cd,B,size,runs', [('fp32', 6, 8, 2, 64, 20), ('bf16', 6, 8, 2, 64, 20), ('bf16x3', 6, 8, 2, 64, 20),
                                                     ('fp32', 21, 64, 4, 128, 4), ('bf16', 21, 64, 8, 256, 4)])
def test_train_step_is_bit_reproducible(C, dtype, nc, cd, B, size, runs):
    """No float atomics anywhere on the path: BatchNorm statistics, the five BatchNorm-backward sums, bias gradients and
    split-K weight gradients are fixed-order sums of per-workgroup / per-tile partial rows.  The same two train steps from
    the same weights on the same batch give bit-identical loss, gradients and updated weights, run after run."""
    ref = _one_step(C, dtype, nc, cd, B, size, steps=2)
    for r in range(1, runs):
        got = _one_step(C, dtype, nc, cd, B, size, steps=2)
        assert torch.equal(got[0], ref[0]), f'run {r}: loss {float(got[0])!r} vs {float(ref[0])!r}'
        assert torch.equal(got[1], ref[1]), f'run {r}: {int((got[1] != ref[1]).sum())} gradient elements differ'
        assert torch.equal(got[2], ref[2]), f'run {r}: weights differ'


@pytest.mark.parametrize('dtype', ['fp32', 'bf16', 'bf16x3'])
def test_scheduling_knobs_do_not_change_results(C, dtype):
    """Scheduling choices -- one workgroup per tile instead of the persistent Winograd grid, CUs left free for RCCL
    (cu_reserve) -- only change how the statistics are split into partial rows: the same sums up to fp32 rounding of the
    rows, every run of one setting bit-identical to itself."""
    ref = _one_step(C, dtype, 6, 16, 4, 128)
    for knob, val in (('wino_persist', 0), ('cu_reserve', 24)):
        a = _one_step(C, dtype, 6, 16, 4, 128, hook=lambda m, o: setattr(m.tuning, knob, val))
        a2 = _one_step(C, dtype, 6, 16, 4, 128, hook=lambda m, o: setattr(m.tuning, knob, val))
        assert torch.equal(a[1], a2[1]) and torch.equal(a[2], a2[2]), knob
        # the statistics agree to 1e-8 between the grids (tools/w24_stats_check.py); a ReLU / max-pool tie that falls the other
        # way moves individual gradient elements, as between any two fp32 implementations: loss tight, gradient loose
        assert abs(float(a[0]) - float(ref[0])) < 1e-5 * abs(float(ref[0])), knob
        assert float((a[1] - ref[1]).norm() / ref[1].norm()) < (2e-2 if dtype == 'bf16' else 5e-3), knob


@pytest.mark.parametrize('dtype', ['fp32', 'bf16x3', 'bf16'])
def test_fused_bn_backward_sums_match_the_separate_reduction(C, dtype, monkeypatch):
    """unet.FUSE_BN_SUMS: the five BatchNorm-backward sums taken in the epilogue of the data-gradient kernel that produces
    the gradient (every direct-kernel dtype, forced on here; 'auto' uses it for the persistent bf16 kernel only) against
    the stand-alone bn_bwd_reduce pass: the same sums over differently shaped partial rows -> the same step up to fp32
    rounding of the rows, and each setting bit-identical to itself."""
    from continual_learning_amd import unet as U
    monkeypatch.setattr(U, 'WINOGRAD', False)               # the Winograd data-gradient kernel has no such epilogue
    monkeypatch.setattr(U, 'FUSE_BN_SUMS', False)
    ref = _one_step(C, dtype, 6, 16, 4, 64)
    assert not any(u.fused_reduce for u in next(iter(ref[3]._engines.values())).convs)
    monkeypatch.setattr(U, 'FUSE_BN_SUMS', True)
    a, a2 = _one_step(C, dtype, 6, 16, 4, 64), _one_step(C, dtype, 6, 16, 4, 64)
    assert sum(u.fused_reduce for u in next(iter(a[3]._engines.values())).convs) >= 13
    assert torch.equal(a[1], a2[1]) and torch.equal(a[2], a2[2])
    assert float((a[1] - ref[1]).norm() / ref[1].norm()) < (1e-4 if dtype != 'bf16' else 2e-2)
    assert abs(float(a[0]) - float(ref[0])) < 1e-5 * abs(float(ref[0]))      # the forward pass is the same launches


def test_misuse_errors(C):
    """There is no silent stock-torch path: (1) a LAYER called on its own raises (the layers only hold parameters; a whole block runs
    through blocks.py, test_blocks_run_on_their_own);
    (2) nn.DataParallel (trainer.py:120-122) on ONE device calls the module itself and works, across devices the replication is
    refused with a pointer to ddp.GradSync; (3) torch's DistributedDataParallel wrapper works (world 1: bit-identical to the
    plain step) -- its reducer hooks see the gradients this path's autograd Function returns."""
    import torch.nn as nn
    dev = torch.device('cuda', 0)
    torch.manual_seed(5)
    m = C.UNet(5, 3, 8).to(dev).train()
    x = torch.from_numpy(C.synth.images(3, 2, 3, 32, 32)).to(dev)
    y = torch.from_numpy(C.synth.labels(3, 2, 32, 32, 5)).to(dev)
    for child, arg in ((m.enc1[0], x), (m.enc1[2], torch.zeros(2, 8, 32, 32, device=dev)), (m.dec1.block[6], torch.zeros(2, 128, 2, 2, device=dev)),
                       (m.enc2.block[0], torch.zeros(2, 8, 32, 32, device=dev)), (m.last[6], torch.zeros(2, 8, 32, 32, device=dev))):
        with pytest.raises(RuntimeError, match='no stock-torch path'):
            child(arg)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m.enc2(torch.zeros(2, 8, 32, 32))
    assert isinstance(m.enc1[0], nn.Conv2d) and isinstance(m.enc1[2], nn.BatchNorm2d) and isinstance(m.dec1.block[6], nn.ConvTranspose2d)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        m(x.cpu())
    # (2) DataParallel
    out_plain = m(x).detach().clone()
    dp = nn.DataParallel(m, device_ids=[0])
    assert torch.equal(dp(x).detach(), out_plain)
    with pytest.raises(RuntimeError, match='GradSync'):
        m._replicate_for_data_parallel()
    # (3) DistributedDataParallel, world 1
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29541')
    created = False
    if not dist.is_initialized():
        dist.init_process_group('gloo', rank=0, world_size=1)
        created = True
    try:
        def run(wrap):
            torch.manual_seed(5)
            mm = C.UNet(5, 3, 8).to(dev).train()
            net = nn.parallel.DistributedDataParallel(mm, device_ids=[0]) if wrap else mm
            opt = C.FusedAdam(mm.parameters(), lr=1e-3, betas=[0.5, 0.99])
            crit = C.CrossEntropyLoss()
            for _ in range(2):
                out = net(x); opt.zero_grad(); loss = crit(out, y); loss.backward(); opt.step()
            torch.cuda.synchronize()
            return loss.detach().clone(), torch.cat([p.detach().reshape(-1) for p in mm.parameters()]).clone()
        (l0, w0), (l1, w1) = run(False), run(True)
        assert torch.equal(l0, l1) and torch.equal(w0, w1), float((w0 - w1).abs().max())
    finally:
        if created:
            dist.destroy_process_group()
    # eval path with more classes than the fused arg-max epilogue holds (64): logits head + arg-max kernel, same answer
    big = C.UNet(70, 3, 4).to(dev).eval()
    xb = torch.from_numpy(C.synth.images(4, 1, 3, 32, 32)).to(dev)
    with torch.no_grad():
        assert torch.equal(big.predict(xb), torch.max(big(xb), 1)[1])


@pytest.mark.parametrize('dtype,cd,size', [('fp32', 16, 64), ('bf16x3', 16, 64), ('bf16', 16, 64), ('fp32', 64, 128)])
def test_batchnorm_folded_into_the_next_convolutions_filters(C, dtype, cd, size, monkeypatch):
    """unet.FOLD_BN_INTO_FILTERS: the BatchNorm between the two convolutions of a block folded algebraically into the second one (bnfold.hip;
    default on the fp32-storage paths, forced here for bf16 too): filters packed with the producer's scale, shift as a border-class bias
    table, weight gradient fixed up from the gradient's border sums -- the normalised tensor is never written.  Same mathematics: ONE train
    step agrees with the unfolded run to rounding (loss) and to the conditioning floor of this network's gradients (ReLU / max-pool tie
    flips; a second step would compare two chaotic trajectories: tests/diag/fold_two_step.py -- stock torch fp32 is
    10-33 % from stock torch fp64 there, and so are both variants); bit-identical run after run over two steps, on one stream or three;
    eval mode and predict() agree with the unfolded model."""
    from continual_learning_amd import unet as U
    B = 4
    monkeypatch.setattr(U, 'FOLD_BN_INTO_FILTERS', False)
    ref = _one_step(C, dtype, 6, cd, B, size, steps=1)
    assert not any(u.fold_on or u.apply_in_filters for u in next(iter(ref[3]._engines.values())).convs)
    monkeypatch.setattr(U, 'FOLD_BN_INTO_FILTERS', True)
    one = _one_step(C, dtype, 6, cd, B, size, steps=1)
    eng = next(iter(one[3]._engines.values()))
    folded = [u.name for u in eng.convs if u.fold_on]
    assert len(folded) >= 4, folded
    assert all(u.fold_a.apply_in_filters and not u.pre_f for u in eng.convs if u.fold_on)
    lt, gt = {'fp32': (2e-6, 1e-2), 'bf16x3': (2e-5, 4e-2), 'bf16': (5e-3, 0.2)}[dtype]
    assert abs(float(one[0]) - float(ref[0])) < lt * abs(float(ref[0])), (float(one[0]), float(ref[0]))
    assert float((one[1] - ref[1]).norm() / ref[1].norm()) < gt, float((one[1] - ref[1]).norm() / ref[1].norm())
    a, a2 = _one_step(C, dtype, 6, cd, B, size, steps=2), _one_step(C, dtype, 6, cd, B, size, steps=2)
    assert torch.equal(a[0], a2[0]) and torch.equal(a[1], a2[1]) and torch.equal(a[2], a2[2])
    monkeypatch.setattr(U, 'KERNEL_TIMING', [])              # one-stream mode (the fix-up stays behind its weight gradient): same results
    b = _one_step(C, dtype, 6, cd, B, size, steps=2)
    monkeypatch.setattr(U, 'KERNEL_TIMING', None)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    # eval mode (running statistics in the filters and the table) and the fused arg-max forward
    x = torch.from_numpy(C.synth.images(5, B, 3, size, size)).cuda()
    m_on, m_off = a[3].eval(), ref[3].eval()
    m_off.load_state_dict(m_on.state_dict())
    with torch.no_grad():
        lo_on, lo_off = m_on(x), m_off(x)
        assert rel_l2(lo_on.cpu().numpy(), lo_off.cpu().numpy()) < {'fp32': 2e-5, 'bf16x3': 1e-4, 'bf16': 3e-2}[dtype]
        assert torch.equal(m_on.predict(x), torch.max(lo_on, 1)[1])


def test_batchnorm_of_an_encoder_block_folded_into_both_readers(C, monkeypatch):
    """unet.FOLD_POOLED (fp32): the BatchNorm at the END of an encoder block folded into both readers of the block's output -- the next
    block's first convolution (through the 2x2 max-pool) and the decoder convolution that takes it through the concat buffer
    (models/unet.py:80-87).  The block's second convolution writes its conv+ReLU output into the concat slice, a pooling pass takes the
    window maximum -- the MINIMUM where the BatchNorm scale is negative: half of the scales are made negative here -- and the readers carry
    scale / shift in their filters and border-class bias tables ([scale | 1], [shift | 0] over the concatenated input).  One step against the
    run without it (loss to rounding, gradients to the conditioning floor), bit-reproducible, eval mode and predict()."""
    from continual_learning_amd import unet as U
    nc, cd, B, size = 6, 64, 2, 64

    def hook(m, opt):
        with torch.no_grad():
            m.enc1[5].weight[::2] *= -1.0           # negative BatchNorm scales: the pooling pass must take the window minimum there

    monkeypatch.setattr(U, 'FOLD_POOLED', False)
    ref = _one_step(C, 'fp32', nc, cd, B, size, hook=hook)
    assert not any(u.pool_fold for u in next(iter(ref[3]._engines.values())).convs)
    monkeypatch.setattr(U, 'FOLD_POOLED', True)
    one = _one_step(C, 'fp32', nc, cd, B, size, hook=hook)
    eng = next(iter(one[3]._engines.values()))
    pf = [u for u in eng.convs if u.pool_fold]
    assert [u.name for u in pf] == ['enc1.3'] and pf[0].y is eng.cat[0] and pf[0].y_ldc == 2 * pf[0].cout_p
    readers = [u.name for u in eng.convs if u.fold_on and isinstance(u.fold_a, U._FoldSource)]
    assert sorted(readers) == ['enc2.block.1', 'last.0'], readers
    assert abs(float(one[0]) - float(ref[0])) < 2e-6 * abs(float(ref[0])), (float(one[0]), float(ref[0]))
    assert float((one[1] - ref[1]).norm() / ref[1].norm()) < 1e-2
    a, a2 = _one_step(C, 'fp32', nc, cd, B, size, hook=hook, steps=2), _one_step(C, 'fp32', nc, cd, B, size, hook=hook, steps=2)
    assert torch.equal(a[0], a2[0]) and torch.equal(a[1], a2[1]) and torch.equal(a[2], a2[2])
    x = torch.from_numpy(C.synth.images(5, B, 3, size, size)).cuda()
    m_on, m_off = a[3].eval(), ref[3].eval()
    m_off.load_state_dict(m_on.state_dict())
    with torch.no_grad():
        lo_on, lo_off = m_on(x), m_off(x)
        assert rel_l2(lo_on.cpu().numpy(), lo_off.cpu().numpy()) < 2e-5
        assert torch.equal(m_on.predict(x), torch.max(lo_on, 1)[1])


def test_winograd44_engine_paths_agree(C, monkeypatch):
    """unet.WINOGRAD44 (round 5): the wide 3x3 layers -- forward, data gradient, weight gradient on the kept forward image, the BatchNorm folded
    into the input transform -- and the narrow layers' data gradients by the pre-transformed F(4x4,3x3) kernels (csrc/wino44g.hip) instead of
    F(2x4).  'auto' takes them only where a launch fills the chip (BASELINE configs[1]: the full-size golden tests run that); here they are
    FORCED onto a small problem (conv_dim 64 at 64x64: 256- to 1024-channel layers at 16x16 down to 4x4, ragged tile blocks included) and
    compared with the F(2x4) engine: loss to rounding, gradients to the conditioning floor, bit-reproducible over two steps, eval mode."""
    from continual_learning_amd import unet as U
    nc, cd, B, size = 6, 64, 2, 64
    monkeypatch.setattr(U, 'WINOGRAD44', False)
    ref = _one_step(C, 'fp32', nc, cd, B, size)
    e0 = next(iter(ref[3]._engines.values()))
    assert not any(u.f44 or u.d44 for u in e0.convs) and any(u.pre_f for u in e0.convs)
    monkeypatch.setattr(U, 'WINOGRAD44', True)
    one = _one_step(C, 'fp32', nc, cd, B, size)
    eng = next(iter(one[3]._engines.values()))
    # (at this size every data gradient has too few F(2x4) work items and runs F(2x2): the F(4x4) data gradients are exercised by the
    # full-size golden tests and, kernel by kernel, by tests/test_wino44_gpu.py)
    assert sum(u.f44 for u in eng.convs) >= 4 and any(u.f44 and u.pre_w for u in eng.convs), [(u.name, u.f44, u.d44, u.pre_w) for u in eng.convs]
    assert any(u.apply_folded for u in eng.convs)                  # a BatchNorm applied by the F(4x4) input transform
    assert abs(float(one[0]) - float(ref[0])) < 2e-6 * abs(float(ref[0])), (float(one[0]), float(ref[0]))
    assert float((one[1] - ref[1]).norm() / ref[1].norm()) < 1e-2
    a, a2 = _one_step(C, 'fp32', nc, cd, B, size, steps=2), _one_step(C, 'fp32', nc, cd, B, size, steps=2)
    assert torch.equal(a[0], a2[0]) and torch.equal(a[1], a2[1]) and torch.equal(a[2], a2[2])
    x = torch.from_numpy(C.synth.images(5, B, 3, size, size)).cuda()
    m_on, m_off = a[3].eval(), ref[3].eval()
    m_off.load_state_dict(m_on.state_dict())
    with torch.no_grad():
        assert rel_l2(m_on(x).cpu().numpy(), m_off(x).cpu().numpy()) < 2e-5


def test_engine_buffers_are_released_with_the_model(C):
    """A model's engine (activations, gradients, workspaces: GBs at full size) must go when the model goes, by reference
    counting -- not whenever the cyclic garbage collector next runs (a trainer that rebuilds models, or begin_task2's
    snapshot, would otherwise pile up device memory)."""
    import gc
    gc.collect(); torch.cuda.synchronize()
    base = torch.cuda.memory_allocated()
    gc.disable()
    try:
        for _ in range(3):
            r = _one_step(C, 'fp32', 6, 16, 2, 64)
            assert torch.cuda.memory_allocated() > base + (1 << 20)
            del r
            assert torch.cuda.memory_allocated() <= base + (64 << 10), torch.cuda.memory_allocated() - base
    finally:
        gc.enable()


def test_gradsync_rccl_world1_on_gpu(C):
    """The RCCL code path of ddp.GradSync (side stream, per-stage buckets, optimiser hook) with a 1-rank "nccl"
    process group on the one GPU of the test box: a world-1 all-reduce is the identity, so loss, gradients and updated
    weights must equal the plain single-GPU step BIT FOR BIT (trainer.py:120-122: replica 0 == single process)."""
    import os
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    created = False
    if not dist.is_initialized():
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
        created = True
    try:
        def ddp_hook(m, opt):
            C.ddp.broadcast_parameters(m)
            C.ddp.GradSync(m, opt, min_bucket_bytes=16 << 10, grad_dtype='fp32')      # bit-level identity needs the fp32 exchange
            assert opt.grad_scale == 1.0 and m.tuning.cu_reserve == 0        # one rank: no channels to make room for
        for dtype in ('fp32', 'bf16', 'bf16x3'):
            plain = _one_step(C, dtype, 6, 8, 2, 64, steps=2)
            synced = _one_step(C, dtype, 6, 8, 2, 64, steps=2, hook=ddp_hook)
            assert torch.equal(plain[0], synced[0]), (dtype, float(plain[0]), float(synced[0]))
            assert torch.equal(plain[1], synced[1]), f'{dtype}: gradients differ, rel {float((plain[1] - synced[1]).norm() / plain[1].norm()):.2e}'
            assert torch.equal(plain[2], synced[2]), f'{dtype}: weights differ'
    finally:
        if created:
            dist.destroy_process_group()


def test_checkpoint_roundtrip_and_resume(C, tmp_path):
    """save_network / load_network keep the reference's file name and keys (trainer.py:68-102); a
    resumed trainer continues exactly like the uninterrupted one, and the file loads into stock torch objects."""
    cfg = C.default_config(n_iters=6, lr=1e-3, num_classes=4, conv_dim=4, compute_dtype='fp32', stats_every=1)
    data = [(torch.from_numpy(C.synth.images(11, 2, 3, 32, 32)), torch.from_numpy(C.synth.labels(11, 2, 32, 32, 4)))]
    torch.manual_seed(0)
    a = C.Trainer(data, cfg)
    a.train_val(epochs=2)
    path = a.save_network('UNET_VOC', 'latest', a.start_epoch - 1, str(tmp_path))
    assert os.path.basename(path) == 'latest_net_UNET_VOC.pth'
    ck = torch.load(path, map_location='cpu', weights_only=False)
    assert set(ck) == {'epoch', 'model_state', 'optimizer_state', 'scheduler_state'} and ck['epoch'] == 2
    ref = TC.build_unet(4, 3, 4)
    ref.load_state_dict(ck['model_state'], strict=True)                      # the reference module layout
    ropt = TC.make_optimizer(ref, lr=1e-3)
    ropt.load_state_dict(ck['optimizer_state'])                               # torch.optim.Adam accepts the state
    assert float(ropt.state[next(iter(ref.parameters()))]['step']) == 2.0
    b = C.Trainer(data, cfg)
    assert b.load_network('UNET_VOC', 'latest', str(tmp_path)) and b.start_epoch == 2
    sa, sb = a.train_val(epochs=1)[0], b.train_val(epochs=1)[0]
    assert abs(sa['lr'] - sb['lr']) < 1e-12
    assert abs(sa['loss'] - sb['loss']) < 2e-3 * abs(sa['loss'])


@pytest.mark.gpu
@pytest.mark.parametrize('dtype,cd,size', [('fp32', 8, 64), ('bf16x3', 8, 64), ('fp32', 64, 64)])
def test_graphed_step_matches_eager(C, dtype, cd, size):
    """Every entry point of the library only enqueues on the caller's stream, so the whole train step CAN be captured in ONE HIP graph
    (tools/graphed_step.py: a measurement tool, not part of the package -- the replay is slower than the three-stream eager step) and replays
    the same kernels as the eager loop:
    the SAME loss sequence, bit for bit (every reduction is a fixed-order sum); the LambdaLR schedule still reaches the
    captured Adam kernel (lr is read from device memory), and the host-side step counter follows the replays."""
    dev = torch.device('cuda', 0)
    # (fp32, 64, 64): 256-channel layers at 16 x 16 -- the pre-transformed Winograd kernels, the BatchNorm folded into their input
    # transform and the direct ConvTranspose GEMMs are inside the captured graph
    x = torch.from_numpy(C.synth.images(5, 2, 3, size, size)).to(dev)
    y = torch.from_numpy(C.synth.labels(5, 2, size, size, 5)).to(dev)

    def make():
        torch.manual_seed(3)
        m = C.UNet(5, 3, cd, compute_dtype=dtype).to(dev).train()
        o = C.FusedAdam(m.parameters(), lr=1e-3, betas=[0.5, 0.99])
        sch = torch.optim.lr_scheduler.LambdaLR(o, lambda n: 0.5 ** n)
        return m, o, sch, C.CrossEntropyLoss()

    m1, o1, s1, c1 = make()
    ref = []
    for i in range(6):
        out = m1(x); o1.zero_grad(); loss = c1(out, y); loss.backward(); o1.step()
        ref.append(float(loss.detach()))
        if i == 3:
            s1.step()                                    # halve the learning rate after the 4th step
    m2, o2, s2, c2 = make()
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    from graphed_step import GraphedStep
    step = GraphedStep(m2, o2, c2, x, y, warmup=3)       # 3 eager steps, then the capture (not executed)
    if cd == 64:
        eng = next(iter(m2._engines.values()))
        assert any(u.pre_f for u in eng.convs) and any(u.pre_w for u in eng.convs) and any(u.apply_folded for u in eng.convs)
    got = [float(l) for l in step.eager_losses]
    flat = lambda m: torch.cat([p.detach().reshape(-1) for p in m.parameters()]).clone()
    moved = []
    for i in range(3, 6):
        w_before = flat(m2)
        got.append(float(step(x, y)))
        moved.append(float((flat(m2) - w_before).abs().mean()))
        if i == 3:
            s2.step()
    assert got == ref, (got, ref)
    assert float(o2.state[next(iter(m2.parameters()))]['step']) == 6.0
    # Adam moves every weight by about lr per step: after the scheduler halved lr, the replayed (captured) Adam kernel
    # must move the weights half as far -- it reads lr from device memory, which sync_hyper() refreshed before the replay
    assert 0.25 < moved[2] / moved[0] < 0.7, moved


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', ['fp32', 'bf16x3'])
def test_training_curve_tracks_stock_torch(C, dtype):
    """40 Adam steps of UNet(21,3,16) at 128x128, bs 4, on one synthetic batch: the loss curve of the HIP path follows the
    stock torch.nn counterpart (oracle/torch_cpu.py, here on the same GPU = MIOpen fp32) — a longer horizon than the
    2-3-step golden captures, through every Winograd/direct kernel choice the engine makes at these sizes.  Adam makes
    the trajectories diverge slowly (sign-like first updates), so the bound is on the curve, not on the weights."""
    dev = torch.device('cuda', 0)
    nc, cd, size, bs, steps = 21, 16, 128, 4, 40
    x = torch.from_numpy(C.synth.images(77, bs, 3, size, size)).to(dev)
    y = torch.from_numpy(C.synth.labels(77, bs, size, size, nc)).to(dev)
    torch.manual_seed(11)
    ref = TC.build_unet(nc, 3, cd).to(dev).train()
    ours = C.UNet(nc, 3, cd, compute_dtype=dtype).to(dev).train()
    ours.load_state_dict(ref.state_dict())
    o_ref = TC.make_optimizer(ref, lr=2e-4)
    o_ours = C.FusedAdam(ours.parameters(), lr=2e-4, betas=[0.5, 0.99])
    c_ref, c_ours = torch.nn.CrossEntropyLoss(), C.CrossEntropyLoss()
    l_ref, l_ours = [], []
    for _ in range(steps):
        out = ref(x); o_ref.zero_grad(); l = c_ref(out, y); l.backward(); o_ref.step(); l_ref.append(float(l.detach()))
        out2 = ours(x); o_ours.zero_grad(); l2 = c_ours(out2, y); l2.backward(); o_ours.step(); l_ours.append(float(l2.detach()))
    assert l_ref[-1] < 0.8 * l_ref[0]                                  # it does train
    assert l_ours == pytest.approx(l_ref, rel=1e-2)
    assert l_ours[:5] == pytest.approx(l_ref[:5], rel=5e-4)
    acc_ref = float((out.argmax(1) == y).float().mean()); acc_ours = float((out2.argmax(1) == y).float().mean())
    assert abs(acc_ref - acc_ours) < 0.02


def _random_model_configs(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        out.append((int(rng.integers(2, 24)), int(rng.choice([3, 5, 8, 12, 16, 20, 24])), 16 * int(rng.integers(2, 9)),
                    16 * int(rng.integers(2, 9)), int(rng.integers(1, 4))))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', ['fp32', 'bf16x3'])
@pytest.mark.parametrize('cfg', _random_model_configs(int(os.environ.get('MODEL_SWEEP', '5')), 5),
                         ids=lambda c: f'nc{c[0]}-cd{c[1]}-{c[2]}x{c[3]}-b{c[4]}')
def test_random_model_configs_vs_stock_torch(C, cfg, dtype):
    """Whole forward + loss + backward on random (classes, conv_dim, H, W, batch) — odd conv_dims (channel padding at every
    level), non-square images, Winograd and direct kernels mixed by size — against the stock torch counterpart on the
    same GPU: logits, loss and every parameter gradient of one step."""
    nc, cd, H, W, B = cfg
    dev = torch.device('cuda', 0)
    x = torch.from_numpy(C.synth.images(31, B, 3, H, W)).to(dev)
    y = torch.from_numpy(C.synth.labels(31, B, H, W, nc)).to(dev)
    torch.manual_seed(cd * 1000 + nc)
    ref = TC.build_unet(nc, 3, cd).to(dev).train()
    ours = C.UNet(nc, 3, cd, compute_dtype=dtype).to(dev).train()
    ours.load_state_dict(ref.state_dict())
    out_r = ref(x); l_r = torch.nn.CrossEntropyLoss()(out_r, y); l_r.backward()
    out_o = ours(x); l_o = C.CrossEntropyLoss()(out_o, y); l_o.backward()
    torch.cuda.synchronize()
    tol = 2e-4 if dtype == 'fp32' else 1e-3
    assert rel_l2(out_o.detach().cpu().numpy(), out_r.detach().cpu().numpy()) < tol
    assert float(l_o.detach()) == pytest.approx(float(l_r.detach()), rel=tol)
    # Gradients of small random nets are decided by ReLU / max-pool TIES: a pre-activation within rounding distance of zero, or two
    # nearly equal values in a pooling window, falls the other way, the gradient takes another route, and at the deep levels
    # (tens of samples per channel here) one such flip moves a whole tensor's gradient by ~1/samples.  This is a property of
    # the network, not of an implementation, and the test DEMONSTRATES it on the reference arithmetic itself: the stock
    # counterpart in float64 (CPU), re-run with every conv weight perturbed by a relative 1e-6 (the size of fp32 Winograd
    # rounding; 1e-5 for bf16x3's split products), moves its own gradients by `floor` -- orders of magnitude more than the
    # perturbation wherever a tie flips.  This path may deviate from the unperturbed fp64 gradients by at most 2x the largest
    # floor seen over 8 perturbation seeds (whole gradient) / 3x (per weight tensor, where single flips are luckier).
    def fp64_grads(delta=0.0, seed=0):
        m64 = TC.build_unet(nc, 3, cd).double()
        sd64 = {k: (v.detach().cpu().double() if v.is_floating_point() else v.detach().cpu()) for k, v in ref.state_dict().items()}
        if delta:
            gen = torch.Generator().manual_seed(1000 + seed)
            for k in sd64:
                if k.endswith('.weight') and sd64[k].dim() == 4:
                    sd64[k] = sd64[k] * (1.0 + delta * torch.randn(sd64[k].shape, generator=gen, dtype=torch.float64))
        m64.load_state_dict(sd64)
        m64.train()
        l64 = torch.nn.CrossEntropyLoss()(m64(x.cpu().double()), y.cpu()); l64.backward()
        return {n: p.grad.numpy() for n, p in m64.named_parameters()}
    g64 = fp64_grads()
    delta = 1e-6 if dtype == 'fp32' else 1e-5
    names = [n for n, p in ours.named_parameters() if p.dim() > 1]       # biases in front of a BatchNorm cancel to ~0: covered by the total
    cat = lambda g: np.concatenate([np.asarray(g[n], np.float64).reshape(-1) for n, _ in ours.named_parameters()])
    floor_t = {n: 0.0 for n in names}
    floor_all = 0.0
    for seed in range(8):
        gp = fp64_grads(delta, seed)
        floor_all = max(floor_all, rel_l2(cat(gp), cat(g64)))
        for n in names:
            floor_t[n] = max(floor_t[n], rel_l2(gp[n], g64[n]))
    go = {n: p.grad.cpu().numpy() for n, p in ours.named_parameters()}
    e_all = rel_l2(cat(go), cat(g64))
    worst = max((rel_l2(go[n].astype(np.float64), g64[n]) / max(floor_t[n], 20 * delta), n) for n in names)
    print(f'gradient vs fp64: whole {e_all:.2e} (fp64 under a {delta:g} weight perturbation: {floor_all:.2e}); '
          f'worst tensor {worst[1]} at {worst[0]:.2f}x its perturbation floor {floor_t[worst[1]]:.2e}')
    assert e_all < 2.0 * max(floor_all, 20 * delta), (e_all, floor_all)
    assert worst[0] < 3.0, worst
