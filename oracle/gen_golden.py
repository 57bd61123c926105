"""Generate tests/golden/*.npz by IMPORTING THE REAL REFERENCE in the build container.

    python -m oracle.gen_golden            # needs /root/reference (read-only), torch CPU

The reference cannot travel to the GPU box, so the vectors captured here are committed as data
(inputs + expected outputs only; no reference source).  What runs below:
  * /root/reference/models/unet.py  UNet            (imported, unmodified)
  * /root/reference/metrics.py      eval_metrics    (imported, unmodified)
  * /root/reference/datasets/voc.py to_mask, to_rgb (imported, unmodified, with an EMPTY stand-in module named
    `torchvision` in sys.modules: voc.py imports torchvision.transforms at :7 but the two palette functions at
    :56-89 use only numpy and torch -- SURVEY.md §8c)
  * trainer.py cannot be imported (needs torchvision, SURVEY.md §8c -- an ordinary
    ModuleNotFoundError, not a denial); its hot loop trainer.py:108-114,147,172-176 is 12 lines of
    stock torch calls, which are issued here verbatim in meaning: Adam(lr, betas=[b1,b2]),
    LambdaLR poly, CrossEntropyLoss, forward -> zero_grad -> loss -> backward -> step.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
OUT = os.path.join(ROOT, 'tests', 'golden')


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _synth():
    return _load('clamd_synth', os.path.join(ROOT, 'continual-learning_amd', 'synth.py'))


def _np(sd):
    return {k: v.detach().cpu().numpy().copy() for k, v in sd.items()}


def _shapes(model):
    return {k: tuple(v.shape) for k, v in model.state_dict().items()
            if not k.endswith(('running_mean', 'running_var', 'num_batches_tracked'))}


def capture_forward_only(ref_unet, ref_metrics, synth, tag, num_classes, conv_dim, batch, size, logits_stride):
    """Train-mode forward + loss of a workload whose full train step does not fit this container's memory (config 5:
    512x512 bs32 is 8x config 2's ~8 GB of saved activations): logits subsample, loss, arg-max histogram, mIoU and the
    BatchNorm running statistics after the one forward.  The full step at this image size is pinned by the bs8 capture."""
    torch.manual_seed(0)
    model = ref_unet.UNet(num_classes=num_classes, in_dim=3, conv_dim=conv_dim)
    state = synth.closed_form_state(_shapes(model), seed=0)
    sd = model.state_dict()
    for k, v in state.items():
        sd[k] = torch.from_numpy(v)
    model.load_state_dict(sd)
    model.train()
    xt = torch.from_numpy(synth.images(1234, batch, 3, size, size))
    yt = torch.from_numpy(synth.labels(1234, batch, size, size, num_classes))
    with torch.no_grad():
        logits = model(xt)
        loss = torch.nn.CrossEntropyLoss()(logits, yt)
    pred = logits.argmax(1)
    oa, pc, miu, mx = ref_metrics.eval_metrics(yt, pred, num_classes)
    out = {'num_classes': num_classes, 'conv_dim': conv_dim, 'batch': batch, 'size': size, 'data_seed': 1234, 'weight_seed': 0,
           'logits_flat_stride': logits_stride, 'logits': logits.numpy().reshape(-1)[::logits_stride].copy(),
           'loss': float(loss), 'pred_hist': np.bincount(pred.numpy().reshape(-1), minlength=num_classes),
           'metrics': np.array([float(oa), float(pc), float(miu), float(mx)], np.float32),
           'stats1': np.concatenate([v.numpy().reshape(-1) for k, v in model.state_dict().items()
                                     if k.endswith(('running_mean', 'running_var'))])}
    np.savez_compressed(os.path.join(OUT, f'unet_{tag}.npz'), **out)
    print(tag, 'loss', float(loss), 'mIoU', float(miu))


def capture_training_miou(ref_unet, ref_metrics, synth, tag, num_classes, conv_dim, batch, size, nimg, epochs, lr):
    """north_star: 'mIoU within +-0.1 of reference on a fixed synthetic 21-class set' (64 images).  The
    reference model is trained with the hot loop of trainer.py:147,165-176 (scheduler-less: constant lr) for `epochs`
    passes over the fixed nimg-image set in batches of `batch`, then evaluated on the same set: the training-time metric of
    trainer.py:183-188 (arg-max of the train-mode outputs of the LAST epoch, confusion accumulated over its batches) and
    the eval-mode pass of trainer.py:270-284 (model.eval())."""
    torch.manual_seed(0)
    model = ref_unet.UNet(num_classes=num_classes, in_dim=3, conv_dim=conv_dim)
    state = synth.closed_form_state(_shapes(model), seed=0)
    sd = model.state_dict()
    for k, v in state.items():
        sd[k] = torch.from_numpy(v)
    model.load_state_dict(sd)
    model.train()
    optim = torch.optim.Adam(model.parameters(), lr=lr, betas=[0.5, 0.99])
    crit = torch.nn.CrossEntropyLoss()
    nb = nimg // batch
    data = []
    for i in range(nb):      # images that carry their labels (synth.images_with_signal): mIoU moves far from chance in a few steps
        lab = synth.labels(1234, batch, size, size, num_classes, first_image=i * batch)
        data.append((torch.from_numpy(synth.images_with_signal(1234, lab, num_classes, first_image=i * batch)), torch.from_numpy(lab)))
    losses, train_conf = [], torch.zeros(num_classes, num_classes)
    for ep in range(epochs):
        for xt, yt in data:
            logits = model(xt)
            optim.zero_grad()
            loss = crit(logits, yt)
            loss.backward()
            optim.step()
            losses.append(float(loss))
            if ep == epochs - 1:
                pred = logits.detach().argmax(1)
                for a, b in zip(yt, pred):
                    train_conf += ref_metrics._fast_conf_matrix(a.flatten(), b.flatten(), num_classes)
    model.eval()
    eval_conf = torch.zeros(num_classes, num_classes)
    with torch.no_grad():
        for xt, yt in data:
            pred = model(xt).argmax(1)
            for a, b in zip(yt, pred):
                eval_conf += ref_metrics._fast_conf_matrix(a.flatten(), b.flatten(), num_classes)
    out = {'num_classes': num_classes, 'conv_dim': conv_dim, 'batch': batch, 'size': size, 'nimg': nimg, 'epochs': epochs,
           'lr': lr, 'losses': np.array(losses), 'train_conf': train_conf.numpy(), 'eval_conf': eval_conf.numpy(),
           'train_miou': float(ref_metrics.mean_IU_2(train_conf)), 'eval_miou': float(ref_metrics.mean_IU_2(eval_conf)),
           'train_acc': float(ref_metrics.overall_pixel_acc(train_conf)), 'eval_acc': float(ref_metrics.overall_pixel_acc(eval_conf))}
    np.savez_compressed(os.path.join(OUT, f'train_{tag}.npz'), **out)
    print(tag, 'losses', losses[0], '->', losses[-1], 'train mIoU', out['train_miou'], 'eval mIoU', out['eval_miou'])


def capture_voc():
    """datasets/voc.py:56-89 to_mask / to_rgb, imported from the reference with an empty `torchvision` stand-in (the module
    only needs to EXIST for voc.py:7's import; neither function touches it)."""
    import types
    tv = types.ModuleType('torchvision')
    tv.transforms = types.ModuleType('torchvision.transforms')
    sys.modules.setdefault('torchvision', tv)
    sys.modules.setdefault('torchvision.transforms', tv.transforms)
    voc = _load('ref_voc', os.path.join(REF, 'datasets', 'voc.py'))
    rng = np.random.RandomState(7)
    pal = np.array(voc.palette, np.uint8)                       # 22 colours, the last one is "void"
    idx = rng.randint(0, 22, (24, 40))
    idx[:3, :5] = 21                                            # void pixels -> class 0 (voc.py:67-68)
    mask_rgb = pal[idx]
    labels = voc.to_mask(mask_rgb).numpy()
    lab_in = rng.randint(0, 22, (3, 10, 12)).astype(np.int64)
    rgb = voc.to_rgb(torch.from_numpy(lab_in)).numpy()
    np.savez_compressed(os.path.join(OUT, 'voc.npz'), palette=pal, mask_rgb=mask_rgb, labels=labels.astype(np.int64),
                        to_rgb_in=lab_in, to_rgb_out=rgb)
    print('voc ok', labels.shape, rgb.shape, rgb.dtype)


def capture_model(ref_unet, ref_metrics, synth, tag, num_classes, conv_dim, batch, size, steps, lr,
                  store_weights, logits_stride=1, store_grads=True):
    torch.manual_seed(0)
    model = ref_unet.UNet(num_classes=num_classes, in_dim=3, conv_dim=conv_dim)
    state = synth.closed_form_state(_shapes(model), seed=0)
    sd = model.state_dict()
    for k, v in state.items():
        sd[k] = torch.from_numpy(v)
    model.load_state_dict(sd)
    model.train()
    x = synth.images(1234, batch, 3, size, size)
    y = synth.labels(1234, batch, size, size, num_classes)
    xt, yt = torch.from_numpy(x), torch.from_numpy(y)

    # trainer.py:108-114
    optim = torch.optim.Adam(model.parameters(), lr=lr, betas=[0.5, 0.99])
    crit = torch.nn.CrossEntropyLoss()
    out = {'num_classes': num_classes, 'conv_dim': conv_dim, 'batch': batch, 'size': size, 'lr': lr,
           'data_seed': 1234, 'weight_seed': 0}
    if store_weights:
        for k, v in state.items():
            out['w0/' + k] = v
    losses = []
    for s in range(steps):
        # trainer.py:172-176
        logits = model(xt)
        optim.zero_grad()
        loss = crit(logits, yt)
        loss.backward()
        if s == 0:
            lg = logits.detach().numpy()
            out['logits_flat_stride'] = logits_stride
            out['logits'] = lg if logits_stride == 1 else lg.reshape(-1)[::logits_stride].copy()
            pred = logits.detach().argmax(1)
            out['pred_hist'] = np.bincount(pred.numpy().reshape(-1), minlength=num_classes)
            oa, pc, miu, mx = ref_metrics.eval_metrics(yt, pred, num_classes)
            out['metrics'] = np.array([float(oa), float(pc), float(miu), float(mx)], np.float32)
            oa2, pc2, miu2, mx2 = ref_metrics.eval_metrics(yt, pred, num_classes + 1)   # trainer.py:188 quirk
            out['metrics_plus1'] = np.array([float(oa2), float(pc2), float(miu2), float(mx2)], np.float32)
            names = [n for n, _ in model.named_parameters()]
            out['grad_norms'] = np.array([float(p.grad.double().norm()) for _, p in model.named_parameters()])
            out['grad_names'] = np.array(names)
            if store_grads:
                for n, p in model.named_parameters():
                    out['g0/' + n] = p.grad.numpy().copy()
            out['stats1'] = np.concatenate([v.numpy().reshape(-1) for k, v in model.state_dict().items()
                                            if k.endswith(('running_mean', 'running_var'))])
        optim.step()
        losses.append(float(loss))
        if store_weights and s in (0, steps - 1):
            for k, v in _np(model.state_dict()).items():
                if not k.endswith('num_batches_tracked'):
                    out[f'w{s + 1}/' + k] = v
    out['losses'] = np.array(losses, np.float64)
    out['param_norms_end'] = np.array([float(p.detach().double().norm()) for p in model.parameters()])
    np.savez_compressed(os.path.join(OUT, f'unet_{tag}.npz'), **out)
    print(tag, 'losses', losses, 'mIoU', out['metrics'][2])


def capture_ops():
    """Per-op vectors from the torch ops the reference's call sites use (models/unet.py:12-18,28-34,72,80;
    trainer.py:113).  Shapes are small; odd channel counts and negative BN gammas on purpose."""
    torch.manual_seed(1)
    F = torch.nn.functional
    out = {}

    def leaf(*s):
        return torch.randn(*s, requires_grad=True)

    # conv3x3 + bias -> relu -> bn(train)
    x, w, b = leaf(2, 5, 8, 12), leaf(7, 5, 3, 3), leaf(7)
    g, be = leaf(7), leaf(7)   # gammas of both signs
    rm, rv = torch.zeros(7), torch.ones(7)
    z = F.conv2d(x, w, b, padding=1)
    yr = F.relu(z)
    u = F.batch_norm(yr, rm, rv, g, be, training=True, momentum=0.1, eps=1e-5)
    go = torch.randn_like(u)
    u.backward(go)
    out.update({'cbr/x': x, 'cbr/w': w, 'cbr/b': b, 'cbr/gamma': g, 'cbr/beta': be, 'cbr/z': z, 'cbr/u': u,
                'cbr/rm': rm, 'cbr/rv': rv, 'cbr/go': go, 'cbr/gx': x.grad, 'cbr/gw': w.grad, 'cbr/gb': b.grad,
                'cbr/ggamma': g.grad, 'cbr/gbeta': be.grad})
    # maxpool with ties
    xp = torch.randint(0, 3, (2, 3, 6, 8)).float().requires_grad_()
    yp = F.max_pool2d(xp, 2, 2)
    gp = torch.randn_like(yp)
    yp.backward(gp)
    out.update({'pool/x': xp, 'pool/y': yp, 'pool/go': gp, 'pool/gx': xp.grad})
    # convT k2 s2
    xt, wt, bt = leaf(2, 6, 4, 5), leaf(6, 3, 2, 2), leaf(3)
    yt = F.conv_transpose2d(xt, wt, bt, stride=2)
    gt = torch.randn_like(yt)
    yt.backward(gt)
    out.update({'convT/x': xt, 'convT/w': wt, 'convT/b': bt, 'convT/y': yt, 'convT/go': gt,
                'convT/gx': xt.grad, 'convT/gw': wt.grad, 'convT/gb': bt.grad})
    # 1x1 head
    xh, wh, bh = leaf(2, 6, 4, 4), leaf(5, 6, 1, 1), leaf(5)
    yh = F.conv2d(xh, wh, bh)
    gh = torch.randn_like(yh)
    yh.backward(gh)
    out.update({'head/x': xh, 'head/w': wh, 'head/b': bh, 'head/y': yh, 'head/go': gh,
                'head/gx': xh.grad, 'head/gw': wh.grad, 'head/gb': bh.grad})
    # cross entropy (mean), plus one with ignore_index pixels
    lg = (torch.randn(2, 21, 4, 6) * 3).requires_grad_()
    lb = torch.randint(0, 21, (2, 4, 6))
    l = F.cross_entropy(lg, lb)
    l.backward()
    out.update({'ce/logits': lg, 'ce/labels': lb, 'ce/loss': l, 'ce/dlogits': lg.grad})
    lg2 = lg.detach().clone().requires_grad_()
    lb2 = lb.clone()
    lb2[0, 0, :3] = -100
    l2 = F.cross_entropy(lg2, lb2)
    l2.backward()
    out.update({'ce_ign/labels': lb2, 'ce_ign/loss': l2, 'ce_ign/dlogits': lg2.grad})
    # Adam, 3 steps, betas (0.5, 0.99) (main.py defaults), and LambdaLR sequence trainer.py:111-112,147
    p = torch.nn.Parameter(torch.randn(37))
    opt = torch.optim.Adam([p], lr=1e-2, betas=[0.5, 0.99])
    out['adam/p0'] = p.detach().clone()
    gs = torch.randn(3, 37)
    out['adam/grads'] = gs
    for i in range(3):
        p.grad = gs[i].clone()
        opt.step()
        out[f'adam/p{i + 1}'] = p.detach().clone()
    st = opt.state[p]
    out['adam/m3'], out['adam/v3'] = st['exp_avg'], st['exp_avg_sq']
    q = torch.nn.Parameter(torch.zeros(1))
    o2 = torch.optim.Adam([q], lr=1e-4, betas=[0.5, 0.99])
    sch = torch.optim.lr_scheduler.LambdaLR(o2, lr_lambda=lambda n: (1 - n / 10) ** 0.9)
    lrs = []
    for e in range(10):
        sch.step()                      # trainer.py:147 (before the epoch's optimiser steps)
        lrs.append(o2.param_groups[0]['lr'])
    out['sched/lrs_n10'] = np.array(lrs)
    np.savez_compressed(os.path.join(OUT, 'ops.npz'),
                        **{k: (v.detach().numpy() if torch.is_tensor(v) else v) for k, v in out.items()})
    print('ops ok; lrs', lrs[:3])


def capture_metrics(ref_metrics):
    rng = np.random.RandomState(3)
    t = rng.randint(0, 21, (4, 32, 32)).astype(np.int64)
    p = np.where(rng.rand(4, 32, 32) < 0.6, t, rng.randint(0, 21, (4, 32, 32))).astype(np.int64)
    t[t == 7] = 3            # leave class 7 absent from the targets -> NaN rows dropped by nanmean
    out = {'target': t, 'pred': p}
    for c in (21, 22):
        oa, pc, miu, mx = ref_metrics.eval_metrics(torch.from_numpy(t), torch.from_numpy(p), c)
        out[f'm{c}'] = np.array([float(oa), float(pc), float(miu), float(mx)], np.float32)
    m = torch.zeros(21, 21)
    for a, b in zip(torch.from_numpy(t), torch.from_numpy(p)):
        m += ref_metrics._fast_conf_matrix(a.flatten(), b.flatten(), 21)
    out['conf21'] = m.numpy()
    np.savez_compressed(os.path.join(OUT, 'metrics.npz'), **out)
    print('metrics', out['m21'], out['m22'])


def main():
    if not os.path.isdir(REF):
        sys.exit('reference not mounted; golden vectors are generated in the build container only')
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(os.cpu_count())
    ref_unet = _load('ref_unet', os.path.join(REF, 'models', 'unet.py'))
    ref_metrics = _load('ref_metrics', os.path.join(REF, 'metrics.py'))
    synth = _synth()
    which = sys.argv[1:] or ['ops', 'metrics', 'voc', 'small', 'mid', 'c1', 'full', 'c5', 'miou64']
    if 'voc' in which:
        capture_voc()
    if 'ops' in which:
        capture_ops()
    if 'metrics' in which:
        capture_metrics(ref_metrics)
    if 'small' in which:     # reference's own smoke shape family (unet.py:94-97): 2 classes, 32x32, bs2
        capture_model(ref_unet, ref_metrics, synth, 'cd4_c2_32', 2, 4, 2, 32, 3, 1e-3, store_weights=True)
    if 'mid' in which:       # config-1-like: 64x64 bs2, 21 classes, conv_dim 8
        capture_model(ref_unet, ref_metrics, synth, 'cd8_c21_64', 21, 8, 2, 64, 3, 1e-3, store_weights=False)
    if 'c1' in which:        # BASELINE.json configs[0] exactly: UNet(2,3,64) 64x64 bs2, 3 steps (CPU-runnable plumbing config)
        capture_model(ref_unet, ref_metrics, synth, 'cd64_c2_64', 2, 64, 2, 64, 3, 1e-4, store_weights=False, store_grads=False)
    if 'full' in which:      # config 2: the real thing, 256x256 bs16 conv_dim 64 (about 1 minute of CPU)
        capture_model(ref_unet, ref_metrics, synth, 'cd64_c21_256', 21, 64, 16, 256, 2, 1e-4,
                      store_weights=False, logits_stride=997, store_grads=False)


    if 'c5' in which:
        # BASELINE.json configs[4] (512x512, bs32 per GPU).  The full train step at bs32 needs ~60 GB of saved activations on
        # the CPU -- more than this container has -- so: (a) the full 2-step capture at 512x512 with bs8 (every 512^2 tile
        # geometry, gradients, Adam), (b) the train-mode forward + loss at the real bs32 (BatchNorm statistics over 32 images).
        capture_model(ref_unet, ref_metrics, synth, 'cd64_c21_512_b8', 21, 64, 8, 512, 2, 1e-4,
                      store_weights=False, logits_stride=1999, store_grads=False)
        capture_forward_only(ref_unet, ref_metrics, synth, 'cd64_c21_512_b32_fwd', 21, 64, 32, 512, logits_stride=7993)
    if 'miou64' in which:    # the fixed 64-image set of SURVEY §8d, 4 epochs of 4 x bs16 steps at 256x256 (about 4 CPU-minutes)
        capture_training_miou(ref_unet, ref_metrics, synth, 'cd64_c21_256_n64', 21, 64, 16, 256, 64, 4, 1e-3)


if __name__ == '__main__':
    main()
