"""CPU oracle #2: the reference's train step rebuilt from STOCK torch.nn layers.

TEST INFRASTRUCTURE ONLY (see oracle/np_unet.py header for the import rule).  Used as
  * the mid/full-size parity checker on the GPU box, where /root/reference does not exist;
  * ``bench.py``'s ``cpu_baseline`` (kind "port"): SURVEY.md §8d asks for "the build's own
    trainer counterpart on CPU PyTorch ... stock torch.nn layers in the A2 structure, CE, Adam".

Parity status: PINNED -- tests/test_oracle_golden.py loads the golden state_dict captured from the
real reference (/root/reference/models/unet.py) into ``build_unet`` (strict=True: the 136 keys must
match) and reproduces the captured logits / loss / grads / post-Adam weights.

Structure follows models/unet.py:49-72 (Conv -> ReLU -> BatchNorm order) but is
generated from oracle.np_unet.layer_table rather than written out by hand.
"""
import os
import time

import torch
import torch.nn as nn

from . import np_unet


class _Wrapped(nn.Module):
    """A module whose only child is ``block`` -- gives the 'encN.block.K' / 'decN.block.K' key names
    of models/unet.py:8-38."""

    def __init__(self, layers):
        super().__init__()
        self.block = nn.Sequential(*layers)

    def forward(self, x):
        return self.block(x)


def _stage_layers(pool_first, convs, tail):
    layers = [nn.MaxPool2d(2, 2)] if pool_first else []
    for _, _, cin, cout in convs:
        layers += [nn.Conv2d(cin, cout, 3, 1, 1), nn.ReLU(), nn.BatchNorm2d(cout)]
    if tail is not None:
        kind, _, cin, cout = tail
        layers.append(nn.ConvTranspose2d(cin, cout, 2, 2) if kind == 'convT' else nn.Conv2d(cin, cout, 1, 1))
    return layers


class TorchUNet(nn.Module):
    """Same constructor, parameter order and state_dict keys as models/unet.py:40-72."""

    def __init__(self, num_classes, in_dim=3, conv_dim=64):
        super().__init__()
        self.num_classes, self.in_dim, self.conv_dim = num_classes, in_dim, conv_dim
        for prefix, pool_first, convs, tail in np_unet.layer_table(num_classes, in_dim, conv_dim):
            layers = _stage_layers(pool_first, convs, tail)
            wrapped = prefix.startswith('enc') and prefix != 'enc1' or prefix.startswith('dec')
            self.add_module(prefix, _Wrapped(layers) if wrapped else nn.Sequential(*layers))

    def forward(self, x):
        e1 = self.enc1(x)
        e2 = self.enc2(e1)
        e3 = self.enc3(e2)
        e4 = self.enc4(e3)
        h = self.dec1(nn.functional.max_pool2d(e4, 2, 2))
        h = self.dec2(torch.cat([e4, h], 1))
        h = self.dec3(torch.cat([e3, h], 1))
        h = self.dec4(torch.cat([e2, h], 1))
        return self.last(torch.cat([e1, h], 1))


def build_unet(num_classes, in_dim=3, conv_dim=64, state=None):
    m = TorchUNet(num_classes, in_dim, conv_dim)
    if state is not None:
        sd = {k: torch.as_tensor(v) for k, v in state.items()}
        m.load_state_dict(sd, strict=True)
    return m


def make_optimizer(model, lr=1e-4, beta1=0.5, beta2=0.99):
    """trainer.py:108-110."""
    return torch.optim.Adam(model.parameters(), lr=lr, betas=[beta1, beta2])


def make_scheduler(optim, n_iters, lr_exp=0.9):
    """trainer.py:111-112."""
    return torch.optim.lr_scheduler.LambdaLR(optim, lr_lambda=lambda n: (1 - n / n_iters) ** lr_exp)


def train_step(model, optim, criterion, images, labels):
    """trainer.py:172-176, in that order (forward, zero_grad, loss, backward, step)."""
    out = model(images)
    optim.zero_grad()
    loss = criterion(out, labels)
    loss.backward()
    optim.step()
    return out, loss


def distill_loss(z_new, z_old, c_old, temperature, lam):
    """BUILD-DEFINED (no reference code; SURVEY.md §8a A12, parity unpinned): the autograd form of
    oracle.np_unet.distill_kl -- lam * mean_px KL(softmax(z_old[:, :c_old]/T) || softmax(z_new[:, :c_old]/T))."""
    la = torch.log_softmax(z_new[:, :c_old] / temperature, 1)
    lb = torch.log_softmax(z_old[:, :c_old].detach() / temperature, 1)
    npx = z_new.shape[0] * z_new.shape[2] * z_new.shape[3]
    return lam * (lb.exp() * (lb - la)).sum() / npx


def continual_two_task(model, task1, task2, c_old, distill_lambda, temperature, l2_lambda, lr):
    """BASELINE.json configs[3] (SURVEY.md §8d "Config 4"), composed from stock torch ops: task 1 = the hot loop of
    trainer.py:172-176 over `task1` [(images, labels restricted to classes < c_old)]; snapshot; task 2 = the same loop over
    `task2` with loss = CE + distillation towards the frozen snapshot (eval mode) + l2_lambda * sum ||theta - theta_old||^2.
    BUILD-DEFINED procedure (the reference has no continual-learning code): parity unpinned.
    Returns dict(losses1, losses2 [(total, ce, kd, l2)], old_state)."""
    import copy
    optim = make_optimizer(model, lr=lr)
    crit = nn.CrossEntropyLoss()
    model.train()
    losses1 = [float(train_step(model, optim, crit, x, y)[1].detach()) for x, y in task1]
    old = copy.deepcopy(model).eval()
    for p in old.parameters():
        p.requires_grad_(False)
    old_params = [p.detach().clone() for p in model.parameters()]
    losses2 = []
    for x, y in task2:
        out = model(x)
        optim.zero_grad()
        with torch.no_grad():
            zo = old(x)
        ce = crit(out, y)
        kd = distill_loss(out, zo, c_old, temperature, distill_lambda) if distill_lambda > 0 else out.new_zeros(())
        l2 = l2_lambda * sum(((p - o) ** 2).sum() for p, o in zip(model.parameters(), old_params)) if l2_lambda > 0 else out.new_zeros(())
        (ce + kd + l2).backward()
        optim.step()
        losses2.append((float((ce + kd).detach()), float(ce.detach()), float(kd.detach()), float(l2.detach())))
    return {'losses1': losses1, 'losses2': losses2, 'old_state': {k: v.detach().clone() for k, v in old.state_dict().items()}}


def usable_cores():
    """Cores this process may actually use: affinity mask, capped by the cgroup CPU quota (os.cpu_count() reports the
    whole host on the GPU box and oversubscribing 256 threads on a 16-core share is ~10x slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))      # a 1-GPU box has a 16-core share


def time_cpu_baseline(batch=16, size=256, num_classes=21, conv_dim=64, steps=1, warmup=1, threads=None,
                      images=None, labels=None):
    """Times the stock-torch CPU train step on this host.  Returns dict(value img/s, cores, sample, s_per_step)."""
    threads = threads or usable_cores()
    torch.set_num_threads(threads)
    m = build_unet(num_classes, 3, conv_dim)
    m.train()
    opt = make_optimizer(m)
    crit = nn.CrossEntropyLoss()
    if images is None:
        images = torch.rand(batch, 3, size, size) * 2 - 1
        labels = torch.randint(0, num_classes, (batch, size, size))
    for _ in range(warmup):
        train_step(m, opt, crit, images, labels)
    best = float('inf')
    for _ in range(steps):
        t0 = time.perf_counter()
        train_step(m, opt, crit, images, labels)
        best = min(best, time.perf_counter() - t0)
    return {'value': batch / best, 'unit': 'images/sec', 'cores': threads, 'kind': 'port',
            's_per_step': best,
            'sample': f'{warmup} warm-up + best of {steps} full train step(s), UNet({num_classes},3,{conv_dim}) '
                      f'fp32 {size}x{size} bs{batch}, stock torch.nn CPU counterpart (oracle/torch_cpu.py)'}
