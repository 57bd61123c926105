"""Two train steps (Adam, lr 1e-3: sign-like first update) on a small random configuration: gradient of the SECOND step of this path with
the algebraic BatchNorm fold on / off and of stock torch fp32 on the GPU, each against stock torch fp64 on the CPU.  Tells a wrong gradient
from the amplification of rounding-level differences by the first update.    python tests/diag/fold_two_step.py [nc cd size B] [dtype]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import continual_learning_amd as C  # noqa: E402
from continual_learning_amd import unet as U  # noqa: E402
from oracle import torch_cpu as TC  # noqa: E402

a = sys.argv[1:]
nc, cd, size, B = (int(v) for v in a[:4]) if len(a) >= 4 else (6, 16, 64, 4)
dtype = a[4] if len(a) > 4 else 'fp32'
x = torch.from_numpy(C.synth.images(3, B, 3, size, size))
y = torch.from_numpy(C.synth.labels(3, B, size, size, nc))
torch.manual_seed(0)
init = C.UNet(nc, 3, cd).state_dict()


def run_torch(dt, dev):
    m = TC.build_unet(nc, 3, cd).to(dt)
    m.load_state_dict({k: (v.to(dt) if v.is_floating_point() else v) for k, v in init.items()})
    m = m.to(dev).train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, betas=(0.5, 0.99))
    out = []
    for _ in range(2):
        opt.zero_grad()
        loss = torch.nn.CrossEntropyLoss()(m(x.to(dev, dt)), y.to(dev))
        loss.backward()
        out.append(torch.cat([p.grad.reshape(-1).double().cpu() for p in m.parameters()]))
        opt.step()
    return out


def run_ours(fold):
    U.FOLD_BN_INTO_FILTERS = fold
    m = C.UNet(nc, 3, cd, compute_dtype=dtype)
    m.load_state_dict(init)
    m = m.cuda().train()
    opt = C.FusedAdam(m.parameters(), lr=1e-3, betas=[0.5, 0.99])
    crit = C.CrossEntropyLoss()
    out = []
    for _ in range(2):
        o = m(x.cuda()); opt.zero_grad(); loss = crit(o, y.cuda()); loss.backward()
        out.append(torch.cat([p.grad.reshape(-1).double().cpu() for p in m.parameters()]))
        opt.step()
    eng = next(iter(m._engines.values()))
    return out, sum(u.fold_on for u in eng.convs)


rel = lambda p, q: float((p - q).norm() / q.norm())
r64 = run_torch(torch.float64, 'cpu')
t32 = run_torch(torch.float32, 'cuda')
print(f'config nc={nc} cd={cd} {size}x{size} B={B} {dtype}')
print(f'torch fp32 (GPU)      step 1 {rel(t32[0], r64[0]):.2e}   step 2 {rel(t32[1], r64[1]):.2e}')
for fold in (False, True):
    g, n = run_ours(fold)
    print(f'this path, fold {"on " if fold else "off"} ({n}) step 1 {rel(g[0], r64[0]):.2e}   step 2 {rel(g[1], r64[1]):.2e}')
