"""Algebraic BatchNorm fold (unet.FOLD_BN_INTO_FILTERS) on / off against the reference capture of BASELINE configs[0]: logits, loss
sequence, gradient norms.  python tools/fold_parity.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import continual_learning_amd as C  # noqa: E402
from continual_learning_amd import unet as U  # noqa: E402

g = np.load(os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden', 'unet_cd64_c2_64.npz'))
rel = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b))
for dtype in ('fp32', 'bf16x3', 'bf16'):
    for fold in (False, True):
        U.FOLD_BN_INTO_FILTERS = fold
        model = C.UNet(2, 3, 64, compute_dtype=dtype)
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items() if not k.endswith(('running_mean', 'running_var', 'num_batches_tracked'))}
        sd = model.state_dict()
        sd.update({k: torch.from_numpy(v) for k, v in C.synth.closed_form_state(shapes, 0).items()})
        model.load_state_dict(sd, strict=True)
        model = model.cuda().train()
        x = torch.from_numpy(C.synth.images(1234, 2, 3, 64, 64)).cuda()
        y = torch.from_numpy(C.synth.labels(1234, 2, 64, 64, 2)).cuda()
        opt = C.FusedAdam(model.parameters(), lr=float(g['lr']), betas=[0.5, 0.99])
        crit = C.CrossEntropyLoss()
        losses = []
        for s in range(3):
            out = model(x); opt.zero_grad(); loss = crit(out, y); loss.backward()
            if s == 0:
                lg = rel(out.detach().cpu().numpy(), g['logits'])
                gn = np.array([float(p.grad.double().norm()) for p in model.parameters()])
                big = g['grad_norms'] > 1e-4 * g['grad_norms'].max()
                gr = float(np.abs(gn[big] / g['grad_norms'][big] - 1).max())
            opt.step()
            losses.append(float(loss.detach()))
        eng = next(iter(model._engines.values()))
        print(f'{dtype:7s} fold={int(fold)} ({sum(u.fold_on for u in eng.convs)} units)  logits {lg:.2e}  grad-norm max rel {gr:.2e}  loss rel '
              + ' '.join(f'{abs(a / b - 1):.1e}' for a, b in zip(losses, g['losses'])))
