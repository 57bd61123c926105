"""Sanity run: 60 train steps per compute dtype on one synthetic batch (loss trend, finite outputs, pixel accuracy)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import continual_learning_amd as C
dev = torch.device('cuda', 0)
x = torch.from_numpy(C.synth.images(1234, 16, 3, 256, 256)).to(dev)
y = torch.from_numpy(C.synth.labels(1234, 16, 256, 256, 21)).to(dev)
for dt in ['fp32', 'bf16x3', 'bf16']:
    torch.manual_seed(0)
    m = C.UNet(21, 3, 64, compute_dtype=dt).to(dev).train()
    o = C.FusedAdam(m.parameters(), lr=2e-4, betas=[0.5, 0.99])
    crit = C.CrossEntropyLoss()
    ls = []
    for i in range(60):
        out = m(x); o.zero_grad(); l = crit(out, y); l.backward(); o.step()
        if i % 10 == 0 or i == 59: ls.append(round(float(l.detach()), 4))
    acc = (out.argmax(1) == y).float().mean().item()
    print(dt, ls, 'pixel acc', round(acc, 4), 'finite', bool(torch.isfinite(out).all()))
