"""Interleaved A/B of whole train steps in ONE process: two models that differ in an engine switch.

    python tools/step_ab.py bf16 FUSE_BN_SUMS False auto        # module attribute of continual_learning_amd.unet
    python tools/step_ab.py bf16 tuning:wgrad_dma 0 1           # field of the model's clamd_tuning (model.tuning), one model per value
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import continual_learning_amd as C
from continual_learning_amd import unet as U

dtype, attr = sys.argv[1], sys.argv[2]
vals = [eval(v) if v in ('True', 'False', 'None') or v.isdigit() else v for v in sys.argv[3:]]
dev = torch.device('cuda', 0)
SIZE, BATCH = int(os.environ.get('STEP_AB_SIZE', '256')), int(os.environ.get('STEP_AB_BATCH', '16'))      # BASELINE configs[4]: 512, 32
x = torch.from_numpy(C.synth.images(1234, BATCH, 3, SIZE, SIZE)).to(dev)
y = torch.from_numpy(C.synth.labels(1234, BATCH, SIZE, SIZE, 21)).to(dev)
crit = C.CrossEntropyLoss()
runs = []
tuning = attr.startswith('tuning:')
lib = C._lib.load()
for v in vals:
    if not tuning:
        setattr(U, attr, v)
    torch.manual_seed(1234)
    m = C.UNet(21, 3, 64, compute_dtype=dtype).to(dev).train()
    if tuning:
        setattr(m.tuning, attr[7:], int(v))
    o = C.FusedAdam(m.parameters(), lr=1e-4, betas=[0.5, 0.99])

    def step(m=m, o=o):
        out = m(x); o.zero_grad(); loss = crit(out, y); loss.backward(); o.step()
        return loss
    for _ in range(3): step()          # builds the engine with this setting
    runs.append((v, step))
best = {str(v): 1e9 for v, _ in runs}
for rd in range(5):
    for v, step in runs:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): loss = step()
        torch.cuda.synchronize()
        best[str(v)] = min(best[str(v)], (time.perf_counter() - t0) / 10)
print(dtype, f'{SIZE}x{SIZE} bs{BATCH}', attr, '  '.join(f'{k}: {t * 1e3:.3f} ms/step ({BATCH / t:.1f} img/s)' for k, t in best.items()), f'loss {float(loss.detach()):.4f}')
