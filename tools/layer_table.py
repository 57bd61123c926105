"""Per-layer table of the 3x3 convolution launches of one train step (HIP events around every launch):

    python tools/layer_table.py [dtype=fp32] [size=256] [batch=16]

unit, direction, time, algorithmic TF/s, fraction of the MFMA peak (fp32 Winograd: executed 16/36 of the algorithmic FLOPs).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import continual_learning_amd as C  # noqa: E402
from continual_learning_amd import unet as U  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 16
for kv in sys.argv[4:]:
    k, v = kv.split('=')
    setattr(U, k, eval(v))
PEAK = {'fp32': 157.3, 'bf16': 2500.0, 'bf16x3': 2500.0 / 3}[dtype]
dev = torch.device('cuda', 0)
torch.manual_seed(0)
m = C.UNet(21, 3, 64, compute_dtype=dtype).to(dev).train()
opt = C.FusedAdam(m.parameters(), lr=1e-4, betas=[0.5, 0.99])
crit = C.CrossEntropyLoss()
x = torch.from_numpy(C.synth.images(1234, batch, 3, size, size)).to(dev)
y = torch.from_numpy(C.synth.labels(1234, batch, size, size, 21)).to(dev)


def step():
    o = m(x); opt.zero_grad(); l = crit(o, y); l.backward(); opt.step()


for _ in range(4):
    step()
torch.cuda.synchronize()
acc = {}
for _ in range(3):
    ev = []
    U.KERNEL_TIMING = ev
    step()
    torch.cuda.synchronize()
    U.KERNEL_TIMING = None
    for tag, flops, e0, e1, nbytes, unit, frac in ev:
        if tag.startswith('hbm:'):              # the HBM-bound families bench.py reports as `hbm_kernels`: not rows of this table
            continue
        xf = tag == 'wino_transform'            # the input transform of a pre-transformed launch: its own (HBM-bound) row
        a = acc.setdefault(unit + (' xform' if xf else ''), [0.0, nbytes if xf else flops, 0, frac])
        a[0] += e0.elapsed_time(e1) * 1e-3
        a[2] += 1
eng = next(iter(m._engines.values()))
tot = {}
print(f'{"unit":22s} {"us":>8s} {"alg TF/s":>9s} {"of peak":>8s}')
for unit, (sec, flops, n, ex) in acc.items():
    t = sec / n
    u = [c for c in eng.convs if unit.startswith(c.name + ' ')][0]
    if unit.endswith(' xform'):
        print(f'{unit:22s} {t * 1e6:8.1f} {flops / t / 1e12:6.2f} TB/s          transform of the launch below')
        tt = tot.setdefault('xform', [0.0, 0.0])
        tt[0] += t
        continue
    tf = flops / t / 1e12
    pre = {'fwd': u.pre_f, 'dgrad': u.pre_d, 'wgrad': u.pre_w}.get(unit.split()[-1], False)
    print(f'{unit:22s} {t * 1e6:8.1f} {tf:9.1f} {tf * ex / PEAK:8.3f}   {u.cin}->{u.cout} @{u.h}x{u.w_}{"  pre-transformed" if pre else ""}')
    d = unit.split()[-1]
    tt = tot.setdefault(d, [0.0, 0.0])
    tt[0] += t; tt[1] += flops
for d, (t, f) in tot.items():
    print(f'total {d:6s} {t * 1e3:7.3f} ms' + (f'  {f / t / 1e12:7.1f} alg TF/s' if f else ''))
