"""Whole train step (forward + CE + backward + Adam) captured in ONE HIP graph and replayed, against the eager step:
same losses (same kernels, same order), step time of both.   python tools/graph_step.py [fp32|bf16x3|bf16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import continual_learning_amd as C

dtype = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
dev = torch.device('cuda', 0)
x = torch.from_numpy(C.synth.images(1234, 16, 3, 256, 256)).to(dev)
y = torch.from_numpy(C.synth.labels(1234, 16, 256, 256, 21)).to(dev)


def make():
    torch.manual_seed(1234)
    m = C.UNet(21, 3, 64, compute_dtype=dtype).to(dev).train()
    o = C.FusedAdam(m.parameters(), lr=1e-4, betas=[0.5, 0.99])
    return m, o, C.CrossEntropyLoss()


def timeit(f, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


m1, o1, c1 = make()
def eager():
    out = m1(x); o1.zero_grad(); loss = c1(out, y); loss.backward(); o1.step()
    return loss
ref = [float(eager().detach()) for _ in range(6)]

m2, o2, c2 = make()
from graphed_step import GraphedStep  # noqa: E402 (tools/ is sys.path[0])
step = GraphedStep(m2, o2, c2, x, y, warmup=3)
got = [float(l) for l in (step.eager_losses + [step().clone() for _ in range(3)])]
print('eager ', ref)
print('graph ', got)
print(f'{dtype}: eager {timeit(eager):.3f} ms/step, graph replay {timeit(step):.3f} ms/step')
