"""Host side of the eager train step: how long Python + ctypes + the HIP runtime need to ENQUEUE one step (the GPU idle at the start of
each step, so nothing blocks on a full queue) against the time the GPU needs to run it.   python tools/host_rate.py [dtype] [steps]
A step whose enqueue time approaches its GPU time is launch-bound in the eager loop (bench.py's); tools/graphed_step.py replays it from a hipGraph."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import continual_learning_amd as C  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = C.UNet(21, 3, 64, compute_dtype=dtype).to(dev).train()
opt = C.FusedAdam(model.parameters(), lr=1e-4, betas=[0.5, 0.99])
crit = C.CrossEntropyLoss()
x = torch.from_numpy(C.synth.images(1234, 16, 3, 256, 256)).to(dev)
y = torch.from_numpy(C.synth.labels(1234, 16, 256, 256, 21)).to(dev)


def step(parts=None):
    t = [time.perf_counter()]
    out = model(x); t.append(time.perf_counter())
    opt.zero_grad(); t.append(time.perf_counter())
    loss = crit(out, y); t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    opt.step(); t.append(time.perf_counter())
    if parts is not None:
        parts.append([b - a for a, b in zip(t, t[1:])])
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
host, total, parts = [], [], []
for _ in range(steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step(parts)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append(t1 - t0); total.append(t2 - t0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
free = (time.perf_counter() - t0) / steps
med = lambda v: sorted(v)[len(v) // 2]
names = ['forward', 'zero_grad', 'loss', 'backward', 'adam']
print(f'{dtype}: enqueue {med(host) * 1e3:.3f} ms per step (median; ' + ', '.join(f'{n} {med([p[i] for p in parts]) * 1e3:.3f}' for i, n in enumerate(names)) +
      f'); step from an idle GPU {med(total) * 1e3:.3f} ms; free-running {free * 1e3:.3f} ms per step')

# free-running: where does the host spend a step when nothing synchronises explicitly?  (a part that takes as long as the GPU step is where
# the runtime throttles the host -- the GPU then starts that part's kernels late)
parts = []
torch.cuda.synchronize()
for _ in range(steps):
    step(parts)
torch.cuda.synchronize()
tail = parts[len(parts) // 2:]
print(f'{dtype}: free-running host time per step ' + ', '.join(f'{n} {med([p[i] for p in tail]) * 1e3:.3f}' for i, n in enumerate(names)) +
      f'; sum {sum(med([p[i] for p in tail]) for i in range(5)) * 1e3:.3f} ms')
