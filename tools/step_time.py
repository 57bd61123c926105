"""One process, one build: best-of-5 x 10 train steps of BASELINE configs[1] (or another dtype).  For A/Bs between BUILDS (CLAMD_LIB=build/<variant>/libclamd.so),
run alternately in one gpurun call:   python tools/step_time.py [dtype]"""
import os, sys, time
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/bench.py') else os.environ.get('GRAFT_REPO_ROOT', '.'))
import torch
import continual_learning_amd as C
dtype = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
dev = torch.device('cuda', 0)
x = torch.from_numpy(C.synth.images(1234, 16, 3, 256, 256)).to(dev)
y = torch.from_numpy(C.synth.labels(1234, 16, 256, 256, 21)).to(dev)
torch.manual_seed(1234)
m = C.UNet(21, 3, 64, compute_dtype=dtype).to(dev).train()
o = C.FusedAdam(m.parameters(), lr=1e-4, betas=[0.5, 0.99]); crit = C.CrossEntropyLoss()
def step():
    out = m(x); o.zero_grad(); l = crit(out, y); l.backward(); o.step(); return l
for _ in range(5): step()
best = 1e9
for r in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): l = step()
    torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 10)
print(f'{dtype} {os.environ.get("CLAMD_LIB", "default")}: {best * 1e3:.3f} ms/step ({16 / best:.1f} img/s) loss {float(l):.4f}')
