"""fp32 3x3 weight gradient: direct (clamd_wgrad) vs Winograd (clamd_wgrad_winograd), interleaved, UNet layer shapes.
TF/s are ALGORITHMIC (direct FLOPs / time, split-K reduce included)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import continual_learning_amd as C
from continual_learning_amd._lib import call, ptr
B, iters, rounds = 16, 5, 4
layers = [(64, 64, 256), (128, 64, 256), (128, 128, 128), (256, 128, 128), (256, 256, 64), (512, 256, 64), (512, 512, 32), (1024, 512, 32), (1024, 1024, 16), (512, 1024, 16)]
lib = C._lib.load(); s = C._lib.stream_ptr()
tot = [0.0, 0.0, 0.0]
for cin, cout, hw in layers:
    x = torch.randn(B, hw, hw, cin, device='cuda'); g = torch.randn(B, hw, hw, cout, device='cuda')
    wsb = max(lib.clamd_wgrad_workspace_bytes(0, B, hw, hw, cout, cin, 0), lib.clamd_wgrad_winograd_workspace_bytes(cout, cin))
    ws = torch.empty(wsb // 4 + 4, device='cuda'); g1 = torch.empty(cout, cin, 3, 3, device='cuda'); g2 = torch.empty_like(g1)
    def direct(): call('clamd_wgrad', 0, ptr(g), cout, ptr(x), cin, ptr(ws), wsb, ptr(g1), B, hw, hw, cout, cin, cout, cin, cout, cout, cin, cin, 0, None, s)
    def wino(): call('clamd_wgrad_winograd', ptr(g), cout, ptr(x), cin, ptr(ws), wsb, ptr(g2), B, hw, hw, cout, cin, cout, cin, cout, cout, cin, cin, None, s)
    best = [1e9, 1e9]
    for rd in range(rounds):
        for i, f in enumerate((direct, wino)):
            f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters): f()
            e1.record(); torch.cuda.synchronize()
            best[i] = min(best[i], e0.elapsed_time(e1) / iters * 1e-3)
    err = ((g1 - g2).norm() / g1.norm()).item()
    fl = 2.0 * B * hw * hw * 9 * cin * cout
    print(f'{cin:5d}x{cout:5d} @{hw:3d}: direct {best[0]*1e6:7.1f}us {fl/best[0]/1e12:6.1f}TF   winograd {best[1]*1e6:7.1f}us {fl/best[1]/1e12:6.1f}TF  x{best[0]/best[1]:.2f}  rel diff {err:.1e}')
    tot[0] += fl; tot[1] += best[0]; tot[2] += best[1]
print(f'aggregate: direct {tot[0]/tot[1]/1e12:.1f} TF/s, winograd {tot[0]/tot[2]/1e12:.1f} TF/s (algorithmic)')
