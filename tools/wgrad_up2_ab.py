"""Interleaved A/B of the ConvTranspose2d weight-gradient kernel (WG_UP2) on the four decoder shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import continual_learning_amd as C
from continual_learning_amd._lib import call, ptr
dt = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
variants = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1").split(",")]
key = sys.argv[3] if len(sys.argv) > 3 else 'wgrad_xcd'            # a field of clamd_tuning, passed per call
dc = {'fp32': 0, 'bf16': 1, 'bf16x3': 2}[dt]
T = C.ops.TORCH_DT[dc]
B, iters, rounds = 16, 10, 4
lib = C._lib.load(); s = C._lib.stream_ptr()
for cin, cout, hw in [(1024, 512, 16), (512, 256, 32), (256, 128, 64), (128, 64, 128)]:
    x = C.ops.randn_nhwc(dc, B, hw, hw, cin)
    gy = C.ops.randn_nhwc(dc, B, 2 * hw, 2 * hw, cout)
    wsb = lib.clamd_wgrad_workspace_bytes(2, B, hw, hw, cin, cout, dc)
    ws = torch.empty(wsb // 4 + 4, device='cuda'); gw = torch.empty(cin, cout, 2, 2, device='cuda')
    tun = {v: C._lib.Tuning(**{key: v}) for v in variants}
    def run(v): call('clamd_wgrad', 2, ptr(x), cin, ptr(gy), cout, ptr(ws), wsb, ptr(gw), B, hw, hw, cin, cout, cin, cout, cin, cin, cout, cout, dc, tun[v].ref(), s)
    best = {v: 1e9 for v in variants}; ref = None
    for rd in range(rounds):
        for v in variants:
            run(v)
            if rd == 0:
                torch.cuda.synchronize()
                if ref is None: ref = gw.clone()
                else: assert ((gw - ref).norm() / ref.norm()).item() < 2e-3
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters): run(v)
            e1.record(); torch.cuda.synchronize()
            best[v] = min(best[v], e0.elapsed_time(e1) / iters * 1e-3)
    fl = 2.0 * B * hw * hw * cin * 4 * cout
    print(f'{cin:5d}->{cout:4d} @{hw:3d}: ' + '  '.join(f'v{v} {best[v]*1e6:7.1f}us {fl/best[v]/1e12:6.1f}TF' for v in variants))
