"""Interleaved A/B of the wgrad split-K target (blocks) in one process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import continual_learning_amd as C
from continual_learning_amd._lib import call, ptr
dt = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
variants = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else '512,384,256').split(',')]
key = sys.argv[3] if len(sys.argv) > 3 else 'wgrad_blocks'        # a field of clamd_tuning, passed per call
dc = {'fp32': 0, 'bf16': 1, 'bf16x3': 2}[dt]
T = C.ops.TORCH_DT[dc]
B, iters, rounds = 16, 10, 4
layers = [(64, 64, 256), (128, 64, 256), (128, 128, 128), (256, 128, 128), (256, 256, 64), (512, 512, 32), (1024, 512, 32), (1024, 1024, 16)]
lib = C._lib.load(); s = C._lib.stream_ptr()
tot = {v: [0.0, 0.0] for v in variants}
for cin, cout, hw in layers:
    x = C.ops.randn_nhwc(dc, B, hw, hw, cin)
    g = C.ops.randn_nhwc(dc, B, hw, hw, cout)
    wsb = lib.clamd_wgrad_workspace_bytes(0, B, hw, hw, cout, cin, dc)
    ws = torch.empty(wsb // 4 + 4, device='cuda'); gw = torch.empty(cout, cin, 3, 3, device='cuda')
    best = {v: 1e9 for v in variants}
    tun = {v: C._lib.Tuning(**{key: v}) for v in variants}
    def run(v):
        call('clamd_wgrad', 0, ptr(g), cout, ptr(x), cin, ptr(ws), wsb, ptr(gw), B, hw, hw, cout, cin, cout, cin, cout, cout, cin, cin, dc, tun[v].ref(), s)
    ref = None
    for rd in range(rounds):
        for v in variants:
            run(v)
            if rd == 0:      # variants may differ in split-K / summation order, never by more than rounding
                torch.cuda.synchronize()
                if ref is None: ref = gw.clone()
                else:
                    err = ((gw - ref).norm() / ref.norm()).item()
                    assert err < 2e-3, f'variant {v}: rel diff {err}'
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters): run(v)
            e1.record(); torch.cuda.synchronize()
            best[v] = min(best[v], e0.elapsed_time(e1) / iters * 1e-3)
    fl = 2.0 * B * hw * hw * 9 * cin * cout
    print(f'{cin:5d}x{cout:5d} @{hw:3d}: ' + '  '.join(f'b{v} {best[v]*1e6:7.1f}us {fl/best[v]/1e12:7.1f}TF' for v in variants))
    for v in variants: tot[v][0] += fl; tot[v][1] += best[v]
print(dt, 'wgrad+reduce aggregate: ' + '  '.join(f'b{v} {tot[v][0]/tot[v][1]/1e12:.1f} TF/s' for v in variants))
