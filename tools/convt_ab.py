"""Interleaved A/B of the ConvTranspose2d(k2,s2) kernels (forward GEMM + pixel-shuffle store, data gradient) on the
four UNet decoder shapes.     python tools/convt_ab.py fp32 0,1   (the variants are repeat labels: these kernels have no structure knob)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import continual_learning_amd as C
from continual_learning_amd._lib import call, ptr
dt = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
variants = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else '0').split(',')]
dc = {'fp32': 0, 'bf16': 1, 'bf16x3': 2}[dt]
T = C.ops.TORCH_DT[dc]
B, iters, rounds = 16, 10, 4
lib = C._lib.load(); s = C._lib.stream_ptr()
tot = {v: [0.0, 0.0] for v in variants}
for cin, cout, hw in [(1024, 512, 16), (512, 256, 32), (256, 128, 64), (128, 64, 128)]:
    x = C.ops.randn_nhwc(dc, B, hw, hw, cin)
    w = torch.randn(cin, cout, 2, 2, device='cuda') / (cin ** 0.5)
    wf = torch.zeros(4 * cout * cin, dtype=T, device='cuda'); wd = torch.zeros(cin * 4 * cout, dtype=T, device='cuda')
    bias = torch.zeros(cout, device='cuda')
    tab = C.ops.PackTable(dc); tab.convT(w, wf, wd, cin, cout); tab.finalize('cuda').run(dc)
    y = torch.empty(B, 2 * hw, 2 * hw, cout, dtype=T, device='cuda')
    gx = torch.empty(B, hw, hw, cin, dtype=T, device='cuda')
    def fwd(): call('clamd_convT2x2_fwd', ptr(x), cin, ptr(wf), ptr(bias), ptr(y), cout, B, hw, hw, cin, cout, dc, s)
    def bwd(): call('clamd_convT2x2_dgrad', ptr(y), cout, ptr(wd), ptr(gx), cin, None, None, 0, B, hw, hw, cin, cout, dc, s)
    best = {(v, n): 1e9 for v in variants for n in 'fb'}
    for rd in range(rounds):
        for v in variants:
            for n, f in (('f', fwd), ('b', bwd)):
                f()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters): f()
                e1.record(); torch.cuda.synchronize()
                best[(v, n)] = min(best[(v, n)], e0.elapsed_time(e1) / iters * 1e-3)
    fl = 2.0 * B * hw * hw * cin * 4 * cout
    print(f'{cin:5d}->{cout:4d} @{hw:3d}: ' + '  '.join(f'v{v} fwd {best[(v,"f")]*1e6:7.1f}us {fl/best[(v,"f")]/1e12:6.1f}TF dgrad {best[(v,"b")]*1e6:7.1f}us {fl/best[(v,"b")]/1e12:6.1f}TF' for v in variants))
