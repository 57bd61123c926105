"""fp32 3x3 convolution forward: direct implicit GEMM (clamd_conv3x3) vs Winograd F(2x2,3x3) (clamd_conv3x3_winograd),
interleaved in one process, UNet layer shapes.  TF/s are ALGORITHMIC (direct-convolution FLOPs / time)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import continual_learning_amd as C
from continual_learning_amd._lib import call, ptr
B, iters, rounds = 16, 5, 4
layers = [(64, 64, 256), (128, 64, 256), (64, 128, 128), (128, 128, 128), (256, 128, 128), (128, 256, 64), (256, 256, 64),
          (512, 256, 64), (512, 512, 32), (1024, 512, 32), (1024, 1024, 16), (512, 1024, 16)]
if os.environ.get('CONV_LAYERS'):
    layers = [tuple(int(v) for v in l.split(',')) for l in os.environ['CONV_LAYERS'].split(';')]
s = C._lib.stream_ptr()
tn = C._lib.Tuning(**{kv.split('=')[0]: int(kv.split('=')[1]) for kv in filter(None, os.environ.get('TUNING', '').split(','))})   # TUNING=wino_band=4,wino_persist=0
tot = [0.0, 0.0, 0.0]
for cin, cout, hw in layers:
    x = torch.randn(B, hw, hw, cin, device='cuda')
    w = torch.randn(cout, cin, 3, 3, device='cuda') / (3 * cin ** 0.5)
    wf = torch.zeros(9 * cout * cin, device='cuda'); ww = torch.zeros(16 * cout * cin, device='cuda')
    bias = torch.zeros(cout, device='cuda')
    t1 = C.ops.PackTable(0); t1.conv3x3(w, wf, None, [(cin, cin)], cout); t1.finalize('cuda').run(0)
    t2 = C.ops.WinoPackTable(); t2.conv3x3(w, ww, None, [(cin, cin)], cout); t2.finalize('cuda').run()
    y1 = torch.empty(B, hw, hw, cout, device='cuda'); y2 = torch.empty_like(y1)
    r1 = C._lib.stat_rows(C._lib.OP_CONV3X3, B, hw, hw, cin, cout, 0, tuning=tn); r2 = C._lib.stat_rows(C._lib.OP_CONV3X3_WINOGRAD, B, hw, hw, cin, cout, 0, tuning=tn)
    stats = torch.empty(max(r1, r2), 2, cout, device='cuda')
    def direct(): call('clamd_conv3x3', ptr(x), cin, ptr(wf), ptr(bias), ptr(y1), cout, ptr(stats), None, None, r1, B, hw, hw, cin, cout, 1, 0, 0, tn.ref(), s)
    def wino(): call('clamd_conv3x3_winograd', ptr(x), cin, ptr(ww), ptr(bias), ptr(y2), cout, ptr(stats), r2, B, hw, hw, cin, cout, 1, tn.ref(), s)
    best = [1e9, 1e9]
    for rd in range(rounds):
        for i, f in enumerate((direct, wino)):
            f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters): f()
            e1.record(); torch.cuda.synchronize()
            best[i] = min(best[i], e0.elapsed_time(e1) / iters * 1e-3)
    err = ((y1 - y2).norm() / y1.norm()).item()
    fl = 2.0 * B * hw * hw * 9 * cin * cout
    print(f'{cin:5d}->{cout:5d} @{hw:3d}: direct {best[0]*1e6:7.1f}us {fl/best[0]/1e12:6.1f}TF   winograd {best[1]*1e6:7.1f}us {fl/best[1]/1e12:6.1f}TF  x{best[0]/best[1]:.2f}  rel diff {err:.1e}')
    tot[0] += fl; tot[1] += best[0]; tot[2] += best[1]
print(f'aggregate: direct {tot[0]/tot[1]/1e12:.1f} TF/s, winograd {tot[0]/tot[2]/1e12:.1f} TF/s (algorithmic)')
