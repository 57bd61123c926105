"""Micro-benchmark of the conv3x3 kernels (forward implicit GEMM and weight gradient) on the UNet layer shapes.
    python tools/conv_bench.py [bf16|fp32] [iters]
Prints TFLOP/s per layer (HIP events around `iters` back-to-back launches, inputs random)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import continual_learning_amd as C
from continual_learning_amd._lib import call, ptr

dt = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dc = {'fp32': 0, 'bf16': 1, 'bf16x3': 2}[dt]
T = C.ops.TORCH_DT[dc]
B = 16
layers = [(64, 64, 256), (128, 64, 256), (128, 128, 128), (256, 128, 128), (256, 256, 64), (512, 256, 64),
          (512, 512, 32), (1024, 512, 32), (1024, 1024, 16), (512, 1024, 16)]
s = C._lib.stream_ptr()
lib = C._lib.load()
tot_f, tot_t = 0.0, 0.0
for cin, cout, hw in layers:
    x = C.ops.randn_nhwc(dc, B, hw, hw, cin)
    w = torch.randn(cout, cin, 3, 3, device='cuda') / (3 * cin ** 0.5)
    wf = torch.zeros(9 * cout * cin, dtype=T, device='cuda'); wd = torch.zeros(9 * cin * cout, dtype=T, device='cuda')
    bias = torch.zeros(cout, device='cuda')
    tab = C.ops.PackTable(dc); tab.conv3x3(w, wf, wd, [(cin, cin)], cout); tab.finalize('cuda').run(dc)
    y = torch.empty(B, hw, hw, cout, dtype=T, device='cuda')
    rows = C._lib.stat_rows(C._lib.OP_CONV3X3, B, hw, hw, cin, cout, dc)
    stats = torch.empty(rows, 2, cout, device='cuda')
    mf = 1 if 9 * cout > B * hw * hw else 0
    def fwd():
        call('clamd_conv3x3', ptr(x), cin, ptr(wf), ptr(bias), ptr(y), cout, ptr(stats), None, None, rows, B, hw, hw, cin, cout, 1, mf, dc, None, s)
    wsb = lib.clamd_wgrad_workspace_bytes(0, B, hw, hw, cout, cin, dc)
    ws = torch.empty(wsb // 4 + 4, device='cuda'); gw = torch.empty_like(w)
    g = C.ops.randn_nhwc(dc, B, hw, hw, cout)
    def wgr():
        call('clamd_wgrad', 0, ptr(g), cout, ptr(x), cin, ptr(ws), wsb, ptr(gw), B, hw, hw, cout, cin, cout, cin, cout, cout, cin, cin, dc, None, s)
    res = []
    for fn in (fwd, wgr):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / iters * 1e-3)
    fl = 2.0 * B * hw * hw * 9 * cin * cout
    tot_f += 2 * fl; tot_t += res[0] + res[1]
    print(f'{cin:5d}->{cout:5d} @{hw:3d}: fwd {res[0]*1e6:7.1f} us {fl/res[0]/1e12:7.1f} TF/s | wgrad(+reduce) {res[1]*1e6:7.1f} us {fl/res[1]/1e12:7.1f} TF/s')
print(f'{dt} aggregate {tot_f/tot_t/1e12:.1f} TF/s')
