"""F(2x2,3x3) vs F(2x4,3x3) Winograd forward kernels on the UNet's layer shapes (fp32, bs16, 256x256 input):
time per launch and algorithmic TF/s, interleaved in one process.   python tools/wino24_ab.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import continual_learning_amd as C  # noqa: E402

lib, ptr = C._lib, C._lib.ptr
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = 16
SH = [(64, 64, 256), (64, 128, 128), (128, 128, 128), (128, 256, 64), (256, 256, 64), (256, 512, 32), (512, 512, 32), (512, 1024, 16),
      (1024, 1024, 16), (1024, 512, 16), (1024, 512, 32), (512, 256, 64), (256, 128, 128), (128, 64, 256)]
tot = {16: 0.0, 24: 0.0}
print(f'{"layer":>18s} {"F2x2 us":>9s} {"TF/s":>7s} {"F2x4 us":>9s} {"TF/s":>7s} {"speedup":>8s}')
for cin, cout, hw in SH:
    x = torch.randn(B, hw, hw, cin, device='cuda')
    w = torch.randn(cout, cin, 3, 3, device='cuda') / (3 * cin ** 0.5)
    bias = torch.zeros(cout, device='cuda')
    y = torch.empty(B, hw, hw, cout, device='cuda')
    s = lib.stream_ptr()
    res = {}
    pk = {}
    for planes in (16, 24):
        wf = torch.zeros(planes * cout * cin, device='cuda')
        tab = C.ops.WinoPackTable(planes); tab.conv3x3(w, wf, None, [(cin, cin)], cout); tab.finalize('cuda').run()
        op = lib.OP_CONV3X3_WINOGRAD if planes == 16 else lib.OP_CONV3X3_WINOGRAD24
        rows = lib.stat_rows(op, B, hw, hw, cin, cout, 0)
        pk[planes] = (wf, torch.empty(rows, 2, cout, device='cuda'), rows)
    fl = 2.0 * B * hw * hw * 9 * cin * cout
    for rnd in range(3):
        for planes in (16, 24):
            wf, st, rows = pk[planes]
            name = 'clamd_conv3x3_winograd' if planes == 16 else 'clamd_conv3x3_winograd24'
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                lib.call(name, ptr(x), cin, ptr(wf), ptr(bias), ptr(y), cout, ptr(st), rows, B, hw, hw, cin, cout, 1, None, s)
            e1.record(); e1.synchronize()
            res[planes] = e0.elapsed_time(e1) / reps * 1e3
    for pl in (16, 24):
        tot[pl] += res[pl]
    print(f'{cin:5d}->{cout:5d} @{hw:3d} {res[16]:9.1f} {fl / res[16] / 1e6:7.1f} {res[24]:9.1f} {fl / res[24] / 1e6:7.1f} {res[16] / res[24]:8.3f}')
print(f'total {tot[16] / 1e3:.3f} ms vs {tot[24] / 1e3:.3f} ms: {tot[16] / tot[24]:.3f}x')
