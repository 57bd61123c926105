"""wino24_kernel (filters staged through LDS) against wino24h_kernel (filters straight into the operand registers) on the
narrow layer shapes (fp32, bs16, 256x256 input), interleaved in one process.   python tools/wino24h_ab.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import continual_learning_amd as C  # noqa: E402

lib, ptr = C._lib, C._lib.ptr
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = 16
SH = [(64, 64, 256), (64, 128, 128), (128, 128, 128), (128, 256, 64), (256, 128, 128), (128, 64, 256), (256, 256, 64), (512, 512, 32)]
tot = {'lds': 0.0, 'direct': 0.0}
print(f'{"layer":>18s} {"lds us":>9s} {"exec":>6s} {"direct us":>10s} {"exec":>6s} {"speedup":>8s}')
for cin, cout, hw in SH:
    x = torch.randn(B, hw, hw, cin, device='cuda')
    w = torch.randn(cout, cin, 3, 3, device='cuda') / (3 * cin ** 0.5)
    bias = torch.zeros(cout, device='cuda')
    y = torch.empty(B, hw, hw, cout, device='cuda')
    s = lib.stream_ptr()
    wf = torch.zeros(24 * cout * cin, device='cuda')
    tab = C.ops.WinoPackTable(24); tab.conv3x3(w, wf, None, [(cin, cin)], cout); tab.finalize('cuda').run()
    rows = lib.stat_rows(lib.OP_CONV3X3_WINOGRAD24, B, hw, hw, cin, cout, 0)
    st = torch.empty(rows, 2, cout, device='cuda')
    fl = 2.0 * B * hw * hw * 9 * cin * cout / 3
    res = {}
    for rnd in range(3):
        for key, name in (('lds', 'clamd_conv3x3_winograd24'), ('direct', 'clamd_conv3x3_winograd24_direct_filters')):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                extra = ()
                lib.call(name, ptr(x), cin, ptr(wf), ptr(bias), ptr(y), cout, ptr(st), rows, *extra, B, hw, hw, cin, cout, 1, None, s)
            e1.record(); e1.synchronize()
            res[key] = e0.elapsed_time(e1) / reps * 1e3
    for k in tot:
        tot[k] += res[k]
    print(f'{cin:5d}->{cout:5d} @{hw:3d} {res["lds"]:9.1f} {fl / res["lds"] / 1e6 / 157.3:6.3f} {res["direct"]:10.1f} {fl / res["direct"] / 1e6 / 157.3:6.3f} {res["lds"] / res["direct"]:8.3f}')
print(f'total {tot["lds"] / 1e3:.3f} ms vs {tot["direct"] / 1e3:.3f} ms: {tot["lds"] / tot["direct"]:.3f}x')
