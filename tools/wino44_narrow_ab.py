"""Where does the pre-transformed F(4x4,3x3) path (transform + transform-free kernel, wino44g.hip) beat the in-kernel-transform F(2x4,3x3)
kernels (wino24_kernel / wino24h_kernel) on the NARROW layer shapes of the fp32 step (bs16, 256x256 input)?  Forward-form launches; a data
gradient is the same launch with the channel counts exchanged.  Interleaved in one process.   python tools/wino44_narrow_ab.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import continual_learning_amd as C  # noqa: E402

lib, ptr = C._lib, C._lib.ptr
L = lib.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = 16
SH = [(64, 64, 256), (128, 64, 256), (64, 128, 256), (64, 128, 128), (128, 128, 128), (256, 128, 128), (128, 256, 128), (256, 128, 64), (128, 256, 64)]


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print(f'{"K -> N @ size":>18s} | {"in-kernel F(2x4)":>17s} | {"xf44":>7s} {"pre44":>7s} {"sum":>7s} | in-kernel / pre-transformed F(4x4)')
for cin, cout, hw in SH:
    x = torch.randn(B, hw, hw, cin, device='cuda')
    w = torch.randn(cout, cin, 3, 3, device='cuda') / (3 * cin ** 0.5)
    y = torch.empty(B, hw, hw, cout, device='cuda')
    s = lib.stream_ptr()
    wf24 = torch.zeros(24 * cout * cin, device='cuda')
    wf44 = torch.zeros(36 * cout * cin, device='cuda')
    for pl, wf in ((24, wf24), (36, wf44)):
        tab = C.ops.WinoPackTable(pl); tab.conv3x3(w, wf, None, [(cin, cin)], cout); tab.finalize('cuda').run()
    v = torch.empty(L.clamd_winograd44_input_elems(B, hw, hw, cin), device='cuda')
    name24 = 'clamd_conv3x3_winograd24_direct_filters' if cin == 64 else 'clamd_conv3x3_winograd24'
    r = {}
    for rnd in range(3):
        r['k24'] = timed(lambda: lib.call(name24, ptr(x), cin, ptr(wf24), None, ptr(y), cout, None, 0, B, hw, hw, cin, cout, 0, None, s))
        r['xf'] = timed(lambda: lib.call('clamd_winograd44_transform_input', ptr(x), cin, None, None, ptr(v), B, hw, hw, cin, s))
        r['pre'] = timed(lambda: lib.call('clamd_conv3x3_winograd44_pre', ptr(v), ptr(wf44), None, ptr(y), cout, None, 0, B, hw, hw, cin, cout, 0, None, s))
    print(f'{cin:5d}->{cout:5d} @{hw:3d} | {r["k24"]:17.1f} | {r["xf"]:7.1f} {r["pre"]:7.1f} {r["xf"] + r["pre"]:7.1f} | {r["k24"] / (r["xf"] + r["pre"]):.3f}')
