"""Pre-transformed F(2x4,3x3) kernels (csrc/wino24g.hip) against the in-kernel-transform ones on the UNet's wide layer shapes
(fp32, bs16, 256x256 input): forward launch, input transform, weight gradient; interleaved in one process.
    python tools/wino24g_ab.py [reps] [min_channels]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import continual_learning_amd as C  # noqa: E402

lib, ptr = C._lib, C._lib.ptr
L = lib.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
minc = int(sys.argv[2]) if len(sys.argv) > 2 else 256
B = 16
SH = [(64, 64, 256), (128, 128, 128), (128, 256, 64), (256, 256, 64), (256, 512, 32), (512, 512, 32), (512, 1024, 16), (1024, 1024, 16),
      (1024, 512, 32), (512, 256, 64), (256, 128, 128)]
PEAK = 157.3


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print(f'{"layer":>18s} | {"w24 us":>8s} {"exec":>5s} | {"xform":>7s} {"TB/s":>5s} {"pre us":>8s} {"exec":>5s} {"x+pre":>7s} | {"wg us":>8s} {"exec":>5s} {"wg_pre":>8s} {"exec":>5s} (xf {"":>3s})')
tot = dict(w24=0.0, xf=0.0, pre=0.0, wg=0.0, wgp=0.0)
for cin, cout, hw in SH:
    if cin < minc:
        continue
    x = torch.randn(B, hw, hw, cin, device='cuda')
    gz = torch.randn(B, hw, hw, cout, device='cuda')
    w = torch.randn(cout, cin, 3, 3, device='cuda') / (3 * cin ** 0.5)
    bias = torch.zeros(cout, device='cuda')
    y = torch.empty(B, hw, hw, cout, device='cuda')
    s = lib.stream_ptr()
    wf = torch.zeros(24 * cout * cin, device='cuda')
    tab = C.ops.WinoPackTable(24); tab.conv3x3(w, wf, None, [(cin, cin)], cout); tab.finalize('cuda').run()
    rows = lib.stat_rows(lib.OP_CONV3X3_WINOGRAD24, B, hw, hw, cin, cout, 0)
    st = torch.empty(rows, 2, cout, device='cuda')
    v = torch.empty(L.clamd_winograd24_input_elems(B, hw, hw, cin), device='cuda')
    fl = 2.0 * B * hw * hw * 9 * cin * cout
    res = {}
    use24 = hw >= 64          # the shipped choice for the weight gradient ('auto')
    wg_name = 'clamd_wgrad_winograd24' if use24 else 'clamd_wgrad_winograd'
    wsb = max(L.clamd_wgrad_winograd24_workspace_bytes(cout, cin), L.clamd_wgrad_winograd_workspace_bytes(cout, cin),
              L.clamd_wgrad_winograd24_pre_workspace_bytes(B, hw, hw, cout, cin) if cout % 256 == 0 and cin % 256 == 0 else 0)
    ws = torch.empty(wsb // 4 + 4, device='cuda')
    gw = torch.empty(cout, cin, 3, 3, device='cuda')
    pre_wg = cout % 256 == 0 and cin % 256 == 0
    if pre_wg:
        yt = torch.empty(L.clamd_wgrad_winograd24_pre_operand_elems(B, hw, hw, cout), device='cuda')
    for rnd in range(3):
        res['w24'] = timed(lambda: lib.call('clamd_conv3x3_winograd24', ptr(x), cin, ptr(wf), ptr(bias), ptr(y), cout, ptr(st), rows, B, hw, hw, cin, cout, 1, None, s))
        res['xf'] = timed(lambda: lib.call('clamd_winograd24_transform_input', ptr(x), cin, None, None, ptr(v), B, hw, hw, cin, s))
        res['pre'] = timed(lambda: lib.call('clamd_conv3x3_winograd24_pre', ptr(v), ptr(wf), ptr(bias), ptr(y), cout, ptr(st), rows, B, hw, hw, cin, cout, 1, None, s))
        res['wg'] = timed(lambda: lib.call(wg_name, ptr(gz), cout, ptr(x), cin, ptr(ws), wsb, ptr(gw), B, hw, hw, cout, cin, cout, cin, cout, cout, cin, cin, None, s))
        if pre_wg:
            res['wgp'] = timed(lambda: lib.call('clamd_wgrad_winograd24_pre', ptr(gz), cout, ptr(v), ptr(yt), ptr(ws), wsb, ptr(gw), B, hw, hw,
                                                cout, cin, cout, cin, cout, cout, cin, cin, None, s))
        else:
            res['wgp'] = float('nan')
    for k in tot:
        tot[k] += res[k] if res[k] == res[k] else res['wg']
    ex = lambda us, frac: fl * frac / us / 1e6 / PEAK          # executed fraction of the fp32 MFMA peak
    xbytes = (x.numel() + v.numel()) * 4
    print(f'{cin:5d}->{cout:5d} @{hw:3d} | {res["w24"]:8.1f} {ex(res["w24"], 1 / 3):5.2f} | {res["xf"]:7.1f} {xbytes / res["xf"] / 1e6:5.2f} {res["pre"]:8.1f} {ex(res["pre"], 1 / 3):5.2f} '
          f'{res["xf"] + res["pre"]:7.1f} | {res["wg"]:8.1f} {ex(res["wg"], 1 / 3 if use24 else 4 / 9):5.2f} {res["wgp"]:8.1f} {ex(res["wgp"], 1 / 3):5.2f}')
print('total: w24 %.3f ms | xform %.3f + pre %.3f = %.3f ms | wgrad %.3f ms -> pre %.3f ms' %
      (tot['w24'] / 1e3, tot['xf'] / 1e3, tot['pre'] / 1e3, (tot['xf'] + tot['pre']) / 1e3, tot['wg'] / 1e3, tot['wgp'] / 1e3))
