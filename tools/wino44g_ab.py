"""F(4x4,3x3) with pre-transformed operands (csrc/wino44g.hip) against F(2x4,3x3) with pre-transformed operands (csrc/wino24g.hip) on the
UNet's wide layer shapes (fp32, bs16, 256x256 input): input transform, forward launch, weight gradient (gradient-side transform + plane GEMM +
reduce), interleaved in one process.  `exec` = executed fraction of the fp32 MFMA peak (1/3 resp. 1/4 of the direct FLOP).
    python tools/wino44g_ab.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import continual_learning_amd as C  # noqa: E402

lib, ptr = C._lib, C._lib.ptr
L = lib.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = 16
SH = [(128, 256, 64), (256, 256, 64), (512, 256, 64), (256, 512, 32), (512, 512, 32), (1024, 512, 32), (512, 1024, 32), (512, 1024, 16), (1024, 1024, 16)]
PEAK = 157.3
_TN0, _TN2 = lib.Tuning(wgrad_streamk=0), lib.Tuning(wgrad_streamk=2)      # kept alive: ref() is the address
SK = {1: _TN2.ref(), 0: _TN0.ref()}


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print(f'{"layer":>18s} | {"xf24":>6s} {"pre24":>7s} {"exec":>5s} | {"xf44":>6s} {"TB/s":>5s} {"pre44":>7s} {"exec":>5s} | {"fwd 24/44":>9s} | {"wg24":>7s} {"wg44":>7s} {"24/44":>6s} | split-K: {"wg24":>7s} {"wg44":>7s}')
tot = dict(f24=0.0, f44=0.0, w24=0.0, w44=0.0)
for cin, cout, hw in SH:
    x = torch.randn(B, hw, hw, cin, device='cuda')
    gz = torch.randn(B, hw, hw, cout, device='cuda')
    w = torch.randn(cout, cin, 3, 3, device='cuda') / (3 * cin ** 0.5)
    bias = torch.zeros(cout, device='cuda')
    y = torch.empty(B, hw, hw, cout, device='cuda')
    s = lib.stream_ptr()
    fl = 2.0 * B * hw * hw * 9 * cin * cout
    res = {}
    K = {}
    for form, op in ((24, lib.OP_CONV3X3_WINOGRAD24), (44, lib.OP_CONV3X3_WINOGRAD44)):
        pl = 24 if form == 24 else 36
        wf = torch.zeros(pl * cout * cin, device='cuda')
        tab = C.ops.WinoPackTable(pl); tab.conv3x3(w, wf, None, [(cin, cin)], cout); tab.finalize('cuda').run()
        rows = lib.stat_rows(op, B, hw, hw, cin, cout, 0)
        st = torch.empty(rows, 2, cout, device='cuda')
        v = torch.empty(getattr(L, f'clamd_winograd{form}_input_elems')(B, hw, hw, cin), device='cuda')
        pre_wg = cout % 256 == 0 and cin % 256 == 0
        yt = ws = None
        wsb = 0
        if pre_wg:
            yt = torch.empty(getattr(L, f'clamd_wgrad_winograd{form}_pre_operand_elems')(B, hw, hw, cout), device='cuda')
            wsb = getattr(L, f'clamd_wgrad_winograd{form}_pre_workspace_bytes')(B, hw, hw, cout, cin)
            ws = torch.empty(wsb // 4 + 4, device='cuda')
        K[form] = (wf, rows, st, v, yt, ws, wsb, pre_wg)
    gw = torch.empty(cout, cin, 3, 3, device='cuda')
    for rnd in range(3):
        for form in (24, 44):
            wf, rows, st, v, yt, ws, wsb, pre_wg = K[form]
            res[f'xf{form}'] = timed(lambda: lib.call(f'clamd_winograd{form}_transform_input', ptr(x), cin, None, None, ptr(v), B, hw, hw, cin, s))
            res[f'pre{form}'] = timed(lambda: lib.call(f'clamd_conv3x3_winograd{form}_pre', ptr(v), ptr(wf), ptr(bias), ptr(y), cout, ptr(st), rows,
                                                      B, hw, hw, cin, cout, 1, None, s))
            for sk in (1, 0):        # stream-K plane GEMM (default) / split-K plan in whole rounds
                res[f'wg{form}' + ('' if sk else 's')] = timed(lambda: lib.call(f'clamd_wgrad_winograd{form}_pre', ptr(gz), cout, ptr(v), ptr(yt), ptr(ws), wsb, ptr(gw),
                                                                             B, hw, hw, cout, cin, cout, cin, cout, cout, cin, cin, SK[sk], s)) if pre_wg else float('nan')
    f24, f44 = res['xf24'] + res['pre24'], res['xf44'] + res['pre44']
    tot['f24'] += f24; tot['f44'] += f44
    if res['wg24'] == res['wg24']:
        tot['w24'] += res['wg24']; tot['w44'] += res['wg44']
    xb = (x.numel() + K[44][3].numel()) * 4
    print(f'{cin:5d}->{cout:5d} @{hw:3d} | {res["xf24"]:6.1f} {res["pre24"]:7.1f} {fl / 3 / res["pre24"] / 1e6 / PEAK:5.2f} | {res["xf44"]:6.1f} {xb / res["xf44"] / 1e6:5.2f} '
          f'{res["pre44"]:7.1f} {fl / 4 / res["pre44"] / 1e6 / PEAK:5.2f} | {f24 / f44:9.3f} | {res["wg24"]:7.1f} {res["wg44"]:7.1f} {res["wg24"] / res["wg44"]:6.3f} |          {res["wg24s"]:7.1f} {res["wg44s"]:7.1f}')
print('total: transform + forward F(2x4) %.3f ms, F(4x4) %.3f ms (%.3fx) | weight gradient F(2x4) %.3f ms, F(4x4) %.3f ms (%.3fx)' %
      (tot['f24'] / 1e3, tot['f44'] / 1e3, tot['f24'] / tot['f44'], tot['w24'] / 1e3, tot['w44'] / 1e3, tot['w24'] / max(tot['w44'], 1e-9)))
