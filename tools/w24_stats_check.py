"""Statistics rows of the F(2x4,3x3) kernels (clamd_conv3x3_winograd24 and its pre-transformed / direct-filter forms) against sums taken from the
activations they wrote: one layer shape, fp32.   python tools/w24_stats_check.py"""
import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import continual_learning_amd as C
from continual_learning_amd._lib import call, ptr
L = C._lib; s = L.stream_ptr()
torch.manual_seed(0)
for B, hw, cin, cout in [(4, 128, 32, 32), (4, 64, 32, 32), (4, 64, 64, 64), (4, 32, 64, 64), (4, 16, 128, 128), (4, 8, 256, 256), (16, 256, 64, 64)]:
    x = torch.randn(B, hw, hw, cin, device='cuda'); w = torch.randn(cout, cin, 3, 3, device='cuda') / (3 * cin ** 0.5)
    bias = torch.randn(cout, device='cuda')
    wf = torch.zeros(24 * cout * cin, device='cuda')
    tab = C.ops.WinoPackTable(24); tab.conv3x3(w, wf, None, [(cin, cin)], cout); tab.finalize('cuda').run()
    res = {}
    for name, tn in [('default', L.Tuning()), ('reserve24', L.Tuning(cu_reserve=24)), ('pertile', L.Tuning(wino_persist=0)), ('band1', L.Tuning(wino_band=1))]:
        rows = L.stat_rows(L.OP_CONV3X3_WINOGRAD24, B, hw, hw, cin, cout, 0, tuning=tn)
        st = torch.full((rows, 2, cout), 7.0, device='cuda'); y = torch.empty(B, hw, hw, cout, device='cuda')
        call('clamd_conv3x3_winograd24', ptr(x), cin, ptr(wf), ptr(bias), ptr(y), cout, ptr(st), rows, B, hw, hw, cin, cout, 1, tn.ref(), s)
        torch.cuda.synchronize()
        res[name] = (st.double().sum(0), y.double().sum((0, 1, 2)), (y.double() ** 2).sum((0, 1, 2)), rows)
    for name, (st, s1, s2, rows) in res.items():
        e1 = float((st[0] - s1).abs().max() / s1.abs().max()); e2 = float((st[1] - s2).abs().max() / s2.abs().max())
        print(f'B{B} {hw}^2 {cin}->{cout} {name:10s} rows {rows:5d}: sum err {e1:.2e}  sumsq err {e2:.2e}')
