"""Border-class bias epilogue (relu flag CLAMD_BIAS_BORDER_CLASSES, bnfold.hip) against the plain bias on the folded layer shapes: the same
launch with relu = 1 and relu = 3, interleaved.   python tools/cls_ab.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import continual_learning_amd as C  # noqa: E402

lib, ptr = C._lib, C._lib.ptr
L = lib.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = 16


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for dcode, name in ((0, 'fp32'), (1, 'bf16'), (2, 'bf16x3')):
    for cin, cout, hw in ((64, 64, 256), (128, 128, 128)):
        T = C.ops.TORCH_DT[dcode]
        x = C.ops.randn_nhwc(dcode, B, hw, hw, cin).abs_() if dcode != 2 else C.ops.split_encode(torch.randn(B, hw, hw, cin, device='cuda').abs_())
        w = torch.randn(cout, cin, 3, 3, device='cuda') / (3 * cin ** 0.5)
        table = torch.randn(9, cout, device='cuda')
        y = torch.empty(B, hw, hw, cout, dtype=T, device='cuda')
        s = lib.stream_ptr()
        res = {}
        if dcode == 0:
            wf = torch.zeros(24 * cout * cin, device='cuda')
            tab = C.ops.WinoPackTable(24); tab.conv3x3(w, wf, None, [(cin, cin)], cout); tab.finalize('cuda').run()
            rows = lib.stat_rows(lib.OP_CONV3X3_WINOGRAD24, B, hw, hw, cin, cout, 0)
            st = torch.empty(rows, 2, cout, device='cuda')
            fn = 'clamd_conv3x3_winograd24_direct_filters' if cin == 64 else 'clamd_conv3x3_winograd24'
            extra = ()
            run = lambda fl: lib.call(fn, ptr(x), cin, ptr(wf), ptr(table), ptr(y), cout, ptr(st), rows, *extra, B, hw, hw, cin, cout, fl, None, s)
        else:
            wf = torch.zeros(9 * cout * cin, dtype=T, device='cuda')
            tab = C.ops.PackTable(dcode); tab.conv3x3(w, wf, None, [(cin, cin)], cout); tab.finalize('cuda').run(dcode)
            rows = lib.stat_rows(lib.OP_CONV3X3, B, hw, hw, cin, cout, dcode)
            st = torch.empty(rows, 2, cout, device='cuda')
            run = lambda fl: lib.call('clamd_conv3x3', ptr(x), cin, ptr(wf), ptr(table), ptr(y), cout, ptr(st), None, None, rows, B, hw, hw, cin, cout,
                                      fl, 0, dcode, None, s)
        for rnd in range(3):
            res['plain'] = timed(lambda: run(1))
            res['cls'] = timed(lambda: run(3))
        print(f'{name:7s} {cin:4d}->{cout:4d} @{hw:3d}: plain {res["plain"]:7.1f} us   border classes {res["cls"]:7.1f} us   ({res["cls"] / res["plain"] - 1:+.1%})')
