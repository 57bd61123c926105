"""F(2x4,3x3) (wino24.hip / the direct-filter form of wino24g.hip, whichever the engine uses) against F(4,3)-along-the-row x 3 kernel rows
(wino41.hip) on the narrow layer shapes (fp32, bs16, 256x256 input): forward launches (bias + ReLU + statistics rows) and plain data-gradient
launches, interleaved in one process.  `exec` = fraction of the fp32 MFMA peak by EXECUTED multiply-adds (1/3 resp. 1/2 of the direct count).
    python tools/wino41_ab.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import continual_learning_amd as C  # noqa: E402

lib, ptr = C._lib, C._lib.ptr
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = 16
SH = [(64, 64, 256), (64, 128, 128), (128, 128, 128), (128, 256, 64), (256, 128, 128), (128, 64, 256), (256, 256, 64)]
for mode in ('fwd', 'dgrad'):
    tot = {'w24': 0.0, 'w41': 0.0}
    print(f'{mode}: {"layer":>18s} {"F(2x4) us":>10s} {"exec":>6s} {"F(4,3)x3 us":>12s} {"exec":>6s} {"speedup":>8s}')
    for cin, cout, hw in SH:
        x = torch.randn(B, hw, hw, cin, device='cuda')
        w = torch.randn(cout, cin, 3, 3, device='cuda') / (3 * cin ** 0.5)
        bias = torch.zeros(cout, device='cuda')
        y = torch.empty(B, hw, hw, cout, device='cuda')
        s = lib.stream_ptr()
        w24, w41 = torch.zeros(24 * cout * cin, device='cuda'), torch.zeros(18 * cout * cin, device='cuda')
        for planes, buf in ((24, w24), (18, w41)):
            tab = C.ops.WinoPackTable(planes); tab.conv3x3(w, buf, None, [(cin, cin)], cout); tab.finalize('cuda').run()
        r24 = lib.stat_rows(lib.OP_CONV3X3_WINOGRAD24, B, hw, hw, cin, cout, 0)
        r41 = lib.stat_rows(lib.OP_CONV3X3_WINOGRAD41, B, hw, hw, cin, cout, 0)
        st = torch.empty(max(r24, r41), 2, cout, device='cuda')
        fwd = mode == 'fwd'
        n24 = 'clamd_conv3x3_winograd24_direct_filters' if cin == 64 else 'clamd_conv3x3_winograd24'
        calls = {'w24': lambda: lib.call(n24, ptr(x), cin, ptr(w24), ptr(bias) if fwd else None, ptr(y), cout, ptr(st) if fwd else None, r24 if fwd else 0,
                                         B, hw, hw, cin, cout, 1 if fwd else 0, None, s),
                 'w41': lambda: lib.call('clamd_conv3x3_winograd41', ptr(x), cin, ptr(w41), ptr(bias) if fwd else None, ptr(y), cout, ptr(st) if fwd else None,
                                         r41 if fwd else 0, B, hw, hw, cin, cout, 1 if fwd else 0, None, s)}
        res, ref = {}, None
        for rd in range(3):
            for key, fn in calls.items():
                fn()
                if rd == 0:
                    torch.cuda.synchronize()
                    if ref is None:
                        ref = y.clone()
                    else:
                        err = float((ref - y).norm() / ref.norm())
                        assert err < 1e-5, err
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    fn()
                e1.record(); e1.synchronize()
                res[key] = min(res.get(key, 1e9), e0.elapsed_time(e1) / reps * 1e3)
        fl = 2.0 * B * hw * hw * 9 * cin * cout
        for k in tot:
            tot[k] += res[k]
        print(f'      {cin:5d}->{cout:5d} @{hw:3d} {res["w24"]:10.1f} {fl / 3 / res["w24"] / 1e6 / 157.3:6.3f} {res["w41"]:12.1f} {fl / 2 / res["w41"] / 1e6 / 157.3:6.3f} '
              f'{res["w24"] / res["w41"]:8.3f}')
    print(f'      total {tot["w24"] / 1e3:.3f} ms vs {tot["w41"] / 1e3:.3f} ms: {tot["w24"] / tot["w41"]:.3f}x')
