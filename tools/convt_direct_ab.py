"""ConvTranspose2d forward / data gradient on the fp32 path: igemm_kernel (LDS-staged) against pw_direct_kernel (operands
straight into the MFMA registers) on the UNet's four shapes at bs16, interleaved in one process.   python tools/convt_direct_ab.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import continual_learning_amd as C  # noqa: E402

lib, ptr = C._lib, C._lib.ptr
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = 16
tot = dict(f_old=0.0, f_new=0.0, d_old=0.0, d_new=0.0)
print(f'{"layer":>20s} | {"fwd old":>8s} {"of pk":>6s} {"fwd new":>8s} {"of pk":>6s} | {"dgrad old":>9s} {"of pk":>6s} {"dgrad new":>9s} {"of pk":>6s}')
for cin, cout, hw in [(1024, 512, 16), (512, 256, 32), (256, 128, 64), (128, 64, 128)]:
    x = torch.randn(B, hw, hw, cin, device='cuda')
    w = torch.randn(cin, cout, 2, 2, device='cuda') / cin ** 0.5
    bias = torch.randn(cout, device='cuda')
    wf = torch.zeros(4 * cout * cin, device='cuda'); wd = torch.zeros(cin * 4 * cout, device='cuda'); bp = torch.zeros(cout, device='cuda')
    tab = C.ops.PackTable(0); tab.convT(w, wf, wd, cin, cout); tab.vector(bias, bp, cout); tab.finalize('cuda').run(0)
    cat = torch.zeros(B, 2 * hw, 2 * hw, 2 * cout, device='cuda')
    gcat = torch.randn(B, 2 * hw, 2 * hw, 2 * cout, device='cuda')
    gx = torch.empty(B, hw, hw, cin, device='cuda')
    s = lib.stream_ptr()
    fl = 2.0 * B * hw * hw * cin * 4 * cout
    runs = {
        'f_old': lambda: lib.call('clamd_convT2x2_fwd', ptr(x), cin, ptr(wf), ptr(bp), ptr(cat[..., cout:]), 2 * cout, B, hw, hw, cin, cout, 0, s),
        'f_new': lambda: lib.call('clamd_convT2x2_fwd_direct', ptr(x), cin, ptr(wf), ptr(bp), ptr(cat[..., cout:]), 2 * cout, B, hw, hw, cin, cout, s),
        'd_old': lambda: lib.call('clamd_convT2x2_dgrad', ptr(gcat[..., cout:]), 2 * cout, ptr(wd), ptr(gx), cin, None, None, 0, B, hw, hw, cin, cout, 0, s),
        'd_new': lambda: lib.call('clamd_convT2x2_dgrad_direct', ptr(gcat[..., cout:]), 2 * cout, ptr(wd), ptr(gx), cin, B, hw, hw, cin, cout, s),
    }
    res = {}
    for rnd in range(3):
        for k, fn in runs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record(); e1.synchronize()
            res[k] = e0.elapsed_time(e1) / reps * 1e3
    for k in tot:
        tot[k] += res[k]
    pk = lambda us: fl / us / 1e6 / 157.3
    print(f'{cin:5d}->{cout:4d} @{hw:3d}->{2 * hw:3d} | {res["f_old"]:8.1f} {pk(res["f_old"]):6.3f} {res["f_new"]:8.1f} {pk(res["f_new"]):6.3f} | '
          f'{res["d_old"]:9.1f} {pk(res["d_old"]):6.3f} {res["d_new"]:9.1f} {pk(res["d_new"]):6.3f}')
print('total: fwd %.3f -> %.3f ms, dgrad %.3f -> %.3f ms' % (tot['f_old'] / 1e3, tot['f_new'] / 1e3, tot['d_old'] / 1e3, tot['d_new'] / 1e3))
