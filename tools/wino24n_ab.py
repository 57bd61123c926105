"""Half-width F(2x4,3x3) workgroups (wino24n.hip: 32 tiles x 32 channels, two workgroups per CU; clamd_tuning::wino_half) against the shipped
full-width kernels (wino24_kernel, and wino24h_kernel where the engine runs it: 64 input channels) on the narrow layer shapes of the fp32 step
(bs16, 256x256 input), forward (bias + ReLU + statistics) and plain data-gradient launches, interleaved in one process.
    python tools/wino24n_ab.py [reps]
Run under `rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES` / `SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS` for the counters."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import continual_learning_amd as C  # noqa: E402

lib, ptr = C._lib, C._lib.ptr
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = 16
SH = [(64, 64, 256), (64, 128, 128), (128, 128, 128), (128, 64, 256), (256, 128, 128), (128, 256, 64)]
half = lib.Tuning(wino_half=1)
print(f'{"layer":>18s} {"mode":>6s} {"full us":>9s} {"exec":>6s} {"direct us":>10s} {"half us":>9s} {"exec":>6s} {"full/half":>9s}')
tot = {}
for cin, cout, hw in SH:
    x = torch.randn(B, hw, hw, cin, device='cuda')
    w = torch.randn(cout, cin, 3, 3, device='cuda') / (3 * cin ** 0.5)
    bias = torch.zeros(cout, device='cuda')
    y = torch.empty(B, hw, hw, cout, device='cuda')
    s = lib.stream_ptr()
    wf = torch.zeros(24 * cout * cin, device='cuda')
    tab = C.ops.WinoPackTable(24); tab.conv3x3(w, wf, None, [(cin, cin)], cout); tab.finalize('cuda').run()
    rows_f = lib.stat_rows(lib.OP_CONV3X3_WINOGRAD24, B, hw, hw, cin, cout, 0)
    rows_h = lib.stat_rows(lib.OP_CONV3X3_WINOGRAD24, B, hw, hw, cin, cout, 0, tuning=half)
    st_f, st_h = torch.empty(rows_f, 2, cout, device='cuda'), torch.empty(rows_h, 2, cout, device='cuda')
    fl = 2.0 * B * hw * hw * 9 * cin * cout / 3
    for mode in ('fwd', 'plain'):
        fwd = mode == 'fwd'
        res = {}
        variants = [('full', 'clamd_conv3x3_winograd24', None, st_f, rows_f), ('half', 'clamd_conv3x3_winograd24', half, st_h, rows_h)]
        if cin == 64:
            variants.insert(1, ('direct', 'clamd_conv3x3_winograd24_direct_filters', None, st_f, rows_f))
        for rnd in range(3):
            for key, name, tn, st, rows in variants:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    lib.call(name, ptr(x), cin, ptr(wf), ptr(bias) if fwd else None, ptr(y), cout, ptr(st) if fwd else None, rows if fwd else 0,
                             B, hw, hw, cin, cout, 1 if fwd else 0, tn.ref() if tn else None, s)
                e1.record(); e1.synchronize()
                res[key] = e0.elapsed_time(e1) / reps * 1e3
        best = min(res['full'], res.get('direct', 1e30))
        for k, v in res.items():
            tot[k] = tot.get(k, 0.0) + (v if k != 'direct' else 0.0)
        tot['best'] = tot.get('best', 0.0) + best
        d = f'{res["direct"]:10.1f}' if 'direct' in res else f'{"-":>10s}'
        print(f'{cin:5d}->{cout:5d} @{hw:3d} {mode:>6s} {res["full"]:9.1f} {fl / res["full"] / 1e6 / 157.3:6.3f} {d} {res["half"]:9.1f} '
              f'{fl / res["half"] / 1e6 / 157.3:6.3f} {best / res["half"]:9.3f}')
print(f'total: shipped (best of full / direct) {tot["best"] / 1e3:.3f} ms, half-width {tot["half"] / 1e3:.3f} ms: {tot["best"] / tot["half"]:.3f}x')
