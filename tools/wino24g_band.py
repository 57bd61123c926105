"""clamd_conv3x3_winograd24_pre (transform-free F(2x4,3x3) loop, wino24g.hip) under every block order clamd_tuning::wino_band allows, one layer
shape: time per launch.  Run under `rocprofv3 --pmc FETCH_SIZE --output-format csv -d DIR -o pmc -- python3 tools/wino24g_band.py ...` for the bytes
the launches fetch through L2 (tools/wino24g_band.sh does both and prints the table).   python tools/wino24g_band.py [cin cout hw] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import continual_learning_amd as C  # noqa: E402

lib, ptr = C._lib, C._lib.ptr
L = lib.load()
cin, cout, hw = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (512, 512, 32)
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
B = 16
x = torch.randn(B, hw, hw, cin, device='cuda')
w = torch.randn(cout, cin, 3, 3, device='cuda') / (3 * cin ** 0.5)
bias = torch.zeros(cout, device='cuda')
y = torch.empty(B, hw, hw, cout, device='cuda')
s = lib.stream_ptr()
wf = torch.zeros(24 * cout * cin, device='cuda')
tab = C.ops.WinoPackTable(24); tab.conv3x3(w, wf, None, [(cin, cin)], cout); tab.finalize('cuda').run()
v = torch.empty(L.clamd_winograd24_input_elems(B, hw, hw, cin), device='cuda')
lib.call('clamd_winograd24_transform_input', ptr(x), cin, None, None, ptr(v), B, hw, hw, cin, s)
print(f'{cin}->{cout} @{hw}: V {v.numel() * 4 / 1e6:.0f} MB, filters {wf.numel() * 4 / 1e6:.0f} MB, output {y.numel() * 4 / 1e6:.0f} MB')
for band in (0, 1, 2, 4, 8, 16):
    if band > cout // 64:
        continue
    tn = lib.Tuning(wino_band=band)
    rows = lib.stat_rows(lib.OP_CONV3X3_WINOGRAD24, B, hw, hw, cin, cout, 0, tuning=tn)
    st = torch.empty(rows, 2, cout, device='cuda')
    f = lambda: lib.call('clamd_conv3x3_winograd24_pre', ptr(v), ptr(wf), ptr(bias), ptr(y), cout, ptr(st), rows, B, hw, hw, cin, cout, 1, tn.ref(), s)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); e1.synchronize()
    print(f'BAND {band:2d} ({"per-launch choice" if band == 0 else "slabs per band"}): {e0.elapsed_time(e1) / reps * 1e3:7.1f} us per launch, {reps + 1} launches')
