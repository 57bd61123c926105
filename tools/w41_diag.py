"""Diagnostic (python continual-learning_amd/build.py --variant diag --diag; CLAMD_LIB=build/diag/libclamd.so): where a workgroup of the
F(4,3)-along-the-row kernel (wino41.hip) spends its cycles -- wave 0's s_memtime stamps, per workgroup."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import continual_learning_amd as C
lib, ptr = C._lib, C._lib.ptr
raw = ctypes.CDLL(C._lib.LIB_PATH)
out = (ctypes.c_ulonglong * 8)()
B = 16
for cin, cout, hw in [(64, 64, 256), (128, 128, 128), (256, 256, 64)]:
    x = torch.randn(B, hw, hw, cin, device='cuda')
    w = torch.randn(cout, cin, 3, 3, device='cuda') / (3 * cin ** 0.5)
    w41 = torch.zeros(18 * cout * cin, device='cuda')
    tab = C.ops.WinoPackTable(18); tab.conv3x3(w, w41, None, [(cin, cin)], cout); tab.finalize('cuda').run()
    y = torch.empty(B, hw, hw, cout, device='cuda'); bias = torch.zeros(cout, device='cuda')
    rows = lib.stat_rows(lib.OP_CONV3X3_WINOGRAD41, B, hw, hw, cin, cout, 0); st = torch.empty(rows, 2, cout, device='cuda')
    s = lib.stream_ptr()
    for _ in range(2):
        lib.call('clamd_conv3x3_winograd41', ptr(x), cin, ptr(w41), ptr(bias), ptr(y), cout, ptr(st), rows, B, hw, hw, cin, cout, 1, None, s)
    torch.cuda.synchronize(); raw.clamd_debug_w41_diag(out, 1)
    lib.call('clamd_conv3x3_winograd41', ptr(x), cin, ptr(w41), ptr(bias), ptr(y), cout, ptr(st), rows, B, hw, hw, cin, cout, 1, None, s)
    torch.cuda.synchronize(); raw.clamd_debug_w41_diag(out, 1)
    v = list(out); nb = max(v[7], 1)
    tiles = B * (hw // 16) * (hw // 32) * ((cout + 63) // 64) / nb
    names = ['load issue', 'MFMA block', 'transform + stage stores', 'barrier', 'epilogues']
    print(f'{cin}->{cout}@{hw}: {nb} workgroups, {tiles:.1f} tiles x {cin // 8} chunks each; cycles per workgroup: ' + ', '.join(f'{n} {v[i] / nb:.0f}' for i, n in enumerate(names))
          + f' | per chunk: MFMA block {v[1] / nb / tiles / (cin // 8):.0f} (144 MFMAs = 9216), per tile: epilogue {v[4] / nb / tiles:.0f}')
