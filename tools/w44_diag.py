"""Diagnostic (python continual-learning_amd/build.py --variant diag --diag; CLAMD_LIB=build/diag/libclamd.so): where a tile of the
pre-transformed F(4x4,3x3) kernel (wino44g_kernel) spends its cycles -- stamps of wave 0, summed over workgroups."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import continual_learning_amd as C
from continual_learning_amd._lib import call, ptr
lib = ctypes.CDLL(C._lib.LIB_PATH); L = C._lib; s = L.stream_ptr()
LL = L.load()
out = (ctypes.c_ulonglong * 8)()
names = ['setup + first loads', 'K loop', 'write 0 + barrier', 'read 0 + barrier', 'write 1 + barrier', 'read 1 + barrier']
for cin, cout, hw in [(64, 128, 256), (128, 256, 128), (256, 256, 64), (512, 512, 32), (1024, 512, 32)]:
    x = torch.randn(16, hw, hw, cin, device='cuda'); w = torch.randn(cout, cin, 3, 3, device='cuda') / (3 * cin ** 0.5)
    y = torch.empty(16, hw, hw, cout, device='cuda'); bias = torch.zeros(cout, device='cuda')
    wf = torch.zeros(36 * cout * cin, device='cuda')
    tab = C.ops.WinoPackTable(36); tab.conv3x3(w, wf, None, [(cin, cin)], cout); tab.finalize('cuda').run()
    rows = L.stat_rows(L.OP_CONV3X3_WINOGRAD44, 16, hw, hw, cin, cout, 0)
    st = torch.empty(rows, 2, cout, device='cuda')
    v = torch.empty(LL.clamd_winograd44_input_elems(16, hw, hw, cin), device='cuda')
    call('clamd_winograd44_transform_input', ptr(x), cin, None, None, ptr(v), 16, hw, hw, cin, s)
    for fwd in (1, 0):
        def run(): call('clamd_conv3x3_winograd44_pre', ptr(v), ptr(wf), ptr(bias) if fwd else None, ptr(y), cout, ptr(st) if fwd else None, rows if fwd else 0,
                        16, hw, hw, cin, cout, fwd, None, s)
        for _ in range(20): run()
        torch.cuda.synchronize(); lib.clamd_debug_w44_diag(out, 1)
        run(); torch.cuda.synchronize(); lib.clamd_debug_w44_diag(out, 1)
        d = list(out); nt = max(d[6], 1)
        tot = sum(d[:6])
        print(f'{cin}->{cout}@{hw} {"fwd" if fwd else "plain"}: {tot / nt:.0f} cycles per tile ({d[6]} tiles), {d[7] / nt:.0f} chunks (MFMA minimum for 3 waves per SIMD {3 * 24 * 64 * d[7] / nt:.0f}), '
              f'K loop {d[1] / max(d[7], 1):.0f} per chunk')
        print('   ' + ', '.join(f'{n} {d[i] / nt:.0f} ({100 * d[i] / tot:.0f}%)' for i, n in enumerate(names)))
