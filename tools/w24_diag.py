"""Diagnostic (python continual-learning_amd/build.py --diag): where a tile of the F(2x4,3x3) Winograd kernel spends its cycles."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import continual_learning_amd as C
from continual_learning_amd._lib import call, ptr
lib = ctypes.CDLL(C._lib.LIB_PATH); L = C._lib; s = L.stream_ptr()
out = (ctypes.c_ulonglong * 8)()
names = ['setup+issue+store', 'wait stage0+frags', 'K loop', 'exchange write+barrier', 'readback+transform+stores', 'statistics']
for cin, cout, hw in [(64, 64, 256), (128, 128, 128), (256, 256, 64), (1024, 1024, 16)]:
    x = torch.randn(16, hw, hw, cin, device='cuda'); w = torch.randn(cout, cin, 3, 3, device='cuda') / (3 * cin ** 0.5)
    y = torch.empty(16, hw, hw, cout, device='cuda'); bias = torch.zeros(cout, device='cuda')
    wf = torch.zeros(24 * cout * cin, device='cuda')
    tab = C.ops.WinoPackTable(24); tab.conv3x3(w, wf, None, [(cin, cin)], cout); tab.finalize('cuda').run()
    rows = L.stat_rows(L.OP_CONV3X3_WINOGRAD24, 16, hw, hw, cin, cout, 0)
    st = torch.empty(rows, 2, cout, device='cuda')
    def run(): call('clamd_conv3x3_winograd24', ptr(x), cin, ptr(wf), ptr(bias), ptr(y), cout, ptr(st), rows, 16, hw, hw, cin, cout, 1, None, s)
    for _ in range(50): run()
    torch.cuda.synchronize(); lib.clamd_debug_w24_diag(out, 1)
    run(); torch.cuda.synchronize(); lib.clamd_debug_w24_diag(out, 1)
    v = list(out); nt = max(v[6], 1)
    tot = sum(v[:6])
    print(f'{cin}->{cout}@{hw}: {tot / nt:.0f} cycles per tile, {v[7] / nt:.0f} chunks (MFMA minimum {48 * 64 * v[7] / nt:.0f}), K loop {v[2] / max(v[7], 1):.0f} per chunk')
    print('   ' + ', '.join(f'{n} {v[i] / nt:.0f} ({100 * v[i] / tot:.0f}%)' for i, n in enumerate(names)))
