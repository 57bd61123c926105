"""Diagnostic (build.py --diag): cycles per pixel tile and in-kernel clock of the Winograd weight-gradient kernel."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import continual_learning_amd as C
from continual_learning_amd._lib import call, ptr
lib = ctypes.CDLL(C._lib.LIB_PATH); l = C._lib.load(); s = C._lib.stream_ptr()
out = (ctypes.c_ulonglong * 4)()
for cin, cout, hw in [(64, 64, 256), (512, 512, 32)]:
    x = torch.randn(16, hw, hw, cin, device='cuda'); g = torch.randn(16, hw, hw, cout, device='cuda')
    wsb = l.clamd_wgrad_winograd_workspace_bytes(cout, cin)
    ws = torch.empty(wsb // 4 + 4, device='cuda'); gw = torch.empty(cout, cin, 3, 3, device='cuda')
    def run(): call('clamd_wgrad_winograd', ptr(g), cout, ptr(x), cin, ptr(ws), wsb, ptr(gw), 16, hw, hw, cout, cin, cout, cin, cout, cout, cin, cin, None, s)
    for _ in range(200): run()
    torch.cuda.synchronize(); lib.clamd_debug_ww_diag(out, 1)
    run(); torch.cuda.synchronize(); lib.clamd_debug_ww_diag(out, 1)
    v = list(out)
    print(f'{cin}x{cout}@{hw}: {v[0] / max(v[2], 1):.0f} cycles per tile (ideal 16384), in-kernel clock {v[0] / max(v[1], 1) * 0.1:.2f} GHz, {v[2] / max(v[3], 1):.1f} tiles per workgroup')
