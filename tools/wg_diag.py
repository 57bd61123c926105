"""Diagnostic (needs `python continual-learning_amd/build.py --diag`): cycle shares of the producer/consumer wgrad."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import continual_learning_amd as C
from continual_learning_amd._lib import call, ptr
lib = ctypes.CDLL(C._lib.LIB_PATH)
dt = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
dc = {'fp32': 0, 'bf16': 1, 'bf16x3': 2}[dt]
T = C.ops.TORCH_DT[dc]
B = 16
l = C._lib.load()
out = (ctypes.c_ulonglong * 8)()
for cin, cout, hw in [(64, 64, 256), (256, 256, 64), (512, 512, 32), (1024, 1024, 16)]:
    x = C.ops.randn_nhwc(dc, B, hw, hw, cin)
    g = C.ops.randn_nhwc(dc, B, hw, hw, cout)
    wsb = l.clamd_wgrad_workspace_bytes(0, B, hw, hw, cout, cin, dc)
    ws = torch.empty(wsb // 4 + 4, device='cuda'); gw = torch.empty(cout, cin, 3, 3, device='cuda')
    s = C._lib.stream_ptr()
    def run():
        call('clamd_wgrad', 0, ptr(g), cout, ptr(x), cin, ptr(ws), wsb, ptr(gw), B, hw, hw, cout, cin, cout, cin, cout, cout, cin, cin, dc, None, s)
    for _ in range(300): run()
    torch.cuda.synchronize(); lib.clamd_debug_wg_diag(out, 1)
    run(); torch.cuda.synchronize(); lib.clamd_debug_wg_diag(out, 1)
    v = list(out); nb = max(v[7], 1)
    names = ['prod prologue', 'prod wait+store+issue', None, None, 'prod at barrier', 'cons multiply', 'cons at barrier']
    clk = v[2] / max(v[3], 1) * 0.1
    print(f'   in-kernel clock {clk:.2f} GHz (shader cycles / 100-MHz ticks over the tile loop, wave 0 of each workgroup)')
    print(f'{cin}x{cout}@{hw}: blocks {nb}; per-wave cycles per block: ' + ', '.join(f'{n} {v[i] / (nb * 4):.0f}' for i, n in enumerate(names) if n))
