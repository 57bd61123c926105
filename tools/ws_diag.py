"""Diagnostic (needs `python continual-learning_amd/build.py --diag`): cycle shares of the producer/consumer igemm."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import continual_learning_amd as C
from continual_learning_amd._lib import call, ptr
lib = ctypes.CDLL(C._lib.LIB_PATH)
dt = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 3
pws = len(sys.argv) > 3 and sys.argv[3] == 'pws'      # persistent kernel: consumer-side shares only
dc = {'fp32': 0, 'bf16': 1, 'bf16x3': 2}[dt]
T = C.ops.TORCH_DT[dc]
B = 16
tn = C._lib.Tuning(igemm_ws=mode, igemm_pws=2 if pws else 0)
diag = lib.clamd_debug_pws_diag if pws else lib.clamd_debug_ws_diag
out = (ctypes.c_ulonglong * 8)()
for cin, cout, hw in [(64, 64, 256), (128, 64, 256), (128, 128, 128), (256, 256, 64), (1024, 512, 32)]:
    x = C.ops.randn_nhwc(dc, B, hw, hw, cin)
    w = torch.randn(cout, cin, 3, 3, device='cuda') / (3 * cin ** 0.5)
    wf = torch.zeros(9 * cout * cin, dtype=T, device='cuda'); bias = torch.zeros(cout, device='cuda')
    tab = C.ops.PackTable(dc); tab.conv3x3(w, wf, None, [(cin, cin)], cout); tab.finalize('cuda').run(dc)
    y = torch.empty(B, hw, hw, cout, dtype=T, device='cuda')
    rows = C._lib.stat_rows(C._lib.OP_CONV3X3, B, hw, hw, cin, cout, dc, tuning=tn); stats = torch.empty(rows, 2, cout, device='cuda')
    s = C._lib.stream_ptr()
    for _ in range(2):
        call('clamd_conv3x3', ptr(x), cin, ptr(wf), ptr(bias), ptr(y), cout, ptr(stats), None, None, rows, B, hw, hw, cin, cout, 1, 0, dc, tn.ref(), s)
    torch.cuda.synchronize(); diag(out, 1)
    call('clamd_conv3x3', ptr(x), cin, ptr(wf), ptr(bias), ptr(y), cout, ptr(stats), None, None, rows, B, hw, hw, cin, cout, 1, 0, dc, tn.ref(), s)
    torch.cuda.synchronize(); diag(out, 1)
    v = list(out); nb = max(v[7], 1)
    if pws:
        print(f'{cin}->{cout}@{hw}: workgroups {nb}; consumer cycles per wave per workgroup: ' + ', '.join(f'{n} {v[i] / (nb * 4):.0f}' for i, n in enumerate(['wait first stage', 'MFMA loops', 'at K-step barriers', 'epilogues'])) +
              '  | epilogue phases: ' + ', '.join(f'{n} {v[i] / (nb * 4):.0f}' for i, n in [(4, 'bias/ReLU/stats + LDS writes'), (5, 'LDS round trip'), (6, 'pack + stores (+BN sums)')]))
        continue
    names = ['prod prologue', 'prod load-wait+store+issue', 'prod at barrier', 'cons wait stage0 + PROD vmcnt wait', 'cons MFMA loop', 'cons at barrier', 'epilogue + PROD wait+store']
    print(f'{cin}->{cout}@{hw}: blocks {nb}; per-wave cycles per block: ' + ', '.join(f'{n} {v[i] / (nb * 4):.0f}' for i, n in enumerate(names)))
