"""bn_bwd_reduce_kernel (+ its fixed-order finalize): time per launch for several grid caps (clamd_tuning::bn_reduce_blocks), UNet layer shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import continual_learning_amd as C
from continual_learning_amd._lib import call, ptr
dt = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
dc = {'fp32': 0, 'bf16': 1}[dt]
T = C.ops.TORCH_DT[dc]
lib = C._lib.load(); s = C._lib.stream_ptr(); B = 16
caps = [0, 256, 512, 1024, 2048]
for ch, hw in [(64, 256), (128, 128), (256, 64), (512, 32), (1024, 16)]:
    g = torch.randn(B, hw, hw, ch, device='cuda').to(T); y = torch.randn(B, hw, hw, ch, device='cuda').to(T)
    out = []
    for cap in caps:
        tn = C._lib.Tuning(bn_reduce_blocks=cap)
        rows = C._lib.stat_rows(C._lib.OP_BN_BWD_REDUCE, B, hw, hw, 0, ch, dc, tuning=tn)
        sums = torch.empty(rows, 5, ch, device='cuda')
        f = lambda: call('clamd_bn_bwd_reduce', ptr(g), ch, None, 0, ptr(y), ch, None, None, ptr(sums), rows, B, hw, hw, ch, dc, tn.ref(), s)
        f(); best = 1e9
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): f()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
        out.append(f'{cap}: {best:6.1f}us')
    nb = 2 * g.numel() * g.element_size()
    print(f'{dt} C={ch:5d} @{hw:3d} ({nb / 1e6:6.1f} MB)  ' + '  '.join(out))
print('channel_sum:')
for ch, hw in [(32, 256), (64, 256), (128, 128), (256, 64), (512, 32)]:
    g = torch.randn(B, hw, hw, ch, device='cuda').to(T); o = torch.zeros(ch, device='cuda')
    out = []
    csb = lib.clamd_channel_sum_workspace_bytes(ch); cws = torch.empty(csb // 4, device='cuda')
    for cap in [64, 128, 256, 512, 1024]:
        tn = C._lib.Tuning(chsum_blocks=cap)
        f = lambda: call('clamd_channel_sum', ptr(g), ch, ptr(o), B * hw * hw, ch, ch, dc, ptr(cws), csb, tn.ref(), s)
        f(); best = 1e9
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): f()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
        out.append(f'{cap}: {best:6.1f}us')
    print(f'{dt} C={ch:5d} @{hw:3d} ({g.numel() * g.element_size() / 1e6:6.1f} MB)  ' + '  '.join(out))
