"""The two boundary kernels of the bf16 / bf16x3 / fp32 step, each alone (HIP events): the first layer's im2col (NCHW fp32 images -> 27 (32)
NHWC channels) and the loss kernel with its NHWC copy of d logits, BASELINE configs[1] sizes.   python tools/boundary_time.py [dtype]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import continual_learning_amd as C  # noqa: E402
from continual_learning_amd._lib import call, ptr  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
dc = {'fp32': 0, 'bf16': 1, 'bf16x3': 2}[dtype]
T = C.ops.TORCH_DT[dc]
L = C._lib.load(); s = C._lib.stream_ptr()
B, K, H, W = 16, 21, 256, 256
x = torch.from_numpy(C.synth.images(1234, B, 3, H, W)).cuda()
y = torch.from_numpy(C.synth.labels(1234, B, H, W, K)).cuda()
xin = torch.empty(B, H, W, 32, dtype=T, device='cuda')
z = torch.randn(B, K, H, W, device='cuda')
d = torch.empty_like(z); nh = torch.empty(B, H, W, 32, dtype=T, device='cuda'); l3 = torch.empty(3, device='cuda')
wsb = L.clamd_ce_workspace_bytes(); ws = torch.empty(wsb // 4, device='cuda')


def t(f, n=30):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


esz = 2 if dc == 1 else 4
print(f'{dtype}: im2col {t(lambda: call("clamd_nchw_im2col3", ptr(x), ptr(xin), 32, B, 3, H, W, 32, dc, s)):.1f} us '
      f'({(x.numel() * 4 + xin.numel() * esz) / 1e6:.0f} MB)')
call('clamd_ce_count', ptr(y), B, K, H, W, -100, ptr(ws), wsb, s)
print(f'{dtype}: loss + NHWC copy {t(lambda: call("clamd_ce_fwd_bwd_counted", ptr(z), ptr(y), ptr(d), ptr(nh), 32, dc, ptr(l3), ptr(ws), wsb, B, K, H, W, -100, 1.0, s)):.1f} us '
      f'({(2 * z.numel() * 4 + nh.numel() * esz + y.numel() * 8) / 1e6:.0f} MB; with its finalize launch)')
print(f'{dtype}: loss alone       {t(lambda: call("clamd_ce_fwd_bwd_counted", ptr(z), ptr(y), ptr(d), None, 0, 0, ptr(l3), ptr(ws), wsb, B, K, H, W, -100, 1.0, s)):.1f} us')
