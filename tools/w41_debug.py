import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import continual_learning_amd as C
lib, ptr = C._lib, C._lib.ptr
torch.manual_seed(0)
B, H, W, cin, cout = 1, 16, 32, 64, 64
x = torch.randn(B, H, W, cin, device='cuda')
w = torch.randn(cout, cin, 3, 3, device='cuda') / (3 * cin ** 0.5)
w41 = torch.zeros(18 * cout * cin, device='cuda')
tab = C.ops.WinoPackTable(18); tab.conv3x3(w, w41, None, [(cin, cin)], cout); tab.finalize('cuda').run()
ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), padding=1).permute(0, 2, 3, 1).float()
zb = torch.zeros(cout, device='cuda')
s = lib.stream_ptr()
for name, bias in (('EPI0 plain', None), ('EPI1 zero bias, no relu', zb)):
    y = torch.full((B, H, W, cout), 7.0, device='cuda')
    lib.call('clamd_conv3x3_winograd41', ptr(x), cin, ptr(w41), ptr(bias), ptr(y), cout, None, 0, B, H, W, cin, cout, 0, None, s)
    torch.cuda.synchronize()
    err = (y - ref).abs()
    print(name, 'rel', float((y - ref).norm() / ref.norm()), 'max', float(err.max()))
    bad = err > 1e-3
    print('  bad fraction', float(bad.float().mean()), 'by row', bad.float().mean((0, 2, 3)).cpu().numpy().round(2), '\n  by col', bad.float().mean((0, 1, 3)).cpu().numpy().round(2),
          '\n  by channel', bad.float().mean((0, 1, 2)).cpu().numpy().round(2))
