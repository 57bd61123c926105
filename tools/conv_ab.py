"""Interleaved A/B of igemm schedule variants in ONE process (conv3x3, UNet layer shapes).

    python tools/conv_ab.py bf16 0,1,2 igemm_variant            # forward launches (bias + ReLU + statistics rows)
    CONV_MODE=dgrad python tools/conv_ab.py bf16 0,1 pws_wres    # plain data-gradient launches (no bias / ReLU / statistics)
    CONV_MODE=dgrad_bn ...                                   # ... with the BatchNorm-backward sums of the consumer in the epilogue
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import continual_learning_amd as C
from continual_learning_amd._lib import call, ptr
dt = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
variants = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else '0,1,2').split(',')]
key = sys.argv[3] if len(sys.argv) > 3 else 'igemm_variant'      # a field of clamd_tuning, passed per call
dc = {'fp32': 0, 'bf16': 1, 'bf16x3': 2}[dt]
T = C.ops.TORCH_DT[dc]
B, iters, rounds = 16, 10, 5
layers = [(64, 64, 256), (128, 64, 256), (128, 128, 128), (256, 128, 128), (256, 256, 64), (512, 512, 32), (1024, 512, 32), (1024, 1024, 16)]
if os.environ.get('CONV_LAYERS'):      # e.g. CONV_LAYERS='1024,512,16;512,1024,16'
    layers = [tuple(int(v) for v in l.split(',')) for l in os.environ['CONV_LAYERS'].split(';')]
lib = C._lib.load(); s = C._lib.stream_ptr()
mode = os.environ.get('CONV_MODE', 'fwd')
tot = {v: [0.0, 0.0] for v in variants}
for cin, cout, hw in layers:
    x = C.ops.randn_nhwc(dc, B, hw, hw, cin)
    w = torch.randn(cout, cin, 3, 3, device='cuda') / (3 * cin ** 0.5)
    wf = torch.zeros(9 * cout * cin, dtype=T, device='cuda')
    bias = torch.zeros(cout, device='cuda')
    tab = C.ops.PackTable(dc); tab.conv3x3(w, wf, None, [(cin, cin)], cout); tab.finalize('cuda').run(dc)
    y = torch.empty(B, hw, hw, cout, dtype=T, device='cuda')
    tun = {v: C._lib.Tuning(**{key: v}) for v in variants}
    rows = {v: C._lib.stat_rows(C._lib.OP_CONV3X3, B, hw, hw, cin, cout, dc, tuning=tun[v]) for v in variants}
    stats = torch.empty(max(rows.values()), 2, cout, device='cuda')
    mf = 1 if 9 * cout > B * hw * hw else 0
    best = {v: 1e9 for v in variants}
    ref = None
    if mode != 'fwd':
        bias = None
        brows = {v: C._lib.stat_rows(C._lib.OP_CONV3X3, B, hw, hw, cin, cout, dc, True, tuning=tun[v]) for v in variants}
        ysave = C.ops.randn_nhwc(dc, B, hw, hw, cout).clamp_min(0) if dc != 2 else C.ops.randn_nhwc(dc, B, hw, hw, cout)
        bsums = torch.empty(max(brows.values()), 5, cout, device='cuda')

    def launch(v):
        if mode == 'fwd':
            call('clamd_conv3x3', ptr(x), cin, ptr(wf), ptr(bias), ptr(y), cout, ptr(stats), None, None, rows[v], B, hw, hw, cin, cout, 1, mf, dc, tun[v].ref(), s)
        elif mode == 'dgrad':
            call('clamd_conv3x3', ptr(x), cin, ptr(wf), None, ptr(y), cout, None, None, None, 0, B, hw, hw, cin, cout, 0, mf, dc, tun[v].ref(), s)
        else:
            call('clamd_conv3x3', ptr(x), cin, ptr(wf), None, ptr(y), cout, None, ptr(ysave), ptr(bsums), brows[v], B, hw, hw, cin, cout, 0, mf, dc, tun[v].ref(), s)
    for rd in range(rounds):
        for v in variants:
            launch(v)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                launch(v)
            e1.record(); torch.cuda.synchronize()
            best[v] = min(best[v], e0.elapsed_time(e1) / iters * 1e-3)
            if rd == 0:
                if ref is None: ref = y.float().clone()
                elif v < 3: assert torch.equal(ref, y.float()), f'variant {v} changed the result'
    fl = 2.0 * B * hw * hw * 9 * cin * cout
    print(f'{cin:5d}->{cout:5d} @{hw:3d}: ' + '  '.join(f'v{v} {best[v]*1e6:7.1f}us {fl/best[v]/1e12:7.1f}TF' for v in variants))
    for v in variants: tot[v][0] += fl; tot[v][1] += best[v]
print(dt, 'aggregate: ' + '  '.join(f'v{v} {tot[v][0]/tot[v][1]/1e12:.1f} TF/s' for v in variants))
