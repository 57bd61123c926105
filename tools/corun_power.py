"""Does running an MFMA kernel and an HBM-bound pass on two streams buy anything, and is the power cap what limits it?  Three phases of ~1 s
on one GPU -- the F(4x4) weight-gradient plane GEMM (512 -> 512 @32x32) alone, one pass of the 64-channel 256x256 layer alone, both on two
streams (20 GEMM launches and 40 pass launches per round) -- while a thread samples the busy card's power and shader clock from sysfs (hwmon
power1_average / power1_input, freq1_input).  Prints launches per second and, for the third phase, the sum of the two rates as a fraction
of the alone rates: 1.0 = the two kernels gained nothing from sharing the chip.
    python tools/corun_power.py [reduce|apply|xform] [bn_reduce_blocks]"""
import glob
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import continual_learning_amd as C  # noqa: E402

lib, ptr = C._lib, C._lib.ptr
L = lib.load()
B = 16
NS = L.clamd_bn_bwd_nsums()
samples, stop = [], False


def read(path):
    try:
        with open(path) as f:
            return float(f.read().split()[0])
    except Exception:
        return None


def sampler():
    hw = sorted(glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*'))
    while not stop:
        row = []
        for h in hw:
            p = read(h + '/power1_average') or read(h + '/power1_input')
            f = read(h + '/freq1_input')
            row.append((p, f))
        samples.append((time.perf_counter(), row))
        time.sleep(0.02)


cin = cout = 512; hw = 32
v = torch.randn(L.clamd_winograd44_input_elems(B, hw, hw, cin), device='cuda')
yt = torch.randn(L.clamd_wgrad_winograd44_pre_operand_elems(B, hw, hw, cout), device='cuda')
wsb = L.clamd_wgrad_winograd44_pre_workspace_bytes(B, hw, hw, cout, cin)
ws = torch.empty(wsb // 4 + 4, device='cuda'); gw = torch.empty(cout, cin, 3, 3, device='cuda')
gemm = lambda s: lib.call('clamd_wgrad_winograd44_pre', None, cout, ptr(v), ptr(yt), ptr(ws), wsb, ptr(gw), B, hw, hw, cout, cin, cout, cin, cout, cout, cin, cin, None, s)
c, hh = 64, 256
g = torch.randn(B, hh, hh, c, device='cuda'); y = torch.relu(torch.randn(B, hh, hh, c, device='cuda'))
rows = lib.stat_rows(lib.OP_BN_BWD_REDUCE, B, hh, hh, 0, c, 0)
sums = torch.zeros(rows, NS, c, device='cuda'); one, zero = torch.ones(c, device='cuda'), torch.zeros(c, device='cuda')
which = sys.argv[1] if len(sys.argv) > 1 else 'reduce'
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 0
tn = lib.Tuning(bn_reduce_blocks=blocks) if blocks else None
tp = tn.ref() if tn is not None else None
rows = lib.stat_rows(lib.OP_BN_BWD_REDUCE, B, hh, hh, 0, c, 0, False, tn)
sums = torch.zeros(rows, NS, c, device='cuda')
gz = torch.empty_like(g); k012 = torch.randn(3, c, device='cuda')
vx = torch.empty(L.clamd_winograd44_input_elems(B, hh, hh, c), device='cuda')
red = {'reduce': lambda s: lib.call('clamd_bn_bwd_reduce', ptr(g), c, None, 0, ptr(y), c, ptr(one), ptr(zero), ptr(sums), rows, B, hh, hh, c, 0, tp, s),
       'apply': lambda s: lib.call('clamd_bn_bwd_apply', ptr(g), c, None, 0, ptr(y), c, ptr(one), ptr(zero), ptr(k012), ptr(gz), c, B, hh, hh, c, 0, s),
       'xform': lambda s: lib.call('clamd_winograd44_transform_input', ptr(g), c, None, None, ptr(vx), B, hh, hh, c, s)}[which]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
gemm(lib.stream_ptr()); red(lib.stream_ptr()); torch.cuda.synchronize()
th = threading.Thread(target=sampler); th.start()
phases = []
for name, fa, fb in (('plane GEMM alone', gemm, None), (which + ' pass alone', None, red), ('both', gemm, red)):
    t0 = time.perf_counter()
    na = nb = 0
    if fa is None and fb is None:
        time.sleep(0.7)
    while (fa or fb) and time.perf_counter() - t0 < 1.0:
        for _ in range(20):                                         # ~4 ms of work per stream, then wait: the queues stay short
            if fa: fa(s1.cuda_stream); na += 1
            if fb: fb(s2.cuda_stream); fb(s2.cuda_stream); nb += 2
        torch.cuda.synchronize()
    t1 = time.perf_counter()
    phases.append((name, t0, t1, na, nb))
stop = True; th.join()
ncard = len(samples[0][1])
base = [1.0, 1.0]
busy = max(range(ncard), key=lambda k: max((r[k][0] or 0) for _, r in samples))
for name, t0, t1, na, nb in phases:
    sel = [r for t, r in samples if t0 + 0.2 <= t <= t1]
    line = f'{name:20s} GEMM {na / (t1 - t0):7.0f}/s  pass {nb / (t1 - t0):7.0f}/s'
    if na and nb:
        line += f'  = {na / (t1 - t0) / base[0] + nb / (t1 - t0) / base[1]:.3f} of the two alone rates'
    elif na: base[0] = na / (t1 - t0)
    elif nb: base[1] = nb / (t1 - t0)
    for k in (busy,):
        ps = [r[k][0] for r in sel if r[k][0] is not None]; fs = [r[k][1] for r in sel if r[k][1] is not None]
        if ps or fs:
            line += f' | card{k}: {sum(ps) / max(len(ps), 1) / 1e6:6.0f} W {sum(fs) / max(len(fs), 1) / 1e6:6.0f} MHz'
    print(line)
