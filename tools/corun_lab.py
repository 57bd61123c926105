"""What does an HBM-bound pass of the critical chain cost BESIDE an MFMA kernel of the second stream?  (fp32, BASELINE configs[1] shapes.)
A background stream runs one weight-gradient / convolution kernel back to back; once it is under way the foreground stream runs `reps` launches
of one pass between two events.  Printed: the pass alone, the pass beside each background kernel (us per launch and the ratio), and what the
background lost (its launches' average duration with / without the foreground).
    python tools/corun_lab.py [reps=12]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import continual_learning_amd as C  # noqa: E402

lib, ptr = C._lib, C._lib.ptr
L = lib.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
B = 16
bg_stream, fg_stream = torch.cuda.Stream(), torch.cuda.Stream()
NS = L.clamd_bn_bwd_nsums()


def background():
    """name -> (launch(stream_ptr), approximate us)"""
    out = {}
    # plane GEMM of the F(4x4) weight gradient, 512 -> 512 @32x32 (stream-K, no LDS, one 168-register wave per SIMD)
    cin = cout = 512; hw = 32
    v = torch.randn(L.clamd_winograd44_input_elems(B, hw, hw, cin), device='cuda')
    yt = torch.randn(L.clamd_wgrad_winograd44_pre_operand_elems(B, hw, hw, cout), device='cuda')
    wsb = L.clamd_wgrad_winograd44_pre_workspace_bytes(B, hw, hw, cout, cin)
    ws = torch.empty(wsb // 4 + 4, device='cuda'); gw = torch.empty(cout, cin, 3, 3, device='cuda')
    out['plane GEMM 512->512@32'] = lambda s: lib.call('clamd_wgrad_winograd44_pre', None, cout, ptr(v), ptr(yt), ptr(ws), wsb, ptr(gw), B, hw, hw, cout, cin,
                                                       cout, cin, cout, cout, cin, cin, None, s)
    # F(2x4) weight gradient with in-kernel transforms, 64 -> 64 @256x256 (111 KB LDS, 172 registers)
    c2 = 64; hw2 = 256
    x2 = torch.randn(B, hw2, hw2, c2, device='cuda'); g2 = torch.randn(B, hw2, hw2, c2, device='cuda')
    wsb2 = L.clamd_wgrad_winograd24_workspace_bytes(c2, c2); ws2 = torch.empty(wsb2 // 4 + 4, device='cuda'); gw2 = torch.empty(c2, c2, 3, 3, device='cuda')
    out['wino24_wgrad 64->64@256'] = lambda s: lib.call('clamd_wgrad_winograd24', ptr(g2), c2, ptr(x2), c2, ptr(ws2), wsb2, ptr(gw2), B, hw2, hw2, c2, c2, c2, c2,
                                                        c2, c2, c2, c2, None, s)
    # F(4x4) forward / data-gradient kernel, 256 -> 256 @64x64 (768 threads, 143 KB LDS)
    c3 = 256; hw3 = 64
    v3 = torch.randn(L.clamd_winograd44_input_elems(B, hw3, hw3, c3), device='cuda')
    w3 = torch.randn(c3, c3, 3, 3, device='cuda') / 48
    wf3 = torch.zeros(36 * c3 * c3, device='cuda')
    tab = C.ops.WinoPackTable(36); tab.conv3x3(w3, wf3, None, [(c3, c3)], c3); tab.finalize('cuda').run()
    y3 = torch.empty(B, hw3, hw3, c3, device='cuda')
    out['wino44g conv 256->256@64'] = lambda s: lib.call('clamd_conv3x3_winograd44_pre', ptr(v3), ptr(wf3), None, ptr(y3), c3, None, 0, B, hw3, hw3, c3, c3, 0, None, s)
    out['_keep'] = (v, yt, ws, gw, x2, g2, ws2, gw2, v3, w3, wf3, y3, tab)
    return out


def passes():
    out = {}
    for hw, c in ((256, 64), (64, 256), (16, 1024)):
        g = torch.randn(B, hw, hw, c, device='cuda'); y = torch.relu(torch.randn(B, hw, hw, c, device='cuda')); gz = torch.empty_like(g)
        rows = lib.stat_rows(lib.OP_BN_BWD_REDUCE, B, hw, hw, 0, c, 0)
        sums = torch.zeros(rows, NS, c, device='cuda')
        one, zero = torch.ones(c, device='cuda'), torch.zeros(c, device='cuda')
        k012 = torch.randn(3, c, device='cuda'); vec = torch.ones(4, c, device='cuda')
        dg, db, dcb = torch.zeros(c, device='cuda'), torch.zeros(c, device='cuda'), torch.zeros(c, device='cuda')
        nr = L.clamd_bn_bwd_apply_sums_rows(B, hw, hw, c); gzr = torch.empty(nr, c, device='cuda')
        vx = torch.empty(L.clamd_winograd44_input_elems(B, hw, hw, c), device='cuda')
        mb = g.numel() * 4 / 1e6
        tag = f'{c}ch@{hw}'
        out[f'reduce5 {tag}'] = (lambda s, g=g, y=y, sums=sums, rows=rows, one=one, zero=zero, hw=hw, c=c:
                                 lib.call('clamd_bn_bwd_reduce', ptr(g), c, None, 0, ptr(y), c, ptr(one), ptr(zero), ptr(sums), rows, B, hw, hw, c, 0, None, s), 2 * mb)
        out[f'finalize {tag}'] = (lambda s, sums=sums, rows=rows, vec=vec, k012=k012, dg=dg, db=db, dcb=dcb, hw=hw, c=c:
                                  lib.call('clamd_bn_bwd_finalize', ptr(sums), rows, ptr(vec[0]), ptr(vec[1]), ptr(vec[2]), ptr(k012), ptr(dg), ptr(db), ptr(dcb), c, c,
                                           float(B * hw * hw), s), 0.0)
        out[f'apply {tag}'] = (lambda s, g=g, y=y, one=one, zero=zero, k012=k012, gz=gz, hw=hw, c=c:
                               lib.call('clamd_bn_bwd_apply', ptr(g), c, None, 0, ptr(y), c, ptr(one), ptr(zero), ptr(k012), ptr(gz), c, B, hw, hw, c, 0, s), 3 * mb)
        out[f'apply_sums {tag}'] = (lambda s, g=g, y=y, k012=k012, gz=gz, gzr=gzr, nr=nr, hw=hw, c=c:
                                    lib.call('clamd_bn_bwd_apply_sums', ptr(g), c, ptr(y), c, ptr(k012), ptr(gz), c, ptr(gzr), nr, B, hw, hw, c, 0, s), 3 * mb)
        out[f'xform44 {tag}'] = (lambda s, gz=gz, vx=vx, hw=hw, c=c:
                                 lib.call('clamd_winograd44_transform_input', ptr(gz), c, None, None, ptr(vx), B, hw, hw, c, s), 3.25 * mb)
        out[f'_keep{tag}'] = (g, y, gz, sums, one, zero, k012, vec, dg, db, dcb, gzr, vx)
    return out


def ev():
    return torch.cuda.Event(enable_timing=True)


def run_pair(fg, bg, nbg):
    """fg reps on fg_stream beside nbg launches of bg on bg_stream (either may be None); returns (fg us per launch, bg us per launch)"""
    torch.cuda.synchronize()
    b0, b1, f0, f1, started = ev(), ev(), ev(), ev(), torch.cuda.Event()
    if bg is not None:
        with torch.cuda.stream(bg_stream):
            bg(bg_stream.cuda_stream)                 # the first launch: the foreground starts behind it
            started.record(bg_stream)
            b0.record(bg_stream)
            for _ in range(nbg):
                bg(bg_stream.cuda_stream)
            b1.record(bg_stream)
    if fg is not None:
        with torch.cuda.stream(fg_stream):
            if bg is not None:
                fg_stream.wait_event(started)
            f0.record(fg_stream)
            for _ in range(reps):
                fg(fg_stream.cuda_stream)
            f1.record(fg_stream)
    torch.cuda.synchronize()
    return (f0.elapsed_time(f1) / reps * 1e3 if fg is not None else 0.0, b0.elapsed_time(b1) / nbg * 1e3 if bg is not None else 0.0)


bgs = background(); keep_b = bgs.pop('_keep')
ps = passes()
keeps = [ps.pop(k) for k in list(ps) if k.startswith('_keep')]
for f in list(bgs.values()) + [p[0] for p in ps.values()]:      # warm up
    f(lib.stream_ptr())
torch.cuda.synchronize()
bg_alone = {n: run_pair(None, f, 20)[1] for n, f in bgs.items()}
print('background alone (us per launch): ' + ', '.join(f'{n} {t:.1f}' for n, t in bg_alone.items()))
print(f'{"pass":>22s} {"alone":>8s} {"TB/s":>5s} | ' + ' | '.join(f'{n:>30s}' for n in bgs))
for pn, (pf, mb) in ps.items():
    alone = min(run_pair(pf, None, 0)[0] for _ in range(3))
    cells = []
    for bn, bf in bgs.items():
        nbg = max(4, int(alone * reps * 4 / bg_alone[bn]) + 2)       # the background outlasts a 4x slower foreground
        t, tb = run_pair(pf, bf, nbg)
        # the background's launches that overlapped the foreground ran (tb * nbg - (nbg - n_ov) * alone) / n_ov: report the time it lost per foreground launch
        lost = (tb - bg_alone[bn]) * nbg / reps
        cells.append(f'{t:8.1f} us {t / alone:5.2f}x  bg -{lost:6.1f} us')
    print(f'{pn:>22s} {alone:8.1f} {mb / alone if mb else 0:5.2f} | ' + ' | '.join(cells))
