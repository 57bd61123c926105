"""GPU parity tests of the F(4x4,3x3) family with pre-transformed operands (csrc/wino44g.hip), through the C ABI, against the CPU oracle
(oracle/np_unet.py) on the same seeded inputs -- at the 2e-5 bound of every other fp32 kernel -- and against the F(2x4) kernels the engine
ran on these layers before (models/unet.py:28-33,50-55; loss.backward(), trainer.py:175)."""
import numpy as np
import pytest
import torch

from conftest import rel_l2
from oracle import np_unet as O
from test_kernels_gpu import C, dev, nhwc_with_segs, phys_map, rnd, stat_buf, sync  # noqa: F401 (C is a fixture)

pytestmark = pytest.mark.gpu

W44_SHAPES = [  # B, Cin segs, Cout, H, W (H, W multiples of 4; Cout_p % 64 == 0)
    (1, [(64, 64)], 64, 16, 32),                 # exactly one 16x32 tile block, 8 chunks
    (2, [(64, 64)], 128, 32, 16),                # the narrow block (32x16 pixels), two output slabs
    (2, [(40, 64), (50, 64)], 100, 24, 40),      # concat input, ragged blocks in both directions, padded output channels
    (3, [(256, 256)], 256, 16, 16),              # half-empty narrow blocks, 32 chunks, four slabs
    (1, [(96, 128)], 192, 36, 68),               # ragged with the wide block, three slabs
    (5, [(64, 64)], 640, 48, 64),                # 5*3*2 blocks x 10 slabs = 300 work items: the persistent loop and its cross-tile prefetch
]


@pytest.mark.parametrize('shape', W44_SHAPES, ids=lambda sh: f'{sh[0]}x{"+".join(str(a) for a, _ in sh[1])}->{sh[2]}@{sh[3]}x{sh[4]}')
def test_conv3x3_winograd44_pretransformed(C, shape):
    """Forward (+bias, ReLU, statistics rows) and data gradient: transform once (clamd_winograd44_transform_input), transform-free K loop
    (clamd_conv3x3_winograd44_pre).  Against the oracle's direct convolution; every element of V written; bit-reproducible; identical
    activations under every grid / block-order choice."""
    B, segs, cout, H, W = shape
    rng = np.random.default_rng(44)
    cin = sum(s[0] for s in segs)
    x = rnd(rng, B, cin, H, W)
    w = rnd(rng, cout, cin, 3, 3) * (1.0 / np.sqrt(9 * cin))
    b = rnd(rng, cout)
    cin_p, cout_p = sum(s[1] for s in segs), C.ops.cpad(cout)
    if cout_p % 64:
        cout_p = 64
    xt = nhwc_with_segs(C, x, segs, 0)
    wt, bt = dev(w), dev(b)
    wf = torch.zeros(36 * cout_p * cin_p, device='cuda')
    wd = torch.zeros(36 * cin_p * cout_p, device='cuda')
    bp = torch.zeros(cout_p, device='cuda')
    tab = C.ops.WinoPackTable(36); tab.conv3x3(wt, wf, wd, segs, cout); tab.finalize('cuda').run()
    pt = C.ops.PackTable(0); pt.vector(bt, bp, cout); pt.finalize('cuda').run(0)
    lib, ptr, s = C._lib, C._lib.ptr, C._lib.stream_ptr()
    L = lib.load()
    gz = rnd(rng, B, cout, H, W)
    gzt = C.ops.to_nhwc(dev(gz), 0, cp=cout_p)
    ref = O.relu_fwd(O.conv3x3_fwd(x, w, b))
    rgx = O.conv3x3_bwd(x, w, gz)[0]
    pm = phys_map(segs)
    y0 = None
    for tn in (None, lib.Tuning(wino_persist=0), lib.Tuning(cu_reserve=120), lib.Tuning(wino_band=1)):
        tp = tn.ref() if tn else None
        stats, rows = stat_buf(C, lib.OP_CONV3X3_WINOGRAD44, B, H, W, cin_p, cout_p, 0, tuning=tn)
        y = torch.full((B, H, W, cout_p), 7.0, device='cuda')
        v = torch.full((L.clamd_winograd44_input_elems(B, H, W, cin_p),), float('nan'), device='cuda')
        lib.call('clamd_winograd44_transform_input', ptr(xt), cin_p, None, None, ptr(v), B, H, W, cin_p, s)
        lib.call('clamd_conv3x3_winograd44_pre', ptr(v), ptr(wf), ptr(bp), ptr(y), cout_p, ptr(stats), rows, B, H, W, cin_p, cout_p, 1, tp, s)
        sync()
        assert not bool(torch.isnan(v).any()), 'the transform must write every element of V'
        err = rel_l2(C.ops.from_nhwc(y, cout, 0).cpu().numpy(), ref)
        assert err < 2e-5, err
        st = stats.double().sum(0).cpu().numpy()
        np.testing.assert_allclose(st[0, :cout], ref.sum((0, 2, 3)), rtol=1e-4, atol=1e-3)
        np.testing.assert_allclose(st[1, :cout], (ref ** 2).sum((0, 2, 3)), rtol=1e-4, atol=1e-3)
        assert float(y[..., cout:].abs().max()) == 0.0 if cout < cout_p else True
        if y0 is None:
            y0 = y
        assert torch.equal(y, y0), 'activations must not depend on the grid or the block order'
        stats_q = torch.full_like(stats, float('nan'))
        y_q = torch.full_like(y, 3.0)
        lib.call('clamd_conv3x3_winograd44_pre', ptr(v), ptr(wf), ptr(bp), ptr(y_q), cout_p, ptr(stats_q), rows, B, H, W, cin_p, cout_p, 1, tp, s)
        sync()
        assert torch.equal(y, y_q) and torch.equal(stats, stats_q), 'two identical launches must be bit-identical'
        # data gradient: the same two calls on the gradient tensor and the tap-flipped filters
        gx = torch.full((B, H, W, cin_p), 4.0, device='cuda')
        vg = torch.empty(L.clamd_winograd44_input_elems(B, H, W, cout_p), device='cuda')
        lib.call('clamd_winograd44_transform_input', ptr(gzt), cout_p, None, None, ptr(vg), B, H, W, cout_p, s)
        lib.call('clamd_conv3x3_winograd44_pre', ptr(vg), ptr(wd), None, ptr(gx), cin_p, None, 0, B, H, W, cout_p, cin_p, 0, tp, s)
        sync()
        got_gx = gx.cpu().numpy().transpose(0, 3, 1, 2)
        errg = rel_l2(got_gx[:, [p_ for p_, l in enumerate(pm) if l >= 0]], rgx)
        assert errg < 2e-5, errg
        pad = [p_ for p_, l in enumerate(pm) if l < 0]
        assert not pad or float(np.abs(got_gx[:, pad]).max()) == 0.0
    # and against the F(2x4) kernel on the same problem (both within 2e-5 of the oracle: within 4e-5 of each other)
    if H % 2 == 0 and cin_p >= 64 and cin_p % 32 == 0:
        wf24 = torch.zeros(24 * cout_p * cin_p, device='cuda')
        t24 = C.ops.WinoPackTable(24); t24.conv3x3(wt, wf24, None, segs, cout); t24.finalize('cuda').run()
        y24 = torch.empty_like(y0)
        lib.call('clamd_conv3x3_winograd24', ptr(xt), cin_p, ptr(wf24), ptr(bp), ptr(y24), cout_p, None, 0, B, H, W, cin_p, cout_p, 1, None, s)
        sync()
        assert rel_l2(y0.cpu().numpy(), y24.cpu().numpy()) < 1e-5
    # the BatchNorm in front of the convolution folded into the transform: x * scale + shift on load, zero padding AFTER the affine
    scale = dev(rng.standard_normal(cin_p).astype(np.float32))
    shift = dev(rng.standard_normal(cin_p).astype(np.float32))
    applied = torch.empty_like(xt)
    lib.call('clamd_bn_apply', ptr(xt), cin_p, ptr(scale), ptr(shift), ptr(applied), cin_p, None, 0, B, H, W, cin_p, 0, s)
    v_ref = torch.empty_like(v)
    v_fold = torch.full_like(v, float('nan'))
    lib.call('clamd_winograd44_transform_input', ptr(applied), cin_p, None, None, ptr(v_ref), B, H, W, cin_p, s)
    lib.call('clamd_winograd44_transform_input', ptr(xt), cin_p, ptr(scale), ptr(shift), ptr(v_fold), B, H, W, cin_p, s)
    sync()
    assert torch.equal(v_ref, v_fold)
    # refused shapes
    with pytest.raises(RuntimeError, match='multiples of 4'):
        lib.call('clamd_conv3x3_winograd44_pre', ptr(v), ptr(wf), ptr(bp), ptr(y0), cout_p, None, 0, B, H + 2, W, cin_p, cout_p, 1, None, s)
    with pytest.raises(RuntimeError, match='Cout_p % 64'):
        lib.call('clamd_conv3x3_winograd44_pre', ptr(v), ptr(wf), ptr(bp), ptr(y0), 32, None, 0, B, H, W, cin_p, 32, 1, None, s)


W44_WGRAD_SHAPES = [  # B, Cin segs, Cout, H, W (Cin_p, Cout_p multiples of 256)
    (2, [(256, 256)], 256, 16, 32),
    (1, [(100, 256), (130, 256)], 200, 12, 20),      # concat input with padding, padded output channels, ragged narrow blocks
    (3, [(256, 256)], 512, 32, 32),                  # several splits
    (2, [(256, 256)], 256, 20, 72),                  # ragged wide blocks in both directions
    (2, [(128, 128)], 128, 16, 32),                  # multiples of 128: the wave-level stream-K plan (one 128 x 128 wave tile per plane)
    (3, [(256, 256)], 128, 32, 32),                  # 128 x 256: two wave tiles per plane
    (1, [(60, 64), (50, 64)], 200, 24, 40),          # 256 x 128 with a concat input, padding on both sides, ragged blocks
]


@pytest.mark.parametrize('shape', W44_WGRAD_SHAPES, ids=lambda sh: f'{sh[0]}x{"+".join(str(a) for a, _ in sh[1])}->{sh[2]}@{sh[3]}x{sh[4]}')
def test_wgrad_winograd44_pretransformed(C, shape):
    """Weight gradient as the batched plane GEMM over the 36 planes of F(4x4,3x3): the x side is the forward image V read in place, the
    gradient side A6 dY A6^T is written once, fixed-order reduce with G6^T . G6.  Against the oracle at 2e-5, bit-reproducible, the same
    (to rounding) under another split plan and against the F(2x4) plane GEMM; the two-call form (transform on another stream) included."""
    B, segs, cout, H, W = shape
    rng = np.random.default_rng(45)
    cin = sum(s[0] for s in segs)
    x = rnd(rng, B, cin, H, W)
    w = rnd(rng, cout, cin, 3, 3) * (1.0 / np.sqrt(9 * cin))
    gz = rnd(rng, B, cout, H, W)
    cin_p, cout_p = sum(s[1] for s in segs), (cout + 127) // 128 * 128
    xt = nhwc_with_segs(C, x, segs, 0)
    gzt = C.ops.to_nhwc(dev(gz), 0, cp=cout_p)
    lib, ptr, s = C._lib, C._lib.ptr, C._lib.stream_ptr()
    L = lib.load()
    v = torch.empty(L.clamd_winograd44_input_elems(B, H, W, cin_p), device='cuda')
    lib.call('clamd_winograd44_transform_input', ptr(xt), cin_p, None, None, ptr(v), B, H, W, cin_p, s)
    yt = torch.full((L.clamd_wgrad_winograd44_pre_operand_elems(B, H, W, cout_p),), float('nan'), device='cuda')
    wsb = L.clamd_wgrad_winograd44_pre_workspace_bytes(B, H, W, cout_p, cin_p)
    ws = torch.empty(wsb // 4 + 4, device='cuda')
    c_seg0, c_seg0p = (segs[0][0], segs[0][1]) if len(segs) == 2 else (cin, cin_p)
    rgw = O.conv3x3_bwd(x, w, gz)[1]
    outs = []
    # wgrad_streamk: 1 (default) per launch, 0 the split-K plan in whole rounds of the chip, 2 the stream-K plane GEMM
    for tn in (None, None, lib.Tuning(cu_reserve=100), lib.Tuning(wgrad_streamk=0), lib.Tuning(wgrad_streamk=0, cu_reserve=100), lib.Tuning(wgrad_streamk=2),
               lib.Tuning(wgrad_streamk=2, cu_reserve=100)):
        gw = torch.full((cout, cin, 3, 3), 5.0, device='cuda')
        lib.call('clamd_wgrad_winograd44_pre', ptr(gzt), cout_p, ptr(v), ptr(yt), ptr(ws), wsb, ptr(gw), B, H, W,
                 cout_p, cin_p, cout, cin, cout, cout_p, c_seg0, c_seg0p, tn.ref() if tn else None, s)
        sync()
        outs.append(gw)
        err = rel_l2(gw.cpu().numpy(), rgw)
        assert err < 2e-5, err
    assert not bool(torch.isnan(yt).any())
    assert torch.equal(outs[0], outs[1]), 'two identical launches must be bit-identical'
    for o_ in outs[2:]:
        assert rel_l2(o_.cpu().numpy(), outs[0].cpu().numpy()) < 5e-6      # another plan: another summation order (measured 2.1e-6)
    # two-call form: the gradient-side transform alone, then the GEMM with gz == NULL
    yt2 = torch.full_like(yt, float('nan'))
    lib.call('clamd_wgrad_winograd44_pre_transform', ptr(gzt), cout_p, ptr(yt2), B, H, W, cout_p, s)
    gw3 = torch.full((cout, cin, 3, 3), 6.0, device='cuda')
    lib.call('clamd_wgrad_winograd44_pre', None, cout_p, ptr(v), ptr(yt2), ptr(ws), wsb, ptr(gw3), B, H, W,
             cout_p, cin_p, cout, cin, cout, cout_p, c_seg0, c_seg0p, None, s)
    sync()
    assert torch.equal(yt, yt2) and torch.equal(gw3, outs[0])
    # and against the F(2x4) plane GEMM
    if H % 2 == 0:
        v24 = torch.empty(L.clamd_winograd24_input_elems(B, H, W, cin_p), device='cuda')
        lib.call('clamd_winograd24_transform_input', ptr(xt), cin_p, None, None, ptr(v24), B, H, W, cin_p, s)
        yt24 = torch.empty(L.clamd_wgrad_winograd24_pre_operand_elems(B, H, W, cout_p), device='cuda')
        wsb24 = L.clamd_wgrad_winograd24_pre_workspace_bytes(B, H, W, cout_p, cin_p)
        ws24 = torch.empty(wsb24 // 4 + 4, device='cuda')
        gw24 = torch.empty(cout, cin, 3, 3, device='cuda')
        lib.call('clamd_wgrad_winograd24_pre', ptr(gzt), cout_p, ptr(v24), ptr(yt24), ptr(ws24), wsb24, ptr(gw24), B, H, W,
                 cout_p, cin_p, cout, cin, cout, cout_p, c_seg0, c_seg0p, None, s)
        sync()
        assert rel_l2(outs[0].cpu().numpy(), gw24.cpu().numpy()) < 1e-5
