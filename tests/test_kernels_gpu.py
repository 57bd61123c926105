"""GPU parity tests, one kernel family at a time, THROUGH THE C ABI (ctypes -> libclamd.so), checked against the
CPU oracle (oracle/np_unet.py) on the same seeded inputs and against golden vectors captured from the reference.

Tolerances: fp32 path (exact-fp32 MFMA) differs from the oracle only by summation order -> rel L2 < 2e-5.
bf16 path: the oracle is run on bf16-rounded inputs/weights, so what remains is accumulation order plus the bf16
rounding of the stored output (2^-9 relative per element) -> rel L2 < 6e-3.
"""
import os

import numpy as np
import pytest
import torch

from conftest import rel_l2
from oracle import np_unet as O

pytestmark = pytest.mark.gpu

DT = [('fp32', 0), ('bf16', 1), ('bf16x3', 2)]
TOL = {0: 2e-5, 1: 6e-3, 2: 6e-5}     # bf16x3: split-bf16 products carry ~2^-17 relative error


@pytest.fixture(scope='module')
def C():
    import continual_learning_amd as C
    C._lib.load()
    return C


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a)).to('cuda', dtype)


def rb(a, dcode):
    """Round a numpy fp32 array to what the compute dtype stores and back: identity for fp32, rne bf16 for bf16,
    hi + lo (two bf16, ~17 mantissa bits) for bf16x3."""
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    if dcode == 0:
        return t.numpy()
    hi = t.to(torch.bfloat16).to(torch.float32)
    if dcode == 1:
        return hi.numpy()
    return (hi + (t - hi).to(torch.bfloat16).to(torch.float32)).numpy()


def nf(C, t, dcode):
    """Device NHWC activation (whole tensor or a 16-channel-aligned slice) -> fp32 torch tensor of its logical values."""
    return C.ops.split_decode(t) if dcode == 2 else t.float()


def nd(C, a, dcode):
    """fp32 NHWC numpy array (padded channel count) -> device tensor in the storage layout of the compute dtype."""
    t = dev(a)
    return C.ops.split_encode(t) if dcode == 2 else t.to(C.ops.TORCH_DT[dcode])


def rnd(rng, *shape):
    return rng.standard_normal(shape).astype(np.float32)


def phys_map(segs):
    """[(logical, physical), ...] -> list: physical index -> logical index or -1."""
    out, base = [], 0
    for lg, ph in segs:
        out += [base + i if i < lg else -1 for i in range(ph)]
        base += lg
    return out


def nhwc_with_segs(C, x_nchw, segs, dcode):
    """Builds the padded NHWC tensor of a (possibly two-segment concat) activation."""
    pm = phys_map(segs)
    B, _, H, W = x_nchw.shape
    full = np.zeros((B, len(pm), H, W), np.float32)
    for p, l in enumerate(pm):
        if l >= 0:
            full[:, p] = x_nchw[:, l]
    return C.ops.to_nhwc(dev(full), dcode, cp=len(pm))


def sync():
    torch.cuda.synchronize()


# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('name,dcode', DT)
def test_layout_roundtrip(C, name, dcode):
    rng = np.random.default_rng(0)
    x = rb(rnd(rng, 2, 5, 6, 10), dcode)
    t = C.ops.to_nhwc(dev(x), dcode)
    assert t.shape == (2, 6, 10, 32)
    back = C.ops.from_nhwc(t, 5, dcode).cpu().numpy()
    assert np.array_equal(back, x)
    assert float(nf(C, t, dcode)[..., 5:].abs().max()) == 0.0        # padded channels are zero


@pytest.mark.parametrize('name,dcode', [d for d in DT if d[1] != 0])
@pytest.mark.parametrize('cout,segs,scaled', [(128, [(64, 64)], False), (96, [(64, 64), (40, 64)], False), (64, [(128, 128)], True), (21, [(72, 96)], True)])
def test_pack_3x3_whole_run_path(C, name, dcode, cout, segs, scaled):
    """3x3 filters into the K-chunk-major layouts: the pack kernel's whole-run path (288 contiguous floats per row of a tile, 16-byte loads
    when aligned, 16-byte stores) against the layout written out in numpy -- full tiles, a second segment that ends inside a tile, padded
    output channels, the folded BatchNorm scale; forward and (tap-flipped, transposed) data-gradient layouts, bf16 and hi/lo pair storage."""
    rng = np.random.default_rng(5)
    T = C.ops.TORCH_DT[dcode]
    cin, cin_p, cout_p = sum(a for a, _ in segs), sum(b for _, b in segs), C.ops.cpad(cout)
    w = rnd(rng, cout, cin, 3, 3)
    ks = (rnd(rng, cin_p) if scaled else None)
    wt, kst = dev(w), (dev(ks) if scaled else None)
    if scaled:
        pm_ = phys_map(segs)
        kst[torch.tensor([i for i, l in enumerate(pm_) if l < 0], dtype=torch.long, device='cuda')] = float('nan')      # padded entries must never matter
    esz = 2 if dcode == 2 else 1                              # hi/lo pairs: 4 bytes per element = two bf16
    wf = torch.zeros(9 * cout_p * cin_p * esz, dtype=torch.bfloat16, device='cuda')
    wd = torch.zeros(9 * cin_p * cout_p * esz, dtype=torch.bfloat16, device='cuda')
    tab = C.ops.PackTable(dcode)
    tab.conv3x3(wt, wf.view(T) if dcode != 2 else wf, None, segs, cout, kscale=kst)
    tab.conv3x3(wt, None, wd.view(T) if dcode != 2 else wd, segs, cout)
    wf2, wd2 = torch.zeros_like(wf), torch.zeros_like(wd)      # both layouts from ONE job (PackJob::dst_t): the same bits, the scale on the forward one only
    tab.conv3x3(wt, wf2.view(T) if dcode != 2 else wf2, wd2.view(T) if dcode != 2 else wd2, segs, cout, kscale=kst)
    assert len(tab.jobs) == 3
    tab.finalize('cuda').run(dcode)
    sync()
    assert torch.equal(wf2.view(torch.int16), wf.view(torch.int16)) and torch.equal(wd2.view(torch.int16), wd.view(torch.int16))
    pm = phys_map(segs)
    exp_f = np.zeros((9, cout_p, cin_p), np.float32); exp_d = np.zeros((9, cin_p, cout_p), np.float32)
    for t in range(9):
        ky, kx = divmod(t, 3)
        for kp, kl in enumerate(pm):
            if kl >= 0:
                exp_f[t, :cout, kp] = w[:, kl, ky, kx] * (ks[kp] if scaled else np.float32(1))
                exp_d[t, kp, :cout] = w[:, kl, 2 - ky, 2 - kx]
    kc = tab.kc

    def chunked(buf, n, k):       # [K/kc][tap][n][kc] -> [tap][n][k]
        if dcode == 2:
            raw = buf.view(torch.int16).cpu().numpy().view(np.uint16).reshape(k // 16, 9, n, 2, 16)
            f = (raw.astype(np.uint32) << 16).view(np.float32)
            return [a.transpose(1, 2, 0, 3).reshape(9, n, k) for a in (f[..., 0, :], f[..., 0, :] + f[..., 1, :])]
        return [buf.float().cpu().numpy().reshape(k // kc, 9, n, kc).transpose(1, 2, 0, 3).reshape(9, n, k)] * 2
    for buf, exp, n, k in ((wf, exp_f, cout_p, cin_p), (wd, exp_d, cin_p, cout_p)):
        hi, full = chunked(buf, n, k)
        assert np.array_equal(hi, rb(exp, 1))
        if dcode == 2:
            assert np.abs(full - exp).max() <= 2.0 ** -16 * np.abs(exp).max()


@pytest.mark.parametrize('name,dcode', DT)
def test_pack_layouts(C, name, dcode):
    rng = np.random.default_rng(1)
    T = C.ops.TORCH_DT[dcode]
    cpad = C.ops.cpad
    # conv 3x3 with a two-segment (concat) input: logical 5+5 channels, physical 32+32
    w = rnd(rng, 7, 10, 3, 3)
    segs = [(5, 32), (5, 32)]
    wf = torch.zeros(9 * 32 * 64, dtype=T, device='cuda')
    wd = torch.zeros(9 * 64 * 32, dtype=T, device='cuda')
    wt = dev(w)
    wc = rnd(rng, 6, 3, 2, 2); wct = dev(wc)
    cf = torch.zeros(4 * 32 * 32, dtype=T, device='cuda'); cd = torch.zeros(32 * 4 * 32, dtype=T, device='cuda')
    wh = rnd(rng, 21, 40, 1, 1); wht = dev(wh)
    hf = torch.zeros(32 * 64, dtype=T, device='cuda'); hd = torch.zeros(64 * 32, dtype=T, device='cuda')
    b = rnd(rng, 7); bt = dev(b); bp = torch.full((32,), 9.0, device='cuda')
    tab = C.ops.PackTable(dcode)
    tab.conv3x3(wt, wf, wd, segs, 7)
    tab.convT(wct, cf, cd, 6, 3)
    tab.head(wht, hf, hd, 40, 21)
    tab.vector(bt, bp, 7)
    tab.finalize('cuda').run(dcode)
    sync()
    if dcode == 2:
        # split layout: every 16-channel chunk of a row is [16 x bf16 hi][16 x bf16 lo]; hi + lo reproduces the fp32
        # weight to ~2^-17 and hi is exactly rne_bf16(w)
        def unsplit(buf, *shape):
            raw = buf.view(torch.int16).cpu().numpy().view(np.uint16).reshape(*shape[:-1], shape[-1] // 16, 2, 16)
            f = (raw.astype(np.uint32) << 16).view(np.float32)
            return f[..., 0, :].reshape(shape), (f[..., 0, :] + f[..., 1, :]).reshape(shape)
        hi, full = unsplit(wf, 4, 9, 32, 16)                 # [K-chunk][tap][n][16]
        hi, full = [a.transpose(1, 2, 0, 3).reshape(9, 32, 64) for a in (hi, full)]
        pm = phys_map(segs)
        exp_f = np.zeros((9, 32, 64), np.float32)
        for t in range(9):
            for kp, kl in enumerate(pm):
                if kl >= 0:
                    exp_f[t, :7, kp] = w[:, kl, t // 3, t % 3]
        assert np.array_equal(hi, rb(exp_f, 1))
        assert np.abs(full - exp_f).max() <= 2.0 ** -16 * np.abs(exp_f).max()
        exp_b = np.zeros(32, np.float32); exp_b[:7] = b
        assert np.array_equal(bp.cpu().numpy(), exp_b)
        return
    pm = phys_map(segs)
    exp_f = np.zeros((9, 32, 64), np.float32); exp_d = np.zeros((9, 64, 32), np.float32)
    for t in range(9):
        ky, kx = divmod(t, 3)
        for kp, kl in enumerate(pm):
            if kl < 0:
                continue
            exp_f[t, :7, kp] = w[:, kl, ky, kx]
            exp_d[t, kp, :7] = w[:, kl, 2 - ky, 2 - kx]
    kc = tab.kc                                           # K-chunk-major: [K/kc][tap][n][kc]
    got_f = wf.float().cpu().numpy().reshape(64 // kc, 9, 32, kc).transpose(1, 2, 0, 3).reshape(9, 32, 64)
    got_d = wd.float().cpu().numpy().reshape(32 // kc, 9, 64, kc).transpose(1, 2, 0, 3).reshape(9, 64, 32)
    assert np.array_equal(got_f, rb(exp_f, dcode))
    assert np.array_equal(got_d, rb(exp_d, dcode))
    exp_cf = np.zeros((4, 32, 32), np.float32); exp_cd = np.zeros((32, 4, 32), np.float32)
    for q in range(4):
        exp_cf[q, :3, :6] = wc[:, :, q // 2, q % 2].T
        exp_cd[:6, q, :3] = wc[:, :, q // 2, q % 2]
    assert np.array_equal(cf.float().cpu().numpy().reshape(4, 32, 32), rb(exp_cf, dcode))
    assert np.array_equal(cd.float().cpu().numpy().reshape(32, 4, 32), rb(exp_cd, dcode))
    exp_hf = np.zeros((32, 64), np.float32); exp_hf[:21, :40] = wh[:, :, 0, 0]
    assert np.array_equal(hf.float().cpu().numpy().reshape(32, 64), rb(exp_hf, dcode))
    assert np.array_equal(hd.float().cpu().numpy().reshape(64, 32), rb(exp_hf.T, dcode))
    exp_b = np.zeros(32, np.float32); exp_b[:7] = b
    assert np.array_equal(bp.cpu().numpy(), exp_b)
    assert cpad(21) == 32 and cpad(64) == 64 and cpad(65) == 128


def stat_buf(C, op, B, H, W, cin_p, cout_p, dcode, nk=2, fused=False, tuning=None):
    """Partial-row buffer of a statistics-producing launch, sized by clamd_stat_rows and pre-filled with NaN: a row the
    launch fails to write shows up in every sum."""
    rows = C._lib.stat_rows(op, B, H, W, cin_p, cout_p, dcode, fused, tuning)
    return torch.full((rows, nk, cout_p), float('nan'), device='cuda'), rows


CONV_SHAPES = [  # B, Cin segs, Cout, H, W
    (2, [(5, 32)], 7, 8, 12),                 # tiny ragged image, heavy channel padding
    (1, [(64, 64)], 64, 16, 16),              # exact 16x16 tile (the centre of the full-size net)
    (2, [(20, 32), (20, 32)], 130, 40, 64),   # concat input, 3 Cout tiles with a ragged last one, TW=32 tiles
    (1, [(3, 32)], 64, 64, 96),               # first layer shape class
    (3, [(128, 128)], 32, 4, 4),              # image smaller than a tile
]


def _conv_case(C, rng, B, segs, cout, H, W, dcode):
    cin = sum(s[0] for s in segs)
    x = rb(rnd(rng, B, cin, H, W), dcode)
    w = rb(rnd(rng, cout, cin, 3, 3) * (1.0 / np.sqrt(9 * cin)), dcode)
    b = rnd(rng, cout)
    T = C.ops.TORCH_DT[dcode]
    cin_p, cout_p = sum(s[1] for s in segs), C.ops.cpad(cout)
    xt = nhwc_with_segs(C, x, segs, dcode)
    wt, bt = dev(w), dev(b)
    wf = torch.zeros(9 * cout_p * cin_p, dtype=T, device='cuda')
    wd = torch.zeros(9 * cin_p * cout_p, dtype=T, device='cuda')
    bp = torch.zeros(cout_p, device='cuda')
    tab = C.ops.PackTable(dcode)
    tab.conv3x3(wt, wf, wd, segs, cout)
    tab.vector(bt, bp, cout)
    tab.finalize('cuda').run(dcode)
    return x, w, b, xt, wf, wd, bp, cin_p, cout_p


@pytest.mark.parametrize('name,dcode', DT)
@pytest.mark.parametrize('shape', CONV_SHAPES)
@pytest.mark.parametrize('m_fastest', [0, 1])
def test_conv3x3_fwd_relu_stats(C, name, dcode, shape, m_fastest):
    B, segs, cout, H, W = shape
    rng = np.random.default_rng(10)
    x, w, b, xt, wf, wd, bp, cin_p, cout_p = _conv_case(C, rng, B, segs, cout, H, W, dcode)
    lib, ptr, s = C._lib, C._lib.ptr, C._lib.stream_ptr()
    T = C.ops.TORCH_DT[dcode]
    y = torch.full((B, H, W, cout_p), 7.0, dtype=T, device='cuda')
    stats, rows = stat_buf(C, lib.OP_CONV3X3, B, H, W, cin_p, cout_p, dcode)
    lib.call('clamd_conv3x3', ptr(xt), cin_p, ptr(wf), ptr(bp), ptr(y), cout_p, ptr(stats), None, None, rows, B, H, W, cin_p, cout_p, 1,
             m_fastest, dcode, None, s)
    sync()
    ref = O.relu_fwd(O.conv3x3_fwd(x, w, b))
    got = C.ops.from_nhwc(y, cout, dcode).cpu().numpy()
    assert rel_l2(got, ref) < TOL[dcode]
    st = stats.double().sum(0).cpu().numpy()
    np.testing.assert_allclose(st[0, :cout], ref.sum((0, 2, 3)), rtol=2e-3 if dcode == 1 else 1e-4, atol=1e-2 if dcode == 1 else 1e-3)
    np.testing.assert_allclose(st[1, :cout], (ref ** 2).sum((0, 2, 3)), rtol=4e-3 if dcode == 1 else 1e-4, atol=1e-2 if dcode == 1 else 1e-3)
    assert float(nf(C, y, dcode)[..., cout:].abs().max()) == 0.0 if cout < cout_p else True


@pytest.mark.parametrize('name,dcode', DT)
@pytest.mark.parametrize('shape', CONV_SHAPES)
def test_conv3x3_dgrad_and_wgrad(C, name, dcode, shape):
    B, segs, cout, H, W = shape
    rng = np.random.default_rng(11)
    x, w, b, xt, wf, wd, bp, cin_p, cout_p = _conv_case(C, rng, B, segs, cout, H, W, dcode)
    cin = x.shape[1]
    gz = rb(rnd(rng, B, cout, H, W), dcode)
    gzt = C.ops.to_nhwc(dev(gz), dcode)
    lib, ptr, s = C._lib, C._lib.ptr, C._lib.stream_ptr()
    T = C.ops.TORCH_DT[dcode]
    gx = torch.zeros(B, H, W, cin_p, dtype=T, device='cuda')
    # the data-gradient launch also accumulates the 5 per-channel sums of the downstream ReLU/BN backward (fused epilogue)
    pm_ = phys_map(segs)
    yb = np.maximum(rnd(rng, B, cin, H, W), 0)
    ybt = nhwc_with_segs(C, rb(yb, dcode), segs, dcode)
    bsums, brows = stat_buf(C, lib.OP_CONV3X3, B, H, W, cout_p, cin_p, dcode, nk=5, fused=True)
    lib.call('clamd_conv3x3', ptr(gzt), cout_p, ptr(wd), None, ptr(gx), cin_p, None, ptr(ybt), ptr(bsums), brows, B, H, W, cout_p, cin_p, 0, 0,
             dcode, None, s)
    wsb = lib.load().clamd_wgrad_workspace_bytes(0, B, H, W, cout_p, cin_p, dcode)
    ws = torch.empty(wsb // 4 + 4, device='cuda')
    gw = torch.full((cout, cin, 3, 3), 5.0, device='cuda')
    c_seg0, c_seg0p = (segs[0][0], segs[0][1]) if len(segs) == 2 else (cin, cin_p)
    lib.call('clamd_wgrad', 0, ptr(gzt), cout_p, ptr(xt), cin_p, ptr(ws), wsb, ptr(gw), B, H, W, cout_p, cin_p, cout, cin,
             cout, cout_p, c_seg0, c_seg0p, dcode, None, s)
    sync()
    rgx, rgw, _ = O.conv3x3_bwd(x, w, gz)
    pm = phys_map(segs)
    got_gx = nf(C, gx, dcode).cpu().numpy().transpose(0, 3, 1, 2)[:, [p for p, l in enumerate(pm) if l >= 0]]
    assert rel_l2(got_gx, rgx) < TOL[dcode]
    assert rel_l2(gw.cpu().numpy(), rgw) < (6e-5 if dcode == 2 else 2e-5)   # wgrad output is fp32 in every path
    ybr = rb(yb, dcode)
    pos = (ybr > 0).astype(np.float32)
    want = np.stack([rgx.sum((0, 2, 3)), (rgx * ybr).sum((0, 2, 3)), (rgx * pos).sum((0, 2, 3)), pos.sum((0, 2, 3)), ybr.sum((0, 2, 3))])
    got = bsums.sum(0).cpu().numpy()[:, [p for p, l in enumerate(pm_) if l >= 0]]
    nsums = lib.load().clamd_conv3x3_bn_sums(B, H, W, cout_p, cin_p, dcode, None)
    assert nsums in (2, 5) and (nsums == 5 or dcode == 1)
    assert lib.load().clamd_conv3x3_bn_sums(B, H, W, cout_p, cin_p, dcode, lib.Tuning(igemm_pws=0).ref()) == 5      # the other structures: all five
    if nsums == 2:      # the persistent bf16 kernel: sum g and sum g y only, rows 2-4 written as NaN (clamd_bn_bwd_apply_sums gives d conv-bias;
        #                 a finalize that still asks for dbias from these rows gets NaN, not a silent zero)
        bs = bsums.cpu().numpy()
        assert not np.isnan(bs[:, :2]).any() and np.isnan(bs[:, 2:]).all()
        want, got = want[:2], got[:2]
    scale = np.abs(want).max(1, keepdims=True) + 1e-6
    assert np.abs(got - want).max() <= (2e-2 if dcode == 1 else 1e-3) * scale.max(), np.abs(got - want).max()
    if nsums == 5:
        np.testing.assert_allclose(got[3:], want[3:], rtol=1e-5, atol=1e-3)


# Every structure of the 3x3 kernels (baseline two-workgroups-per-CU, producer/consumer with 128/256/512-pixel tiles,
# persistent producer/consumer) is forced in turn through a per-call clamd_tuning and must (a) match the oracle, (b) give
# bit-identical activations -- they share the tile, the LDS image and the summation order of every output element -- and
# (c) write bit-identical statistics rows when launched twice (no float atomics).
VARIANT_SHAPES = [  # B, Cin, Cout, H, W
    (2, 64, 64, 40, 64),      # short K, ragged tile rows
    (1, 256, 96, 16, 16),     # long K, 16-wide tiles, ragged Cout tile
    (3, 128, 130, 24, 40),    # 3 Cout slabs (persistent kernel: several workgroups per slab), ragged in x and y
    (5, 64, 40, 128, 160),    # 400 tiles > CUs: persistent workgroups walk several tiles (resident filter slab in bf16)
]
CONV_VARIANTS = [('igemm_pws', 0, 'igemm_ws', 0), ('igemm_pws', 0, 'igemm_ws', 1), ('igemm_pws', 0, 'igemm_ws', 3),
                 ('igemm_pws', 0, 'igemm_ws', 4), ('igemm_pws', 2, 'igemm_ws', 2), ('igemm_pws', 2, 'pws_wres', 0),
                 ('igemm_pws', 1, 'pws_wres', 1)]


@pytest.mark.parametrize('name,dcode', DT)
@pytest.mark.parametrize('shape', VARIANT_SHAPES)
def test_conv3x3_kernel_structures_agree(C, name, dcode, shape):
    B, cin, cout, H, W = shape
    rng = np.random.default_rng(12)
    segs = [(cin, C.ops.cpad(cin))]
    x, w, b, xt, wf, wd, bp, cin_p, cout_p = _conv_case(C, rng, B, segs, cout, H, W, dcode)
    lib, ptr, s = C._lib, C._lib.ptr, C._lib.stream_ptr()
    T = C.ops.TORCH_DT[dcode]
    ref = O.relu_fwd(O.conv3x3_fwd(x, w, b))
    first, first_cl = None, None
    for k1, v1, k2, v2 in CONV_VARIANTS:
        tn = lib.Tuning(**{k1: v1, k2: v2})
        y = torch.full((B, H, W, cout_p), 7.0, dtype=T, device='cuda')
        stats, rows = stat_buf(C, lib.OP_CONV3X3, B, H, W, cin_p, cout_p, dcode, tuning=tn)
        lib.call('clamd_conv3x3', ptr(xt), cin_p, ptr(wf), ptr(bp), ptr(y), cout_p, ptr(stats), None, None, rows, B, H, W, cin_p,
                 cout_p, 1, 0, dcode, tn.ref(), s)
        stats2 = torch.full_like(stats, float('nan'))
        lib.call('clamd_conv3x3', ptr(xt), cin_p, ptr(wf), ptr(bp), ptr(y), cout_p, ptr(stats2), None, None, rows, B, H, W, cin_p,
                 cout_p, 1, 0, dcode, tn.ref(), s)
        sync()
        assert torch.equal(stats, stats2), f'{k1}={v1} {k2}={v2}: statistics rows differ between two identical launches'
        got = C.ops.from_nhwc(y, cout, dcode).cpu().numpy()
        assert rel_l2(got, ref) < TOL[dcode], (k1, v1, k2, v2)
        st = stats.double().sum(0).cpu().numpy()
        np.testing.assert_allclose(st[0, :cout], ref.sum((0, 2, 3)), rtol=2e-3 if dcode == 1 else 1e-4, atol=1e-2 if dcode == 1 else 1e-3)
        np.testing.assert_allclose(st[1, :cout], (ref ** 2).sum((0, 2, 3)), rtol=4e-3 if dcode == 1 else 1e-4, atol=1e-2 if dcode == 1 else 1e-3)
        # bf16, persistent kernel (channels-in-the-lane epilogue): the same MFMA chains started at the bias
        # instead of at zero with the bias added last -- the same sum rounded at another place, so those launches agree bit for bit among
        # themselves and with the other structures to the last bit of the stored bf16 value
        cl = dcode == 1 and k1 == 'igemm_pws' and v1 > 0
        if first is None:
            first = y.clone()
        elif cl and first_cl is None:
            first_cl = y.clone()
            d = (first.float() - y.float()).abs()
            assert float((d / first.float().abs().clamp_min(1e-3)).max()) <= 2.0 ** -7 and float((d > 0).float().mean()) < 0.02, (k1, v1, k2, v2)
        else:
            assert torch.equal(first_cl if cl else first, y), f'{k1}={v1} {k2}={v2} changed the activations'
    # a wrong row count is refused, not silently mis-summed
    with pytest.raises(RuntimeError, match='partial rows'):
        lib.call('clamd_conv3x3', ptr(xt), cin_p, ptr(wf), ptr(bp), ptr(y), cout_p, ptr(stats), None, None, rows + 1, B, H, W, cin_p,
                 cout_p, 1, 0, dcode, tn.ref(), s)


@pytest.mark.parametrize('name,dcode', DT)
@pytest.mark.parametrize('shape', [(2, 64, 64, 40, 64), (1, 256, 96, 16, 16), (5, 40, 130, 24, 40)])
def test_wgrad_kernel_structures_agree(C, name, dcode, shape):
    """wgrad: two-workgroups-per-CU vs producer/consumer kernel (different split-K widths: equal up to rounding)."""
    B, cin, cout, H, W = shape
    rng = np.random.default_rng(13)
    x = rb(rnd(rng, B, cin, H, W), dcode)
    gz = rb(rnd(rng, B, cout, H, W), dcode)
    xt, gzt = C.ops.to_nhwc(dev(x), dcode), C.ops.to_nhwc(dev(gz), dcode)
    cin_p, cout_p = C.ops.cpad(cin), C.ops.cpad(cout)
    lib, ptr, s = C._lib, C._lib.ptr, C._lib.stream_ptr()
    wsb = lib.load().clamd_wgrad_workspace_bytes(0, B, H, W, cout_p, cin_p, dcode)
    ws = torch.empty(wsb // 4 + 4, device='cuda')
    rgw = O.conv3x3_bwd(x, np.zeros((cout, cin, 3, 3), np.float32), gz)[1]
    for v in (0, 1, 2):          # 2 = producer/consumer kernel with LDS-DMA staging forced (bf16 only; same as 1 otherwise)
        tn = lib.Tuning(wgrad_ws=min(v, 1), wgrad_dma=2 if v == 2 else 0)
        gw = torch.full((cout, cin, 3, 3), 5.0, device='cuda')
        lib.call('clamd_wgrad', 0, ptr(gzt), cout_p, ptr(xt), cin_p, ptr(ws), wsb, ptr(gw), B, H, W, cout_p, cin_p, cout, cin,
                 cout, cout_p, cin, cin_p, dcode, tn.ref(), s)
        gw2 = torch.full((cout, cin, 3, 3), 5.0, device='cuda')
        lib.call('clamd_wgrad', 0, ptr(gzt), cout_p, ptr(xt), cin_p, ptr(ws), wsb, ptr(gw2), B, H, W, cout_p, cin_p, cout, cin,
                 cout, cout_p, cin, cin_p, dcode, tn.ref(), s)
        sync()
        assert torch.equal(gw, gw2), f'wgrad variant {v} is not bit-reproducible'
        assert rel_l2(gw.cpu().numpy(), rgw) < (6e-5 if dcode == 2 else 2e-5), v


WINO_SHAPES = [  # B, Cin segs, Cout, H, W (H, W even)
    (1, [(64, 64)], 64, 16, 16),              # exactly one workgroup tile
    (2, [(20, 32), (20, 32)], 130, 40, 64),   # concat input, three Cout slabs with a ragged last one, ragged tiles in y
    (2, [(5, 32)], 7, 8, 12),                 # image smaller than a tile, heavy channel padding
    (1, [(128, 128)], 96, 34, 18),            # ragged in both directions
    (3, [(32, 32)], 64, 128, 128),            # 192 workgroups: the 16x16-pixel tile variant (smaller grids take 8x16)
]


def _random_wino_shapes(n, seed):
    """Even, ragged image sizes, one- or two-segment (concat) inputs, any channel counts."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        B = int(rng.integers(1, 4))
        H, W = 2 * int(rng.integers(4, 37)), 2 * int(rng.integers(4, 37))
        cout = int(rng.integers(1, 200))
        if rng.random() < 0.4:
            c1, c2 = int(rng.integers(1, 80)), int(rng.integers(1, 80))
            pp = max(32, 1 << (max(c1, c2) - 1).bit_length())                   # equal halves, as in the UNet's concat buffers
            segs = [(c1, pp), (c2, pp)]
        else:
            c = int(rng.integers(1, 150))
            segs = [(c, max(32, 1 << (c - 1).bit_length()))]
        out.append((B, segs, cout, H, W))
    return out


@pytest.mark.parametrize('shape', WINO_SHAPES + _random_wino_shapes(int(os.environ.get('WINO_SWEEP', '8')), 99),
                         ids=lambda sh: f'{sh[0]}x{"+".join(str(a) for a, _ in sh[1])}->{sh[2]}@{sh[3]}x{sh[4]}')
def test_conv3x3_winograd_fp32(C, shape):
    """Winograd F(2x2,3x3) forward (+bias, ReLU, BN statistics) and data gradient, fp32, against the oracle's direct
    convolution at the SAME bound as the direct kernels (fp32 transforms add ~1e-7 relative error)."""
    B, segs, cout, H, W = shape
    rng = np.random.default_rng(21)
    cin = sum(s[0] for s in segs)
    x = rnd(rng, B, cin, H, W)
    w = rnd(rng, cout, cin, 3, 3) * (1.0 / np.sqrt(9 * cin))
    b = rnd(rng, cout)
    cin_p, cout_p = sum(s[1] for s in segs), C.ops.cpad(cout)
    xt = nhwc_with_segs(C, x, segs, 0)
    wt, bt = dev(w), dev(b)
    wf = torch.zeros(16 * cout_p * cin_p, device='cuda')
    wd = torch.zeros(16 * cin_p * cout_p, device='cuda')
    bp = torch.zeros(cout_p, device='cuda')
    tab = C.ops.WinoPackTable(); tab.conv3x3(wt, wf, wd, segs, cout); tab.finalize('cuda').run()
    pt = C.ops.PackTable(0); pt.vector(bt, bp, cout); pt.finalize('cuda').run(0)
    lib, ptr, s = C._lib, C._lib.ptr, C._lib.stream_ptr()
    y = torch.full((B, H, W, cout_p), 7.0, device='cuda')
    stats, rows = stat_buf(C, lib.OP_CONV3X3_WINOGRAD, B, H, W, cin_p, cout_p, 0)
    lib.call('clamd_conv3x3_winograd', ptr(xt), cin_p, ptr(wf), ptr(bp), ptr(y), cout_p, ptr(stats), rows, B, H, W, cin_p, cout_p, 1, None, s)
    gz = rnd(rng, B, cout, H, W)
    gzt = C.ops.to_nhwc(dev(gz), 0)
    gx = torch.full((B, H, W, cin_p), 3.0, device='cuda')
    lib.call('clamd_conv3x3_winograd', ptr(gzt), cout_p, ptr(wd), None, ptr(gx), cin_p, None, 0, B, H, W, cout_p, cin_p, 0, None, s)
    sync()
    ref = O.relu_fwd(O.conv3x3_fwd(x, w, b))
    assert rel_l2(C.ops.from_nhwc(y, cout, 0).cpu().numpy(), ref) < TOL[0]
    st = stats.double().sum(0).cpu().numpy()
    np.testing.assert_allclose(st[0, :cout], ref.sum((0, 2, 3)), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(st[1, :cout], (ref ** 2).sum((0, 2, 3)), rtol=1e-4, atol=1e-3)
    assert float(y[..., cout:].abs().max()) == 0.0 if cout < cout_p else True
    rgx = O.conv3x3_bwd(x, w, gz)[0]
    pm = phys_map(segs)
    got_gx = gx.cpu().numpy().transpose(0, 3, 1, 2)
    assert rel_l2(got_gx[:, [p_ for p_, l in enumerate(pm) if l >= 0]], rgx) < TOL[0]
    pad = [p_ for p_, l in enumerate(pm) if l < 0]
    assert not pad or float(np.abs(got_gx[:, pad]).max()) == 0.0
    # block order, persistence and tile height are scheduling choices: bit-identical activations under every setting; the
    # statistics rows are per workgroup (persistent grid) or per tile: same totals
    for key, val in (('wino_band', 1), ('wino_band', 32), ('wino_persist', 0), ('wino_mt', 1), ('wino_mt', 2), ('cu_reserve', 37)):
        tn = lib.Tuning(**{key: val})
        y2 = torch.full((B, H, W, cout_p), 7.0, device='cuda')
        stats2, rows2 = stat_buf(C, lib.OP_CONV3X3_WINOGRAD, B, H, W, cin_p, cout_p, 0, tuning=tn)
        lib.call('clamd_conv3x3_winograd', ptr(xt), cin_p, ptr(wf), ptr(bp), ptr(y2), cout_p, ptr(stats2), rows2, B, H, W, cin_p, cout_p, 1,
                 tn.ref(), s)
        sync()
        assert torch.equal(y, y2), (key, val)
        if rows2 == rows:
            assert torch.equal(stats, stats2), (key, val)
        else:       # another tile height: other rows, same totals
            np.testing.assert_allclose(stats2.double().sum(0).cpu().numpy(), stats.double().sum(0).cpu().numpy(), rtol=1e-5, atol=1e-3)
    # weight gradient by Winograd
    wsb = lib.load().clamd_wgrad_winograd_workspace_bytes(cout_p, cin_p)
    ws = torch.empty(wsb // 4 + 4, device='cuda')
    gw = torch.full((cout, cin, 3, 3), 5.0, device='cuda')
    c_seg0, c_seg0p = (segs[0][0], segs[0][1]) if len(segs) == 2 else (cin, cin_p)
    lib.call('clamd_wgrad_winograd', ptr(gzt), cout_p, ptr(xt), cin_p, ptr(ws), wsb, ptr(gw), B, H, W, cout_p, cin_p, cout, cin,
             cout, cout_p, c_seg0, c_seg0p, None, s)
    sync()
    rgw = O.conv3x3_bwd(x, w, gz)[1]
    assert rel_l2(gw.cpu().numpy(), rgw) < 2e-5


W24_SHAPES = [  # B, Cin segs, Cout, H, W (H even, W % 4 == 0)
    (1, [(64, 64)], 64, 16, 16),              # 16x16-pixel workgroup tile (images narrower than 32)
    (1, [(64, 64)], 64, 8, 32),               # exactly one 8x32 tile
    (2, [(20, 32), (20, 32)], 130, 40, 64),   # concat input, three Cout slabs with a ragged last one
    (2, [(5, 32)], 7, 8, 12),                 # image smaller than a tile, heavy channel padding
    (1, [(128, 128)], 96, 34, 20),            # ragged in both directions, narrow tile
    (3, [(32, 32)], 64, 64, 96),              # 72 workgroups
    (2, [(256, 256)], 64, 24, 72),            # long K, ragged columns with the wide tile
    (5, [(32, 32)], 448, 40, 64),             # 50 tiles x 7 slabs: the persistent loops (350 > 256 workgroups; half-width: 700 > 512)
]


def _random_w24_shapes(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        B = int(rng.integers(1, 4))
        H, W = 2 * int(rng.integers(2, 30)), 4 * int(rng.integers(1, 24))
        cout = int(rng.integers(1, 200))
        if rng.random() < 0.4:
            c1, c2 = int(rng.integers(1, 80)), int(rng.integers(1, 80))
            pp = max(32, 1 << (max(c1, c2) - 1).bit_length())
            segs = [(c1, pp), (c2, pp)]
        else:
            c = int(rng.integers(1, 150))
            segs = [(c, max(32, 1 << (c - 1).bit_length()))]
        out.append((B, segs, cout, H, W))
    return out


@pytest.mark.parametrize('shape', W24_SHAPES + _random_w24_shapes(int(os.environ.get('WINO_SWEEP', '8')), 77),
                         ids=lambda sh: f'{sh[0]}x{"+".join(str(a) for a, _ in sh[1])}->{sh[2]}@{sh[3]}x{sh[4]}')
def test_conv3x3_winograd24_fp32(C, shape):
    """Hybrid Winograd F(2x4,3x3) (wino24.hip): forward (+bias, ReLU, BN statistics rows) and data gradient, fp32, against
    the oracle's direct convolution at the SAME 2e-5 bound as the direct and F(2x2) kernels; bit-identical activations and
    statistics rows under every scheduling choice; bit-reproducible."""
    B, segs, cout, H, W = shape
    rng = np.random.default_rng(23)
    cin = sum(s[0] for s in segs)
    x = rnd(rng, B, cin, H, W)
    w = rnd(rng, cout, cin, 3, 3) * (1.0 / np.sqrt(9 * cin))
    b = rnd(rng, cout)
    cin_p, cout_p = sum(s[1] for s in segs), C.ops.cpad(cout)
    xt = nhwc_with_segs(C, x, segs, 0)
    wt, bt = dev(w), dev(b)
    wf = torch.zeros(24 * cout_p * cin_p, device='cuda')
    wd = torch.zeros(24 * cin_p * cout_p, device='cuda')
    bp = torch.zeros(cout_p, device='cuda')
    tab = C.ops.WinoPackTable(24); tab.conv3x3(wt, wf, wd, segs, cout); tab.finalize('cuda').run()
    pt = C.ops.PackTable(0); pt.vector(bt, bp, cout); pt.finalize('cuda').run(0)
    lib, ptr, s = C._lib, C._lib.ptr, C._lib.stream_ptr()
    y = torch.full((B, H, W, cout_p), 7.0, device='cuda')
    stats, rows = stat_buf(C, lib.OP_CONV3X3_WINOGRAD24, B, H, W, cin_p, cout_p, 0)
    lib.call('clamd_conv3x3_winograd24', ptr(xt), cin_p, ptr(wf), ptr(bp), ptr(y), cout_p, ptr(stats), rows, B, H, W, cin_p, cout_p, 1, None, s)
    gz = rnd(rng, B, cout, H, W)
    gzt = C.ops.to_nhwc(dev(gz), 0)
    gx = torch.full((B, H, W, cin_p), 3.0, device='cuda')
    lib.call('clamd_conv3x3_winograd24', ptr(gzt), cout_p, ptr(wd), None, ptr(gx), cin_p, None, 0, B, H, W, cout_p, cin_p, 0, None, s)
    sync()
    ref = O.relu_fwd(O.conv3x3_fwd(x, w, b))
    assert rel_l2(C.ops.from_nhwc(y, cout, 0).cpu().numpy(), ref) < TOL[0]
    st = stats.double().sum(0).cpu().numpy()
    np.testing.assert_allclose(st[0, :cout], ref.sum((0, 2, 3)), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(st[1, :cout], (ref ** 2).sum((0, 2, 3)), rtol=1e-4, atol=1e-3)
    assert float(y[..., cout:].abs().max()) == 0.0 if cout < cout_p else True
    rgx = O.conv3x3_bwd(x, w, gz)[0]
    pm = phys_map(segs)
    got_gx = gx.cpu().numpy().transpose(0, 3, 1, 2)
    assert rel_l2(got_gx[:, [p_ for p_, l in enumerate(pm) if l >= 0]], rgx) < TOL[0]
    pad = [p_ for p_, l in enumerate(pm) if l < 0]
    assert not pad or float(np.abs(got_gx[:, pad]).max()) == 0.0
    # weight gradient by the same hybrid form (wino24_wgrad.hip): fixed-order split-K, bit-reproducible
    wsb = lib.load().clamd_wgrad_winograd24_workspace_bytes(cout_p, cin_p)
    ws = torch.empty(wsb // 4 + 4, device='cuda')
    c_seg0, c_seg0p = (segs[0][0], segs[0][1]) if len(segs) == 2 else (cin, cin_p)
    rgw = O.conv3x3_bwd(x, w, gz)[1]
    for tn in (None, lib.Tuning(wgrad_blocks=64), lib.Tuning(cu_reserve=37), lib.Tuning(wgrad_blocks=1024)):
        gw = torch.full((cout, cin, 3, 3), 5.0, device='cuda')
        gw2 = torch.full((cout, cin, 3, 3), 6.0, device='cuda')
        for o_ in (gw, gw2):
            lib.call('clamd_wgrad_winograd24', ptr(gzt), cout_p, ptr(xt), cin_p, ptr(ws), wsb, ptr(o_), B, H, W, cout_p, cin_p, cout, cin,
                     cout, cout_p, c_seg0, c_seg0p, tn.ref() if tn else None, s)
        sync()
        assert torch.equal(gw, gw2)
        assert rel_l2(gw.cpu().numpy(), rgw) < 2e-5
    # ('wino_half', 1): the half-width workgroups of wino24n.hip (32 tiles x 32 channels, two per CU) -- same filters, formulas and MFMA
    # chains: bit-identical activations; rows per workgroup of ITS grid
    for key, val in (('wino_band', 1), ('wino_band', 32), ('wino_persist', 0), ('cu_reserve', 37), ('wino_persist', 1), ('wino_half', 1),
                     ('wino_half', 2)):
        tn = lib.Tuning(**({key: val} if (key, val) != ('wino_half', 2) else {'wino_half': 1, 'wino_persist': 0}))
        y2 = torch.full((B, H, W, cout_p), 7.0, device='cuda')
        stats2, rows2 = stat_buf(C, lib.OP_CONV3X3_WINOGRAD24, B, H, W, cin_p, cout_p, 0, tuning=tn)
        lib.call('clamd_conv3x3_winograd24', ptr(xt), cin_p, ptr(wf), ptr(bp), ptr(y2), cout_p, ptr(stats2), rows2, B, H, W, cin_p, cout_p, 1,
                 tn.ref(), s)
        stats3 = torch.full_like(stats2, float('nan'))
        lib.call('clamd_conv3x3_winograd24', ptr(xt), cin_p, ptr(wf), ptr(bp), ptr(y2), cout_p, ptr(stats3), rows2, B, H, W, cin_p, cout_p, 1,
                 tn.ref(), s)
        sync()
        assert torch.equal(y, y2), (key, val)
        assert torch.equal(stats2, stats3), f'{key}={val}: statistics rows differ between two identical launches'
        ph_, pw_ = (8, 32) if W >= 32 else (16, 16)
        per_tile = rows == B * (-(-H // ph_)) * (-(-W // pw_))       # one row per pixel tile: independent of the block order
        if rows2 == rows and key != 'wino_half' and (per_tile or key == 'wino_persist'):
            assert torch.equal(stats, stats2), (key, val)
        else:       # other grid: other rows (per workgroup / per tile), same totals
            np.testing.assert_allclose(stats2.double().sum(0).cpu().numpy(), stats.double().sum(0).cpu().numpy(), rtol=1e-5, atol=1e-3)


FOLD_SHAPES = [  # B, Cin, Cout, H, W
    (2, 64, 64, 24, 40),       # 8x32 tiles ragged along x: every tile touches the border
    (1, 128, 128, 16, 16),     # one 16x16 tile per image: all nine border classes inside one tile
    (2, 64, 40, 32, 96),       # interior tiles exist (fast path) next to border tiles; padded output channels
    (3, 40, 64, 64, 64),       # padded input channels (scale / shift of the padding never read as data)
]


@pytest.mark.parametrize('kernel,dcode', [('w24', 0), ('w24n', 0), ('w24h', 0), ('pws', 0), ('pws', 1), ('pws', 2)])
@pytest.mark.parametrize('shape', FOLD_SHAPES, ids=lambda sh: 'x'.join(str(a) for a in sh))
def test_batchnorm_folded_into_conv3x3(C, kernel, dcode, shape):
    """nn.BatchNorm2d folded algebraically into the nn.Conv2d behind it (bnfold.hip; models/unet.py:15-16): filters packed with the
    per-input-channel scale, the shift as a border-class bias table in the epilogue, the weight gradient fixed up from the
    gradient's border sums -- against the stock fp64 convolution ON THE NORMALISED TENSOR (zero padding after the affine), at the
    bounds of the unfolded kernels; statistics rows included; negative and zero scales included."""
    import torch.nn.functional as F
    B, cin, cout, H, W = shape
    rng = np.random.default_rng(321)
    lib, ptr, s = C._lib, C._lib.ptr, C._lib.stream_ptr()
    L = lib.load()
    T = C.ops.TORCH_DT[dcode]
    cin_p, cout_p = C.ops.cpad(cin), C.ops.cpad(cout)
    r = rb(np.maximum(rnd(rng, B, cin, H, W) + 0.3, 0), dcode)                 # a post-ReLU activation as the kernels store it
    scale = rnd(rng, cin) * 0.8
    scale[0] = 0.0                                                             # gamma = 0 is legal
    shift = rnd(rng, cin) * 0.5
    w = rnd(rng, cout, cin, 3, 3) * (1.0 / np.sqrt(9 * cin))
    b = rnd(rng, cout)
    gz = rb(rnd(rng, B, cout, H, W), dcode)
    x64 = torch.from_numpy(r).double() * torch.from_numpy(scale).double().view(1, -1, 1, 1) + torch.from_numpy(shift).double().view(1, -1, 1, 1)
    # forward reference: what the compute dtype stores are the FOLDED filters w * scale (the shift term is an fp32 table of the fp32 master filters)
    wfold = torch.from_numpy(rb(w * scale[None, :, None, None], dcode)).double()
    shift_img = torch.from_numpy(shift).double().view(1, -1, 1, 1).expand(B, cin, H, W)
    ref = torch.relu(F.conv2d(torch.from_numpy(r).double(), wfold, None, padding=1)
                     + F.conv2d(shift_img, torch.from_numpy(w).double(), torch.from_numpy(b).double(), padding=1)).numpy()
    xg = x64.clone().requires_grad_(False)
    wv = torch.from_numpy(w).double().requires_grad_(True)
    F.conv2d(xg, wv, None, padding=1).backward(torch.from_numpy(gz).double())
    rgw = wv.grad.numpy()

    rt = nd(C, np.ascontiguousarray(np.pad(r, ((0, 0), (0, cin_p - cin), (0, 0), (0, 0))).transpose(0, 2, 3, 1)), dcode)
    gzt = nd(C, np.ascontiguousarray(np.pad(gz, ((0, 0), (0, cout_p - cout), (0, 0), (0, 0))).transpose(0, 2, 3, 1)), dcode)
    sc = torch.full((cin_p,), float('nan'), device='cuda'); sc[:cin] = dev(scale)       # padded entries must never matter
    sh = torch.full((cin_p,), float('nan'), device='cuda'); sh[:cin] = dev(shift)
    wt, bt = dev(w), dev(b)
    table = torch.full((9, cout_p), float('nan'), device='cuda')
    lib.call('clamd_bn_fold_bias', ptr(wt), ptr(sh), ptr(bt), ptr(table), cout, cin, cout_p, s)
    y = torch.full((B, H, W, cout_p), 7.0, dtype=T, device='cuda')
    flags = 1 | 2                                                              # ReLU | CLAMD_BIAS_BORDER_CLASSES
    if kernel == 'pws':
        tn = lib.Tuning(igemm_pws=2)
        if not L.clamd_conv3x3_border_bias_ok(B, H, W, cin_p, cout_p, dcode, tn.ref()):
            pytest.skip('the persistent kernel declines this shape')
        wf = torch.zeros(9 * cout_p * cin_p, dtype=T, device='cuda')
        tab = C.ops.PackTable(dcode); tab.conv3x3(wt, wf, None, [(cin, cin_p)], cout, kscale=sc); tab.finalize('cuda').run(dcode)
        stats, rows = stat_buf(C, lib.OP_CONV3X3, B, H, W, cin_p, cout_p, dcode, tuning=tn)
        lib.call('clamd_conv3x3', ptr(rt), cin_p, ptr(wf), ptr(table), ptr(y), cout_p, ptr(stats), None, None, rows, B, H, W, cin_p, cout_p,
                 flags, 0, dcode, tn.ref(), s)
        # the other structures refuse the flag instead of ignoring it
        tn0 = lib.Tuning(igemm_pws=0)
        with pytest.raises(RuntimeError, match='border-class'):
            lib.call('clamd_conv3x3', ptr(rt), cin_p, ptr(wf), ptr(table), ptr(y), cout_p, None, None, None, 0, B, H, W, cin_p, cout_p,
                     flags, 0, dcode, tn0.ref(), s)
    else:
        wf = torch.zeros(24 * cout_p * cin_p, device='cuda')
        tab = C.ops.WinoPackTable(24); tab.conv3x3(wt, wf, None, [(cin, cin_p)], cout, kscale=sc); tab.finalize('cuda').run()
        tnw = lib.Tuning(wino_half=1) if kernel == 'w24n' else None      # w24n: the half-width workgroups (wino24n.hip)
        stats, rows = stat_buf(C, lib.OP_CONV3X3_WINOGRAD24, B, H, W, cin_p, cout_p, 0, tuning=tnw if kernel == 'w24n' else None)
        if kernel in ('w24', 'w24n'):
            lib.call('clamd_conv3x3_winograd24', ptr(rt), cin_p, ptr(wf), ptr(table), ptr(y), cout_p, ptr(stats), rows, B, H, W, cin_p, cout_p,
                     flags, tnw.ref() if tnw else None, s)
        else:
            lib.call('clamd_conv3x3_winograd24_direct_filters', ptr(rt), cin_p, ptr(wf), ptr(table), ptr(y), cout_p, ptr(stats), rows,
                     B, H, W, cin_p, cout_p, flags, None, s)
    sync()
    assert bool(torch.isfinite(table).all()) and float(table[:, cout:].abs().max() if cout < cout_p else 0.0) == 0.0
    got = C.ops.from_nhwc(y, cout, dcode).cpu().numpy()
    assert rel_l2(got, ref) < TOL[dcode], rel_l2(got, ref)
    st = stats.double().sum(0).cpu().numpy()
    np.testing.assert_allclose(st[0, :cout], ref.sum((0, 2, 3)), rtol=2e-3 if dcode == 1 else 1e-4, atol=2e-2 if dcode == 1 else 1e-3)
    np.testing.assert_allclose(st[1, :cout], (ref ** 2).sum((0, 2, 3)), rtol=4e-3 if dcode == 1 else 1e-4, atol=2e-2 if dcode == 1 else 1e-3)
    assert float(nf(C, y, dcode)[..., cout:].abs().max()) == 0.0 if cout < cout_p else True

    # weight gradient on r, then the fix-up in place; twice: bit-identical (fixed-order border sums)
    sum_gz = dev(gz.astype(np.float64).sum((0, 2, 3)).astype(np.float32))
    fws = L.clamd_bn_fold_wgrad_workspace_bytes(B, cout_p)
    fw = torch.empty(fws // 4 + 4, device='cuda')
    outs = []
    for _ in range(2):
        gw = torch.full((cout, cin, 3, 3), 5.0, device='cuda')
        if kernel == 'pws':
            wsb = L.clamd_wgrad_workspace_bytes(0, B, H, W, cout_p, cin_p, dcode)
            ws = torch.empty(wsb // 4 + 4, device='cuda')
            lib.call('clamd_wgrad', 0, ptr(gzt), cout_p, ptr(rt), cin_p, ptr(ws), wsb, ptr(gw), B, H, W, cout_p, cin_p, cout, cin,
                     cout, cout_p, cin, cin_p, dcode, None, s)
        else:
            wsb = L.clamd_wgrad_winograd24_workspace_bytes(cout_p, cin_p)
            ws = torch.empty(wsb // 4 + 4, device='cuda')
            lib.call('clamd_wgrad_winograd24', ptr(gzt), cout_p, ptr(rt), cin_p, ptr(ws), wsb, ptr(gw), B, H, W, cout_p, cin_p, cout, cin,
                     cout, cout_p, cin, cin_p, None, s)
        lib.call('clamd_bn_fold_wgrad', ptr(gzt), cout_p, ptr(sum_gz), ptr(sc), ptr(sh), ptr(gw), ptr(fw), fws, B, H, W, cout_p, cout, cin,
                 dcode, s)
        sync()
        outs.append(gw)
    assert torch.equal(outs[0], outs[1])
    assert rel_l2(outs[0].cpu().numpy(), rgw) < (6e-5 if dcode == 2 else 2e-5), rel_l2(outs[0].cpu().numpy(), rgw)


W24G_SHAPES = [  # B, Cin segs, Cout, H, W (Cin_p >= 64, Cout_p % 64 == 0)
    (1, [(64, 64)], 64, 16, 16),                 # 16x16-pixel workgroup tile, 8 chunks
    (2, [(128, 128)], 128, 8, 32),               # exactly one 8x32 tile per image, two output slabs
    (2, [(40, 64), (50, 64)], 100, 24, 40),      # concat input, ragged rows and columns, padded output channels
    (3, [(256, 256)], 256, 16, 16),              # 32 chunks, four slabs, persistent grid not reached
    (1, [(96, 128)], 192, 34, 68),               # ragged in both directions with the wide tile, three slabs
    (5, [(64, 64)], 640, 40, 64),                # 5*5*2 tiles x 10 slabs = 500 work items: the persistent loop and its
    #                                              cross-tile load stream (next tile's chunks fetched by the last chunks)
]


@pytest.mark.parametrize('shape', W24G_SHAPES, ids=lambda sh: f'{sh[0]}x{"+".join(str(a) for a, _ in sh[1])}->{sh[2]}@{sh[3]}x{sh[4]}')
def test_conv3x3_winograd24_pretransformed(C, shape):
    """wino24g.hip: input transformed ONCE (clamd_winograd24_transform_input), transform-free K loop
    (clamd_conv3x3_winograd24_pre).  Forward (+bias, ReLU, statistics rows) and data gradient against the oracle at the 2e-5
    bound, and BIT-IDENTICAL to clamd_conv3x3_winograd24 (same filters, same MFMA chains, same epilogue, same rows)."""
    B, segs, cout, H, W = shape
    rng = np.random.default_rng(29)
    cin = sum(s[0] for s in segs)
    x = rnd(rng, B, cin, H, W)
    w = rnd(rng, cout, cin, 3, 3) * (1.0 / np.sqrt(9 * cin))
    b = rnd(rng, cout)
    cin_p, cout_p = sum(s[1] for s in segs), C.ops.cpad(cout)
    if cout_p % 64:
        cout_p = 64
    xt = nhwc_with_segs(C, x, segs, 0)
    wt, bt = dev(w), dev(b)
    wf = torch.zeros(24 * cout_p * cin_p, device='cuda')
    wd = torch.zeros(24 * cin_p * cout_p, device='cuda')
    bp = torch.zeros(cout_p, device='cuda')
    tab = C.ops.WinoPackTable(24); tab.conv3x3(wt, wf, wd, segs, cout); tab.finalize('cuda').run()
    pt = C.ops.PackTable(0); pt.vector(bt, bp, cout); pt.finalize('cuda').run(0)
    lib, ptr, s = C._lib, C._lib.ptr, C._lib.stream_ptr()
    L = lib.load()
    gz = rnd(rng, B, cout, H, W)
    gzt = C.ops.to_nhwc(dev(gz), 0, cp=cout_p)
    ref = O.relu_fwd(O.conv3x3_fwd(x, w, b))
    rgx = O.conv3x3_bwd(x, w, gz)[0]
    pm = phys_map(segs)
    for tn in (None, lib.Tuning(wino_persist=0), lib.Tuning(cu_reserve=120), lib.Tuning(wino_band=1)):
        tp = tn.ref() if tn else None
        stats, rows = stat_buf(C, lib.OP_CONV3X3_WINOGRAD24, B, H, W, cin_p, cout_p, 0, tuning=tn)
        stats_p = torch.full_like(stats, float('nan'))
        y = torch.full((B, H, W, cout_p), 7.0, device='cuda')
        y_p = torch.full((B, H, W, cout_p), 8.0, device='cuda')
        lib.call('clamd_conv3x3_winograd24', ptr(xt), cin_p, ptr(wf), ptr(bp), ptr(y), cout_p, ptr(stats), rows, B, H, W, cin_p, cout_p, 1, tp, s)
        v = torch.full((L.clamd_winograd24_input_elems(B, H, W, cin_p),), float('nan'), device='cuda')
        lib.call('clamd_winograd24_transform_input', ptr(xt), cin_p, None, None, ptr(v), B, H, W, cin_p, s)
        lib.call('clamd_conv3x3_winograd24_pre', ptr(v), ptr(wf), ptr(bp), ptr(y_p), cout_p, ptr(stats_p), rows, B, H, W, cin_p, cout_p, 1, tp, s)
        sync()
        assert not bool(torch.isnan(v).any()), 'the transform must write every element of V'
        assert torch.equal(y, y_p)
        # rows: per tile, or per workgroup of the persistent grid.  The block order (band of output slabs) follows each kernel's own
        # traffic model, so per-workgroup rows partition the tiles differently unless the band is forced: same totals always,
        # identical rows under a forced band and on the one-workgroup-per-tile grid
        np.testing.assert_allclose(stats_p.double().sum(0).cpu().numpy(), stats.double().sum(0).cpu().numpy(), rtol=1e-5, atol=1e-3)
        if tn is not None and (tn.wino_band == 1 or tn.wino_persist == 0):
            assert torch.equal(stats, stats_p)
        stats_q = torch.full_like(stats, float('nan'))
        lib.call('clamd_conv3x3_winograd24_pre', ptr(v), ptr(wf), ptr(bp), ptr(y_p), cout_p, ptr(stats_q), rows, B, H, W, cin_p, cout_p, 1, tp, s)
        sync()
        assert torch.equal(stats_p, stats_q), 'statistics rows differ between two identical launches'
        assert rel_l2(C.ops.from_nhwc(y_p, cout, 0).cpu().numpy(), ref) < TOL[0]
        # data gradient: the same two calls on the gradient tensor and the tap-flipped filters
        gx = torch.full((B, H, W, cin_p), 3.0, device='cuda')
        gx_p = torch.full((B, H, W, cin_p), 4.0, device='cuda')
        lib.call('clamd_conv3x3_winograd24', ptr(gzt), cout_p, ptr(wd), None, ptr(gx), cin_p, None, 0, B, H, W, cout_p, cin_p, 0, tp, s)
        vg = torch.empty(L.clamd_winograd24_input_elems(B, H, W, cout_p), device='cuda')
        lib.call('clamd_winograd24_transform_input', ptr(gzt), cout_p, None, None, ptr(vg), B, H, W, cout_p, s)
        lib.call('clamd_conv3x3_winograd24_pre', ptr(vg), ptr(wd), None, ptr(gx_p), cin_p, None, 0, B, H, W, cout_p, cin_p, 0, tp, s)
        sync()
        assert torch.equal(gx, gx_p)
        got_gx = gx_p.cpu().numpy().transpose(0, 3, 1, 2)
        assert rel_l2(got_gx[:, [p_ for p_, l in enumerate(pm) if l >= 0]], rgx) < TOL[0]
        # the narrow-layer variant: in-kernel transform, filters straight into the operand registers -- bit-identical as well
        y_h = torch.full((B, H, W, cout_p), 9.0, device='cuda')
        stats_h = torch.full_like(stats, float('nan'))
        lib.call('clamd_conv3x3_winograd24_direct_filters', ptr(xt), cin_p, ptr(wf), ptr(bp), ptr(y_h), cout_p, ptr(stats_h), rows,
                 B, H, W, cin_p, cout_p, 1, tp, s)
        gx_h = torch.full((B, H, W, cin_p), 5.0, device='cuda')
        lib.call('clamd_conv3x3_winograd24_direct_filters', ptr(gzt), cout_p, ptr(wd), None, ptr(gx_h), cin_p, None, 0, B, H, W,
                 cout_p, cin_p, 0, tp, s)
        sync()
        assert torch.equal(y, y_h) and torch.equal(gx, gx_h)
        assert torch.equal(stats, stats_h), 'same block order as clamd_conv3x3_winograd24: identical rows'
    # BatchNorm folded into the transform: V(raw * scale + shift, zero padding AFTER the affine) == V of the materialised tensor
    scale = torch.rand(cin_p, device='cuda') + 0.5
    shift = torch.randn(cin_p, device='cuda')
    applied = torch.empty_like(xt)                                 # the materialised BatchNorm output: clamd_bn_apply (one fma per element)
    lib.call('clamd_bn_apply', ptr(xt), cin_p, ptr(scale), ptr(shift), ptr(applied), cin_p, None, 0, B, H, W, cin_p, 0, s)
    v_ref = torch.empty_like(v)
    v_fold = torch.full_like(v, float('nan'))
    lib.call('clamd_winograd24_transform_input', ptr(applied), cin_p, None, None, ptr(v_ref), B, H, W, cin_p, s)
    lib.call('clamd_winograd24_transform_input', ptr(xt), cin_p, ptr(scale), ptr(shift), ptr(v_fold), B, H, W, cin_p, s)
    sync()
    assert torch.equal(v_ref, v_fold)
    # refused shapes
    with pytest.raises(RuntimeError, match='Cout_p % 64'):
        lib.call('clamd_conv3x3_winograd24_pre', ptr(v), ptr(wf), ptr(bp), ptr(y_p), 32, None, 0, B, H, W, cin_p, 32, 1, None, s)


W24G_WGRAD_SHAPES = [  # B, Cin segs, Cout, H, W (Cin_p, Cout_p multiples of 256)
    (2, [(256, 256)], 256, 16, 16),
    (1, [(100, 256), (130, 256)], 200, 12, 20),      # concat input with padding, padded output channels, ragged 16x16 tile blocks
    (3, [(256, 256)], 512, 32, 32),                  # several splits
    (2, [(256, 256)], 256, 20, 72),                  # ragged 8x32 tile blocks in both directions
    (2, [(128, 128)], 128, 16, 32),                  # multiples of 128 (round 5): the wave-level stream-K plan
    (1, [(60, 64), (50, 64)], 200, 24, 40),          # 256 x 128 with a concat input, padding on both sides
]


@pytest.mark.parametrize('shape', W24G_WGRAD_SHAPES, ids=lambda sh: f'{sh[0]}x{"+".join(str(a) for a, _ in sh[1])}->{sh[2]}@{sh[3]}x{sh[4]}')
def test_wgrad_winograd24_pretransformed(C, shape):
    """Weight gradient as a batched GEMM over the 24 Winograd planes (wino24g.hip): the x side is the forward image V read in
    place, the gradient side is transformed once.  Against the oracle at the 2e-5 bound of the other weight-gradient kernels,
    bit-reproducible, same result (to rounding) under another split plan."""
    B, segs, cout, H, W = shape
    rng = np.random.default_rng(31)
    cin = sum(s[0] for s in segs)
    x = rnd(rng, B, cin, H, W)
    w = rnd(rng, cout, cin, 3, 3) * (1.0 / np.sqrt(9 * cin))
    gz = rnd(rng, B, cout, H, W)
    cin_p, cout_p = sum(s[1] for s in segs), (cout + 127) // 128 * 128
    xt = nhwc_with_segs(C, x, segs, 0)
    gzt = C.ops.to_nhwc(dev(gz), 0, cp=cout_p)
    lib, ptr, s = C._lib, C._lib.ptr, C._lib.stream_ptr()
    L = lib.load()
    v = torch.empty(L.clamd_winograd24_input_elems(B, H, W, cin_p), device='cuda')
    lib.call('clamd_winograd24_transform_input', ptr(xt), cin_p, None, None, ptr(v), B, H, W, cin_p, s)
    yt = torch.full((L.clamd_wgrad_winograd24_pre_operand_elems(B, H, W, cout_p),), float('nan'), device='cuda')
    wsb = L.clamd_wgrad_winograd24_pre_workspace_bytes(B, H, W, cout_p, cin_p)
    ws = torch.empty(wsb // 4 + 4, device='cuda')
    c_seg0, c_seg0p = (segs[0][0], segs[0][1]) if len(segs) == 2 else (cin, cin_p)
    rgw = O.conv3x3_bwd(x, w, gz)[1]
    outs = []
    # wgrad_streamk: 1 (default) per launch, 0 the split-K plan in whole rounds of the chip, 2 the stream-K plane GEMM (round 5)
    for tn in (None, None, lib.Tuning(cu_reserve=100), lib.Tuning(wgrad_streamk=0), lib.Tuning(wgrad_streamk=0, cu_reserve=100), lib.Tuning(wgrad_streamk=2),
               lib.Tuning(wgrad_streamk=2, cu_reserve=100)):
        gw = torch.full((cout, cin, 3, 3), 5.0, device='cuda')
        lib.call('clamd_wgrad_winograd24_pre', ptr(gzt), cout_p, ptr(v), ptr(yt), ptr(ws), wsb, ptr(gw), B, H, W,
                 cout_p, cin_p, cout, cin, cout, cout_p, c_seg0, c_seg0p, tn.ref() if tn else None, s)
        sync()
        outs.append(gw)
        assert rel_l2(gw.cpu().numpy(), rgw) < 2e-5
    assert not bool(torch.isnan(yt).any())
    assert torch.equal(outs[0], outs[1]), 'two identical launches must be bit-identical'
    for o_ in outs[2:]:
        assert rel_l2(o_.cpu().numpy(), outs[0].cpu().numpy()) < 2e-6
    # and against the in-kernel-transform weight gradient of the same form
    wsb2 = L.clamd_wgrad_winograd24_workspace_bytes(cout_p, cin_p)
    ws2 = torch.empty(wsb2 // 4 + 4, device='cuda')
    gw2 = torch.empty(cout, cin, 3, 3, device='cuda')
    lib.call('clamd_wgrad_winograd24', ptr(gzt), cout_p, ptr(xt), cin_p, ptr(ws2), wsb2, ptr(gw2), B, H, W, cout_p, cin_p, cout, cin,
             cout, cout_p, c_seg0, c_seg0p, None, s)
    sync()
    assert rel_l2(outs[0].cpu().numpy(), gw2.cpu().numpy()) < 1e-5


def _random_conv_shapes(n, seed):
    """Seeded random problem sizes that hit ragged tiles, several channel slabs, K-step pairs / fours / odd counts and
    the split-K tail of every 3x3 kernel."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        B = int(rng.integers(1, 4))
        cin = int(rng.choice([3, 20, 33, 64, 96, 128, 200, 256, 300]))
        cout = int(rng.choice([5, 32, 64, 70, 128, 130]))
        H, W = int(rng.integers(3, 41)), int(rng.integers(3, 70))
        out.append((B, cin, cout, H, W))
    return out


@pytest.mark.parametrize('name,dcode', DT)
@pytest.mark.parametrize('shape', _random_conv_shapes(10, 2024))
def test_conv3x3_random_shapes(C, name, dcode, shape):
    """Forward (+ReLU, BN statistics), data gradient and weight gradient of one 3x3 convolution on a random ragged shape."""
    B, cin, cout, H, W = shape
    rng = np.random.default_rng(hash(shape) % (2 ** 31))
    segs = [(cin, C.ops.cpad(cin))]
    x, w, b, xt, wf, wd, bp, cin_p, cout_p = _conv_case(C, rng, B, segs, cout, H, W, dcode)
    lib, ptr, s = C._lib, C._lib.ptr, C._lib.stream_ptr()
    T = C.ops.TORCH_DT[dcode]
    y = torch.full((B, H, W, cout_p), 7.0, dtype=T, device='cuda')
    stats, rows = stat_buf(C, lib.OP_CONV3X3, B, H, W, cin_p, cout_p, dcode)
    lib.call('clamd_conv3x3', ptr(xt), cin_p, ptr(wf), ptr(bp), ptr(y), cout_p, ptr(stats), None, None, rows, B, H, W, cin_p, cout_p, 1, 0, dcode, None, s)
    gz = rb(rnd(rng, B, cout, H, W), dcode)
    gzt = C.ops.to_nhwc(dev(gz), dcode)
    gx = torch.zeros(B, H, W, cin_p, dtype=T, device='cuda')
    lib.call('clamd_conv3x3', ptr(gzt), cout_p, ptr(wd), None, ptr(gx), cin_p, None, None, None, 0, B, H, W, cout_p, cin_p, 0, 0, dcode, None, s)
    wsb = lib.load().clamd_wgrad_workspace_bytes(0, B, H, W, cout_p, cin_p, dcode)
    ws = torch.empty(wsb // 4 + 4, device='cuda')
    gw = torch.full((cout, cin, 3, 3), 5.0, device='cuda')
    lib.call('clamd_wgrad', 0, ptr(gzt), cout_p, ptr(xt), cin_p, ptr(ws), wsb, ptr(gw), B, H, W, cout_p, cin_p, cout, cin,
             cout, cout_p, cin, cin_p, dcode, None, s)
    sync()
    ref = O.relu_fwd(O.conv3x3_fwd(x, w, b))
    assert rel_l2(C.ops.from_nhwc(y, cout, dcode).cpu().numpy(), ref) < TOL[dcode]
    st = stats.double().sum(0).cpu().numpy()
    np.testing.assert_allclose(st[0, :cout], ref.sum((0, 2, 3)), rtol=2e-3 if dcode == 1 else 1e-4, atol=2e-2 if dcode == 1 else 1e-3)
    assert float(nf(C, y, dcode)[..., cout:].abs().max()) == 0.0 if cout < cout_p else True
    rgx, rgw, _ = O.conv3x3_bwd(x, w, gz)
    assert rel_l2(nf(C, gx, dcode).cpu().numpy().transpose(0, 3, 1, 2)[:, :cin], rgx) < TOL[dcode]
    assert rel_l2(gw.cpu().numpy(), rgw) < (6e-5 if dcode == 2 else 2e-5)


@pytest.mark.parametrize('name,dcode', DT)
@pytest.mark.parametrize('shape', [(2, 6, 3, 4, 5), (1, 128, 64, 16, 16), (2, 70, 40, 8, 48)])
def test_convT2x2_fwd_dgrad_wgrad(C, name, dcode, shape):
    B, cin, cout, h, w_ = shape
    rng = np.random.default_rng(12)
    lib, ptr, s = C._lib, C._lib.ptr, C._lib.stream_ptr()
    T = C.ops.TORCH_DT[dcode]
    cin_p, cout_p = C.ops.cpad(cin), C.ops.cpad(cout)
    x = rb(rnd(rng, B, cin, h, w_), dcode)
    w = rb(rnd(rng, cin, cout, 2, 2) * (1 / np.sqrt(cin)), dcode)
    b = rnd(rng, cout)
    xt, wt, bt = C.ops.to_nhwc(dev(x), dcode), dev(w), dev(b)
    wf = torch.zeros(4 * cout_p * cin_p, dtype=T, device='cuda'); wd = torch.zeros(cin_p * 4 * cout_p, dtype=T, device='cuda')
    bp = torch.zeros(cout_p, device='cuda')
    tab = C.ops.PackTable(dcode); tab.convT(wt, wf, wd, cin, cout); tab.vector(bt, bp, cout); tab.finalize('cuda').run(dcode)
    # output goes into the SECOND half of a concat buffer (pitch 2*cout_p), the first half must stay untouched
    cat = torch.full((B, 2 * h, 2 * w_, 2 * cout_p), 3.0, dtype=T, device='cuda')
    ysl = cat[..., cout_p:]
    lib.call('clamd_convT2x2_fwd', ptr(xt), cin_p, ptr(wf), ptr(bp), ptr(ysl), 2 * cout_p, B, h, w_, cin_p, cout_p, dcode, s)
    sync()
    ref = O.convT2x2_fwd(x, w, b)
    got = nf(C, cat[..., cout_p:], dcode)[..., :cout].cpu().numpy().transpose(0, 3, 1, 2)
    assert rel_l2(got, ref) < TOL[dcode]
    assert float((cat[..., :cout_p].float() - 3.0).abs().max()) == 0.0          # raw storage of the other half: untouched
    # backward: gradient arrives in the same slice
    gy = rb(rnd(rng, B, cout, 2 * h, 2 * w_), dcode)
    gfull = np.zeros((B, 2 * h, 2 * w_, 2 * cout_p), np.float32)
    gfull[..., cout_p:cout_p + cout] = gy.transpose(0, 2, 3, 1)
    gcat = nd(C, gfull, dcode)
    gsl = gcat[..., cout_p:]
    gx = torch.zeros(B, h, w_, cin_p, dtype=T, device='cuda')
    lib.call('clamd_convT2x2_dgrad', ptr(gsl), 2 * cout_p, ptr(wd), ptr(gx), cin_p, None, None, 0, B, h, w_, cin_p, cout_p, dcode, s)
    wsb = lib.load().clamd_wgrad_workspace_bytes(2, B, h, w_, cin_p, cout_p, dcode)
    ws = torch.empty(wsb // 4 + 4, device='cuda')
    gw = torch.zeros(cin, cout, 2, 2, device='cuda')
    lib.call('clamd_wgrad', 2, ptr(xt), cin_p, ptr(gsl), 2 * cout_p, ptr(ws), wsb, ptr(gw), B, h, w_, cin_p, cout_p, cin, cout,
             cin, cin_p, cout, cout_p, dcode, None, s)
    gb = torch.full((cout,), 9.0, device='cuda')          # overwritten, not accumulated into
    csb = lib.load().clamd_channel_sum_workspace_bytes(cout_p)
    cws = torch.empty(csb // 4, device='cuda')
    lib.call('clamd_channel_sum', ptr(gsl), 2 * cout_p, ptr(gb), B * 4 * h * w_, cout_p, cout, dcode, ptr(cws), csb, None, s)
    gb2 = torch.full((cout,), 9.0, device='cuda')
    lib.call('clamd_channel_sum', ptr(gsl), 2 * cout_p, ptr(gb2), B * 4 * h * w_, cout_p, cout, dcode, ptr(cws), csb, None, s)
    sync()
    assert torch.equal(gb, gb2)
    rgx, rgw, rgb = O.convT2x2_bwd(x, w, gy)
    assert rel_l2(C.ops.from_nhwc(gx, cin, dcode).cpu().numpy(), rgx) < TOL[dcode]
    assert rel_l2(gw.cpu().numpy(), rgw) < (6e-5 if dcode == 2 else 2e-5)
    assert rel_l2(gb.cpu().numpy(), rgb) < 2e-5


@pytest.mark.parametrize('name,dcode', DT)
@pytest.mark.parametrize('shape', [(2, 6, 5, 16, 16), (1, 64, 21, 32, 64), (2, 40, 2, 16, 48)])
def test_head_fwd_bwd(C, name, dcode, shape):
    B, cin, K, H, W = shape
    rng = np.random.default_rng(13)
    lib, ptr, s = C._lib, C._lib.ptr, C._lib.stream_ptr()
    T = C.ops.TORCH_DT[dcode]
    cin_p, kp = C.ops.cpad(cin), C.ops.cpad(K)
    x = rb(rnd(rng, B, cin, H, W), dcode)
    w = rb(rnd(rng, K, cin, 1, 1) * (1 / np.sqrt(cin)), dcode)
    b = rnd(rng, K)
    xt, wt, bt = C.ops.to_nhwc(dev(x), dcode), dev(w), dev(b)
    wf = torch.zeros(kp * cin_p, dtype=T, device='cuda'); wd = torch.zeros(cin_p * kp, dtype=T, device='cuda')
    bp = torch.zeros(kp, device='cuda')
    tab = C.ops.PackTable(dcode); tab.head(wt, wf, wd, cin, K); tab.vector(bt, bp, K); tab.finalize('cuda').run(dcode)
    logits = torch.full((B, K, H, W), 9.0, device='cuda')
    lib.call('clamd_conv1x1_logits', ptr(xt), cin_p, ptr(wf), ptr(bp), ptr(logits), B, H, W, cin_p, kp, K, dcode, s)
    sync()
    ref = O.conv1x1_fwd(x, w, b)
    assert rel_l2(logits.cpu().numpy(), ref) < (6e-5 if dcode == 2 else 2e-5)   # fp32 output, fp32 accumulate
    g = rb(rnd(rng, B, K, H, W), dcode)
    gt = C.ops.to_nhwc(dev(g), dcode)
    gx = torch.zeros(B, H, W, cin_p, dtype=T, device='cuda')
    lib.call('clamd_conv1x1', ptr(gt), kp, ptr(wd), None, ptr(gx), cin_p, None, None, None, 0, B, H, W, kp, cin_p, 0, dcode, s)
    wsb = lib.load().clamd_wgrad_workspace_bytes(1, B, H, W, kp, cin_p, dcode)
    ws = torch.empty(wsb // 4 + 4, device='cuda')
    gw = torch.zeros(K, cin, 1, 1, device='cuda')
    lib.call('clamd_wgrad', 1, ptr(gt), kp, ptr(xt), cin_p, ptr(ws), wsb, ptr(gw), B, H, W, kp, cin_p, K, cin, K, kp, cin, cin_p, dcode, None, s)
    sync()
    rgx, rgw, _ = O.conv1x1_bwd(x, w, g)
    assert rel_l2(C.ops.from_nhwc(gx, cin, dcode).cpu().numpy(), rgx) < TOL[dcode]
    assert rel_l2(gw.cpu().numpy(), rgw) < (6e-5 if dcode == 2 else 2e-5)


@pytest.mark.parametrize('name,dcode', DT)
@pytest.mark.parametrize('shape', [(2, 3, 64, 16, 48), (1, 1, 5, 32, 32), (2, 3, 7, 8, 12), (1, 3, 8, 6, 10), (3, 2, 40, 40, 72)])      # (widths that are not a multiple of 4 take the piece-per-thread kernel in bf16 too; 3 x 40 x 72 / 4 quads end inside a workgroup)
def test_first_layer_im2col_path(C, name, dcode, shape):
    """enc1.0 (Cin = 3, models/unet.py:50) as im2col + pointwise GEMM: forward (+ReLU, BN stats) and weight gradient
    against the oracle's 3x3 conv."""
    B, cin, cout, H, W = shape
    rng = np.random.default_rng(21)
    lib, ptr, s = C._lib, C._lib.ptr, C._lib.stream_ptr()
    T = C.ops.TORCH_DT[dcode]
    x = rnd(rng, B, cin, H, W)
    w = rb(rnd(rng, cout, cin, 3, 3) * (1.0 / np.sqrt(9 * cin)), dcode)
    b = rnd(rng, cout)
    kp, cout_p = C.ops.cpad(9 * cin), C.ops.cpad(cout)
    xt_ = dev(x)
    xcol = torch.empty(B, H, W, kp, dtype=T, device='cuda')
    lib.call('clamd_nchw_im2col3', ptr(xt_), ptr(xcol), kp, B, cin, H, W, kp, dcode, s)
    wt, bt = dev(w), dev(b)
    wf = torch.zeros(cout_p * kp, dtype=T, device='cuda'); bp = torch.zeros(cout_p, device='cuda')
    tab = C.ops.PackTable(dcode); tab.head(wt, wf, None, 9 * cin, cout); tab.vector(bt, bp, cout); tab.finalize('cuda').run(dcode)
    y = torch.zeros(B, H, W, cout_p, dtype=T, device='cuda')
    stats, rows = stat_buf(C, lib.OP_CONV1X1, B, H, W, kp, cout_p, dcode)
    lib.call('clamd_conv1x1', ptr(xcol), kp, ptr(wf), ptr(bp), ptr(y), cout_p, ptr(stats), None, None, rows, B, H, W, kp, cout_p, 1, dcode, s)
    sync()
    xr = rb(x, dcode)                                    # the gather rounds the image to the compute dtype
    ref = O.relu_fwd(O.conv3x3_fwd(xr, w, b))
    assert rel_l2(C.ops.from_nhwc(y, cout, dcode).cpu().numpy(), ref) < TOL[dcode]
    np.testing.assert_allclose(stats.sum(0).cpu().numpy()[0, :cout], ref.sum((0, 2, 3)), rtol=3e-3, atol=1e-2)
    gz = rb(rnd(rng, B, cout, H, W), dcode)
    gzt = C.ops.to_nhwc(dev(gz), dcode)
    wsb = lib.load().clamd_wgrad_workspace_bytes(1, B, H, W, cout_p, kp, dcode)
    ws = torch.empty(wsb // 4 + 4, device='cuda'); gw = torch.zeros(cout, cin, 3, 3, device='cuda')
    lib.call('clamd_wgrad', 1, ptr(gzt), cout_p, ptr(xcol), kp, ptr(ws), wsb, ptr(gw), B, H, W, cout_p, kp, cout, 9 * cin,
             cout, cout_p, 9 * cin, kp, dcode, None, s)
    sync()
    _, rgw, _ = O.conv3x3_bwd(xr, w, gz, need_gx=False)
    assert rel_l2(gw.cpu().numpy(), rgw) < (6e-5 if dcode == 2 else 2e-5)


@pytest.mark.parametrize('name,dcode', DT)
@pytest.mark.parametrize('shape', [(2, 7, 8, 12), (1, 64, 16, 16), (2, 200, 4, 6), (1, 1024, 2, 2)])
@pytest.mark.parametrize('pool', [False, True])
def test_bn_fwd_bwd_pool(C, name, dcode, shape, pool):
    """BatchNorm(train) apply (+2x2 max-pool) and the fused ReLU/BN backward (+pool routing) vs the oracle;
    gammas of both signs, values with ties inside pooling windows."""
    B, Cc, H, W = shape
    rng = np.random.default_rng(14)
    lib, ptr, s = C._lib, C._lib.ptr, C._lib.stream_ptr()
    T = C.ops.TORCH_DT[dcode]
    cp = C.ops.cpad(Cc)
    R, NS = 16, lib.load().clamd_bn_bwd_nsums()          # R: any number of partial rows
    y = np.maximum(rnd(rng, B, Cc, H, W), 0)                       # a ReLU output: many exact zeros (ties)
    y = rb(y, dcode)
    gamma, beta = rnd(rng, Cc), rnd(rng, Cc)
    rm0, rv0 = rnd(rng, Cc), np.abs(rnd(rng, Cc)) + 0.5
    yt = C.ops.to_nhwc(dev(y), dcode)
    # statistics as a conv epilogue would have produced them, spread over partial rows
    stats = torch.zeros(R, 2, cp, device='cuda')
    stats[0, 0, :Cc] = dev(y.sum((0, 2, 3)) * 0.25); stats[3, 0, :Cc] = dev(y.sum((0, 2, 3)) * 0.75)
    stats[1, 1, :Cc] = dev((y.astype(np.float64) ** 2).sum((0, 2, 3)).astype(np.float32))
    vec = torch.zeros(7, cp, device='cuda')
    gt_, bt_, rm, rv = dev(gamma), dev(beta), dev(rm0), dev(rv0)
    n = B * H * W
    nbt = torch.tensor(41, dtype=torch.int64, device='cuda')
    lib.call('clamd_bn_finalize', ptr(stats), R, ptr(gt_), ptr(bt_), ptr(rm), ptr(rv), ptr(vec[0]), ptr(vec[1]), ptr(vec[2]),
             ptr(vec[3]), cp, Cc, float(n), 0.1, 1e-5, ptr(nbt), s)
    assert int(nbt) == 42                                  # nn.BatchNorm2d.num_batches_tracked, incremented by the launch itself
    cat = torch.full((B, H, W, 2 * cp), 2.0, dtype=T, device='cuda')       # BN output goes to the FIRST half of a concat buffer
    pooled = torch.zeros(B, H // 2, W // 2, cp, dtype=T, device='cuda') if pool else None
    lib.call('clamd_bn_apply', ptr(yt), cp, ptr(vec[0]), ptr(vec[1]), ptr(cat), 2 * cp, ptr(pooled), cp, B, H, W, cp, dcode, s)
    sync()
    u_ref, cache, rm_ref, rv_ref = O.bn_train_fwd(y, gamma, beta, rm0, rv0)
    got_u = nf(C, cat[..., :cp], dcode)[..., :Cc].cpu().numpy().transpose(0, 3, 1, 2)
    assert rel_l2(got_u, u_ref) < (1e-5 if dcode != 1 else 4e-3)
    assert float((cat[..., cp:].float() - 2.0).abs().max()) == 0.0
    np.testing.assert_allclose(rm.cpu().numpy(), rm_ref, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rv.cpu().numpy(), rv_ref, rtol=1e-4, atol=1e-6)
    if pool:
        # the kernel takes max/arg-max on the fp32 value fma(y, scale, shift) BEFORE rounding to the storage dtype;
        # emulate that fma exactly (double product of two floats is exact) with the kernel's own scale/shift
        sc = vec[0, :Cc].cpu().numpy().astype(np.float64).reshape(1, -1, 1, 1)
        sh = vec[1, :Cc].cpu().numpy().astype(np.float64).reshape(1, -1, 1, 1)
        u32 = (y.astype(np.float64) * sc + sh).astype(np.float32)
        p_ref, idx = O.maxpool2x2_fwd(u32)
        assert np.array_equal(nf(C, pooled, dcode)[..., :Cc].cpu().numpy().transpose(0, 3, 1, 2), rb(p_ref, dcode))
    # ---- backward ----
    ga = rb(rnd(rng, B, Cc, H, W), dcode)
    gfull = np.zeros((B, H, W, 2 * cp), np.float32)
    gfull[..., :Cc] = ga.transpose(0, 2, 3, 1)
    gat = nd(C, gfull, dcode)
    gp = rb(rnd(rng, B, Cc, H // 2, W // 2), dcode) if pool else None
    gpt = C.ops.to_nhwc(dev(gp), dcode) if pool else None
    sums, srows = stat_buf(C, lib.OP_BN_BWD_REDUCE, B, H, W, 1 if pool else 0, cp, dcode, nk=NS)
    lib.call('clamd_bn_bwd_reduce', ptr(gat), 2 * cp, ptr(gpt), cp, ptr(yt), cp, ptr(vec[0]), ptr(vec[1]), ptr(sums), srows, B, H, W, cp,
             dcode, None, s)
    dg, db, dcb = torch.zeros(Cc, device='cuda'), torch.zeros(Cc, device='cuda'), torch.zeros(Cc, device='cuda')
    lib.call('clamd_bn_bwd_finalize', ptr(sums), srows, ptr(gt_), ptr(vec[2]), ptr(vec[3]), ptr(vec[4]), ptr(dg), ptr(db), ptr(dcb), cp, Cc, float(n), s)
    gz = torch.zeros(B, H, W, cp, dtype=T, device='cuda')
    lib.call('clamd_bn_bwd_apply', ptr(gat), 2 * cp, ptr(gpt), cp, ptr(yt), cp, ptr(vec[0]), ptr(vec[1]), ptr(vec[4]), ptr(gz), cp, B, H, W, cp, dcode, s)
    sync()
    gu = ga.copy()
    if pool:
        gu = gu + O.maxpool2x2_bwd(gp, idx)          # routing by the arg-max of the kernel's own u (first max wins)
    gy_ref, gg_ref, gb_ref = O.bn_train_bwd(gu, gamma, cache)
    gz_ref = gy_ref * (y > 0)
    tol = 1e-4 if dcode != 1 else 8e-3
    assert rel_l2(C.ops.from_nhwc(gz, Cc, dcode).cpu().numpy(), gz_ref) < tol
    scale = np.abs(gg_ref).max() + 1e-6
    np.testing.assert_allclose(dg.cpu().numpy(), gg_ref, rtol=1e-3 if dcode != 1 else 2e-2, atol=1e-3 * scale if dcode != 1 else 2e-2 * scale)
    np.testing.assert_allclose(db.cpu().numpy(), gb_ref, rtol=1e-3, atol=1e-3 * (np.abs(gb_ref).max() + 1e-6))
    np.testing.assert_allclose(dcb.cpu().numpy(), gz_ref.sum((0, 2, 3)), rtol=1e-3 if dcode != 1 else 3e-2,
                               atol=(1e-3 if dcode != 1 else 3e-2) * (np.abs(gz_ref.sum((0, 2, 3))).max() + 1e-6))
    if not pool:
        # two-sum form: finalize without the bias gradient (rows 2-4 unused), the apply pass writes the same g_z bit for bit and the
        # partial rows of its per-channel sum; clamd_rows_sum adds them in a fixed order
        sums2 = sums.clone(); sums2[:, 2:] = 0.0
        vec2 = vec.clone()
        dg2, db2, dcb2 = torch.zeros(Cc, device='cuda'), torch.zeros(Cc, device='cuda'), torch.full((Cc,), 3.0, device='cuda')
        lib.call('clamd_bn_bwd_finalize', ptr(sums2), srows, ptr(gt_), ptr(vec2[2]), ptr(vec2[3]), ptr(vec2[4]), ptr(dg2), ptr(db2), None, cp, Cc, float(n), s)
        nr = lib.load().clamd_bn_bwd_apply_sums_rows(B, H, W, cp)
        gz2 = torch.zeros(B, H, W, cp, dtype=T, device='cuda')
        out = []
        for _ in range(2):
            rows_ = torch.full((nr, cp), float('nan'), device='cuda')
            lib.call('clamd_bn_bwd_apply_sums', ptr(gat), 2 * cp, ptr(yt), cp, ptr(vec2[4]), ptr(gz2), cp, ptr(rows_), nr, B, H, W, cp, dcode, s)
            lib.call('clamd_rows_sum', ptr(rows_), nr, ptr(dcb2), cp, Cc, s)
            sync()
            out.append((rows_.clone(), dcb2.clone()))
        assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
        assert torch.equal(gz2, gz) and torch.equal(dg2, dg) and torch.equal(db2, db) and torch.equal(vec2[4:7], vec[4:7])
        np.testing.assert_allclose(dcb2.cpu().numpy(), gz_ref.sum((0, 2, 3)), rtol=1e-3 if dcode != 1 else 3e-2,
                                   atol=(1e-3 if dcode != 1 else 3e-2) * (np.abs(gz_ref.sum((0, 2, 3))).max() + 1e-6))
        with pytest.raises(RuntimeError, match='nrows'):
            lib.call('clamd_bn_bwd_apply_sums', ptr(gat), 2 * cp, ptr(yt), cp, ptr(vec2[4]), ptr(gz2), cp, ptr(rows_), nr + 1, B, H, W, cp, dcode, s)


@pytest.mark.parametrize('name,dcode', DT)
def test_maxpool2x2_alone_and_signed(C, name, dcode):
    """nn.MaxPool2d(2,2) on its own (models/unet.py:12; blocks.py) and its signed form (window minimum where sign < 0: the pool of a
    BatchNorm output scale * x + shift taken on x, bnfold.hip): bit-exact against torch on the stored values, gradient to the first
    extremum of the window; pitched input and output (a channel slice of a concat buffer)."""
    import torch.nn.functional as F
    lib, ptr, s = C._lib, C._lib.ptr, C._lib.stream_ptr()
    rng = np.random.default_rng(5)
    B, Cc, H, W = 2, 64, 12, 20
    x = rb(rnd(rng, B, H, W, 2 * Cc), dcode)                                   # [B,H,W,128]: the op works on the first 64 channels
    xt = nd(C, x, dcode)
    sign = dev(np.where(rng.random(Cc) < 0.5, -1.0, 1.0).astype(np.float32))
    T = C.ops.TORCH_DT[dcode]
    for sg in (None, sign):
        out = torch.zeros(B, H // 2, W // 2, Cc, dtype=T, device='cuda')
        lib.call('clamd_maxpool2x2', ptr(xt), 2 * Cc, ptr(sg), ptr(out), Cc, B, H, W, Cc, dcode, s)
        xs = torch.from_numpy(x[..., :Cc]).cuda().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
        sv = (sg if sg is not None else torch.ones(Cc, device='cuda')).view(1, -1, 1, 1)
        ref = F.max_pool2d(xs * sv, 2, 2) * sv                                 # max for +1, min for -1 (exact: multiplication by +-1)
        assert torch.equal(nf(C, out, dcode).permute(0, 3, 1, 2), ref.detach())
        gp = rb(rnd(rng, B, H // 2, W // 2, Cc), dcode)
        gx = torch.zeros(B, H, W, Cc, dtype=T, device='cuda')
        lib.call('clamd_maxpool2x2_bwd', ptr(xt), 2 * Cc, ptr(sg), ptr(nd(C, gp, dcode)), Cc, ptr(gx), Cc, B, H, W, Cc, dcode, s)
        ref.backward(torch.from_numpy(gp).cuda().permute(0, 3, 1, 2))         # d(pool)/dx = s * s = 1 at the selected element of each window
        sync()
        assert torch.equal(nf(C, gx, dcode).permute(0, 3, 1, 2), xs.grad)


def test_bn_eval_mode(C):
    rng = np.random.default_rng(15)
    lib, ptr, s = C._lib, C._lib.ptr, C._lib.stream_ptr()
    y = rnd(rng, 2, 9, 4, 6); gamma, beta = rnd(rng, 9), rnd(rng, 9); rm0, rv0 = rnd(rng, 9), np.abs(rnd(rng, 9)) + 0.3
    yt = C.ops.to_nhwc(dev(y), 0)
    vec = torch.zeros(4, 32, device='cuda'); rm, rv = dev(rm0), dev(rv0)
    gt_, bt_ = dev(gamma), dev(beta)          # keep the tensors alive: raw pointers are only borrowed
    nbt = torch.tensor(7, dtype=torch.int64, device='cuda')      # eval mode: the counter stays
    lib.call('clamd_bn_finalize', None, 0, ptr(gt_), ptr(bt_), ptr(rm), ptr(rv), ptr(vec[0]), ptr(vec[1]), ptr(vec[2]), ptr(vec[3]), 32, 9, 48.0, 0.1, 1e-5, ptr(nbt), s)
    out = torch.zeros(2, 4, 6, 32, device='cuda')
    lib.call('clamd_bn_apply', ptr(yt), 32, ptr(vec[0]), ptr(vec[1]), ptr(out), 32, None, 0, 2, 4, 6, 32, 0, s)
    sync()
    assert int(nbt) == 7
    assert rel_l2(C.ops.from_nhwc(out, 9, 0).cpu().numpy(), O.bn_eval_fwd(y, gamma, beta, rm0, rv0)) < 1e-5
    assert np.array_equal(rm.cpu().numpy(), rm0) and np.array_equal(rv.cpu().numpy(), rv0)   # eval updates nothing


def test_ce_counts_out_of_range_labels_and_confusion_guards_small_matrix(C):
    """Labels that are neither ignore_index nor a class: torch's CrossEntropyLoss asserts; here they are excluded from the
    mean AND counted on the criterion (a label bug in a class split cannot hide).  argmax_confusion with a matrix smaller
    than the class count drops predictions outside it instead of writing past the histogram."""
    rng = np.random.default_rng(3)
    logits = torch.from_numpy(rnd(rng, 2, 5, 8, 8)).cuda()
    labels = torch.from_numpy(rng.integers(0, 5, (2, 8, 8))).cuda()
    crit = C.CrossEntropyLoss()
    crit(logits, labels)
    assert int(crit.bad_labels) == 0
    bad = labels.clone(); bad[0, 0, :3] = 7; bad[1, 2, 2] = -100          # three out-of-range labels, one ignored pixel
    loss = crit(logits, bad)
    assert int(crit.bad_labels) == 3
    keep = (bad >= 0) & (bad < 5)
    ref = torch.nn.functional.cross_entropy(logits.permute(0, 2, 3, 1)[keep], bad[keep])
    assert abs(float(loss) - float(ref)) < 1e-5
    conf, _ = C.metrics.argmax_confusion(logits, labels.clamp(max=2), 3)       # 3x3 matrix, 5 classes predicted
    pred = logits.argmax(1)
    m = (pred < 3)
    want = np.zeros((3, 3), np.int64)
    np.add.at(want, (labels.clamp(max=2)[m].cpu().numpy(), pred[m].cpu().numpy()), 1)
    assert np.array_equal(conf.cpu().numpy(), want)


def test_integration_md_snippet_runs(C):
    """The binding example of INTEGRATION.md, executed as written (its Conv2d -> ReLU -> BatchNorm2d unit, models/unet.py:13-15)
    and compared with the oracle: a stale example would put an int into a pointer slot."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc = open(os.path.join(root, 'INTEGRATION.md')).read()
    code = doc[doc.index('```python\nimport ctypes, torch') + len('```python\n'):]
    code = code[:code.index('```')].replace("'continual-learning_amd/libclamd.so'", repr(C._lib.LIB_PATH))
    ns = {}
    exec(compile(code, 'INTEGRATION.md', 'exec'), ns)
    rng = np.random.default_rng(5)
    B, cin, cout, H, W = 2, 20, 40, 24, 40
    segs = [(cin, C.ops.cpad(cin))]
    x, w, b, xt, wf, wd, bp, cin_p, cout_p = _conv_case(C, rng, B, segs, cout, H, W, 0)
    gamma, beta = rnd(rng, cout), rnd(rng, cout)
    rm0, rv0 = np.zeros(cout, np.float32), np.ones(cout, np.float32)
    g_, b_, rm, rv = dev(gamma), dev(beta), dev(rm0), dev(rv0)
    out, y, vec = ns['conv_relu_bn'](xt, wf, bp, g_, b_, rm, rv, B, H, W, cin_p, cout_p, cout)
    sync()
    yr = O.relu_fwd(O.conv3x3_fwd(x, w, b))
    u_ref, _, rm_ref, rv_ref = O.bn_train_fwd(yr, gamma, beta, rm0, rv0)
    assert rel_l2(C.ops.from_nhwc(out, cout, 0).cpu().numpy(), u_ref) < 2e-5
    np.testing.assert_allclose(rm.cpu().numpy(), rm_ref, rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------------------------------------------------
def test_cross_entropy_golden(C, golden):
    g = golden('ops.npz')
    crit = C.CrossEntropyLoss()
    for tag in ('ce', 'ce_ign'):
        lg = dev(g['ce/logits']).requires_grad_()
        loss = crit(lg, dev(g[f'{tag}/labels'], torch.int64))
        loss.backward()
        sync()
        assert abs(float(loss) - float(g[f'{tag}/loss'])) < 2e-6 * max(1.0, abs(float(g[f'{tag}/loss'])))
        assert rel_l2(lg.grad.cpu().numpy(), g[f'{tag}/dlogits']) < 1e-5


def test_cross_entropy_edge_cases(C):
    """All pixels ignored -> loss 0 / zero gradient (no NaN); large logits; 2 classes; 32 classes."""
    crit = C.CrossEntropyLoss()
    lg = (torch.randn(1, 2, 16, 16, device='cuda') * 50).requires_grad_()
    lb = torch.full((1, 16, 16), -100, dtype=torch.int64, device='cuda')
    loss = crit(lg, lb); loss.backward(); sync()
    assert float(loss) == 0.0 and float(lg.grad.abs().max()) == 0.0
    rng = np.random.default_rng(3)
    for K in (2, 32):
        z = (rnd(rng, 2, K, 16, 32) * 30).astype(np.float32)
        y = rng.integers(0, K, (2, 16, 32))
        ref_l, ref_d = O.cross_entropy(z.astype(np.float64), y)
        t = dev(z).requires_grad_()
        l = crit(t, dev(y, torch.int64)); l.backward(); sync()
        assert abs(float(l) - ref_l) < 1e-5 * max(1, abs(ref_l))
        assert rel_l2(t.grad.cpu().numpy(), ref_d) < 1e-5
    with pytest.raises(RuntimeError):
        crit(torch.randn(1, 33, 16, 16, device='cuda'), torch.zeros(1, 16, 16, dtype=torch.int64, device='cuda'))


@pytest.mark.parametrize('name,dcode', DT)
def test_cross_entropy_counted_form_and_nhwc_copy(C, name, dcode):
    """clamd_ce_count + clamd_ce_fwd_bwd_counted (the training-step form: partial-row count, no memset / atomics) against
    clamd_ce_fwd_bwd bit for bit -- loss, d logits, both counters -- for 5 / 21 / 32 classes with ignored and out-of-range labels; the
    second copy of d logits (NHWC, compute dtype, channels K .. 31 zero) equals clamd_nchw_to_nhwc of the NCHW one bit for bit, and
    clamd_scale_by_device_scalar_nhwc scales it as the fp32 kernel scales the NCHW tensor."""
    lib, ptr, s = C._lib, C._lib.ptr, C._lib.stream_ptr()
    L = lib.load()
    rng = np.random.default_rng(21)
    wsb = L.clamd_ce_workspace_bytes()
    off = L.clamd_ce_bad_label_count_offset() // 4
    for K, B, H, W in ((5, 2, 8, 12), (21, 3, 16, 20), (32, 1, 4, 4)):
        z = dev(rnd(rng, B, K, H, W) * 4)
        y = rng.integers(0, K, (B, H, W))
        y[0, 0, :3] = -100; y[0, 1, 0] = K + 2; y[-1, -1, -1] = -7
        yt = dev(y, torch.int64)
        d0, l0, w0 = torch.empty_like(z), torch.empty(3, device='cuda'), torch.zeros(wsb // 4, device='cuda')
        lib.call('clamd_ce_fwd_bwd', ptr(z), ptr(yt), None, 0, 0, 1.0, 0.0, ptr(d0), ptr(l0), ptr(w0), wsb, B, K, H, W, -100, 1.0, s)
        d1, l1, w1 = torch.empty_like(z), torch.empty(3, device='cuda'), torch.full((wsb // 4,), float('nan'), device='cuda')
        nh = torch.full((B, H, W, 32), 5.0, dtype=C.ops.TORCH_DT[dcode], device='cuda')
        lib.call('clamd_ce_count', ptr(yt), B, K, H, W, -100, ptr(w1), wsb, s)
        lib.call('clamd_ce_fwd_bwd_counted', ptr(z), ptr(yt), ptr(d1), ptr(nh), 32, dcode, ptr(l1), ptr(w1), wsb, B, K, H, W, -100, 1.0, s)
        sync()
        assert torch.equal(d0, d1) and torch.equal(l0, l1)
        assert torch.equal(w0[off - 1:off + 1].view(torch.int32), w1[off - 1:off + 1].view(torch.int32)) and int(w1[off:off + 1].view(torch.int32)) == 2
        conv = C.ops.to_nhwc(d1, dcode, cp=32)
        nbad = int((conv.view(torch.int16) != nh.view(torch.int16)).sum())
        assert nbad == 0, (K, nbad, float((nf(C, conv, dcode) - nf(C, nh, dcode)).abs().max()))
        g = torch.tensor([0.5], device='cuda')
        lib.call('clamd_scale_by_device_scalar_nhwc', ptr(nh), nh.numel(), dcode, ptr(g), ptr(d1), d1.numel(), s)      # both copies in one launch
        assert torch.equal(d1, d0 * 0.5)
        sync()
        conv = C.ops.to_nhwc(d1, dcode, cp=32)
        if dcode != 2:       # a power of two: exact.  bf16x3: hi + lo is re-split after the multiplication -- the same value, possibly another (hi, lo) pair
            assert torch.equal(conv.view(torch.int16), nh.view(torch.int16))
        assert float((nf(C, conv, dcode) - nf(C, nh, dcode)).abs().max()) <= 2.0 ** -17 * float(nf(C, conv, dcode).abs().max())
        one = torch.ones(1, device='cuda')
        before = nh.clone()
        lib.call('clamd_scale_by_device_scalar_nhwc', ptr(nh), nh.numel(), dcode, ptr(one), None, 0, s)
        sync()
        assert torch.equal(before.view(torch.int16), nh.view(torch.int16))
    with pytest.raises(RuntimeError, match='H \\* W % 4'):
        lib.call('clamd_ce_fwd_bwd_counted', ptr(z), ptr(yt), ptr(d1), None, 0, 0, ptr(l1), ptr(w1), wsb, 1, 32, 1, 6, -100, 1.0, s)


def test_distillation_loss_vs_oracle(C):
    """Build-defined term (parity unpinned vs the reference, which has none): checked against oracle.distill_kl."""
    rng = np.random.default_rng(4)
    z, zo = rnd(rng, 2, 21, 16, 16) * 3, rnd(rng, 2, 21, 16, 16) * 3
    y = rng.integers(0, 21, (2, 16, 16))
    crit = C.DistillationCrossEntropy(c_old=11, temperature=2.0, lam=0.7)
    t = dev(z).requires_grad_()
    loss = crit(t, dev(y, torch.int64), dev(zo)); loss.backward(); sync()
    l_ce, d_ce = O.cross_entropy(z, y)
    l_kd, d_kd = O.distill_kl(z, zo, 11, 2.0, 0.7)
    assert abs(float(loss) - float(l_ce + l_kd)) < 1e-5
    assert rel_l2(t.grad.cpu().numpy(), d_ce + d_kd) < 1e-5


def test_fused_adam_golden_and_l2(C, golden):
    g = golden('ops.npz')
    p = torch.nn.Parameter(dev(g['adam/p0']))
    opt = C.FusedAdam([p], lr=1e-2, betas=[0.5, 0.99])
    for i in range(3):
        p.grad = dev(g['adam/grads'][i])
        opt.step()
        sync()
        assert rel_l2(p.detach().cpu().numpy(), g[f'adam/p{i + 1}']) < 1e-6
    st = opt.state[p]
    assert rel_l2(st['exp_avg'].cpu().numpy(), g['adam/m3']) < 1e-6
    assert rel_l2(st['exp_avg_sq'].cpu().numpy(), g['adam/v3']) < 1e-6
    assert float(st['step']) == 3.0
    sd = opt.state_dict()
    assert set(sd['state'][0].keys()) == {'step', 'exp_avg', 'exp_avg_sq'}      # torch.optim.Adam's checkpoint layout
    # lr change through param_groups (what LambdaLR does) is picked up
    opt.param_groups[0]['lr'] = 0.0
    before = p.detach().clone(); p.grad = dev(g['adam/grads'][0]); opt.step(); sync()
    assert torch.equal(before, p.detach())
    # L2-to-old-weights (build-defined): grad += 2*lam*(theta - theta_old)
    rng = np.random.default_rng(5)
    p0, old, gr = rnd(rng, 5000), rnd(rng, 5000), rnd(rng, 5000)
    q = torch.nn.Parameter(dev(p0)); o2 = C.FusedAdam([q], lr=1e-2, betas=[0.5, 0.99]); o2.set_l2_anchor([dev(old)], 0.3)
    q.grad = dev(gr); o2.step(); sync()
    ref, _, _ = O.adam_step(p0, gr + 2 * 0.3 * (p0 - old), np.zeros_like(p0), np.zeros_like(p0), 1, 1e-2)
    assert rel_l2(q.detach().cpu().numpy(), ref) < 1e-6
    assert abs(float(o2.l2_penalty()) - 0.3 * float(((p0 - old).astype(np.float64) ** 2).sum())) < 1e-2
    # the penalty is a fixed-order sum of per-workgroup partials (no float atomics): bit-identical run after run, also over
    # many workgroups (3M elements = hundreds of chunks)
    big_p, big_old, big_g = rnd(rng, 3_000_017), rnd(rng, 3_000_017), rnd(rng, 3_000_017)
    pens = []
    for _ in range(5):
        q = torch.nn.Parameter(dev(big_p)); o3 = C.FusedAdam([q], lr=1e-2, betas=[0.5, 0.99]); o3.set_l2_anchor([dev(big_old)], 0.3)
        q.grad = dev(big_g); o3.step(); sync()
        pens.append(o3.l2_penalty().clone())
    assert all(torch.equal(pens[0], t) for t in pens[1:]), [float(t) for t in pens]
    want = 0.3 * float(((big_p - big_old).astype(np.float64) ** 2).sum())
    assert float(pens[0]) == pytest.approx(want, rel=2e-6)


def test_metrics_golden(C, golden):
    g = golden('metrics.npz')
    t, p = g['target'], g['pred']
    onehot = np.zeros((4, 21, 32, 32), np.float32)
    np.put_along_axis(onehot, p[:, None], 1.0, 1)
    lg = dev(onehot)
    conf, pred = C.argmax_confusion(lg, dev(t, torch.int64), 21, want_pred=True)
    sync()
    assert np.array_equal(pred.cpu().numpy(), p)
    assert np.array_equal(conf.cpu().numpy().astype(np.float32), g['conf21'])
    for c in (21, 22):   # trainer.py:188 passes 22
        m = C.eval_metrics(dev(t, torch.int64), lg, c)
        np.testing.assert_allclose([float(v) for v in m], g[f'm{c}'], rtol=1e-6)
    # first maximum wins on ties
    tie = torch.zeros(1, 5, 16, 16, device='cuda')
    _, pr = C.argmax_confusion(tie, torch.zeros(1, 16, 16, dtype=torch.int64, device='cuda'), 5, want_pred=True)
    assert int(pr.max()) == 0


def test_voc_data_path_vs_oracle(C):
    """Pad/CenterCrop/ToTensor/Normalize + palette lookup on the GPU vs the oracle restatement of
    main.py:18-23 and datasets/voc.py:56-72 (incl. void -> 0, an undersized image, an off-palette colour)."""
    rng = np.random.default_rng(30)
    pal = np.array(O.VOC_PALETTE, np.uint8)
    for hs, ws, h, w in [(120, 200, 64, 96), (40, 50, 64, 64), (77, 131, 32, 48)]:
        img = rng.integers(0, 256, (hs, ws, 3)).astype(np.uint8)
        mask = pal[rng.integers(0, 22, (hs, ws))]
        ri, rl = O.voc_prepare(img, mask, h, w)
        gi, gl = C.data.prepare_sample(torch.from_numpy(img).cuda(), torch.from_numpy(mask).cuda(), (h, w))
        sync()
        np.testing.assert_allclose(gi.cpu().numpy(), ri, rtol=0, atol=1e-6)
        assert np.array_equal(gl.cpu().numpy(), rl)
        back = C.data.to_rgb(gl[None])
        assert np.array_equal(back.cpu().numpy(), O.to_rgb(rl[None]))
    bad = mask.copy(); bad[hs // 2, ws // 2] = (1, 2, 3)          # a pixel inside the centre crop
    with pytest.raises(ValueError):
        C.data.prepare_sample(torch.from_numpy(img).cuda(), torch.from_numpy(bad).cuda(), (h, w))


def test_voc_palette_kernels_vs_reference_fixture(C, golden):
    """The palette kernels against vectors captured from the reference's own voc.to_mask / voc.to_rgb (tests/golden/voc.npz):
    Pad(10) + CenterCrop to the image's own size is the identity, so prepare_sample's labels must equal to_mask's."""
    g = golden('voc.npz')
    mask = torch.from_numpy(g['mask_rgb']).cuda()
    hs, ws = mask.shape[:2]
    _, labels = C.data.prepare_sample(None, mask, (hs, ws))
    sync()
    assert np.array_equal(labels.cpu().numpy(), g['labels'])
    rgb = C.data.to_rgb(torch.from_numpy(g['to_rgb_in']).cuda())
    assert np.array_equal(rgb.cpu().numpy().astype(np.float64), g['to_rgb_out'])
