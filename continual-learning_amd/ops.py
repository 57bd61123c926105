"""Thin host-side helpers over the C ABI: weight-packing job tables and single-kernel wrappers.

Used by the UNet engine (unet.py) and by the per-kernel parity tests; every function here ends in a libclamd launch.
"""
import numpy as np
import torch

from . import _lib
from ._lib import call, ptr

TORCH_DT = {_lib.F32: torch.float32, _lib.BF16: torch.bfloat16, _lib.SPLIT: torch.float32}

_PACK_DT = np.dtype({'names': ['src', 'dst', 'T', 'Np', 'Kp', 'N', 'K', 'st', 'sn', 'sk', 'dt', 'dn', 'dk',
                               'n_seg0', 'n_seg0p', 'k_seg0', 'k_seg0p', 'flip', 'dst_f32', 'block0', 'kc', 'kscale', 'dst_t'],
                     'formats': ['u8', 'u8', 'i4', 'i4', 'i4', 'i4', 'i4', 'i8', 'i8', 'i8', 'i8', 'i8', 'i8',
                                 'i4', 'i4', 'i4', 'i4', 'i4', 'i4', 'i4', 'i4', 'u8', 'u8'],
                     'offsets': [0, 8, 16, 20, 24, 28, 32, 40, 48, 56, 64, 72, 80, 88, 92, 96, 100, 104, 108, 112, 116, 120, 128],
                     'itemsize': 136})


def cpad(c):
    """Physical channel count: next power of two, at least 32 (kernels move 16-byte channel groups)."""
    return max(32, 1 << (int(c) - 1).bit_length())


class PackTable:
    """Job table for clamd_pack: one fused launch re-packs every fp32 master parameter (see include/clamd.h)."""

    KC = {_lib.F32: 16, _lib.BF16: 32, _lib.SPLIT: 16}     # channels per 64-byte K-chunk of the 3x3 kernels

    def __init__(self, dcode):
        self.jobs = []
        self.dev_table = None
        self.dcode = dcode
        self.kc = self.KC[dcode]

    def add(self, src, dst, T, Np, Kp, N, K, st, sn, sk, dt, dn, dk, nseg=None, kseg=None, flip=0, f32=0, kc=0, kscale=None, dst_t=None):
        nseg = nseg or (N, Np)
        kseg = kseg or (K, Kp)
        self.jobs.append((src.data_ptr(), dst.data_ptr(), T, Np, Kp, N, K, st, sn, sk, dt, dn, dk,
                          nseg[0], nseg[1], kseg[0], kseg[1], flip, f32, 0, kc, kscale.data_ptr() if kscale is not None else 0,
                          dst_t.data_ptr() if dst_t is not None else 0))

    # ---- the layouts of include/clamd.h ----
    def conv3x3(self, w, wf, wd, cin_segs, cout, kscale=None):
        """w [Cout][Cin][3][3] fp32 -> wf [Cin_p/kc][9][Cout_p][kc] (forward) and wd [Cout_p/kc][9][Cin_p][kc] (data
        gradient, taps flipped): K-chunk-major, so the [tap][n] slab of one 64-byte K-chunk is contiguous.
        cin_segs: [(logical, physical), ...] one or two channel segments (concat inputs).
        kscale (fp32 [Cin_p], optional): the forward filters are multiplied by kscale[ci] (a folded BatchNorm, bnfold.hip)."""
        cin = sum(s[0] for s in cin_segs)
        cin_p = sum(s[1] for s in cin_segs)
        cout_p = cpad(cout)
        seg = (cin_segs[0][0], cin_segs[0][1]) if len(cin_segs) == 2 else None
        if wf is not None:      # both layouts from ONE read of the source tile (PackJob::dst_t)
            self.add(w, wf, 9, cout_p, cin_p, cout, cin, 1, cin * 9, 9, cout_p * cin_p, cin_p, 1, kseg=seg, kc=self.kc, kscale=kscale, dst_t=wd)
        elif wd is not None:
            self.add(w, wd, 9, cin_p, cout_p, cin, cout, 1, 9, cin * 9, cin_p * cout_p, cout_p, 1, nseg=seg, flip=1,
                     kc=self.kc)

    def convT(self, w, wf, wd, cin, cout):
        """w [Cin][Cout][2][2] -> wf [4][Cout_p][Cin_p] and wd [Cin_p][4][Cout_p]."""
        cin_p, cout_p = cpad(cin), cpad(cout)
        self.add(w, wf, 4, cout_p, cin_p, cout, cin, 1, 4, cout * 4, cout_p * cin_p, cin_p, 1)
        if wd is not None:
            self.add(w, wd, 4, cin_p, cout_p, cin, cout, 1, cout * 4, 4, cout_p, 4 * cout_p, 1)

    def head(self, w, wf, wd, cin, k, kscale=None):
        """w [K][Cin][1][1] -> wf [K_p][Cin_p] and wd [Cin_p][K_p]; kscale as in conv3x3."""
        cin_p, kp = cpad(cin), cpad(k)
        if wf is not None:
            self.add(w, wf, 1, kp, cin_p, k, cin, 0, cin, 1, 0, cin_p, 1, kscale=kscale)
        if wd is not None:
            self.add(w, wd, 1, cin_p, kp, cin, k, 0, 1, cin, 0, kp, 1)

    def vector(self, v, dst, c):
        """1-D fp32 vector -> zero-padded fp32 [cpad(c)]."""
        self.add(v, dst, 1, 1, cpad(c), 1, c, 0, 0, 1, 0, 0, 1, f32=1)

    def finalize(self, device):
        lib = _lib.load()
        assert lib.clamd_sizeof_pack_job() == _PACK_DT.itemsize
        tile = lib.clamd_pack_tile()
        arr = np.zeros(len(self.jobs), dtype=_PACK_DT)
        blk = 0
        for i, j in enumerate(self.jobs):
            assert j[2] <= 9, 'pack: at most 9 taps'
            arr[i] = j
            arr[i]['block0'] = blk
            blk += ((j[3] + tile - 1) // tile) * ((j[4] + tile - 1) // tile)
        self.dev_table = torch.from_numpy(arr.view(np.uint8).copy()).to(device)
        self.nblocks = blk
        return self

    def run(self, dcode=None, stream=None):
        assert dcode is None or dcode == self.dcode
        call('clamd_pack', ptr(self.dev_table), len(self.jobs), self.nblocks, self.dcode, stream or _lib.stream_ptr())


_WINO_DT = np.dtype({'names': ['w', 'dst', 'Np', 'Kp', 'N', 'K', 'n_seg0', 'n_seg0p', 'k_seg0', 'k_seg0p', 'dgrad', 'block0', 'kscale'],
                     'formats': ['u8', 'u8', 'i4', 'i4', 'i4', 'i4', 'i4', 'i4', 'i4', 'i4', 'i4', 'i4', 'u8'],
                     'offsets': [0, 8, 16, 20, 24, 28, 32, 36, 40, 44, 48, 52, 56], 'itemsize': 64})


class WinoPackTable:
    """Job table for clamd_wino_pack / clamd_wino24_pack: Winograd filter transforms of every 3x3 conv of one form in one
    launch (fp32 path).  ``planes`` = 16: F(2x2,3x3) (wino.hip); 24: F(2x4,3x3) (wino24.hip); 36: F(4x4,3x3) (wino44g.hip; no kscale)."""

    def __init__(self, planes=16):
        assert planes in (16, 24, 36)
        self.jobs = []
        self.planes = planes

    def conv3x3(self, w, wf, wd, cin_segs, cout, kscale=None):
        """w [Cout][Cin][3][3] fp32 -> wf [Cin_p/8][planes][Cout_p][8] (forward) and wd [Cout_p/8][planes][Cin_p][8] (data
        gradient: tap-flipped, transposed).  cin_segs and kscale as in PackTable.conv3x3."""
        cin = sum(s[0] for s in cin_segs)
        cin_p = sum(s[1] for s in cin_segs)
        cout_p = cpad(cout)
        seg = (cin_segs[0][0], cin_segs[0][1]) if len(cin_segs) == 2 else (cin, cin_p)
        if wf is not None:
            self.jobs.append((w.data_ptr(), wf.data_ptr(), cout_p, cin_p, cout, cin, cout, cout_p, seg[0], seg[1], 0, 0,
                              kscale.data_ptr() if kscale is not None else 0))
        if wd is not None:
            self.jobs.append((w.data_ptr(), wd.data_ptr(), cin_p, cout_p, cin, cout, seg[0], seg[1], cout, cout_p, 1, 0, 0))

    def finalize(self, device):
        lib = _lib.load()
        assert lib.clamd_sizeof_wino_pack_job() == _WINO_DT.itemsize
        arr = np.zeros(len(self.jobs), dtype=_WINO_DT)
        blk = 0
        for i, j in enumerate(self.jobs):
            arr[i] = j
            arr[i]['block0'] = blk
            blk += (j[2] * j[3] + 255) // 256
        self.dev_table = torch.from_numpy(arr.view(np.uint8).copy()).to(device)
        self.nblocks = blk
        return self

    def run(self, stream=None):
        call({16: 'clamd_wino_pack', 24: 'clamd_wino24_pack', 36: 'clamd_wino44_pack'}[self.planes], ptr(self.dev_table), len(self.jobs), self.nblocks,
             stream or _lib.stream_ptr())


# ---- single-kernel wrappers (tests, small tools) ----------------------------------------------------------------
def to_nhwc(x_nchw, dcode, cp=None):
    B, C, H, W = x_nchw.shape
    cp = cp or cpad(C)
    out = torch.empty(B, H, W, cp, dtype=TORCH_DT[dcode], device=x_nchw.device)
    call('clamd_nchw_to_nhwc', ptr(x_nchw.contiguous().float()), ptr(out), cp, B, C, H, W, cp, 1.0, dcode, _lib.stream_ptr())
    return out


def split_encode(x):
    """fp32 [..., Cp] (Cp % 16 == 0) -> the bf16x3 device layout of the same logical tensor: per 16-channel group
    [16 x bf16 hi][16 x bf16 lo] with hi = rne_bf16(x), lo = rne_bf16(x - hi), held in a float32-typed container of the same
    shape (csrc/common.hip.h Vec8<split_t>).  Host-side helper for tests and tools; the step itself never calls it."""
    assert x.dtype == torch.float32 and x.shape[-1] % 16 == 0
    g = x.contiguous().reshape(*x.shape[:-1], x.shape[-1] // 16, 16)
    hi = g.to(torch.bfloat16)
    lo = (g - hi.float()).to(torch.bfloat16)
    return torch.stack((hi, lo), dim=-2).contiguous().view(torch.int16).reshape(*x.shape[:-1], 2 * x.shape[-1]) \
        .view(torch.float32)


def split_decode(t):
    """Inverse of split_encode: the fp32 values hi + lo of a bf16x3 tensor (any 16-channel-aligned slice of one)."""
    assert t.dtype == torch.float32 and t.shape[-1] % 16 == 0
    raw = t.contiguous().view(torch.int16).reshape(*t.shape[:-1], t.shape[-1] // 16, 2, 16)
    f = (raw.to(torch.int32) << 16).view(torch.float32)
    return (f[..., 0, :] + f[..., 1, :]).reshape(t.shape)


def randn_nhwc(dcode, *shape, device='cuda'):
    """Standard-normal NHWC activation in the storage layout of compute dtype `dcode` (timing tools)."""
    x = torch.randn(*shape, device=device)
    return split_encode(x) if dcode == 2 else x.to(TORCH_DT[dcode])


def from_nhwc(t, C, dcode):
    B, H, W, ldc = t.shape
    out = torch.empty(B, C, H, W, dtype=torch.float32, device=t.device)
    call('clamd_nhwc_to_nchw', ptr(t), ldc, ptr(out), B, C, H, W, dcode, _lib.stream_ptr())
    return out
