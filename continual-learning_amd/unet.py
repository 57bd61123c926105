"""Drop-in ``UNet(num_classes, in_dim=3, conv_dim=64)`` whose forward/backward run on libclamd's HIP kernels.

Mirrors the reference module surface (models/unet.py:40-92): same constructor, the same 82 parameters in the same
registration order, the same 136 ``state_dict`` keys ('enc1.0.weight', 'enc2.block.1.weight', 'dec1.block.6.weight',
'last.6.bias', ...), fp32 NCHW in / fp32 NCHW logits out, ``.train()/.eval()`` BatchNorm semantics, autograd-attached.
The child modules (nn.Conv2d, nn.BatchNorm2d, ...) are only PARAMETER CONTAINERS with torch's default init; their own
``forward`` is never called.  ``UNet.forward`` runs the whole network as ONE ``torch.autograd.Function`` whose forward
and backward are fixed schedules of C-ABI kernel launches on NHWC activations (fp32 or bf16) kept in buffers owned by
PyTorch's allocator.  There is no CPU path: calling it with CPU tensors raises.
"""
import os
import weakref

import torch
import torch.nn as nn

from . import _lib
from ._lib import call, ptr, tune_ptr
from .ops import PackTable, WinoPackTable, cpad

BN_EPS, BN_MOMENTUM = 1e-5, 0.1

# Fused BN-backward sums in the epilogue of the data-gradient kernel that produces the BN-output gradient:
# 'auto' = where that kernel is the persistent bf16 kernel (sums stay in registers across tiles, one flush per workgroup,
# the saved activation is prefetched under the last K-step); True = everywhere the kernels support it (every dgrad
# launch of the other kernels got 30-80 us slower than the 33-us reduce pass it replaced); False = never.
FUSE_BN_SUMS = {'0': False, 'false': False, '1': True, 'true': True}.get(os.environ.get('CLAMD_FUSE_BN_SUMS', 'auto').lower(), 'auto')

# fp32 path: forward and data gradient of the 3x3 convolutions by Winograd F(2x2,3x3) (csrc/wino.hip): 2.25x fewer MFMA
# cycles, fp32 transforms (error vs fp64 3.5e-7 against 2.3e-7 for the direct sum).  False = direct implicit GEMM.
WINOGRAD = True
# ... and, where the image width is a multiple of 4, by the hybrid F(2x4,3x3) (csrc/wino24.hip): 3 instead of 4 multiply-adds
# per output (1.17x faster launches, error vs fp64 1e-6).  False = F(2x2,3x3) everywhere.
WINOGRAD24 = True
# ... and the weight gradient too (csrc/wino24_wgrad.hip) where it measures faster: 'auto' = images of at least 64x64
# (tools/wino24_wgrad_ab.py: 1.02-1.04x there, 0.88-1.00x on the deep layers -- both operands are transformed in the loop,
# 3.5 transform VALU per MFMA, so the 25 % fewer MFMAs buy little); True = everywhere it applies; False = F(2x2,3x3).
WINOGRAD24_WGRAD = 'auto'
# ... and, for the wide layers, with the operands transformed ONCE per tensor (csrc/wino24g.hip): the in-kernel transform is
# redone by every output-slab workgroup (8-16x per tile at 512/1024 channels) and costs 2-3.5 VALU per fp32 MFMA; the
# transform-free K loop runs at 0.77-0.90 of the MFMA pipe instead of 0.57-0.66 (tools/wino24g_ab.py).  The transformed
# input (3x the activation) is kept from the forward pass and is also the x-side operand of the weight-gradient GEMM.
# 'auto' = where the transform pass pays for itself (see _Engine.unit); False = never.
PRETRANSFORM = os.environ.get('CLAMD_PRETRANSFORM', 'auto')
PRETRANSFORM = {'0': False, 'false': False, '1': True, 'true': True}.get(str(PRETRANSFORM).lower(), 'auto')
# ... and those pre-transformed layers by the 2-D F(4x4,3x3) (csrc/wino44g.hip: 2.25 instead of 3 multiply-adds per output, a transformed
# input of 2.25x instead of 3x the activation; error vs fp64 2-3.5e-6) where a launch has a chip's worth of its 512-pixel x 64-channel work
# items: the 32x32 and 64x64 levels at config 2 (tools/wino44g_ab.py: transform + forward 1.23-1.30x, weight gradient 1.0-1.26x faster
# there; 0.67-0.70x at 16x16, where 128 work items leave half the chip idle).  'auto' = that rule; True = wherever it applies; False = never.
WINOGRAD44 = os.environ.get('CLAMD_WINOGRAD44', 'auto')
WINOGRAD44 = {'0': False, 'false': False, '1': True, 'true': True}.get(str(WINOGRAD44).lower(), 'auto')
# ... and the BatchNorm in front of such a convolution is applied by the transform kernel on load where nothing else reads the
# BatchNorm output (the first unit of enc3/enc4/dec1/dec2/dec3): one HBM pass less per unit.  False = always run clamd_bn_apply.
FOLD_BN_INTO_TRANSFORM = os.environ.get('CLAMD_FOLD_BN', '1') != '0'
NARROW_DIRECT = os.environ.get('CLAMD_NARROW_DIRECT', '1') != '0'
WGRAD_TAIL_EARLY = os.environ.get('CLAMD_WGRAD_TAIL_EARLY', '1') != '0'      # see _Engine._conv_bwd
NARROW_PRE_WGRAD = os.environ.get('CLAMD_NARROW_PRE_WGRAD', '1') != '0'      # the 128-channel layers pre-transformed (forward + weight gradient), see _Engine
# The narrow layers (in-kernel transform / bf16 direct kernels) cannot take the affine on load -- their loops are VALU-bound -- so there
# the BatchNorm between the two convolutions of a block (models/unet.py:13-18) is folded ALGEBRAICALLY into the second one
# (csrc/bnfold.hip): filters packed with scale[ci] once the statistics are final, the shift as a border-class bias table in the
# epilogue, the weight gradient fixed up from the gradient's border sums.  The normalised tensor is never written: the bn_apply pass
# of the first unit of enc1 / enc2 / dec4 / last (268 + 268 MB at level 0 in fp32) leaves the forward pass.  Up to
# FOLD_FILTERS_MAX_CHANNELS input channels: the per-step filter pack is on the critical path, the apply pass shrinks with depth.
# Interleaved A/B (bench.py, ms per step): fp32 21.30 -> 21.08, bf16x3 15.17 -> 14.84; bf16 6.81 = 6.81 in round 3 (the apply passes are half
# the bytes while the pack launch and the border tiles' table lookups cost the same) and 6.489 -> 6.415 in round 4, with the
# channels-in-the-lane epilogue (igemm_pws.hip: the class-4 bias is the accumulator's start, only the lanes of border pixels of border
# tiles add a difference row) and the pack launch at 7 us: every compute dtype folds now.  False = never.
FOLD_BN_INTO_FILTERS = os.environ.get('CLAMD_FOLD_FILTERS', '1').lower() not in ('0', 'false')
FOLD_FILTERS_MAX_CHANNELS = int(os.environ.get('CLAMD_FOLD_FILTERS_MAX_CHANNELS', '128'))
# fp32 path: the same fold for the OUTPUT of an encoder block -- pooled into the next block, concatenated into the decoder (models/unet.py:80-87)
# -- where both readers are narrow F(2x4) convolutions (enc1 -> enc2's first conv and last's first conv at config 2): the block's second conv
# writes its conv+ReLU output straight into the concat slice, one pass pools it (window minimum where the BatchNorm scale is negative:
# max(s x + t) = s min(x) + t), and the two readers take scale / shift in their filters and bias tables.  The pooled bn_apply pass of enc1
# (603 MB, the largest elementwise pass of the step) becomes a 335 MB pooling pass.
FOLD_POOLED = os.environ.get('CLAMD_FOLD_POOLED', '1') != '0'
# fp32 path, pre-transformed weight gradients: the gradient-side transform (HBM-bound) on a THIRD stream, so that it runs beside the
# weight-gradient GEMM of the unit before (which leaves 188 registers per SIMD free) instead of in front of its own GEMM on the second
# stream, and that GEMM can start the moment the data gradient of its unit has finished.
WGRAD_XFORM_STREAM = os.environ.get('CLAMD_WGRAD_XFORM_STREAM', '1') != '0'
# Weight-gradient kernels (and the bias-gradient channel sums of the ConvTranspose / head layers) go to a second HIP stream:
# they are off the critical chain of the backward pass (dgrad -> BatchNorm-backward reduce / finalize / apply -> dgrad ...),
# and the HBM-bound BatchNorm passes of the NEXT unit fit beside a weight-gradient workgroup on the same CU (one wave per
# SIMD, <= 64 registers, <= 8 KB LDS), so they run under the MFMA-bound kernel instead of after it.  Results are unchanged
# (same kernels, same arguments); joined back before backward() returns.  Off while bench.py times single launches.
WGRAD_STREAM = os.environ.get('CLAMD_WGRAD_STREAM', '1') != '0'      # =0: everything on one stream (kernel-trace profiles)
# The plain filter pack of everything behind enc3 (96 % of the parameters; HBM-bound) on the second stream under enc1-enc3 instead of in
# front of the forward pass (=0: one launch chain on the main stream, as before round 4)
PACK_LATE_STREAM = os.environ.get('CLAMD_PACK_LATE_STREAM', '1') != '0'
PACK_LATE_AT = int(os.environ.get('CLAMD_PACK_LATE_AT', '2'))      # index of the convolution unit it is released beside (2 = enc2's first)

# bench.py sets this to a list to get per-launch HIP-event timings of the MFMA kernels:
# entries (tag, algorithmic_flops, start_event, end_event, algorithmic_bytes), recorded on the stream the kernel is
# launched on.  Algorithmic bytes of a 3x3 convolution launch = both activations once + the filters once.
KERNEL_TIMING = None


_TIMED_UNIT = ['', 1.0]   # conv unit being launched and the fraction of its algorithmic FLOPs the kernel executes (Winograd:
#                         # 16/36 or 24/72); only read while KERNEL_TIMING is set (bench.py, tools/layer_table.py)


_SECOND_STREAM = {}


def _second_stream(dev):
    """ONE second stream per device and process, shared by every engine: HIP multiplexes streams onto a handful of hardware
    queues (4 by default) in creation order, and a kernel queues behind whatever shares its hardware queue -- a stream per
    engine would sooner or later land on the queue RCCL's kernels use."""
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _SECOND_STREAM:
        # default priority: the device offers only (normal, high), and giving either stream the high one changed nothing
        # measurable (tools/cu_steal.py, base and held-CU cases within 0.5 %)
        _SECOND_STREAM[key] = torch.cuda.Stream(device=dev)
    return _SECOND_STREAM[key]


_THIRD_STREAM = {}


def _third_stream(dev):
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _THIRD_STREAM:
        _THIRD_STREAM[key] = torch.cuda.Stream(device=dev)
    return _THIRD_STREAM[key]


def _timed(tag, flops, nbytes, name, *args):
    kt = KERNEL_TIMING
    if kt is None:
        call(name, *args)
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    call(name, *args)
    e1.record()
    kt.append((tag, flops, e0, e1, nbytes, _TIMED_UNIT[0], _TIMED_UNIT[1]))


def _hbm(family, nbytes, name, *args):
    """A launch of an HBM-bound kernel family (A5 / A6 BatchNorm and pooling passes, A7 ConvTranspose, A9 head, A10 loss,
    A13 Adam, enc1.0) with its ALGORITHMIC bytes -- every tensor it has to read or write, once: timed per launch by bench.py's instrumented
    steps (`hbm_kernels` on the JSON line), a plain launch otherwise."""
    _timed('hbm:' + family, 0.0, nbytes, name, *args)


def stage_table(num_classes, in_dim=3, conv_dim=64):
    """Structure of models/unet.py:49-72: (name, wrapped_in_block, pool_first, conv/bn module indices, tail)."""
    d = conv_dim
    t = [dict(name='enc1', wrapped=False, pool=False, convs=[(0, 2, in_dim, d), (3, 5, d, d)], tail=None)]
    c = d
    for i in (2, 3, 4):
        t.append(dict(name=f'enc{i}', wrapped=True, pool=True, convs=[(1, 3, c, 2 * c), (4, 6, 2 * c, 2 * c)], tail=None))
        c *= 2
    for i, (cin, mid, cout) in enumerate([(8 * d, 16 * d, 8 * d), (16 * d, 8 * d, 4 * d), (8 * d, 4 * d, 2 * d),
                                          (4 * d, 2 * d, d)], 1):
        t.append(dict(name=f'dec{i}', wrapped=True, pool=False, convs=[(0, 2, cin, mid), (3, 5, mid, mid)],
                      tail=('convT', 6, mid, cout)))
    t.append(dict(name='last', wrapped=False, pool=False, convs=[(0, 2, 2 * d, d), (3, 5, d, d)],
                  tail=('head', 6, d, num_classes)))
    return t


class _NoForward:
    """The LAYERS inside UNet's blocks are parameter containers (torch's default init, the reference's state_dict names).  Their own
    ``forward`` would run stock torch operators (MIOpen) -- a silent fallback this package does not have: it raises.  The BLOCKS
    (models/unet.py:8-38: enc1 ... last, and their ``.block`` sequences) can be called on their own: blocks.py runs them as one
    autograd Function over libclamd kernels."""

    def forward(self, *args, **kwargs):
        raise RuntimeError(f'{type(self).__name__}.forward: the layers of continual-learning_amd.UNet only hold parameters; '
                           'run UNet.forward / UNet.predict, or a whole block (model.enc2(x), model.dec1(x): blocks.py) -- there is no '
                           'stock-torch path for a single layer')


class _Seq(_NoForward, nn.Sequential):
    def forward(self, x):
        spec = getattr(self, '_block_spec', None)
        if spec is None:
            return _NoForward.forward(self, x)
        from . import blocks
        return blocks.run_block(self, spec[0], spec[1], x)


class _Conv2d(_NoForward, nn.Conv2d):
    pass


class _ConvTranspose2d(_NoForward, nn.ConvTranspose2d):
    pass


class _BatchNorm2d(_NoForward, nn.BatchNorm2d):
    pass


class _ReLU(_NoForward, nn.ReLU):
    pass


class _MaxPool2d(_NoForward, nn.MaxPool2d):
    pass


class _Stage(_NoForward, nn.Module):
    """Gives the 'encN.block.K' / 'decN.block.K' key names of models/unet.py:8-38."""

    def __init__(self, layers):
        super().__init__()
        self.block = _Seq(*layers)

    def forward(self, x):
        return self.block(x)


def _stage_modules(st):
    layers = [_MaxPool2d(2, 2)] if st['pool'] else []
    for _, _, cin, cout in st['convs']:
        layers += [_Conv2d(cin, cout, 3, 1, 1), _ReLU(), _BatchNorm2d(cout)]
    if st['tail'] is not None:
        kind, _, cin, cout = st['tail']
        layers.append(_ConvTranspose2d(cin, cout, 2, 2) if kind == 'convT' else _Conv2d(cin, cout, 1, 1))
    return layers


_DTYPES = {'fp32': (_lib.F32, torch.float32), 'float32': (_lib.F32, torch.float32),
           'bf16': (_lib.BF16, torch.bfloat16), 'bfloat16': (_lib.BF16, torch.bfloat16),
           'bf16x3': (_lib.SPLIT, torch.float32)}


class UNet(nn.Module):
    """models/unet.py:40-92.  ``compute_dtype``: 'fp32' (exact-fp32 MFMA, the reference's arithmetic), 'bf16'
    (bf16 activations/packed weights, fp32 accumulation, fp32 master weights and BatchNorm statistics) or 'bf16x3'
    (fp32 activations; every MFMA operand split into bf16 hi+lo, three bf16 MFMAs per product, fp32 accumulation)."""

    def __init__(self, num_classes, in_dim=3, conv_dim=64, compute_dtype='fp32'):
        super().__init__()
        self.num_classes, self.in_dim, self.conv_dim = num_classes, in_dim, conv_dim
        if compute_dtype not in _DTYPES:
            raise ValueError(f'compute_dtype must be one of {sorted(_DTYPES)}')
        self.compute_dtype = compute_dtype
        self._table = stage_table(num_classes, in_dim, conv_dim)
        for st in self._table:
            layers = _stage_modules(st)
            mod = _Stage(layers) if st['wrapped'] else _Seq(*layers)
            (mod.block if st['wrapped'] else mod)._block_spec = (st, _DTYPES[compute_dtype][0])      # stand-alone call of the block: blocks.py
            self.add_module(st['name'], mod)
        self._engines = {}
        self.grad_sync = None          # set by ddp.GradSync to overlap RCCL all-reduce with backward
        self._tuning = None

    @property
    def tuning(self):
        """Kernel-structure selection of THIS model (a `clamd_tuning`, see include/clamd.h), passed to every launch: the
        library has no process-wide knobs.  Fields may be changed between steps (A/B tools, ddp.GradSync)."""
        if self._tuning is None:
            self._tuning = _lib.Tuning()
        return self._tuning

    def _seq(self, st):
        m = getattr(self, st['name'])
        return m.block if st['wrapped'] else m

    def _replicate_for_data_parallel(self):
        """``nn.DataParallel(model)`` (trainer.py:120-122) on ONE device calls the module directly and works as is.  On several
        devices it would replicate this module into threads of one process every forward; the engine (activation buffers,
        streams, flat gradient buffer) belongs to one device, and the reference's scheme is what ddp.GradSync replaces:
        one process per GPU, RCCL all-reduce overlapped with backward."""
        raise RuntimeError('continual-learning_amd.UNet cannot be replicated by nn.DataParallel across devices: run one process per '
                           'GPU (torch.distributed.run) with continual-learning_amd.ddp.init_rccl + ddp.GradSync(model, optimizer)')

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError('continual-learning_amd.UNet runs only on an MI355X GPU tensor: there is no CPU fallback')
        if x.dim() != 4 or x.shape[1] != self.in_dim:
            raise ValueError(f'expected input [B,{self.in_dim},H,W], got {tuple(x.shape)}')
        B, _, H, W = x.shape
        assert H % 16 == 0 and W % 16 == 0, 'input size(H, W) must be a multiple of 16 (four 2x2 pools and matching skip concats)'
        key = (B, H, W, x.device.index)
        eng = self._engines.get(key)
        if eng is None:
            eng = _Engine(self, B, H, W, x.device)
            self._engines = {key: eng}     # one shape at a time: activations are sized for it
        params = [p for p in self.parameters()]
        out = _UNetFn.apply(x.contiguous().float(), eng, *params)
        out._clamd_engine = (weakref.ref(eng), eng.generation)      # lets this package's loss write d logits where the backward pass reads it
        return out

    @torch.no_grad()
    def predict(self, x):
        """``torch.max(self(x), 1)[1]`` (trainer.py:279) without materialising the logits: the arg-max over classes runs
        in the epilogue of the head kernel.  int64 [B,H,W]; BatchNorm follows ``self.training``."""
        if not x.is_cuda:
            raise RuntimeError('continual-learning_amd.UNet runs only on an MI355X GPU tensor: there is no CPU fallback')
        if x.dim() != 4 or x.shape[1] != self.in_dim:
            raise ValueError(f'expected input [B,{self.in_dim},H,W], got {tuple(x.shape)}')
        B, _, H, W = x.shape
        assert H % 16 == 0 and W % 16 == 0, 'input size(H, W) must be a multiple of 16 (four 2x2 pools and matching skip concats)'
        key = (B, H, W, x.device.index)
        eng = self._engines.get(key)
        if eng is None:
            eng = _Engine(self, B, H, W, x.device)
            self._engines = {key: eng}
        return eng.forward(x.contiguous().float(), [p for p in self.parameters()], predict=True)

    def extra_repr(self):
        return f'num_classes={self.num_classes}, in_dim={self.in_dim}, conv_dim={self.conv_dim}, compute={self.compute_dtype}'


def dlogits_sink(logits, B, K, H, W):
    """For loss.CrossEntropyLoss: the engine whose forward produced `logits` (and nothing since), or None.  Its `dl` buffer [B,H,W,Kp] takes a
    second copy of d logits in the head data gradient's own layout; the engine uses it when the gradient autograd hands back is the very
    tensor the loss wrote (same storage, untouched: `dl_src`) and converts that tensor as before otherwise."""
    tag = getattr(logits, '_clamd_engine', None)
    if tag is None:
        return None
    eng, gen = tag[0](), tag[1]
    if eng is None or eng.generation != gen or (eng.B, eng.K, eng.H, eng.W) != (B, K, H, W) or not eng.fwd_training or eng.dl.device != logits.device:
        return None
    return eng


class _UNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eng, *params):
        logits = eng.forward(x, params)
        ctx.eng = eng
        ctx.gen = eng.generation
        ctx.nparams = len(params)
        return logits

    @staticmethod
    def backward(ctx, gout):
        eng = ctx.eng
        if ctx.gen != eng.generation:
            raise RuntimeError('UNet.backward: the saved activations were overwritten by a later forward of the same module')
        grads = eng.backward(gout.contiguous().float())
        return (None, None) + tuple(grads)


class _FoldSource:
    """What a folded convolution reads instead of a normalised tensor: the raw tensor (`y`, pitch `cout_p`) and the per-input-channel
    scale / shift (`vec[0]`, `vec[1]`) that live in its filters and bias table -- the interface of the producing _Conv the pair fold uses."""

    def __init__(self, y, pitch, scale, shift):
        self.y, self.cout_p, self.vec, self.apply_in_filters = y, pitch, [scale, shift], False


class _Conv:
    """One Conv3x3 -> ReLU -> BatchNorm unit and everything it needs in both directions."""
    pass


class _Engine:
    @property
    def model(self):
        return self._model_ref()

    def __init__(self, model, B, H, W, device):
        lib = _lib.load()
        self._model_ref = weakref.ref(model)      # the model owns its engines; a strong reference back would leave the
        #                                           multi-GB activation buffers to the cyclic garbage collector
        self.B, self.H, self.W, self.dev = B, H, W, device
        self.dcode, self.tdtype = _DTYPES[model.compute_dtype]
        self.wino = bool(WINOGRAD) and self.dcode == _lib.F32
        self.pack_late_stream, self.pack_late_at = bool(PACK_LATE_STREAM), int(PACK_LATE_AT)
        self.tuning = model.tuning
        self.NS = lib.clamd_bn_bwd_nsums()
        self.generation = 0
        self.dl_src = None
        self.fwd_training = False
        self.esize = 2 if self.dcode == _lib.BF16 else 4      # activation element size in HBM (bf16x3 stores fp32)
        K, d = model.num_classes, model.conv_dim
        self.K, self.Kp = K, cpad(K)
        T = self.tdtype
        dev = device

        def act(level, c):
            return torch.zeros(B, H >> level, W >> level, c, dtype=T, device=dev)

        named = dict(model.named_parameters())
        bufs = dict(model.named_buffers())
        self.param_names = [n for n, _ in model.named_parameters()]
        # flat gradient buffer in REVERSE registration order (= order gradients are produced): contiguous buckets
        sizes = [named[n].numel() for n in self.param_names]
        self.gflat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
        # NOTE: no tensor views of the gradients are kept alive here.  Fresh views are handed to autograd at the end
        # of every backward so that AccumulateGrad can adopt them as .grad without a copy (it clones when the
        # incoming gradient has other owners).
        self.goffset, self.gshape, off = {}, {}, 0
        for n in reversed(self.param_names):
            k = named[n].numel()
            self.goffset[n] = (off, k)
            self.gshape[n] = tuple(named[n].shape)
            off += k

        # ---- geometry: stages -> conv units ----------------------------------------------------------
        self.x_in = act(0, cpad(model.in_dim))
        convs, self.stages = [], []
        C = [d, 2 * d, 4 * d, 8 * d]                      # encoder output channels, levels 0..3
        self.cat = [act(l, 2 * cpad(C[l])) for l in range(4)]
        self.gcat = [act(l, 2 * cpad(C[l])) for l in range(4)]
        self.pool = [act(l + 1, cpad(C[l])) for l in range(4)]
        self.gpool = [act(l + 1, cpad(C[l])) for l in range(4)]

        def unit(prefix, ci, bi, level, cin_segs, cout, xin, xin_ldc, first_of_net=False):
            u = _Conv()
            u.name = f'{prefix}.{ci}'
            u.w, u.b = named[f'{prefix}.{ci}.weight'], named[f'{prefix}.{ci}.bias']
            u.gamma, u.beta = named[f'{prefix}.{bi}.weight'], named[f'{prefix}.{bi}.bias']
            u.rm, u.rv = bufs[f'{prefix}.{bi}.running_mean'], bufs[f'{prefix}.{bi}.running_var']
            u.nbt = bufs[f'{prefix}.{bi}.num_batches_tracked']
            u.keys = (f'{prefix}.{ci}.weight', f'{prefix}.{ci}.bias', f'{prefix}.{bi}.weight', f'{prefix}.{bi}.bias')
            u.level, u.h, u.w_ = level, H >> level, W >> level
            u.cin_segs = cin_segs                                  # [(logical, physical), ...] one or two segments
            u.cin = sum(s[0] for s in cin_segs)
            u.cin_p = sum(s[1] for s in cin_segs)
            u.cout, u.cout_p = cout, cpad(cout)
            u.xin, u.xin_ldc = xin, xin_ldc
            u.first = first_of_net
            # first conv (Cin = 3): 3x3 neighbourhood folded into 27 (->32) channels, conv runs as a pointwise GEMM
            u.im2col = first_of_net and 9 * u.cin <= cpad(9 * u.cin) == u.cin_p
            u.y = act(level, u.cout_p)
            u.gz = act(level, u.cout_p)
            # Winograd tiles are 2x2 outputs inside 8x16 / 16x16-pixel workgroup tiles: nothing to gain below 8x8 images
            u.wino = self.wino and not u.im2col and (u.h | u.w_) % 2 == 0 and min(u.h, u.w_) >= 8
            u.w24 = u.wino and bool(WINOGRAD24) and u.w_ % 4 == 0          # forward / data gradient by F(2x4,3x3)
            u.w24g = u.w24 and (min(u.h, u.w_) >= 64 if WINOGRAD24_WGRAD == 'auto' else bool(WINOGRAD24_WGRAD))   # ... weight gradient
            # ... data gradient: F(2x4) tiles are 256 pixels x 64 (input) channels; a launch with at most half a chip of them runs
            # the F(2x2) kernel instead, whose 128-pixel tiles give twice the work items (1024 -> 512 @16x16, the data gradient of
            # dec1.block.0: 128 items, 238 us against 160 us, tools/wino24_ab.py)
            items24 = B * ((u.h + 7) // 8 * ((u.w_ + 31) // 32) if u.w_ >= 32 else (u.h + 15) // 16 * ((u.w_ + 15) // 16)) * ((u.cin_p + 63) // 64)
            u.w24d = u.w24 and not (2 * items24 <= (torch.cuda.get_device_properties(dev).multi_processor_count if dev.type == 'cuda' else 256))
            # pre-transformed operands (wino24g.hip).  Weight gradient: both channel counts multiples of 256 (a wave owns a 128 x 128
            # block, a workgroup 256 x 256).  Forward: >= 256 input channels, and either the image is needed by the weight gradient
            # anyway or there are enough output channels to amortise the transform pass (its cost grows with Cin, the kernel's gain
            # with Cin x Cout: 512 -> 256 @64x64 loses 6 %, 256 -> 128 @128x128 26 %, tools/wino24g_ab.py).  Data gradient: the
            # same with the roles of the channel counts exchanged; the transformed gradient is used once and not kept.
            pt = PRETRANSFORM
            # F(4x4,3x3) applies where the launch fills the chip with (16x32-pixel tile block, 64-channel slab) work items (WINOGRAD44)
            ncu = torch.cuda.get_device_properties(dev).multi_processor_count if dev.type == 'cuda' else 256
            blocks44 = B * u.h * u.w_ // 512                       # FULL tile blocks (a 16x16 image fills half of a 32x16 block)
            ok44 = lambda slab_ch, in_ch: (bool(WINOGRAD44) and u.h % 4 == 0 and u.w_ % 4 == 0 and slab_ch % 64 == 0
                                           and (WINOGRAD44 is True or blocks44 * (slab_ch // 64) >= ncu)
                                           and lib.clamd_winograd44_input_elems(B, u.h, u.w_, in_ch) * 4 < (1 << 32))
            u.pre_w = bool(pt) and u.w24 and u.cin_p % 256 == 0 and u.cout_p % 256 == 0
            u.pre_f = bool(pt) and u.w24 and u.cin_p >= 64 and u.cout_p % 64 == 0 and (
                pt is True or (u.cin_p >= 256 and (u.pre_w or 2 * u.cout_p > u.cin_p)) or (u.cin_p >= 128 and u.cout_p >= 2 * u.cin_p))
            # one buffer descriptor spans a whole transformed tensor: below 2^32 bytes (config 2: <= 0.4 GB; 512 x 512 bs32 fp32: 3.2 GB)
            fits = lambda c: lib.clamd_winograd24_input_elems(B, u.h, u.w_, c) * 4 < (1 << 32)
            u.pre_f = u.pre_f and fits(u.cin_p)
            u.pre_w = u.pre_w and u.pre_f and lib.clamd_wgrad_winograd24_pre_operand_elems(B, u.h, u.w_, u.cout_p) * 4 // 24 < (1 << 32)
            u.pre_d = bool(pt) and u.w24d and not first_of_net and u.cout_p >= 64 and u.cin_p % 64 == 0 and fits(u.cout_p) and (
                pt is True or (u.cout_p >= 256 and (2 * u.cin_p > u.cout_p or (2 * u.cin_p == u.cout_p and u.cin_p >= 256))))
            # ... the 128-channel layers too (round 5: enc2.block.4, enc3.block.1, dec4 at config 2): their weight gradients ran the
            # in-kernel-transform kernel at 0.42-0.50 of the pipe; with channel counts that are multiples of 128 the plane GEMM runs them as
            # 128 x 128 wave tiles (wave-level stream-K) on the forward image, so forward AND weight gradient go pre-transformed F(4x4) and the
            # BatchNorm in front is applied by the transform (FOLD_BN_INTO_TRANSFORM) instead of by folded filters and a border-class table
            if (pt == 'auto' and WINOGRAD44 and u.w24 and not (u.pre_f and u.pre_w) and min(u.cin_p, u.cout_p) >= 128
                    and u.cin_p % 128 == 0 and u.cout_p % 128 == 0 and ok44(u.cout_p, u.cin_p) and NARROW_PRE_WGRAD):
                u.pre_f = u.pre_w = True
            u.f44 = u.pre_f and ok44(u.cout_p, u.cin_p)            # forward (and, with pre_w, the weight gradient: it reads the forward image)
            # ... and the data gradients of the NARROW layers whose launch has at least 128 output (= this unit's input) channels: transform of
            # the gradient + transform-free F(4x4) loop against the in-kernel-transform F(2x4) kernel, tools/wino44_narrow_ab.py: 64 -> 128
            # @256x256 1.07x, 128 -> 128 @128x128 1.08x, 128 -> 256 @128x128 1.26x, 256 -> 128 @64x64 1.21x (128 -> 64 and 64 -> 64: 0.83-0.85x)
            if (pt == 'auto' and not u.pre_d and u.w24d and not first_of_net and u.cin_p >= 128 and u.cout_p >= 64 and u.cin_p % 64 == 0
                    and WINOGRAD44 and ok44(u.cin_p, u.cout_p)):
                u.pre_d = True
            u.d44 = u.pre_d and ok44(u.cin_p, u.cout_p)            # data gradient
            if u.f44 and u.pre_w:
                u.pre_w = lib.clamd_wgrad_winograd44_pre_operand_elems(B, u.h, u.w_, u.cout_p) * 4 // 36 < (1 << 32)
            u.vx = torch.empty((lib.clamd_winograd44_input_elems if u.f44 else lib.clamd_winograd24_input_elems)(B, u.h, u.w_, u.cin_p),
                               dtype=torch.float32, device=dev) if u.pre_f else None
            # 64 input channels (8 chunks per tile): the in-kernel-transform kernel with the filters loaded straight into the operand
            # registers (wino24h_kernel) is 4-6 % faster there and 1-4 % slower on longer K loops (tools/wino24h_ab.py)
            u.direct_f = NARROW_DIRECT and u.w24 and not u.pre_f and u.cin_p == 64 and u.cout_p % 64 == 0
            u.direct_d = NARROW_DIRECT and u.w24d and not u.pre_d and not first_of_net and u.cout_p == 64 and u.cin_p % 64 == 0
            ntap = 1 if u.im2col else ((36 if u.f44 else (24 if u.w24 else 16)) if u.wino else 9)   # Winograd: [Cin_p/8][16|24|36][Cout_p][8] transformed filters
            ntap_d = (36 if u.d44 else (24 if u.w24d else 16)) if u.wino else ntap
            u.wf = torch.zeros(ntap * u.cout_p * u.cin_p, dtype=T, device=dev)
            u.wd = None if first_of_net else torch.zeros(ntap_d * u.cin_p * u.cout_p, dtype=T, device=dev)
            u.bias_p = torch.zeros(u.cout_p, dtype=torch.float32, device=dev)
            u.vec = torch.zeros(7, u.cout_p, dtype=torch.float32, device=dev)   # scale, shift, mean, istd, k0, k1, k2
            u.m_fastest = 1 if 9 * u.cout_p > B * u.h * u.w_ else 0
            convs.append(u)
            return u

        table = model._table
        prev = (self.x_in, self.x_in.shape[-1], [(model.in_dim, cpad(model.in_dim))])
        for k in range(4):                                         # encoders
            st = table[k]
            pre = st['name'] + ('.block' if st['wrapped'] else '')
            (c0, b0, cin, cout), (c1, b1, _, _) = st['convs']
            ua = act(k, cpad(cout))
            a = unit(pre, c0, b0, k, prev[2], cout, prev[0], prev[1], first_of_net=(k == 0))
            a.out, a.out_ldc, a.pooled, a.g_out, a.g_out_ldc, a.g_pool = ua, ua.shape[-1], None, None, None, None
            b = unit(pre, c1, b1, k, [(cout, cpad(cout))], cout, ua, ua.shape[-1])
            b.out, b.out_ldc, b.pooled = self.cat[k], self.cat[k].shape[-1], self.pool[k]
            a.g_in = None if k == 0 else self.gpool[k - 1]           # dgrad target of conv a (grad w.r.t. pooled input)
            b.g_in = act(k, cpad(cout))                             # grad w.r.t. ua
            a.g_src = (b.g_in, b.g_in.shape[-1], None)               # where conv a's BN-output gradient comes from
            b.g_src = (self.gcat[k], self.gcat[k].shape[-1], self.gpool[k])
            b.consumer, a.fused_reduce = (a, True) if self._fuse_sums(b) else (None, False)
            a.consumer, b.fused_reduce = None, False                # a's dgrad feeds a pooled / concat gradient: separate reduce
            self.stages.append(dict(kind='enc', convs=(a, b)))
            prev = (self.pool[k], self.pool[k].shape[-1], [(cout, cpad(cout))])
        spec = [(8 * d, 16 * d, 8 * d), (16 * d, 8 * d, 4 * d), (8 * d, 4 * d, 2 * d), (4 * d, 2 * d, d), (2 * d, d, None)]
        for j in range(5):                                         # decoders + last
            st = table[4 + j]
            pre = st['name'] + ('.block' if st['wrapped'] else '')
            level = 4 - j
            (c0, b0, cin, mid), (c1, b1, _, _) = st['convs']
            if j == 0:
                xin, segs, g_in_a = self.pool[3], [(8 * d, cpad(8 * d))], self.gpool[3]
            else:
                half = C[level]
                xin, segs, g_in_a = self.cat[level], [(half, cpad(half)), (half, cpad(half))], self.gcat[level]
            ua, ub = act(level, cpad(mid)), act(level, cpad(mid))
            a = unit(pre, c0, b0, level, segs, mid, xin, xin.shape[-1])
            a.out, a.out_ldc, a.pooled = ua, ua.shape[-1], None
            b = unit(pre, c1, b1, level, [(mid, cpad(mid))], mid, ua, ua.shape[-1])
            b.out, b.out_ldc, b.pooled = ub, ub.shape[-1], None
            a.g_in, b.g_in = g_in_a, act(level, cpad(mid))
            g_ub = act(level, cpad(mid))
            a.g_src = (b.g_in, b.g_in.shape[-1], None)
            b.g_src = (g_ub, g_ub.shape[-1], None)
            b.consumer, a.fused_reduce = (a, True) if self._fuse_sums(b) else (None, False)
            a.consumer, b.fused_reduce = None, FUSE_BN_SUMS is True  # b's gradient comes from the tail's data-gradient kernel
            kind, ti, tcin, tcout = st['tail']
            tail = _Conv()
            tail.kind = kind
            tail.w, tail.b = named[f'{pre}.{ti}.weight'], named[f'{pre}.{ti}.bias']
            tail.keys = (f'{pre}.{ti}.weight', f'{pre}.{ti}.bias')
            tail.cin, tail.cin_p, tail.cout = tcin, cpad(tcin), tcout
            tail.x, tail.g_x, tail.level = ub, g_ub, level
            tail.consumer = b if FUSE_BN_SUMS is True else None
            if kind == 'convT':
                tail.cout_p = cpad(tcout)
                tail.wf = torch.zeros(4 * tail.cout_p * tail.cin_p, dtype=T, device=dev)
                tail.wd = torch.zeros(tail.cin_p * 4 * tail.cout_p, dtype=T, device=dev)
                up = self.cat[level - 1]
                tail.y_slice = up[..., tail.cout_p:]               # second half of the concat buffer one level up
                tail.gy_slice = self.gcat[level - 1][..., tail.cout_p:]
                tail.y_ldc = up.shape[-1]
            else:
                tail.cout_p = self.Kp
                tail.wf = torch.zeros(tail.cout_p * tail.cin_p, dtype=T, device=dev)
                tail.wd = torch.zeros(tail.cin_p * tail.cout_p, dtype=T, device=dev)
                self.dl = act(0, self.Kp)
            tail.bias_p = torch.zeros(tail.cout_p, dtype=torch.float32, device=dev)
            self.stages.append(dict(kind='dec', convs=(a, b), tail=tail))
        self.convs = convs
        for u in convs:
            u.apply_folded, u.fold_src = False, None
        for st in self.stages:
            a, b = st['convs']
            # a's BatchNorm output `ua` is read by b's convolution (forward) and by b's weight gradient only: when both run on b's
            # transformed input, the affine is applied by the transform itself and a's bn_apply pass (and `ua`) disappears
            if FOLD_BN_INTO_TRANSFORM and b.pre_f and b.pre_w and a.pooled is None and b.xin is a.out:
                a.apply_folded, b.fold_src = True, a
        for st in self.stages:
            a, b = st['convs']
            # ... and where b transforms inside its kernel (or is a bf16 direct kernel): the algebraic fold of bnfold.hip
            b.fold_a, a.fold_a, a.fold_on, b.fold_on, a.apply_in_filters, b.apply_in_filters = None, None, False, False, False, False
            if (FOLD_BN_INTO_FILTERS
                    and not a.apply_folded and not b.pre_f and a.pooled is None and b.xin is a.out and len(b.cin_segs) == 1
                    and min(b.h, b.w_) >= 2 and b.cin_p <= FOLD_FILTERS_MAX_CHANNELS):
                b.fold_a = a      # one direction only: a cycle between units would keep the engine's buffers alive until the garbage collector runs
                b.cb = torch.zeros(9, b.cout_p, dtype=torch.float32, device=dev)
        for u in convs:
            u.pool_fold, u.y_ldc = False, u.cout_p
        for k in range(3):
            # ... and the output of an encoder block with two narrow F(2x4) readers (FOLD_POOLED): static (fp32 Winograd kernels take the
            # border-class table under every tuning), because the raw tensor then lives where the normalised one would
            b, nxt = self.stages[k]['convs'][1], self.stages[k + 1]['convs'][0]
            dec = next((st_['convs'][0] for st_ in self.stages if st_['kind'] == 'dec' and st_['convs'][0].xin is self.cat[k]), None)
            ok = (FOLD_POOLED and FOLD_BN_INTO_FILTERS and self.dcode == _lib.F32 and dec is not None and nxt.xin is self.pool[k]
                  and all(c.w24 and not c.pre_f and c.fold_a is None and c.cin_p <= FOLD_FILTERS_MAX_CHANNELS and min(c.h, c.w_) >= 2
                          and all(lg == ph for lg, ph in c.cin_segs) for c in (nxt, dec))
                  and b.w24 and not b.pre_f and b.cout == b.cout_p and len(dec.cin_segs) == 2
                  and dec.cin_segs[0] == (b.cout, b.cout_p))
            if not ok:
                continue
            b.pool_fold, b.apply_in_filters = True, True
            b.y, b.y_ldc = self.cat[k], self.cat[k].shape[-1]               # conv+ReLU output straight into the skip half of the concat buffer
            comp = torch.zeros(2, dec.cin_p, dtype=torch.float32, device=dev)  # scale / shift over the decoder conv's input: [block | up-conv]
            comp[0, b.cout_p:] = 1.0
            vec = b.vec
            b.vec = [comp[0, :b.cout_p], comp[1, :b.cout_p]] + [vec[i] for i in range(2, 7)]     # rows 4-6 (k0, k1, k2) stay contiguous
            nxt.fold_a = _FoldSource(self.pool[k], self.pool[k].shape[-1], b.vec[0], b.vec[1])
            dec.fold_a = _FoldSource(self.cat[k], self.cat[k].shape[-1], comp[0], comp[1])
            for c in (nxt, dec):
                c.cb = torch.zeros(9, c.cout_p, dtype=torch.float32, device=dev)
        for st in self.stages:
            t = st.get('tail')
            if t is None:
                continue
            # ... and the 1x1 head behind the last BatchNorm: pointwise, no border classes -- in every compute dtype
            t.fold_b = None
            b = st['convs'][1]
            if (FOLD_BN_INTO_FILTERS and t.kind == 'head' and b.pooled is None and t.x is b.out
                    and b.cout_p <= FOLD_FILTERS_MAX_CHANNELS):
                t.fold_b = b
                b.apply_in_filters = True
                t.bias_fold = torch.zeros(t.cout_p, dtype=torch.float32, device=dev)
        nfw = max([lib.clamd_bn_fold_wgrad_workspace_bytes(B, u.cout_p) // 4 for u in convs if u.fold_a is not None] + [0])
        self.fold_ws = torch.empty(nfw, dtype=torch.float32, device=dev) if nfw else None
        for s in self.stages:
            t = s.get('tail')
            if t is not None and t.consumer is not None:
                t.consumer.sum_src = t          # that unit's five BN-backward sums come from the tail's data-gradient launch
        self._tune_key = None
        self._plan_stat_rows()
        self.nbts = [u.nbt for u in convs]
        # split-K slabs of the weight-gradient kernels
        ws = 0
        for u in convs:
            ws = max(ws, lib.clamd_wgrad_workspace_bytes(_lib.WGRAD_CONV3, B, u.h, u.w_, u.cout_p, u.cin_p, self.dcode))
            ws = max(ws, lib.clamd_channel_sum_workspace_bytes(u.cout_p))
            if u.wino:
                ws = max(ws, lib.clamd_wgrad_winograd_workspace_bytes(u.cout_p, u.cin_p))
            if u.w24g:
                ws = max(ws, lib.clamd_wgrad_winograd24_workspace_bytes(u.cout_p, u.cin_p))
            if u.pre_w:
                ws = max(ws, (lib.clamd_wgrad_winograd44_pre_workspace_bytes if u.f44 else lib.clamd_wgrad_winograd24_pre_workspace_bytes)(
                    B, u.h, u.w_, u.cout_p, u.cin_p))
        for s in self.stages:
            t = s.get('tail')
            if t is not None:
                mode = _lib.WGRAD_UP2 if t.kind == 'convT' else _lib.WGRAD_PW
                rp, cp_ = (t.cin_p, t.cout_p) if t.kind == 'convT' else (t.cout_p, t.cin_p)
                ws = max(ws, lib.clamd_wgrad_workspace_bytes(mode, B, H >> t.level, W >> t.level, rp, cp_, self.dcode))
        self.ws = torch.empty(ws // 4 + 16, dtype=torch.float32, device=dev)
        # scratch of the pre-transformed kernels: the transformed gradient of the data-gradient launch (main stream) and the
        # gradient-side operand of the weight-gradient GEMM (second stream); launches on one stream are serialised, so one each
        nvg = max([(lib.clamd_winograd44_input_elems if u.d44 else lib.clamd_winograd24_input_elems)(B, u.h, u.w_, u.cout_p) for u in convs if u.pre_d] + [0])
        nyt = max([(lib.clamd_wgrad_winograd44_pre_operand_elems if u.f44 else lib.clamd_wgrad_winograd24_pre_operand_elems)(B, u.h, u.w_, u.cout_p)
                   for u in convs if u.pre_w] + [0])
        self.vg = torch.empty(nvg, dtype=torch.float32, device=dev) if nvg else None
        self.yt = torch.empty(nyt, dtype=torch.float32, device=dev) if nyt else None
        # third stream + a second operand buffer: the transform of unit u runs while the GEMM of unit u+1 still reads the other buffer
        self.x3_stream = (_third_stream(dev) if (WGRAD_XFORM_STREAM and WGRAD_STREAM and dev.type == 'cuda' and (nyt or nfw)) else None)
        self.yt2 = torch.empty(nyt, dtype=torch.float32, device=dev) if (nyt and self.x3_stream is not None) else None
        self._yt_flip = 0
        self._yt_ev = [None, None]
        self._x3_ev = None
        self.wg_stream = _second_stream(dev) if (WGRAD_STREAM and dev.type == 'cuda') else None
        self._wg_used = False
        self._pack_pending = 0
        self.ws_bytes = ws
        self._build_pack_table()
        self._ptrs = None

    # ------------------------------------------------------------------------------------------ statistics rows
    def _plan_stat_rows(self):
        """Partial-row buffers of the deterministic per-channel reductions (include/clamd.h, clamd_stat_rows): every unit
        gets stats [rows][2][Cout_p] (forward launch) and sums [rows][5][Cout_p] (whichever launch produces its five
        BatchNorm-backward sums).  Row counts depend on the kernel structure, i.e. on the tuning: re-planned when it changes."""
        key = tuple(self.tuning.as_dict().values())
        if key == self._tune_key:
            return
        self._tune_key = key
        B, dc, tn = self.B, self.dcode, self.tuning
        rows = _lib.stat_rows
        sizes = []
        for u in self.convs:      # which kernel runs a fold candidate depends on the tuning: F(2x4) Winograd / the persistent direct kernel take the table
            u.fold_on = u.fold_a is not None and (u.w24 if u.wino else
                                                  bool(_lib.load().clamd_conv3x3_border_bias_ok(B, u.h, u.w_, u.cin_p, u.cout_p, dc, tune_ptr(tn))))
            if u.fold_a is not None:
                u.fold_a.apply_in_filters = u.fold_on      # the producer's bn_apply pass is skipped
        for u in self.convs:
            u.gz_nrows = 0
            if u.im2col:
                r = rows(_lib.OP_CONV1X1, B, u.h, u.w_, u.cin_p, u.cout_p, dc)
            elif u.wino:
                r = rows(_lib.OP_CONV3X3_WINOGRAD44 if u.f44 else (_lib.OP_CONV3X3_WINOGRAD24 if u.w24 else _lib.OP_CONV3X3_WINOGRAD),
                         B, u.h, u.w_, u.cin_p, u.cout_p, dc, tuning=tn)
            else:
                r = rows(_lib.OP_CONV3X3, B, u.h, u.w_, u.cin_p, u.cout_p, dc, tuning=tn)
            u.stat_rows_launch = r
            u.stat_rows = r
            if u.fused_reduce:
                src = getattr(u, 'sum_src', None)
                if src is None:       # the 3x3 data-gradient launch of the next conv of this stage (K = its output channels)
                    b = next(c for c in self.convs if c.consumer is u)
                    u.sum_rows = rows(_lib.OP_CONV3X3, B, b.h, b.w_, b.cout_p, b.cin_p, dc, fused_bn=True, tuning=tn)
                    # the persistent bf16 kernel takes sum g and sum g y only: the conv-bias gradient then comes from the apply pass
                    if _lib.load().clamd_conv3x3_bn_sums(B, b.h, b.w_, b.cout_p, b.cin_p, dc, tune_ptr(tn)) == 2:
                        u.gz_nrows = _lib.load().clamd_bn_bwd_apply_sums_rows(B, u.h, u.w_, u.cout_p)
                elif src.kind == 'head':
                    u.sum_rows = rows(_lib.OP_CONV1X1, B, u.h, u.w_, src.cout_p, src.cin_p, dc, fused_bn=True)
                else:                 # ConvTranspose2d data gradient: the launch runs on the convT INPUT grid (= this unit's)
                    u.sum_rows = rows(_lib.OP_CONVT2X2_DGRAD, B, u.h, u.w_, src.cin_p, src.cout_p, dc, fused_bn=True)
            else:
                u.sum_rows = rows(_lib.OP_BN_BWD_REDUCE, B, u.h, u.w_, 1 if u.g_src[2] is not None else 0, u.cout_p, dc, tuning=tn)
            sizes.append((u.stat_rows * 2 + u.sum_rows * self.NS + u.gz_nrows) * u.cout_p)
        self.stat_arena = torch.empty(sum(sizes), dtype=torch.float32, device=self.dev)
        off = 0
        for u, n in zip(self.convs, sizes):
            k, k2 = u.stat_rows * 2 * u.cout_p, u.gz_nrows * u.cout_p
            u.stats = self.stat_arena[off:off + k]
            u.sums = self.stat_arena[off + k:off + n - k2]
            u.gz_rows = self.stat_arena[off + n - k2:off + n] if k2 else None
            off += n

    # ------------------------------------------------------------------------------------------ pack table
    def _build_pack_table(self):
        tab = PackTable(self.dcode)
        # ... and the plain pack in two launches too: `tab` = what the first three encoder stages need (and every bias vector), in front of
        # the forward pass; `late` = the 3x3 filters from enc4 on and the ConvTranspose / head filters (96 % of the parameters: 118 MB read,
        # 62 MB written in bf16, 130 us -- HBM-bound) on the second stream under enc1-enc3, waited for in front of enc4's first convolution
        late = PackTable(self.dcode)
        # Winograd filter transforms in two launches per form: "early" = the first three encoder stages (4 % of the parameters,
        # needed 0.3 ms into the forward pass), "late" = everything else (first needed by enc4, 2 ms in): the forward pass waits for
        # a few microseconds of packing instead of for all of it (see forward())
        wtab = {(pl, late): WinoPackTable(pl) for pl in (16, 24, 36) for late in (False, True)}
        for i, u in enumerate(self.convs):
            u.pack_late = i >= 6                                 # units 0-5 = enc1, enc2, enc3
            if u.im2col:
                tab.head(u.w, u.wf, None, 9 * u.cin, u.cout)     # [Cout][Cin*9] is already the (c*9 + tap) K order
            elif u.wino:
                if u.fold_a is None:
                    wtab[(36 if u.f44 else (24 if u.w24 else 16), u.pack_late)].conv3x3(u.w, u.wf, None, u.cin_segs, u.cout)
                if u.wd is not None:
                    wtab[(36 if u.d44 else (24 if u.w24d else 16), u.pack_late)].conv3x3(u.w, None, u.wd, u.cin_segs, u.cout)
            else:
                (late if u.pack_late else tab).conv3x3(u.w, None if u.fold_a is not None else u.wf, u.wd, u.cin_segs, u.cout)
            tab.vector(u.b, u.bias_p, u.cout)
            if u.fold_a is not None:
                # the forward filters of a fold candidate are packed inside the step, behind the producer's bn_finalize: with its scale
                # (fold on) or plain (a tuning that runs a kernel without the border-class epilogue)
                u.fold_table, u.plain_table = [
                    (WinoPackTable(24 if u.w24 else 16) if u.wino else PackTable(self.dcode)) for _ in range(2)]
                u.fold_table.conv3x3(u.w, u.wf, None, u.cin_segs, u.cout, kscale=u.fold_a.vec[0])
                u.plain_table.conv3x3(u.w, u.wf, None, u.cin_segs, u.cout)
                u.fold_table.finalize(self.dev); u.plain_table.finalize(self.dev)
        for s in self.stages:
            t = s.get('tail')
            if t is None:
                continue
            if t.kind == 'convT':
                late.convT(t.w, t.wf, t.wd, t.cin, t.cout)
            elif t.fold_b is not None:      # forward filters inside the step, with the last BatchNorm's scale (see _fwd_fold)
                late.head(t.w, None, t.wd, t.cin, t.cout)
                t.fold_table = PackTable(self.dcode)
                t.fold_table.head(t.w, t.wf, None, t.cin, t.cout, kscale=t.fold_b.vec[0])
                t.fold_table.finalize(self.dev)
            else:
                late.head(t.w, t.wf, t.wd, t.cin, t.cout)
            tab.vector(t.b, t.bias_p, t.cout)
        self.pack_table = tab.finalize(self.dev)
        self.pack_late = late.finalize(self.dev) if late.jobs else None
        self._ev_pack_late = None
        self.wino_early = [t.finalize(self.dev) for (pl, late), t in wtab.items() if t.jobs and not late]
        self.wino_late = [t.finalize(self.dev) for (pl, late), t in wtab.items() if t.jobs and late]
        self._ev_early = torch.cuda.Event() if self.dev.type == 'cuda' else None
        self._param_ptrs = [p.data_ptr() for p in self.model.parameters()]

    def _check_ptrs(self, params):
        cur = [p.data_ptr() for p in params]
        if cur != self._param_ptrs:
            # parameters were re-allocated (.to(), load from a different storage): rebuild the job table
            named = dict(zip(self.param_names, params))
            for u in self.convs:
                u.w, u.b, u.gamma, u.beta = (named[k] for k in u.keys)
            for s in self.stages:
                t = s.get('tail')
                if t is not None:
                    t.w, t.b = named[t.keys[0]], named[t.keys[1]]
            self._build_pack_table()

    # ------------------------------------------------------------------------------------------ forward
    def forward(self, x, params, predict=False):
        m = self.model
        training = m.training
        self.fwd_training = training
        self.generation += 1
        self.dl_src = None
        self._check_ptrs(params)
        s = _lib.stream_ptr()
        B, H, W, dc = self.B, self.H, self.W, self.dcode
        self._plan_stat_rows()
        self.pack_table.run(dc, s)
        # the Winograd filter transforms (454 MB of HBM traffic per step at config 2) are first needed by the SECOND convolution:
        # they run on the second stream under the first layer's im2col / pointwise conv / BatchNorm passes
        self._pack_pending = 0               # 2: neither part waited for yet, 1: the early part has been waited for
        sp = s
        if self.wg_stream is not None and KERNEL_TIMING is None and (self.wino_early or self.wino_late):
            self.wg_stream.wait_stream(torch.cuda.current_stream())
            sp, self._pack_pending = self.wg_stream.cuda_stream, 2
        for t in self.wino_early:
            t.run(sp)
        if self._pack_pending:
            self._ev_early.record(self.wg_stream)        # the late part is enqueued in front of the first Winograd convolution
        else:
            for t in self.wino_late:
                t.run(sp)
        self._ev_pack_late = None
        self._pack_late_pending = self.pack_late is not None      # released beside convolution PACK_LATE_AT (see _release_pack_late)
        if self.convs[0].im2col:
            _hbm('enc1.0', B * H * W * (4 * m.in_dim + self.esize * self.x_in.shape[-1]),
                 'clamd_nchw_im2col3', ptr(x), ptr(self.x_in), self.x_in.shape[-1], B, m.in_dim, H, W, self.x_in.shape[-1], dc, s)
        else:
            call('clamd_nchw_to_nhwc', ptr(x), ptr(self.x_in), self.x_in.shape[-1], B, m.in_dim, H, W,
                 self.x_in.shape[-1], 1.0, dc, s)
        for st in self.stages:
            for u in st['convs']:
                self._fwd_pre(u, s)
                self._fwd_fold(u, s)
                self._fwd_conv(u, training, s)
                self._fwd_finalize(u, training, s)
                self._fwd_post(u, s)
            t = st.get('tail')
            if t is None:
                continue               # encoder: the pooled output feeds the next stage's first convolution
            self._join_pack_late()
            h, w = H >> t.level, W >> t.level
            tx, tx_ldc, tbias = t.x, t.x.shape[-1], t.bias_p
            if t.fold_b is not None:       # the head reads the last unit's conv+ReLU output: its BatchNorm is in the filters and the bias
                ft, fb = t.fold_table, t.fold_b
                call('clamd_bn_fold_pack', 0, ptr(ft.dev_table), len(ft.jobs), ft.nblocks, dc, ptr(t.w), 1, ptr(fb.vec[1]), ptr(t.b),
                     ptr(t.bias_fold), t.cout, t.cin, t.cout_p, s)
                tx, tx_ldc, tbias = fb.y, fb.cout_p, t.bias_fold
            if t.kind == 'convT':
                _hbm('convT', self.esize * (B * h * w * (t.cin + 4 * t.cout) + 4 * t.cin * t.cout),
                     'clamd_convT2x2_fwd', ptr(t.x), t.x.shape[-1], ptr(t.wf), ptr(t.bias_p), ptr(t.y_slice), t.y_ldc, B, h, w, t.cin_p, t.cout_p, dc, s)
            elif predict and t.cout_p <= 64:      # arg-max fused into the head's epilogue: the logits never reach HBM
                logits = torch.empty(B, H, W, dtype=torch.int64, device=self.dev)
                call('clamd_conv1x1_argmax', ptr(tx), tx_ldc, ptr(t.wf), ptr(tbias), ptr(logits), None, B, h, w,
                     t.cin_p, t.cout_p, self.K, dc, s)
            elif predict:      # more than 64 (padded) classes: the fused epilogue holds one 64-class slab; logits, then arg-max
                lg = torch.empty(B, self.K, H, W, dtype=torch.float32, device=self.dev)
                call('clamd_conv1x1_logits', ptr(tx), tx_ldc, ptr(t.wf), ptr(tbias), ptr(lg), B, h, w,
                     t.cin_p, t.cout_p, self.K, dc, s)
                logits = torch.empty(B, H, W, dtype=torch.int64, device=self.dev)
                call('clamd_argmax_confusion', ptr(lg), None, ptr(logits), None, B, self.K, 1, H, W, s)
            else:
                logits = torch.empty(B, self.K, H, W, dtype=torch.float32, device=self.dev)
                _hbm('head', B * h * w * (self.esize * t.cin + 4 * self.K),
                     'clamd_conv1x1_logits', ptr(tx), tx_ldc, ptr(t.wf), ptr(tbias), ptr(logits), B, h, w, t.cin_p, t.cout_p, self.K, dc, s)
        if self._pack_pending == 2:     # no Winograd layer ran at all: the late part was never enqueued
            for t in self.wino_late:
                t.run(self.wg_stream.cuda_stream)
        if self._pack_pending:          # some part was never waited for (no late Winograd layer in this net): join before returning
            torch.cuda.current_stream().wait_stream(self.wg_stream)
            self._pack_pending = 0
        self._join_pack_late()
        return logits

    @staticmethod
    def executed_fraction(u, direction):
        """Multiply-adds the kernel executes per algorithmic (direct-convolution) multiply-add of unit u."""
        if not u.wino:
            return 1.0
        if {'wgrad': u.pre_w and u.f44, 'dgrad': u.d44}.get(direction, u.f44):
            return 36.0 / 144.0                                   # F(4x4,3x3): 36 per 16 outputs x 9 taps
        return 24.0 / 72.0 if {'wgrad': u.w24g or u.pre_w, 'dgrad': u.w24d}.get(direction, u.w24) else 16.0 / 36.0

    def executed_flop_deficit(self):
        """Algorithmic minus executed FLOPs of one train step (3x3 convolutions by Winograd), for bench.py."""
        d = 0.0
        for u in self.convs:
            f = 2.0 * self.B * u.h * u.w_ * 9 * u.cin * u.cout
            for direction in ('fwd', 'wgrad') + (('dgrad',) if u.g_in is not None else ()):
                d += (1.0 - self.executed_fraction(u, direction)) * f
        return d

    def _conv_bytes(self, u):
        """Algorithmic HBM bytes of one 3x3 launch on unit u (forward, data gradient or weight gradient alike): input and
        output activation once each, filters (or their gradient) once."""
        return self.esize * (self.B * u.h * u.w_ * (u.cin + u.cout) + 9 * u.cin * u.cout)

    def _fwd_pre(self, u, s):
        """Input transform of a pre-transformed convolution (wino24g.hip)."""
        if not u.pre_f:
            return
        # the BatchNorm of the unit in front folded into the transform where nothing else reads its output (u.fold_src)
        f = u.fold_src
        xsrc, xldc, fs, fh = (f.y, f.cout_p, f.vec[0], f.vec[1]) if f is not None else (u.xin, u.xin_ldc, None, None)
        _TIMED_UNIT[:] = [u.name + ' fwd', self.executed_fraction(u, 'fwd')]
        _timed('wino_transform', 0.0, (13 if u.f44 else 16) * self.B * u.h * u.w_ * u.cin_p,   # reads the activation once, writes 3x (F(4x4): 2.25x) its size
               'clamd_winograd44_transform_input' if u.f44 else 'clamd_winograd24_transform_input', ptr(xsrc), xldc, ptr(fs), ptr(fh), ptr(u.vx),
               self.B, u.h, u.w_, u.cin_p, s)

    def _fwd_fold(self, u, s):
        """Forward filters of a fold candidate (bnfold.hip): packed here, behind the producer's bn_finalize -- with its scale and the
        border-class bias table when the fold is on, plain otherwise."""
        a = u.fold_a
        if a is None:
            return
        if not u.fold_on:
            t = u.plain_table
            t.run(s) if u.wino else t.run(self.dcode, s)
            return
        t = u.fold_table        # one launch: the filters times the producer's scale, and the bias table from its shift
        call('clamd_bn_fold_pack', (24 if u.w24 else 16) if u.wino else 0, ptr(t.dev_table), len(t.jobs), t.nblocks, self.dcode,
             ptr(u.w), 9, ptr(a.vec[1]), ptr(u.b), ptr(u.cb), u.cout, u.cin, u.cout_p, s)

    def _release_pack_late(self):
        """Enqueues the late part of the plain filter pack: on the second stream, free to start with the convolution about to be launched on
        the current one.  Released beside enc2's first convolution, not at the start of the forward pass: the pack is HBM-bound (180 MB in
        bf16) and so are the im2col pass and the 3 -> 64-channel first layer -- beside those it only made them longer (kernel trace, round 4:
        the first layer 64 -> 160 us with the pack running), the 128 x 128 levels leave HBM bandwidth."""
        if not self._pack_late_pending:
            return
        self._pack_late_pending = False
        if self.wg_stream is not None and KERNEL_TIMING is None and self.pack_late_stream:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.wg_stream.wait_event(ev)
            self.pack_late.run(self.dcode, self.wg_stream.cuda_stream)
            self._ev_pack_late = torch.cuda.Event()
            self._ev_pack_late.record(self.wg_stream)
        else:
            self.pack_late.run(self.dcode, _lib.stream_ptr())

    def _join_pack_late(self):
        """The late part of the plain filter pack (second stream) is needed from here on."""
        self._release_pack_late()
        if self._ev_pack_late is not None:
            torch.cuda.current_stream().wait_event(self._ev_pack_late)
            self._ev_pack_late = None

    def _fwd_conv(self, u, training, s):
        """conv3x3 + bias + ReLU (+ BatchNorm statistics rows) of unit u."""
        dc, tp = self.dcode, tune_ptr(self.tuning)
        if u.pack_late:
            self._join_pack_late()
        elif u is self.convs[min(self.pack_late_at, len(self.convs) - 1)]:
            self._release_pack_late()
        Bl = self.B
        _TIMED_UNIT[:] = [u.name + ' fwd', self.executed_fraction(u, 'fwd')]
        rows = u.stat_rows_launch
        st = u.stats if training else None
        xin, y = u.xin, u.y
        xin_ldc, bias, relu = u.xin_ldc, u.bias_p, 1
        if u.fold_on:      # reads the producer's conv+ReLU output; its BatchNorm lives in the filters and in the bias table
            xin, xin_ldc, bias, relu = u.fold_a.y, u.fold_a.cout_p, u.cb, 3
        flops = 2.0 * Bl * u.h * u.w_ * 9 * u.cin * u.cout
        nbytes = self.esize * (Bl * u.h * u.w_ * (u.cin + u.cout) + 9 * u.cin * u.cout)
        if u.im2col:
            _hbm('enc1.0', self.esize * Bl * u.h * u.w_ * (u.cin_p + u.cout),
                 'clamd_conv1x1', ptr(xin), u.xin_ldc, ptr(u.wf), ptr(u.bias_p), ptr(y), u.cout_p,
                 ptr(st), None, None, rows, Bl, u.h, u.w_, u.cin_p, u.cout_p, 1, dc, s)
        elif u.wino:
            if self._pack_pending == 2 and (u.level >= 1 or u.pack_late):
                # the late transforms (HBM-bound, 0.3 ms) start here, under this MFMA-bound convolution, instead of beside the
                # HBM-bound first-layer kernels -- and, round 5, behind enc1's pooling pass (level >= 1): started under enc1's second
                # convolution they were still running when that pass came up and it took 141 instead of 64 us (r05h trace)
                torch.cuda.current_stream().wait_event(self._ev_early)
                self.wg_stream.wait_stream(torch.cuda.current_stream())
                for t in self.wino_late:
                    t.run(self.wg_stream.cuda_stream)
                self._pack_pending = 1
            if self._pack_pending == 1 and u.pack_late:
                torch.cuda.current_stream().wait_stream(self.wg_stream)
                self._pack_pending = 0
            if u.pre_f:
                _timed('igemm_conv3x3', flops, nbytes, 'clamd_conv3x3_winograd44_pre' if u.f44 else 'clamd_conv3x3_winograd24_pre',
                       ptr(u.vx), ptr(u.wf), ptr(u.bias_p), ptr(y), u.cout_p,
                       ptr(st), rows, Bl, u.h, u.w_, u.cin_p, u.cout_p, 1, tp, s)
            else:
                name = ('clamd_conv3x3_winograd24_direct_filters' if u.direct_f else 'clamd_conv3x3_winograd24') if u.w24 else 'clamd_conv3x3_winograd'
                _timed('igemm_conv3x3', flops, nbytes, name, ptr(xin), xin_ldc, ptr(u.wf), ptr(bias), ptr(y), u.y_ldc, ptr(st), rows,
                       Bl, u.h, u.w_, u.cin_p, u.cout_p, relu, tp, s)
        else:
            _timed('igemm_conv3x3', flops, nbytes, 'clamd_conv3x3', ptr(xin), xin_ldc, ptr(u.wf), ptr(bias), ptr(y), u.cout_p,
                   ptr(st), None, None, rows, Bl, u.h, u.w_, u.cin_p, u.cout_p, relu, u.m_fastest, dc, tp, s)

    def _fwd_finalize(self, u, training, s):
        v = u.vec
        call('clamd_bn_finalize', ptr(u.stats) if training else None, u.stat_rows, ptr(u.gamma), ptr(u.beta), ptr(u.rm), ptr(u.rv),
             ptr(v[0]), ptr(v[1]), ptr(v[2]), ptr(v[3]), u.cout_p, u.cout, float(self.B * u.h * u.w_), BN_MOMENTUM, BN_EPS,
             ptr(u.nbt) if training else None, s)      # num_batches_tracked += 1 inside the launch (was a torch._foreach_add_ on the critical chain)

    def _fwd_post(self, u, s):
        """BatchNorm apply (+ max-pool, concat placement) of unit u."""
        if u.apply_folded:          # the only reader of the BatchNorm output is the next convolution's input transform
            return
        if u.pool_fold:             # ... or the filters and bias tables of both readers of an encoder block's output: only the pooling is left
            _hbm('bn_fwd', self.esize * self.B * u.h * u.w_ * u.cout * 5 // 4,
                 'clamd_maxpool2x2', ptr(u.y), u.y_ldc, ptr(u.vec[0]), ptr(u.pooled), u.pooled.shape[-1], self.B, u.h, u.w_, u.cout_p, self.dcode, s)
            return
        if u.apply_in_filters:      # ... or its filters and bias table (bnfold.hip)
            return
        v = u.vec
        _hbm('bn_fwd', self.esize * self.B * u.h * u.w_ * u.cout * (9 if u.pooled is not None else 8) // 4,
             'clamd_bn_apply', ptr(u.y), u.cout_p, ptr(v[0]), ptr(v[1]), ptr(u.out), u.out_ldc,
             ptr(u.pooled), u.pooled.shape[-1] if u.pooled is not None else 0, self.B, u.h, u.w_, u.cout_p, self.dcode, s)

    # ------------------------------------------------------------------------------------------ backward
    def _wg_stream_ptr(self):
        """Stream for a parameter-gradient launch whose inputs have just been enqueued on the current stream."""
        if self.wg_stream is None or KERNEL_TIMING is not None:
            return _lib.stream_ptr()
        self.wg_stream.wait_stream(torch.cuda.current_stream())
        self._wg_used = True
        return self.wg_stream.cuda_stream

    def backward(self, gout):
        m = self.model
        s = _lib.stream_ptr()
        self._wg_used = False
        self._x3_fold = False
        self._pack_pending = 0
        self._yt_ev = [None, None]      # events of THIS backward pass only (the previous one was joined before it returned; a captured
        self._x3_ev = None              # graph must not wait on an event recorded outside the capture)
        B, H, W, dc = self.B, self.H, self.W, self.dcode
        if not self.fwd_training:
            raise RuntimeError('UNet.backward after an eval-mode forward is not supported (BatchNorm backward uses batch statistics)')
        if tuple(self.tuning.as_dict().values()) != self._tune_key:
            # the partial-row buffers were planned for the forward's kernel structure
            raise RuntimeError('model.tuning changed between forward and backward: change it between steps (before the forward)')
        p0 = next(iter(m.parameters()))
        if p0.grad is not None:
            lo = self.gflat.data_ptr()
            if lo <= p0.grad.data_ptr() < lo + 4 * self.gflat.numel():
                # the previous gradients are still installed as .grad (no zero_grad since): accumulate semantics
                # need them intact, so this backward writes into a fresh buffer
                if m.grad_sync is not None:
                    # AccumulateGrad would add the new buffer into .grad on this stream while RCCL is still reducing it
                    # on the side stream: refuse instead of racing (the reference zeroes gradients every step, trainer.py:173)
                    raise RuntimeError('gradient accumulation (backward without zero_grad) is not supported together with '
                                       'ddp.GradSync: call optimizer.zero_grad() before every backward')
                self.gflat = torch.empty_like(self.gflat)
        base = self.gflat.data_ptr()
        g = {n: base + 4 * o for n, (o, _) in self.goffset.items()}      # raw device pointers into the flat buffer
        sync = m.grad_sync
        if sync is not None:
            sync.begin()
        self._gp = g
        tp = tune_ptr(self.tuning)
        for st in reversed(self.stages):
            t = st.get('tail')
            if t is not None:
                h, w = H >> t.level, W >> t.level
                if t.kind == 'head':
                    src = self.dl_src
                    if not (src is not None and src[1:] == (gout.data_ptr(), gout._version, self.generation)):
                        # not the tensor this package's loss wrote beside its NHWC copy (another loss, a hook, a sum of gradients): convert
                        call('clamd_nchw_to_nhwc', ptr(gout), ptr(self.dl), self.Kp, B, self.K, H, W, self.Kp, 1.0, dc, s)
                    self.dl_src = None
                    _hbm('head', self.esize * B * h * w * (self.Kp + t.cin),
                         'clamd_conv1x1', ptr(self.dl), self.Kp, ptr(t.wd), None, ptr(t.g_x), t.g_x.shape[-1], None,
                         ptr(t.consumer.y) if t.consumer else None, ptr(t.consumer.sums) if t.consumer else None,
                         t.consumer.sum_rows if t.consumer else 0, B, h, w, t.cout_p, t.cin_p, 0, dc, s)
                    sw = self._wg_stream_ptr()      # parameter gradients on the second stream, behind the data gradient (see _conv_bwd)
                    fb = t.fold_b
                    tx, tx_ldc = (fb.y, fb.cout_p) if fb is not None else (t.x, t.x.shape[-1])
                    _hbm('head', self.esize * B * h * w * (self.Kp + t.cin),
                         'clamd_wgrad', _lib.WGRAD_PW, ptr(self.dl), self.Kp, ptr(tx), tx_ldc, ptr(self.ws),
                         self.ws_bytes, g[t.keys[0]], B, h, w, t.cout_p, t.cin_p, t.cout, t.cin,
                         t.cout, t.cout_p, t.cin, t.cin_p, dc, tp, sw)
                    _hbm('head', 0,                # algorithmically free: d logits was just streamed by the weight gradient above
                         'clamd_channel_sum', ptr(self.dl), self.Kp, g[t.keys[1]], B * h * w, self.Kp, t.cout, dc,
                         ptr(self.ws), self.ws_bytes, tp, sw)
                    if fb is not None:      # the weight gradient ran on the un-normalised tensor: dW = scale * dW + shift * (bias gradient)
                        call('clamd_bn_fold_wgrad_pointwise', g[t.keys[1]], ptr(fb.vec[0]), ptr(fb.vec[1]), g[t.keys[0]], t.cout, t.cin, sw)
                else:
                    ctb = self.esize * (B * h * w * (t.cin + 4 * t.cout) + 4 * t.cin * t.cout)
                    _hbm('convT', ctb,
                         'clamd_convT2x2_dgrad', ptr(t.gy_slice), t.y_ldc, ptr(t.wd), ptr(t.g_x), t.g_x.shape[-1],
                         ptr(t.consumer.y) if t.consumer else None, ptr(t.consumer.sums) if t.consumer else None,
                         t.consumer.sum_rows if t.consumer else 0, B, h, w, t.cin_p, t.cout_p, dc, s)
                    sw = self._wg_stream_ptr()
                    _hbm('convT', ctb,
                         'clamd_wgrad', _lib.WGRAD_UP2, ptr(t.x), t.x.shape[-1], ptr(t.gy_slice), t.y_ldc, ptr(self.ws),
                         self.ws_bytes, g[t.keys[0]], B, h, w, t.cin_p, t.cout_p, t.cin, t.cout,
                         t.cin, t.cin_p, t.cout, t.cout_p, dc, tp, sw)
                    _hbm('convT', 0,               # algorithmically free: the gradient was just streamed by the weight gradient above
                         'clamd_channel_sum', ptr(t.gy_slice), t.y_ldc, g[t.keys[1]], B * 4 * h * w, t.cout_p,
                         t.cout, dc, ptr(self.ws), self.ws_bytes, tp, sw)
            for u in reversed(st['convs']):
                self._conv_bwd(u, s)
            if sync is not None:
                if self._x3_fold:       # the stage's fixed-up weight gradients belong to the bucket: the second stream (which stage_done waits for) joins the third
                    self.wg_stream.wait_stream(self.x3_stream)
                    self._x3_fold = False
                sync.stage_done(self, st)
        if self._x3_fold:
            self.wg_stream.wait_stream(self.x3_stream)
        if self._wg_used:
            torch.cuda.current_stream().wait_stream(self.wg_stream)       # every gradient is complete for whoever comes next
        gf = self.gflat
        return [gf[o:o + k].view(self.gshape[n]) for n, (o, k) in ((n, self.goffset[n]) for n in self.param_names)]

    def _x3_allowed(self):
        """The third stream only where it cannot end up on a hardware queue with RCCL's kernels: HIP multiplexes streams onto a fixed number of
        hardware queues in creation order and a kernel waits behind whatever shares its queue.  A data-parallel rank has the default stream,
        the second and third streams, GradSync's stream and RCCL's: five -- so under ddp.GradSync the third stream needs at least eight queues
        (two rounds of the assignment apart).  The count is MEASURED (ddp.hw_queues: spin kernels on eight streams), not read from
        GPU_MAX_HW_QUEUES -- the runtime reads that variable once, when it starts."""
        if self.model.grad_sync is None:
            return True
        from . import ddp
        return ddp.hw_queues(self.dev) >= 8

    def _fuse_sums(self, b):
        """Does the data-gradient launch of conv `b` (3x3, K = b.cout_p gradient channels) also reduce the BN-backward sums
        of the unit in front of it?"""
        if b.wino:                     # the Winograd data-gradient kernels have no such epilogue (round 4: built with two sums in the statistics
            return False               # registers, measured 21.06 -> 21.11 ms per step, removed: the reduce passes it replaces run beside a weight gradient)
        if FUSE_BN_SUMS == 'auto':     # persistent bf16 kernel: <= 256 input channels, K-steps in pairs (64 channels)
            return self.dcode == _lib.BF16 and b.cout_p <= 256 and b.cout_p % 64 == 0
        return bool(FUSE_BN_SUMS)

    def _conv_bwd(self, u, s):
        B, dc, tp = self.B, self.dcode, tune_ptr(self.tuning)
        v = u.vec
        ga, ga_ldc, gp = u.g_src
        count = float(B * u.h * u.w_)
        g = self._gp
        if not u.fused_reduce:     # otherwise the five sums were accumulated by the epilogue of the kernel that wrote `ga`
            _hbm('bn_bwd', 0,                      # algorithmically free: one backward pass reads g and y once (the apply pass below is charged for it)
                 'clamd_bn_bwd_reduce', ptr(ga), ga_ldc, ptr(gp), gp.shape[-1] if gp is not None else 0, ptr(u.y), u.y_ldc,
                 ptr(v[0]), ptr(v[1]), ptr(u.sums), u.sum_rows, B, u.h, u.w_, u.cout_p, dc, tp, s)
        two = u.fused_reduce and u.gz_nrows > 0      # the producing launch took sum g and sum g y only: d conv-bias = sum g_z, from the apply pass
        call('clamd_bn_bwd_finalize', ptr(u.sums), u.sum_rows, ptr(u.gamma), ptr(v[2]), ptr(v[3]), ptr(v[4]), g[u.keys[2]],
             g[u.keys[3]], None if two else g[u.keys[1]], u.cout_p, u.cout, count, s)
        if two:
            assert gp is None
            _hbm('bn_bwd', self.esize * B * u.h * u.w_ * u.cout * 3,
                 'clamd_bn_bwd_apply_sums', ptr(ga), ga_ldc, ptr(u.y), u.y_ldc, ptr(v[4]), ptr(u.gz), u.cout_p, ptr(u.gz_rows), u.gz_nrows,
                 B, u.h, u.w_, u.cout_p, dc, s)
        else:
            _hbm('bn_bwd', self.esize * B * u.h * u.w_ * u.cout * (13 if gp is not None else 12) // 4,
                 'clamd_bn_bwd_apply', ptr(ga), ga_ldc, ptr(gp), gp.shape[-1] if gp is not None else 0, ptr(u.y), u.y_ldc,
                 ptr(v[0]), ptr(v[1]), ptr(v[4]), ptr(u.gz), u.cout_p, B, u.h, u.w_, u.cout_p, dc, s)
        if len(u.cin_segs) == 2:
            c_seg0, c_seg0p = u.cin_segs[0]
        else:
            c_seg0, c_seg0p = u.cin, u.cin_p
        flops = 2.0 * B * u.h * u.w_ * 9 * u.cin * u.cout
        self._x3_ev = None
        if (u.pre_w and self.x3_stream is not None and self.wg_stream is not None and KERNEL_TIMING is None and self._x3_allowed()
                and not torch.cuda.is_current_stream_capturing()):      # hipStreamEndCapture crashes on this three-stream pattern (ROCm 7.2):
            #                                                            a captured step keeps the transform on the second stream
            # gz is complete on the current stream: its weight-gradient transform goes to the third stream NOW (it runs beside whatever
            # weight-gradient GEMM the second stream is in), into the operand buffer the previous GEMM is not reading
            self._yt_flip ^= 1
            self._x3_buf = self.yt2 if self._yt_flip else self.yt
            x3 = self.x3_stream
            x3.wait_stream(torch.cuda.current_stream())
            if self._yt_ev[self._yt_flip] is not None:      # the GEMM that read this buffer last (two pre-transformed units back)
                x3.wait_event(self._yt_ev[self._yt_flip])
            call('clamd_wgrad_winograd44_pre_transform' if u.f44 else 'clamd_wgrad_winograd24_pre_transform', ptr(u.gz), u.cout_p, ptr(self._x3_buf),
                 B, u.h, u.w_, u.cout_p, x3.cuda_stream)
            self._x3_ev = torch.cuda.Event(); self._x3_ev.record(x3)
        def dgrad():
            _TIMED_UNIT[:] = [u.name + ' dgrad', self.executed_fraction(u, 'dgrad')]
            if u.g_in is not None and u.pre_d:
                _timed('wino_transform', 0.0, (13 if u.d44 else 16) * B * u.h * u.w_ * u.cout_p,      # reads the gradient once, writes 3x (F(4x4): 2.25x) its size
                       'clamd_winograd44_transform_input' if u.d44 else 'clamd_winograd24_transform_input', ptr(u.gz), u.cout_p, None, None, ptr(self.vg),
                       B, u.h, u.w_, u.cout_p, s)
                _timed('igemm_conv3x3', flops, self._conv_bytes(u),
                       'clamd_conv3x3_winograd44_pre' if u.d44 else 'clamd_conv3x3_winograd24_pre', ptr(self.vg), ptr(u.wd), None, ptr(u.g_in), u.g_in.shape[-1], None, 0,
                       B, u.h, u.w_, u.cout_p, u.cin_p, 0, tp, s)
            elif u.g_in is not None and u.wino:
                name = 'clamd_conv3x3_winograd24_direct_filters' if u.direct_d else ('clamd_conv3x3_winograd24' if u.w24d else 'clamd_conv3x3_winograd')
                _timed('igemm_conv3x3', flops, self._conv_bytes(u), name, ptr(u.gz), u.cout_p, ptr(u.wd), None, ptr(u.g_in), u.g_in.shape[-1], None, 0,
                       B, u.h, u.w_, u.cout_p, u.cin_p, 0, tp, s)
            elif u.g_in is not None:
                _timed('igemm_conv3x3', flops, self._conv_bytes(u),
                       'clamd_conv3x3', ptr(u.gz), u.cout_p, ptr(u.wd), None, ptr(u.g_in), u.g_in.shape[-1], None,
                       ptr(u.consumer.y) if u.consumer is not None else None,
                       ptr(u.consumer.sums) if u.consumer is not None else None,
                       u.consumer.sum_rows if u.consumer is not None else 0,
                       B, u.h, u.w_, u.cout_p, u.cin_p, 0, 1 if 9 * u.cin_p > B * u.h * u.w_ else 0, dc, tp, s)

        # The weight gradient may start once the data gradient of the same unit has FINISHED (the second stream's wait is
        # recorded behind it).  Started together, the dispatcher interleaves the workgroups of the two kernels, they end together
        # and the next unit's BatchNorm passes run alone again; started behind it, the weight gradient is what runs beside those
        # passes (tools/trace_gaps.py: 1.80 instead of 2.16 ms per fp32 step without an MFMA kernel).  A/B in one process:
        # bf16 +0.7 %, bf16x3 +1.0 %, fp32 unchanged (the kernels that share the chip with the passes run that much longer).
        # The issue ORDER of the two launches alone makes no difference.
        # ... except for the LAST weight gradients of the backward pass (the level-0 encoder block: nothing of the critical chain is left to run
        # beside them, the step ends with the main stream waiting for the second one -- 249 us in the r05h trace): those start as soon as
        # their gradient is ready, beside their own unit's data gradient
        early = WGRAD_TAIL_EARLY and self.dcode != _lib.BF16 and u.level == 0 and u.name.startswith('enc1')      # (bf16: 6.418 against 6.397 ms: off)
        sw = self._wg_stream_ptr() if early else None
        dgrad()
        if sw is None:
            sw = self._wg_stream_ptr()
        if two:      # off the critical chain: the fixed-order sum of the apply pass's rows, in front of this unit's weight gradient
            call('clamd_rows_sum', ptr(u.gz_rows), u.gz_nrows, g[u.keys[1]], u.cout_p, u.cout, sw)
        if u.im2col:
            _hbm('enc1.0', self.esize * B * u.h * u.w_ * (u.cout + u.cin_p),
                 'clamd_wgrad', _lib.WGRAD_PW, ptr(u.gz), u.cout_p, ptr(u.xin), u.xin_ldc, ptr(self.ws), self.ws_bytes,
                 g[u.keys[0]], B, u.h, u.w_, u.cout_p, u.cin_p, u.cout, 9 * u.cin, u.cout, u.cout_p, 9 * u.cin, u.cin_p, dc, tp, sw)
            return
        _TIMED_UNIT[:] = [u.name + ' wgrad', self.executed_fraction(u, 'wgrad')]
        if u.pre_w and self._x3_ev is not None:
            # the gradient-side operand was transformed on the third stream (enqueued when gz became ready, see below)
            self.wg_stream.wait_event(self._x3_ev)
            call('clamd_wgrad_winograd44_pre' if u.f44 else 'clamd_wgrad_winograd24_pre', None, u.cout_p, ptr(u.vx), ptr(self._x3_buf), ptr(self.ws), self.ws_bytes,
                 g[u.keys[0]], B, u.h, u.w_, u.cout_p, u.cin_p, u.cout, u.cin, u.cout, u.cout_p, c_seg0, c_seg0p, tp, sw)
            ev = torch.cuda.Event(); ev.record(self.wg_stream)
            self._yt_ev[self._yt_flip] = ev
            self._x3_ev = None
        elif u.pre_w:
            _timed('wgrad_conv3x3', flops, self._conv_bytes(u),
                   'clamd_wgrad_winograd44_pre' if u.f44 else 'clamd_wgrad_winograd24_pre', ptr(u.gz), u.cout_p, ptr(u.vx), ptr(self.yt), ptr(self.ws), self.ws_bytes,
                   g[u.keys[0]], B, u.h, u.w_, u.cout_p, u.cin_p, u.cout, u.cin, u.cout, u.cout_p, c_seg0, c_seg0p, tp, sw)
        elif u.wino:
            xin, xin_ldc = (u.fold_a.y, u.fold_a.cout_p) if u.fold_on else (u.xin, u.xin_ldc)
            _timed('wgrad_conv3x3', flops, self._conv_bytes(u),
                   'clamd_wgrad_winograd24' if u.w24g else 'clamd_wgrad_winograd', ptr(u.gz), u.cout_p, ptr(xin), xin_ldc, ptr(self.ws), self.ws_bytes,
                   g[u.keys[0]], B, u.h, u.w_, u.cout_p, u.cin_p, u.cout, u.cin, u.cout, u.cout_p, c_seg0, c_seg0p, tp, sw)
        else:
            xin, xin_ldc = (u.fold_a.y, u.fold_a.cout_p) if u.fold_on else (u.xin, u.xin_ldc)
            _timed('wgrad_conv3x3', flops, self._conv_bytes(u),
                   'clamd_wgrad', _lib.WGRAD_CONV3, ptr(u.gz), u.cout_p, ptr(xin), xin_ldc, ptr(self.ws), self.ws_bytes,
                   g[u.keys[0]], B, u.h, u.w_, u.cout_p, u.cin_p, u.cout, u.cin, u.cout, u.cout_p, c_seg0, c_seg0p, dc, tp, sw)
        if u.fold_on:
            # the weight gradient ran on the producer's conv+ReLU output r instead of x = scale * r + shift: dW = scale * dWr + shift * S, S from
            # the border sums of gz and the conv-bias gradient bn_bwd_finalize wrote above (bnfold.hip); same stream, in place
            a = u.fold_a
            fs = sw
            if (self.x3_stream is not None and self.wg_stream is not None and KERNEL_TIMING is None and self._x3_allowed()
                    and not torch.cuda.is_current_stream_capturing()):
                # two latency-bound launches of a few microseconds: on the third stream they run beside the next unit's weight gradient
                # instead of in front of it (the second stream is the longer one at the end of the bf16 backward pass)
                ev = torch.cuda.Event(); ev.record(self.wg_stream)
                self.x3_stream.wait_event(ev)
                fs, self._x3_fold = self.x3_stream.cuda_stream, True
            call('clamd_bn_fold_wgrad', ptr(u.gz), u.cout_p, g[u.keys[1]], ptr(a.vec[0]), ptr(a.vec[1]), g[u.keys[0]], ptr(self.fold_ws),
                 4 * self.fold_ws.numel(), B, u.h, u.w_, u.cout_p, u.cout, u.cin, dc, fs)
