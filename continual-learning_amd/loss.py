"""Per-pixel cross-entropy (+ build-defined continual-learning distillation) on libclamd's fused kernel.

``CrossEntropyLoss()`` mirrors ``nn.CrossEntropyLoss()`` as the reference uses it (trainer.py:113,174): fp32 NCHW
logits, int64 [B,H,W] labels, mean over non-ignored pixels, ignore_index -100.  Forward and backward are ONE kernel
pass: the forward computes the loss and d loss / d logits; backward only scales it by the incoming gradient.

``DistillationCrossEntropy`` adds  lam * mean_px KL(softmax(z_old[:, :c_old]/T) || softmax(z[:, :c_old]/T))  (LwF-style,
SURVEY.md §8a row A12).  The reference has NO such code: this term is build-defined and its parity is
pinned only by tests against this repo's own CPU restatement.
"""
import os

import torch
import torch.nn as nn

from . import _lib
from ._lib import call, ptr


# d logits handed to the UNet's backward pass in the head data gradient's own layout (see _CEFn.forward); =0: the engine converts the NCHW
# gradient as it does for any other loss (A/B measurements)
HANDOVER = os.environ.get('CLAMD_LOSS_HANDOVER', '1') != '0'


class _CEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, old_logits, c_old, temperature, lam, ignore_index, holder):
        if not logits.is_cuda:
            raise RuntimeError('continual-learning_amd loss runs only on GPU tensors: there is no CPU fallback')
        lib = _lib.load()
        logits_in = logits
        logits = logits.contiguous().float()
        labels = labels.contiguous()
        if labels.dtype != torch.int64:
            raise TypeError('labels must be int64 (datasets/voc.py:72)')
        B, K, H, W = logits.shape
        if tuple(labels.shape) != (B, H, W):
            raise ValueError(f'labels shape {tuple(labels.shape)} does not match logits {tuple(logits.shape)}')
        dl = torch.empty_like(logits)
        out3 = torch.empty(3, dtype=torch.float32, device=logits.device)
        wsb = lib.clamd_ce_workspace_bytes()
        ws = torch.empty(wsb // 4, dtype=torch.float32, device=logits.device)
        kold = 0
        if old_logits is not None:
            old_logits = old_logits.contiguous().float()
            kold = old_logits.shape[1]
            if old_logits.shape[0] != B or tuple(old_logits.shape[2:]) != (H, W):
                raise ValueError('old_logits must be [B, K_old, H, W]')
        ctx.sink = None
        # the four-pixel kernels read 16 bytes of logits / 32 bytes of labels per lane: an odd storage offset takes the scalar kernel
        aligned = logits.data_ptr() % 16 == 0 and dl.data_ptr() % 16 == 0 and labels.data_ptr() % 32 == 0
        if old_logits is None and (H * W) % 4 == 0 and aligned:
            # the training-step form: the count of valid pixels as partial rows (no memset, no atomics), and, when the logits come from this
            # package's UNet, d logits written a second time in the layout (and dtype) its 1x1 head's data gradient reads -- the backward
            # pass then starts without a conversion pass (unet._Engine.backward)
            from . import unet as U
            # (algorithmic bytes, SURVEY 8d: logits read + d logits written + the label; the counting pass reads the labels a second time)
            U._hbm('loss', 0, 'clamd_ce_count', ptr(labels), B, K, H, W, int(ignore_index), ptr(ws), wsb, _lib.stream_ptr())
            eng = U.dlogits_sink(logits_in, B, K, H, W) if HANDOVER else None
            nh, ldc, dcode = (eng.dl, eng.Kp, eng.dcode) if eng is not None else (None, 0, 0)
            U._hbm('loss', B * H * W * (2 * K * 4 + 8 + (ldc * eng.esize if eng is not None else 0)),
                   'clamd_ce_fwd_bwd_counted', ptr(logits), ptr(labels), ptr(dl), ptr(nh), ldc, dcode, ptr(out3), ptr(ws), wsb, B, K, H, W,
                   int(ignore_index), 1.0, _lib.stream_ptr())
            if eng is not None:
                ctx.sink = eng
                # a STRONG reference: while the engine waits for this gradient its storage cannot be freed and handed to another
                # tensor of the same shape (a second loss on the same logits would otherwise pass for this one by address)
                eng.dl_src = (dl, dl.data_ptr(), dl._version, eng.generation)
        else:
            call('clamd_ce_fwd_bwd', ptr(logits), ptr(labels), ptr(old_logits), kold, int(c_old), float(temperature),
                 float(lam), ptr(dl), ptr(out3), ptr(ws), wsb, B, K, H, W, int(ignore_index), 1.0, _lib.stream_ptr())
        ctx.save_for_backward(dl)
        ctx.parts = out3
        # labels outside [0, K) that are not ignore_index: a device counter on the criterion (int(...) synchronises);
        # torch's CrossEntropyLoss asserts on such labels, here they are left out of the mean and counted
        off = lib.clamd_ce_bad_label_count_offset() // 4
        holder.bad_labels = ws[off:off + 1].view(torch.int32)
        return out3[0]

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        # g is the scalar upstream gradient on the DEVICE (exactly 1 for loss.backward()): the kernel tests it there and
        # touches d logits only when it is not 1 -- no host sync, no 176-MB multiply-by-one pass per step
        g = g.contiguous().float()
        eng = ctx.sink
        if eng is not None and eng.dl_src is not None and eng.dl_src[1] == dl.data_ptr():      # the NHWC copy follows: both in one launch
            call('clamd_scale_by_device_scalar_nhwc', ptr(eng.dl), eng.dl.numel(), eng.dcode, ptr(g), ptr(dl), dl.numel(), _lib.stream_ptr())
        else:
            call('clamd_scale_by_device_scalar', ptr(dl), dl.numel(), ptr(g), _lib.stream_ptr())
        return dl, None, None, None, None, None, None, None


class CrossEntropyLoss(nn.Module):
    def __init__(self, ignore_index=-100):
        super().__init__()
        self.ignore_index = ignore_index
        self.bad_labels = None        # after a forward: device int32[1], labels that are neither ignore_index nor a class

    def forward(self, logits, labels):
        return _CEFn.apply(logits, labels, None, 0, 1.0, 0.0, self.ignore_index, self)


class DistillationCrossEntropy(nn.Module):
    """CE(logits, labels) + lam * KL(old || new) over the first ``c_old`` classes at temperature T."""

    def __init__(self, c_old, temperature=2.0, lam=1.0, ignore_index=-100):
        super().__init__()
        self.c_old, self.temperature, self.lam, self.ignore_index = c_old, temperature, lam, ignore_index

    def forward(self, logits, labels, old_logits):
        return _CEFn.apply(logits, labels, old_logits.detach(), self.c_old, self.temperature, self.lam, self.ignore_index, self)
