"""GPU-resident segmentation metrics: argmax over classes fused with the confusion-matrix
histogram, then the reference's formulas (metrics.py:6-63) on the tiny K x K matrix.

``eval_metrics(target, logits_or_pred, num_classes)`` returns the same 4-tuple as the reference's
``metrics.eval_metrics`` (overall acc %, mean per-class acc %, mean_IU_2, max per-class acc %).
"""
import torch

from . import _lib
from ._lib import call, ptr


def argmax_confusion(logits, labels, num_classes_conf=None, want_pred=False):
    """logits fp32 NCHW (GPU), labels int64 [B,H,W] -> (confusion [Kc,Kc] int64 on GPU, pred or None)."""
    if not logits.is_cuda:
        raise RuntimeError('GPU tensors only: there is no CPU fallback')
    logits = logits.contiguous().float()
    B, K, H, W = logits.shape
    Kc = num_classes_conf or K
    conf = torch.zeros(Kc * Kc, dtype=torch.int64, device=logits.device)
    pred = torch.empty(B, H, W, dtype=torch.int64, device=logits.device) if want_pred else None
    call('clamd_argmax_confusion', ptr(logits), ptr(labels.contiguous()), ptr(pred), ptr(conf), B, K, Kc, H, W,
         _lib.stream_ptr())
    return conf.view(Kc, Kc), pred


def _nanmean(x):
    x = x[x == x]
    return x.mean() if x.numel() else torch.tensor(float('nan'))


def metrics_from_confusion(conf):
    """metrics.py:6-29,40-53 on a confusion matrix (float32 arithmetic, NaN classes dropped)."""
    m = conf.detach().to('cpu', torch.float32)
    d = torch.diag(m)
    overall = d.sum() * 100 / m.sum()
    per = 100 * d / m.sum(1)
    jac = d / (m.sum(1) + m.sum(0) - d)
    per_valid = per[per == per]
    return overall, _nanmean(per), _nanmean(jac), (per_valid.max() if per_valid.numel() else torch.tensor(float('nan')))


def eval_metrics(target, logits, num_classes):
    conf, _ = argmax_confusion(logits, target, num_classes)
    return metrics_from_confusion(conf)
