"""GPU data path: what the reference does per sample on the CPU with PIL/torchvision and a
per-pixel Python loop (main.py:18-23; datasets/voc.py:56-72,127-144), as one libclamd kernel on uint8 RGB tensors.

``prepare_sample(img_u8, mask_u8, (h, w))``: ``img_u8`` / ``mask_u8`` are ``[Hs, Ws, 3]`` uint8 GPU tensors (decoded
JPEG / palette PNG converted to RGB, as ``Image.open(...).convert('RGB')`` yields).  Returns ``(image, labels)`` =
fp32 ``[3,h,w]`` normalised to [-1,1] and int64 ``[h,w]`` class indices, exactly what ``VOC.__getitem__`` returns.
"""
import torch

from . import _lib
from ._lib import call, ptr

PAD = 10          # transforms.Pad(10), main.py:19 / voc.py:140


def crop_origin(hs, ws, h, w, pad=PAD):
    """Source coordinate of output pixel (0,0) after Pad(pad) then torchvision CenterCrop((h, w)).
    torchvision.transforms.functional.center_crop: an image smaller than the crop is first zero-padded by
    (crop - size)//2 on the top/left and (crop - size + 1)//2 on the bottom/right; then
    top = int(round((H - h) / 2.0)), left = int(round((W - w) / 2.0)) (Python round, half to even)."""
    def one(src, crop):
        size = src + 2 * pad
        lead = (crop - size) // 2 if crop > size else 0
        trail = (crop - size + 1) // 2 if crop > size else 0
        start = int(round((size + lead + trail - crop) / 2.0))
        return start - lead - pad
    return one(hs, h), one(ws, w)


def prepare_sample(img_u8, mask_u8, size, check=True):
    h, w = size
    src = img_u8 if img_u8 is not None else mask_u8
    if not src.is_cuda or src.dtype != torch.uint8 or src.dim() != 3 or src.shape[2] != 3:
        raise ValueError('expected [Hs, Ws, 3] uint8 GPU tensors (there is no CPU fallback)')
    hs, ws = src.shape[:2]
    oy, ox = crop_origin(hs, ws, h, w)
    dev = src.device
    image = torch.empty(3, h, w, dtype=torch.float32, device=dev) if img_u8 is not None else None
    labels = torch.empty(h, w, dtype=torch.int64, device=dev) if mask_u8 is not None else None
    bad = torch.zeros(1, dtype=torch.int32, device=dev)
    call('clamd_voc_prepare', ptr(img_u8.contiguous() if img_u8 is not None else None),
         ptr(mask_u8.contiguous() if mask_u8 is not None else None), ptr(image), ptr(labels), hs, ws, oy, ox, h, w,
         ptr(bad), _lib.stream_ptr())
    if check and mask_u8 is not None and int(bad) != 0:
        raise ValueError(f'{int(bad)} mask pixels have a colour outside the VOC palette (voc.to_mask would raise)')
    return image, labels


def to_rgb(labels):
    """voc.to_rgb (datasets/voc.py:74-89): int64 [N,H,W] labels -> [N,3,H,W] palette colours (0..255)."""
    if not labels.is_cuda or labels.dtype != torch.int64 or labels.dim() != 3:
        raise ValueError('expected an int64 [N,H,W] GPU tensor')
    n, h, w = labels.shape
    out = torch.empty(n, 3, h, w, dtype=torch.float32, device=labels.device)
    call('clamd_label_to_rgb', ptr(labels.contiguous()), ptr(out), n, h * w, _lib.stream_ptr())
    return out
