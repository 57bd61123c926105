"""Counter-based synthetic data and closed-form weights.

Everything here is a pure function of (seed, flat index) through splitmix64, so the build
container (where the golden vectors are captured from the reference) and the GPU box (where
the reference does not exist) regenerate bit-identical tensors without torch's RNG.

Images: x ~ U(-1, 1) fp32 NCHW -- the range main.py:21-22's Normalize(0.5, 0.5) gives [0,1] pixels.
Labels: int64 [B,H,W], *blocky* (constant over 16x16 cells) so mIoU is learnable.
"""
import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    """Vectorised splitmix64 finaliser on uint64 arrays."""
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over='ignore'):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def _uniform01(seed, n, offset=0):
    """n floats in [0,1) with 24 random bits each (exactly representable in fp32)."""
    idx = np.arange(offset, offset + n, dtype=np.uint64)
    with np.errstate(over='ignore'):
        h = splitmix64(idx ^ (np.uint64(seed) * np.uint64(0xD1342543DE82EF95)))
    return ((h >> np.uint64(40)).astype(np.float32)) * np.float32(1.0 / (1 << 24))


def images(seed, batch, channels, height, width, first_image=0):
    """fp32 NCHW images in [-1, 1); image b depends only on (seed, first_image + b)."""
    per = channels * height * width
    u = _uniform01(seed, batch * per, offset=first_image * per)
    return (u * np.float32(2.0) - np.float32(1.0)).reshape(batch, channels, height, width)


def labels(seed, batch, height, width, num_classes, cell=16, first_image=0, class_lo=0, class_hi=None):
    """int64 [B,H,W]; class = hash(seed, b, y//cell, x//cell) mod C.  ``class_lo/hi`` restrict the
    label set for the continual two-task split (classes outside [lo,hi) map to 0 = background)."""
    b = np.arange(first_image, first_image + batch, dtype=np.uint64)[:, None, None]
    cy = (np.arange(height, dtype=np.uint64) // np.uint64(cell))[None, :, None]
    cx = (np.arange(width, dtype=np.uint64) // np.uint64(cell))[None, None, :]
    with np.errstate(over='ignore'):
        key = (b * np.uint64(1000003) + cy) * np.uint64(1000033) + cx
        h = splitmix64(key ^ (np.uint64(seed) * np.uint64(0xA24BAED4963EE407)))
    lab = (h % np.uint64(num_classes)).astype(np.int64)
    if class_hi is not None:
        lab = np.where((lab >= class_lo) & (lab < class_hi), lab, 0)
    return lab


def class_colours(num_classes, seed=0):
    """One RGB triple in [-1, 1) per class (closed form), for ``images_with_signal``."""
    return (_uniform01(seed ^ 0x5EED, num_classes * 3) * np.float32(2.0) - np.float32(1.0)).reshape(num_classes, 3)


def images_with_signal(seed, lab, num_classes, mix=0.5, first_image=0):
    """Images that carry their labels: x = (1 - mix) * U(-1,1) noise + mix * colour[label] (fp32 NCHW, still inside [-1, 1)).
    ``images`` alone is independent of the labels (a model can only memorise it); the fixed 64-image set of the
    mIoU-after-training check uses this so that a few Adam steps move mIoU far from chance."""
    b, h, w = lab.shape
    noise = images(seed, b, 3, h, w, first_image=first_image)
    col = class_colours(num_classes)[lab].transpose(0, 3, 1, 2)            # [B,3,H,W]
    return (np.float32(1.0 - mix) * noise + np.float32(mix) * col).astype(np.float32)


def closed_form_tensor(name, shape, seed=0):
    """Deterministic stand-in for torch's default init, as a function of (name, flat index):
    conv / convT weights and biases ~ U(+-1/sqrt(fan_in)); BN gamma in [0.5,1.5), beta in [-0.25,0.25)
    (non-trivial affine so BN-apply paths are exercised, including nothing negative-gamma here;
    negative gammas are covered by the per-op tests)."""
    n = int(np.prod(shape))
    s = (zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0xFFFFFFFF
    u = _uniform01(s, n)
    if len(shape) == 4:
        fan_in = shape[1] * shape[2] * shape[3]
        bound = 1.0 / np.sqrt(fan_in)
        return ((u * 2 - 1) * np.float32(bound)).astype(np.float32).reshape(shape)
    return u.reshape(shape)  # caller rescales 1-D tensors (needs to know what they are)


def closed_form_state(param_shapes, seed=0):
    """param_shapes: ordered {name: shape} for weights/biases (conv, convT, BN).  BN tensors are
    recognised by being 1-D with a sibling 4-D '.weight' absent (i.e. prefix has a 1-D weight)."""
    out = {}
    for name, shape in param_shapes.items():
        t = closed_form_tensor(name, shape, seed)
        if len(shape) == 1:
            prefix, leaf = name.rsplit('.', 1)
            wshape = param_shapes.get(prefix + '.weight')
            if wshape is not None and len(wshape) == 4:      # conv / convT bias
                if leaf == 'bias':
                    # torch: fan_in of the sibling weight ([Cout,Cin,k,k]; convT is [Cin,Cout,k,k])
                    fan_in = wshape[1] * wshape[2] * wshape[3]
                    t = (t * 2 - 1) * np.float32(1.0 / np.sqrt(fan_in))
            else:                                            # BatchNorm affine
                t = t + np.float32(0.5) if leaf == 'weight' else (t - np.float32(0.5)) * np.float32(0.5)
        out[name] = t.astype(np.float32)
    return out
