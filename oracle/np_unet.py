"""CPU oracle: NumPy restatement of the reference's UNet train-step path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` may be imported by the product
package (``continual-learning_amd/``).  Allowed importers: ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg, and there only as
the *checker* / reported CPU baseline, never as the thing measured or shipped.

Parity status: PINNED.  ``oracle/gen_golden.py`` imports the real reference
(``/root/reference/models/unet.py``, ``/root/reference/metrics.py``) in the build
container and writes ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every
function below against those fixtures.  The continual-learning regulariser
(``distill_kl`` / ``l2_to_old``) has NO reference code and is
therefore "parity unpinned": it is pinned only by its own closed-form/finite-difference tests.

All arrays are NCHW float32 unless a function says otherwise (the reference's layout,
``models/unet.py:74-92``).  ``dtype`` can be switched to float64 to obtain a
higher-precision "truth" for judging which of two fp32 implementations is closer.

Reference map (file:line relative to /root/reference):
  conv3x3_fwd / conv1x1       models/unet.py:13,16,28,31,50,53,66,69,72  (nn.Conv2d)
  relu                        models/unet.py:14,17,29,32,51,54,67,70     (nn.ReLU, BEFORE BatchNorm)
  bn_train_fwd                models/unet.py:15,18,30,33,52,55,68,71     (nn.BatchNorm2d train mode)
  maxpool2x2                  models/unet.py:12,80                       (nn.MaxPool2d(2,2))
  convT2x2                    models/unet.py:34                          (nn.ConvTranspose2d k2 s2)
  concat (encoder first)      models/unet.py:83-87
  cross_entropy               trainer.py:113,174                         (nn.CrossEntropyLoss, mean)
  adam_step                   trainer.py:108-110,176                     (optim.Adam, betas from cfg)
  poly_lr                     trainer.py:111-112,147                     (LambdaLR, stepped per epoch)
  confusion / mean_iu2        metrics.py:23-38,55-63
"""
import numpy as np

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


# --------------------------------------------------------------------------- layers
def _im2col3(x):
    """[B,C,H,W] -> [B,H,W,C*9] patches for a 3x3 / stride 1 / zero-pad 1 window.
    Column order (c, ky, kx) matches a [Cout, Cin, 3, 3] weight reshaped to [Cout, Cin*9]."""
    B, C, H, W = x.shape
    xp = np.zeros((B, C, H + 2, W + 2), x.dtype)
    xp[:, :, 1:-1, 1:-1] = x
    cols = np.empty((B, H, W, C, 3, 3), x.dtype)
    for ky in range(3):
        for kx in range(3):
            cols[:, :, :, :, ky, kx] = xp[:, :, ky:ky + H, kx:kx + W].transpose(0, 2, 3, 1)
    return cols.reshape(B, H, W, C * 9)


def conv3x3_fwd(x, w, b):
    """nn.Conv2d(k=3, s=1, p=1) forward.  models/unet.py:13 (and the 17 other 3x3 sites)."""
    B, C, H, W = x.shape
    co = w.shape[0]
    y = _im2col3(x).reshape(-1, C * 9) @ w.reshape(co, C * 9).T
    y = y.reshape(B, H, W, co) + b
    return np.ascontiguousarray(y.transpose(0, 3, 1, 2))


def conv3x3_bwd(x, w, gy, need_gx=True):
    """Returns (gx, gw, gb) of conv3x3_fwd."""
    B, C, H, W = x.shape
    co = w.shape[0]
    g2 = gy.transpose(0, 2, 3, 1).reshape(-1, co)
    gw = (g2.T @ _im2col3(x).reshape(-1, C * 9)).reshape(co, C, 3, 3)
    gb = g2.sum(0)
    gx = None
    if need_gx:
        # dgrad = correlation of gy with the spatially flipped, in/out-swapped filter
        wf = np.ascontiguousarray(w[:, :, ::-1, ::-1].transpose(1, 0, 2, 3))
        gx = conv3x3_fwd(gy, wf, np.zeros(C, x.dtype))
    return gx, gw, gb


def conv1x1_fwd(x, w, b):
    """nn.Conv2d(k=1) head.  models/unet.py:72."""
    co, ci = w.shape[:2]
    y = np.einsum('bchw,oc->bohw', x, w.reshape(co, ci), optimize=True)
    return y + b.reshape(1, co, 1, 1)


def conv1x1_bwd(x, w, gy):
    co, ci = w.shape[:2]
    gx = np.einsum('bohw,oc->bchw', gy, w.reshape(co, ci), optimize=True)
    gw = np.einsum('bohw,bchw->oc', gy, x, optimize=True).reshape(w.shape)
    gb = gy.sum((0, 2, 3))
    return gx, gw, gb


def relu_fwd(x):
    return np.maximum(x, 0)


def bn_train_fwd(x, gamma, beta, running_mean=None, running_var=None):
    """nn.BatchNorm2d in train mode (models/unet.py:15): biased variance for normalisation;
    running stats use momentum 0.1 and the UNBIASED variance.  Returns (y, cache, new_rm, new_rv)."""
    n = x.shape[0] * x.shape[2] * x.shape[3]
    mean = x.mean((0, 2, 3), dtype=x.dtype)
    var = ((x - mean.reshape(1, -1, 1, 1)) ** 2).mean((0, 2, 3), dtype=x.dtype)
    istd = 1.0 / np.sqrt(var + x.dtype.type(BN_EPS))
    xh = (x - mean.reshape(1, -1, 1, 1)) * istd.reshape(1, -1, 1, 1)
    y = xh * gamma.reshape(1, -1, 1, 1) + beta.reshape(1, -1, 1, 1)
    new_rm = new_rv = None
    if running_mean is not None:
        m = x.dtype.type(BN_MOMENTUM)
        new_rm = (1 - m) * running_mean + m * mean
        new_rv = (1 - m) * running_var + m * var * (n / max(n - 1, 1))
    return y, (xh, istd), new_rm, new_rv


def bn_train_bwd(gy, gamma, cache):
    xh, istd = cache
    n = gy.shape[0] * gy.shape[2] * gy.shape[3]
    gbeta = gy.sum((0, 2, 3))
    ggamma = (gy * xh).sum((0, 2, 3))
    k = (gamma * istd).reshape(1, -1, 1, 1)
    gx = k * (gy - (gbeta / n).reshape(1, -1, 1, 1) - xh * (ggamma / n).reshape(1, -1, 1, 1))
    return gx, ggamma, gbeta


def bn_eval_fwd(x, gamma, beta, running_mean, running_var):
    istd = 1.0 / np.sqrt(running_var + x.dtype.type(BN_EPS))
    return (x - running_mean.reshape(1, -1, 1, 1)) * (gamma * istd).reshape(1, -1, 1, 1) \
        + beta.reshape(1, -1, 1, 1)


def maxpool2x2_fwd(x):
    """nn.MaxPool2d(2,2) (models/unet.py:12,80).  Returns (y, argmax) with argmax in 0..3 = 2*dy+dx,
    first maximum in row-major window order winning ties (torch's rule)."""
    B, C, H, W = x.shape
    win = x.reshape(B, C, H // 2, 2, W // 2, 2).transpose(0, 1, 2, 4, 3, 5).reshape(B, C, H // 2, W // 2, 4)
    idx = win.argmax(-1)
    return np.take_along_axis(win, idx[..., None], -1)[..., 0], idx


def maxpool2x2_bwd(gy, idx):
    B, C, h, w = gy.shape
    win = np.zeros((B, C, h, w, 4), gy.dtype)
    np.put_along_axis(win, idx[..., None], gy[..., None], -1)
    return win.reshape(B, C, h, w, 2, 2).transpose(0, 1, 2, 4, 3, 5).reshape(B, C, 2 * h, 2 * w)


def convT2x2_fwd(x, w, b):
    """nn.ConvTranspose2d(k=2, s=2) (models/unet.py:34); weight layout [Cin, Cout, 2, 2].
    Non-overlapping: out[b,co,2y+dy,2x+dx] = sum_ci x[b,ci,y,x] * w[ci,co,dy,dx] + b[co]."""
    B, ci, h, wd = x.shape
    co = w.shape[1]
    y = np.einsum('bchw,codx->bohdwx', x, w, optimize=True)  # [B,co,h,dy,w,dx]
    return y.reshape(B, co, 2 * h, 2 * wd) + b.reshape(1, co, 1, 1)


def convT2x2_bwd(x, w, gy):
    B, ci, h, wd = x.shape
    co = w.shape[1]
    g6 = gy.reshape(B, co, h, 2, wd, 2)
    gx = np.einsum('bohdwx,codx->bchw', g6, w, optimize=True)
    gw = np.einsum('bchw,bohdwx->codx', x, g6, optimize=True)
    gb = gy.sum((0, 2, 3))
    return gx, gw, gb


# --------------------------------------------------------------------------- loss
def cross_entropy(logits, labels, ignore_index=-100):
    """nn.CrossEntropyLoss() (trainer.py:113,174): log-softmax over dim 1, NLL, mean over the
    non-ignored pixels.  Returns (loss, dlogits)."""
    B, K, H, W = logits.shape
    z = logits - logits.max(1, keepdims=True)
    lse = np.log(np.exp(z).sum(1, keepdims=True))
    logp = z - lse
    valid = labels != ignore_index
    nvalid = max(int(valid.sum()), 1)
    lab = np.where(valid, labels, 0)
    picked = np.take_along_axis(logp, lab[:, None], 1)[:, 0]
    loss = -(picked * valid).sum(dtype=np.float64) / nvalid
    onehot = np.zeros_like(logits)
    np.put_along_axis(onehot, lab[:, None], 1, 1)
    d = (np.exp(logp) - onehot) * valid[:, None] / logits.dtype.type(nvalid)
    return logits.dtype.type(loss), d.astype(logits.dtype)


def distill_kl(z_new, z_old, c_old, temperature=2.0, lam=1.0):
    """BUILD-DEFINED (no reference code; SURVEY.md §8a A12, parity unpinned).
    lam * mean_px KL( softmax(z_old[:, :c_old]/T) || softmax(z_new[:, :c_old]/T) ).
    Returns (loss, d/dz_new) with the gradient zero for channels >= c_old."""
    T = z_new.dtype.type(temperature)
    a = z_new[:, :c_old] / T
    b = z_old[:, :c_old] / T
    la = a - a.max(1, keepdims=True)
    la = la - np.log(np.exp(la).sum(1, keepdims=True))
    lb = b - b.max(1, keepdims=True)
    lb = lb - np.log(np.exp(lb).sum(1, keepdims=True))
    p = np.exp(lb)
    npx = z_new.shape[0] * z_new.shape[2] * z_new.shape[3]
    loss = lam * (p * (lb - la)).sum(dtype=np.float64) / npx
    g = np.zeros_like(z_new)
    g[:, :c_old] = (lam / (npx * T)) * (np.exp(la) - p)
    return z_new.dtype.type(loss), g


def l2_to_old(params, old_params, lam):
    """BUILD-DEFINED (parity unpinned): lam * sum ||theta - theta_old||^2 ; grad = 2 lam (theta - theta_old)."""
    loss = 0.0
    grads = {}
    for k, v in params.items():
        d = v - old_params[k]
        loss += lam * float((d.astype(np.float64) ** 2).sum())
        grads[k] = (2 * lam) * d
    return loss, grads


# --------------------------------------------------------------------------- optimiser
def adam_step(p, g, m, v, step, lr, beta1=0.5, beta2=0.99, eps=1e-8):
    """torch.optim.Adam (trainer.py:108-110; defaults main.py: beta1 0.5, beta2 0.99), weight decay 0,
    amsgrad off.  ``step`` is the 1-based step count AFTER increment.  Returns (p, m, v)."""
    f = p.dtype.type
    m = m + f(1 - beta1) * (g - m)              # exp_avg.lerp_(grad, 1-beta1)
    v = v * f(beta2) + f(1 - beta2) * g * g     # exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = np.sqrt(v) / f(bc2 ** 0.5) + f(eps)
    p = p - f(lr / bc1) * (m / denom)
    return p, m, v


def poly_lr(base_lr, n, n_iters, lr_exp):
    """LambdaLR lambda of trainer.py:111: lr * (1 - n/n_iters)**lr_exp.  The reference calls
    scheduler.step() BEFORE the first optimiser step of every epoch (trainer.py:147), so epoch e
    trains at n = e + 1."""
    return base_lr * (1 - n / n_iters) ** lr_exp


# --------------------------------------------------------------------------- metrics
def confusion(target, pred, num_classes):
    """metrics.py:32-38 _fast_conf_matrix: bincount(C*t + p) over pixels with 0 <= t < C, as float32."""
    t = target.reshape(-1)
    p = pred.reshape(-1)
    mask = (t >= 0) & (t < num_classes)
    return np.bincount(num_classes * t[mask] + p[mask], minlength=num_classes ** 2) \
        .reshape(num_classes, num_classes).astype(np.float32)


def _nanmean(x):
    x = x[x == x]
    return x.mean(dtype=np.float32) if x.size else np.float32(np.nan)


def mean_iu2(matrix):
    """metrics.py:23-29: diag / (row + col - diag), NaN classes dropped (0/0), float32 arithmetic."""
    with np.errstate(divide='ignore', invalid='ignore'):
        d = np.diag(matrix)
        j = d / (matrix.sum(1) + matrix.sum(0) - d)
    return _nanmean(j.astype(np.float32))


def eval_metrics(targets, preds, num_classes):
    """metrics.py:55-63: accumulate the confusion matrix over the batch, then
    (overall_acc %, avg per-class acc %, mean_IU_2, max per-class acc %)."""
    m = np.zeros((num_classes, num_classes), np.float32)
    for t, p in zip(targets, preds):
        m += confusion(t, p, num_classes)
    with np.errstate(divide='ignore', invalid='ignore'):
        d = np.diag(m)
        overall = d.sum() * 100 / m.sum()
        per = (100 * d / m.sum(1)).astype(np.float32)
    per_valid = per[per == per]
    return (np.float32(overall), _nanmean(per), mean_iu2(m),
            per_valid.max() if per_valid.size else np.float32(np.nan), m)


# --------------------------------------------------------------------------- data path
VOC_PALETTE = [(0, 0, 0), (128, 0, 0), (0, 128, 0), (128, 128, 0), (0, 0, 128), (128, 0, 128), (0, 128, 128),
               (128, 128, 128), (64, 0, 0), (192, 0, 0), (64, 128, 0), (192, 128, 0), (64, 0, 128), (192, 0, 128),
               (64, 128, 128), (192, 128, 128), (0, 64, 0), (128, 64, 0), (0, 192, 0), (128, 192, 0), (0, 64, 128),
               (224, 224, 192)]          # datasets/voc.py:32-53 (data); index 21 = void


def pad_center_crop(a, h, w, pad=10):
    """transforms.Pad(pad) then CenterCrop((h, w)) on an [H,W,C] array (main.py:19-20, voc.py:140-141).  torchvision
    is not installed in the build container, so this follows torchvision.transforms.functional.{pad,center_crop} as
    documented (zero fill; undersized images zero-padded (d//2, (d+1)//2); top = int(round((H-h)/2.0))): this part of
    the data path is restated, not pinned by running the reference."""
    a = np.pad(a, ((pad, pad), (pad, pad), (0, 0)))
    H, W = a.shape[:2]
    if h > H or w > W:
        pt, pb = ((h - H) // 2, (h - H + 1) // 2) if h > H else (0, 0)
        pl, pr = ((w - W) // 2, (w - W + 1) // 2) if w > W else (0, 0)
        a = np.pad(a, ((pt, pb), (pl, pr), (0, 0)))
        H, W = a.shape[:2]
    top, left = int(round((H - h) / 2.0)), int(round((W - w) / 2.0))
    return a[top:top + h, left:left + w]


def to_mask(mask_rgb):
    """voc.to_mask (datasets/voc.py:56-72): [H,W,3] uint8 palette colours -> [H,W] int64 class index, void -> 0; a colour
    outside the palette raises ValueError (list.index).  Pinned by tests/golden/voc.npz (captured from the reference)."""
    H, W = mask_rgb.shape[:2]
    m = mask_rgb.reshape(-1, 3)
    lab = np.empty(m.shape[0], np.int64)
    for i, px in enumerate(map(tuple, m)):           # the reference's per-pixel loop, voc.py:64-70
        lab[i] = 0 if px == (224, 224, 192) else VOC_PALETTE.index(px)
    return lab.reshape(H, W)


def voc_prepare(img_u8, mask_u8, h, w):
    """VOC.__getitem__ (datasets/voc.py:127-144): image -> Pad, CenterCrop, ToTensor, Normalize(0.5, 0.5) [3,h,w] f32;
    mask -> Pad, CenterCrop, to_mask (voc.py:56-72: palette index, void -> 0) [h,w] int64."""
    img = pad_center_crop(img_u8, h, w).astype(np.float32) / np.float32(255)
    image = ((img - np.float32(0.5)) / np.float32(0.5)).transpose(2, 0, 1)
    return np.ascontiguousarray(image), to_mask(pad_center_crop(mask_u8, h, w))


def to_rgb(labels):
    """voc.to_rgb (datasets/voc.py:74-89).  Pinned by tests/golden/voc.npz."""
    pal = np.array(VOC_PALETTE, np.float32)
    return pal[labels].transpose(0, 3, 1, 2)


# --------------------------------------------------------------------------- network
def layer_table(num_classes, in_dim=3, conv_dim=64):
    """state_dict naming of models/unet.py:49-72 as a flat table.
    Each stage: (prefix, pool_first, [(conv_key, bn_key, cin, cout), (conv_key, bn_key, cin, cout)], tail)
    tail = ('convT', key, cin, cout) | ('head', key, cin, cout) | None."""
    d = conv_dim
    t = [('enc1', False, [('enc1.0', 'enc1.2', in_dim, d), ('enc1.3', 'enc1.5', d, d)], None)]
    c = d
    for i in (2, 3, 4):
        t.append((f'enc{i}', True, [(f'enc{i}.block.1', f'enc{i}.block.3', c, 2 * c),
                                    (f'enc{i}.block.4', f'enc{i}.block.6', 2 * c, 2 * c)], None))
        c *= 2
    # c == 8d.  dec1: 8d -> 16d -> convT 8d ; dec2: 16d -> 8d -> 4d ; ...
    spec = [(8 * d, 16 * d, 8 * d), (16 * d, 8 * d, 4 * d), (8 * d, 4 * d, 2 * d), (4 * d, 2 * d, d)]
    for i, (cin, mid, cout) in enumerate(spec, 1):
        p = f'dec{i}.block'
        t.append((f'dec{i}', False, [(f'{p}.0', f'{p}.2', cin, mid), (f'{p}.3', f'{p}.5', mid, mid)],
                  ('convT', f'{p}.6', mid, cout)))
    t.append(('last', False, [('last.0', 'last.2', 2 * d, d), ('last.3', 'last.5', d, d)],
              ('head', 'last.6', d, num_classes)))
    return t


def _block_fwd(x, P, convs, caches, stats_out):
    for ck, bk, _, _ in convs:
        z = conv3x3_fwd(x, P[ck + '.weight'], P[ck + '.bias'])
        y = relu_fwd(z)
        u, bc, rm, rv = bn_train_fwd(y, P[bk + '.weight'], P[bk + '.bias'],
                                     P.get(bk + '.running_mean'), P.get(bk + '.running_var'))
        if rm is not None:
            stats_out[bk + '.running_mean'] = rm
            stats_out[bk + '.running_var'] = rv
        caches.append((ck, bk, x, y, bc))
        x = u
    return x


def unet_forward(P, x, num_classes, in_dim=3, conv_dim=64):
    """models/unet.py:74-92 in train mode.  P: dict of state_dict-named arrays.
    Returns (logits, cache, new_running_stats)."""
    tab = layer_table(num_classes, in_dim, conv_dim)
    cache = {'stages': [], 'tab': tab}
    new_stats = {}
    enc = {}
    h = x
    for prefix, pool_first, convs, tail in tab[:4]:
        st = {'convs': []}
        if pool_first:
            h, st['pool_idx'] = maxpool2x2_fwd(h)
        h = _block_fwd(h, P, convs, st['convs'], new_stats)
        enc[prefix] = h
        cache['stages'].append(st)
    center, cidx = maxpool2x2_fwd(enc['enc4'])
    cache['center_idx'] = cidx
    h = center
    skips = [None, 'enc4', 'enc3', 'enc2', 'enc1']
    for i, (prefix, _, convs, tail) in enumerate(tab[4:]):
        st = {'convs': []}
        if skips[i] is not None:
            h = np.concatenate([enc[skips[i]], h], 1)   # encoder first (unet.py:83-87)
        h = _block_fwd(h, P, convs, st['convs'], new_stats)
        st['tail_in'] = h
        kind, key, _, _ = tail
        if kind == 'convT':
            h = convT2x2_fwd(h, P[key + '.weight'], P[key + '.bias'])
        else:
            h = conv1x1_fwd(h, P[key + '.weight'], P[key + '.bias'])
        cache['stages'].append(st)
    return h, cache, new_stats


def _block_bwd(g, P, st_convs, G, need_gx_first=True):
    for i in range(len(st_convs) - 1, -1, -1):
        ck, bk, xin, y, bc = st_convs[i]
        gy, gg, gb = bn_train_bwd(g, P[bk + '.weight'], bc)
        G[bk + '.weight'] = gg
        G[bk + '.bias'] = gb
        gz = gy * (y > 0)
        need = need_gx_first or i > 0
        g, gw, gcb = conv3x3_bwd(xin, P[ck + '.weight'], gz, need_gx=need)
        G[ck + '.weight'] = gw
        G[ck + '.bias'] = gcb
    return g


def unet_backward(P, cache, glogits):
    """Gradient of every parameter given d loss / d logits.  Returns dict keyed like P."""
    tab = cache['tab']
    G = {}
    g = glogits
    genc = {}
    skips = [None, 'enc4', 'enc3', 'enc2', 'enc1']
    for i in range(4, -1, -1):
        prefix, _, convs, tail = tab[4 + i]
        st = cache['stages'][4 + i]
        kind, key, _, _ = tail
        if kind == 'convT':
            g, gw, gb = convT2x2_bwd(st['tail_in'], P[key + '.weight'], g)
        else:
            g, gw, gb = conv1x1_bwd(st['tail_in'], P[key + '.weight'], g)
        G[key + '.weight'] = gw
        G[key + '.bias'] = gb
        g = _block_bwd(g, P, st['convs'], G)
        if skips[i] is not None:
            c = g.shape[1] // 2
            genc[skips[i]] = g[:, :c]
            g = g[:, c:]
    # g is now d/d center
    g = maxpool2x2_bwd(g, cache['center_idx']) + genc['enc4']
    for i in range(3, -1, -1):
        prefix, pool_first, convs, _ = tab[i]
        st = cache['stages'][i]
        g = _block_bwd(g, P, st['convs'], G, need_gx_first=(i > 0))
        if pool_first:
            g = maxpool2x2_bwd(g, st['pool_idx']) + genc[f'enc{i}']
    return G


def param_keys(num_classes, in_dim=3, conv_dim=64):
    """Parameter names in registration order (= optimiser order, models/unet.py:49-72)."""
    keys = []
    for prefix, _, convs, tail in layer_table(num_classes, in_dim, conv_dim):
        for ck, bk, _, _ in convs:
            keys += [ck + '.weight', ck + '.bias', bk + '.weight', bk + '.bias']
        if tail is not None:
            keys += [tail[1] + '.weight', tail[1] + '.bias']
    return keys


def train_step(P, opt_state, x, labels, step, lr, num_classes, in_dim=3, conv_dim=64,
               beta1=0.5, beta2=0.99):
    """One iteration of trainer.py:172-176: forward, CE, backward, Adam.  Mutates nothing;
    returns (loss, logits, grads, new_P, new_opt_state)."""
    logits, cache, new_stats = unet_forward(P, x, num_classes, in_dim, conv_dim)
    loss, dl = cross_entropy(logits, labels)
    G = unet_backward(P, cache, dl)
    newP = dict(P)
    newP.update(new_stats)
    new_state = {}
    for k in param_keys(num_classes, in_dim, conv_dim):
        m, v = opt_state.get(k, (np.zeros_like(P[k]), np.zeros_like(P[k])))
        p, m, v = adam_step(P[k], G[k], m, v, step, lr, beta1, beta2)
        newP[k] = p
        new_state[k] = (m, v)
    return loss, logits, G, newP, new_state
