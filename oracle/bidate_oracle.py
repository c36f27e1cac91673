"""CPU restatement of the reference's bi-date Siamese U-Net hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the
product package ``fabric_amd``.

Every function cites the reference file:line it follows (paths relative to
/root/reference).  The reference's arithmetic lives in PyTorch (ATen CPU
kernels); this restatement is written as explicit functional code over plain
tensors (own BatchNorm, max-pool, bilinear and loss arithmetic; only the 3x3 /
1x1 correlation itself is delegated to ``F.conv2d``) and is pinned to golden
vectors produced by importing the reference itself (oracle/make_golden.py).

Layout here is the reference's: NCHW float32 (or float64 when asked).
"""
import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-5        # torch.nn.BatchNorm2d default used at models/unet_parts.py:14,17
BN_MOMENTUM = 0.1    # ditto


# --------------------------------------------------------------------------
# leaf ops
# --------------------------------------------------------------------------
def conv3x3(x, w, b=None):
    """nn.Conv2d(ci, co, 3, padding=1) -- models/unet_parts.py:13,16."""
    return F.conv2d(x, w, b, stride=1, padding=1)


def conv1x1(x, w, b=None):
    """nn.Conv2d(ci, co, 1) -- models/unet_parts.py:86."""
    return F.conv2d(x, w, b)


def bn_batch_stats(z):
    """Per-channel mean and *biased* variance over (N,H,W) of one call.
    ATen's CPU batch-norm accumulates float32 statistics in double; so do we."""
    zd = z.double()
    mean = zd.mean(dim=(0, 2, 3))
    var = ((zd - mean[None, :, None, None]) ** 2).mean(dim=(0, 2, 3))
    return mean.to(z.dtype), var.to(z.dtype)


class _BNTrain(torch.autograd.Function):
    """Training-mode batch norm with the explicit backward

        dgamma = sum(g * xhat), dbeta = sum(g),
        dz = gamma * inv_std * (g - dbeta / M - xhat * dgamma / M)

    (reductions accumulated in double like ATen's CPU kernel).  Differentiating
    the unfused mean/var expression instead loses ~1e-2 relative accuracy on the
    preceding conv's weight gradient through cancellation."""

    @staticmethod
    def forward(ctx, z, gamma, beta, eps):
        mean, var = bn_batch_stats(z)
        inv = torch.rsqrt(var + eps)
        xhat = (z - mean[None, :, None, None]) * inv[None, :, None, None]
        ctx.save_for_backward(xhat, gamma, inv)
        ctx.mark_non_differentiable(mean, var)
        return xhat * gamma[None, :, None, None] + beta[None, :, None, None], mean, var

    @staticmethod
    def backward(ctx, g, _gm, _gv):
        xhat, gamma, inv = ctx.saved_tensors
        m = g.numel() // g.shape[1]
        gd = g.double()
        dbeta = gd.sum(dim=(0, 2, 3))
        dgamma = (gd * xhat.double()).sum(dim=(0, 2, 3))
        dz = (gamma.double() * inv.double())[None, :, None, None] * (
            gd - (dbeta / m)[None, :, None, None] - xhat.double() * (dgamma / m)[None, :, None, None])
        return dz.to(g.dtype), dgamma.to(g.dtype), dbeta.to(g.dtype), None


def bn_apply(z, mean, var, gamma, beta, eps=BN_EPS):
    inv = torch.rsqrt(var + eps)
    return (z - mean[None, :, None, None]) * (inv * gamma)[None, :, None, None] \
        + beta[None, :, None, None]


def bn_forward(z, gamma, beta, running_mean, running_var, nbt, training,
               eps=BN_EPS, momentum=BN_MOMENTUM):
    """nn.BatchNorm2d -- models/unet_parts.py:14,17.

    Training: normalise with this call's batch statistics (biased variance)
    and update the running buffers with the *unbiased* variance, momentum 0.1.
    Eval: normalise with the running buffers.  Returns (y, rm, rv, nbt).
    """
    if training:
        y, mean, var = _BNTrain.apply(z, gamma, beta, eps)
        n = z.numel() // z.shape[1]
        unbiased = var * (n / max(n - 1, 1))
        rm = (1 - momentum) * running_mean + momentum * mean
        rv = (1 - momentum) * running_var + momentum * unbiased
        return y, rm, rv, nbt + 1
    return bn_apply(z, running_mean, running_var, gamma, beta, eps), running_mean, running_var, nbt


def maxpool2(x):
    """nn.MaxPool2d(2) -- models/unet_parts.py:40.  Floor mode; the first
    maximum in row-major window order wins ties (gradient routing)."""
    n, c, h, w = x.shape
    ho, wo = h // 2, w // 2
    xw = x[:, :, :2 * ho, :2 * wo].reshape(n, c, ho, 2, wo, 2).permute(0, 1, 2, 4, 3, 5)
    xw = xw.reshape(n, c, ho, wo, 4)
    # torch.max returns the first index among ties on CPU
    return xw.max(dim=-1).values


def upsample2x_align(x):
    """nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) --
    models/unet_parts.py:56-58.  src = dst * (in - 1) / (out - 1)."""
    n, c, h, w = x.shape
    ho, wo = 2 * h, 2 * w

    def taps(n_in, n_out):
        scale = (n_in - 1) / (n_out - 1) if n_out > 1 else 0.0
        src = torch.arange(n_out, dtype=torch.float64) * scale
        i0 = torch.clamp(src.floor().long(), 0, n_in - 1)
        i1 = torch.clamp(i0 + 1, max=n_in - 1)
        lam = (src - i0.to(torch.float64)).to(x.dtype)
        return i0, i1, lam

    y0, y1, ly = taps(h, ho)
    x0, x1, lx = taps(w, wo)
    top = x[:, :, y0, :] * (1 - ly)[None, None, :, None] + x[:, :, y1, :] * ly[None, None, :, None]
    out = top[:, :, :, x0] * (1 - lx)[None, None, None, :] + top[:, :, :, x1] * lx[None, None, None, :]
    return out


def pad_to(x1, x2):
    """F.pad of the upsampled map to the skip's size -- models/unet_parts.py:68-72
    (left = diff // 2, right = diff - diff // 2, same for top/bottom)."""
    dy = x2.shape[2] - x1.shape[2]
    dx = x2.shape[3] - x1.shape[3]
    return F.pad(x1, (dx // 2, dx - dx // 2, dy // 2, dy - dy // 2))


# --------------------------------------------------------------------------
# stages
# --------------------------------------------------------------------------
class State:
    """Flat name->tensor view of the reference state-dict."""

    def __init__(self, sd):
        self.sd = {(k[len('module.'):] if k.startswith('module.') else k): v for k, v in sd.items()}
        self.new_buffers = {}

    def p(self, key):
        return self.sd[key]

    def buf(self, key):
        return self.new_buffers.get(key, self.sd[key])

    def set_buf(self, key, v):
        self.new_buffers[key] = v


def double_conv(st, prefix, x, training):
    """(conv => BN => ReLU) * 2 -- models/unet_parts.py:8-23.
    `prefix` is the nn.Sequential's state-dict prefix, e.g. 'inc.conv.conv'."""
    for ci, bi in ((0, 1), (3, 4)):
        z = conv3x3(x, st.p(f'{prefix}.{ci}.weight'), st.p(f'{prefix}.{ci}.bias'))
        y, rm, rv, nbt = bn_forward(z, st.p(f'{prefix}.{bi}.weight'), st.p(f'{prefix}.{bi}.bias'),
                                    st.buf(f'{prefix}.{bi}.running_mean'),
                                    st.buf(f'{prefix}.{bi}.running_var'),
                                    st.buf(f'{prefix}.{bi}.num_batches_tracked'), training)
        st.set_buf(f'{prefix}.{bi}.running_mean', rm)
        st.set_buf(f'{prefix}.{bi}.running_var', rv)
        st.set_buf(f'{prefix}.{bi}.num_batches_tracked', nbt)
        x = torch.relu(y)
    return x


def encoder(st, x, training):
    """inc, down1..down4 on one date -- models/bidate_model.py:23-27 / 29-33,
    models/unet_parts.py:26-46."""
    x1 = double_conv(st, 'inc.conv.conv', x, training)
    feats = [x1]
    for k in range(1, 5):
        feats.append(double_conv(st, f'down{k}.mpconv.1.conv', maxpool2(feats[-1]), training))
    return feats


def up_stage(st, name, x1, x2, training):
    """up.forward -- models/unet_parts.py:64-80: upsample x1, pad to x2,
    cat([x2, x1]) (skip first), double_conv."""
    x1 = pad_to(upsample2x_align(x1), x2)
    return double_conv(st, f'{name}.conv.conv', torch.cat([x2, x1], dim=1), training)


def bidate_forward(sd, x_d1, x_d2, training=True):
    """BiDateNet.forward -- models/bidate_model.py:22-40.

    The shared encoder (and its BatchNorm modules) runs on date 1 and then on
    date 2, so batch statistics are per date and the running buffers are
    updated twice, d1 then d2.  Returns (logits, new_buffers)."""
    st = State(sd)
    f1 = encoder(st, x_d1, training)
    f2 = encoder(st, x_d2, training)
    fused = [torch.relu(b * a) for a, b in zip(f1, f2)]      # bidate_model.py:35-38
    x = up_stage(st, 'up1', fused[4], fused[3], training)
    x = up_stage(st, 'up2', x, fused[2], training)
    x = up_stage(st, 'up3', x, fused[1], training)
    x = up_stage(st, 'up4', x, fused[0], training)
    logits = conv1x1(x, st.p('outc.conv.weight'), st.p('outc.conv.bias'))   # :39
    return logits, st.new_buffers


# --------------------------------------------------------------------------
# losses -- utils/metrics.py
# --------------------------------------------------------------------------
def _probas_onehot(logits, true):
    """Shared head of dice/jaccard/tversky for num_classes > 1
    (utils/metrics.py:76-79, 111-114, 159-163)."""
    nc = logits.shape[1]
    one_hot = torch.eye(nc)[true.squeeze(1) if true.dim() == 4 else true]
    one_hot = one_hot.permute(0, 3, 1, 2).to(logits.dtype)
    probas = torch.softmax(logits, dim=1)
    dims = (0,) + tuple(range(2, true.dim()))      # metrics.py:164 -- (0,2) for [B,H,W] labels!
    return probas, one_hot, dims


def tversky_loss(logits, true, alpha=0.5, beta=0.5, eps=1e-7):
    """TverskyLoss.forward -- utils/metrics.py:130-171.  With the [B,H,W]
    labels train.py:85 feeds, `dims` is (0,2): the sums run over batch and H
    only, leaving a [C,W] ratio map that is then averaged."""
    probas, one_hot, dims = _probas_onehot(logits, true)
    inter = torch.sum(probas * one_hot, dims)
    fps = torch.sum(probas * (1 - one_hot), dims)
    fns = torch.sum((1 - probas) * one_hot, dims)
    return 1 - (inter / (inter + alpha * fps + beta * fns + eps)).mean()


def dice_loss(logits, true, eps=1e-7):
    """utils/metrics.py:51-83."""
    probas, one_hot, dims = _probas_onehot(logits, true)
    inter = torch.sum(probas * one_hot, dims)
    card = torch.sum(probas + one_hot, dims)
    return 1 - (2. * inter / (card + eps)).mean()


def jaccard_loss(logits, true, eps=1e-7):
    """utils/metrics.py:86-119."""
    probas, one_hot, dims = _probas_onehot(logits, true)
    inter = torch.sum(probas * one_hot, dims)
    card = torch.sum(probas + one_hot, dims)
    return 1 - (inter / (card - inter + eps)).mean()


def focal_loss(logits, true, gamma=0.0, alpha=None, size_average=True):
    """FocalLoss.forward -- utils/metrics.py:8-48.  The modulating factor (1-pt)**gamma is built from
    `logpt.data.exp()` (:35), i.e. it is a CONSTANT for autograd: d loss / d logits has no gamma*(1-pt)**(gamma-1)
    term.  alpha: None, a float a (-> class weights [a, 1-a], :13-14) or a list of class weights (:15-16)."""
    nc = logits.shape[1]
    x = logits.reshape(logits.shape[0], nc, -1).transpose(1, 2).reshape(-1, nc)     # :20-28
    t = true.reshape(-1, 1).long()
    logpt = torch.log_softmax(x, dim=1).gather(1, t).view(-1)                       # :32-34
    pt = logpt.detach().exp()
    if alpha is not None:
        a = torch.tensor([alpha, 1 - alpha]) if isinstance(alpha, (float, int)) else torch.tensor(alpha)
        logpt = logpt * a.to(x.dtype).gather(0, t.view(-1))                         # :37-41
    loss = -1 * (1 - pt) ** gamma * logpt
    return loss.mean() if size_average else loss.sum()


# --------------------------------------------------------------------------
# train step -- train.py:83-101 with optim.SGD(lr) from train.py:55
# --------------------------------------------------------------------------
def train_step(sd, x_d1, x_d2, labels, lr=1e-3, alpha=0.1, beta=0.9):
    """One reference training step on a *copy* of `sd`.

    Returns dict(logits, loss, grads{name}, new_sd, preds).  Parameters are
    every floating entry that is not a BatchNorm buffer."""
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()
              if v.is_floating_point() and 'running_' not in k}
    full = dict(sd)
    full.update(params)
    logits, new_buf = bidate_forward(full, x_d1, x_d2, training=True)
    loss = tversky_loss(logits, labels.long(), alpha, beta)
    names = list(params)
    grads = torch.autograd.grad(loss, [params[k] for k in names])
    new_sd = {k: v.clone() for k, v in sd.items()}
    for k, g in zip(names, grads):
        new_sd[k] = (sd[k] - lr * g).detach()             # SGD, no momentum / wd
    for k, v in new_buf.items():
        new_sd[k] = v.detach()
    preds = torch.max(logits, 1)[1]                       # train.py:96
    return dict(logits=logits.detach(), loss=loss.detach(),
                grads=dict(zip(names, [g.detach() for g in grads])),
                new_sd=new_sd, preds=preds)


# --------------------------------------------------------------------------
# batch metrics -- train.py:98-106, utils/helpers.py:45-89
# --------------------------------------------------------------------------
def binary_prf(labels, preds):
    """sklearn precision_recall_fscore_support(average='binary', pos_label=1)
    with its zero-division -> 0 default, as called at train.py:103-106."""
    labels = labels.reshape(-1).long()
    preds = preds.reshape(-1).long()
    tp = int(((preds == 1) & (labels == 1)).sum())
    fp = int(((preds == 1) & (labels == 0)).sum())
    fn = int(((preds == 0) & (labels == 1)).sum())
    p = tp / (tp + fp) if tp + fp else 0.0
    r = tp / (tp + fn) if tp + fn else 0.0
    f = 2 * p * r / (p + r) if p + r else 0.0
    return p, r, f


def accuracy_percent(labels, preds, patch_size):
    """train.py:98-100 (true division under torch >= 1.6)."""
    return 100.0 * float((preds.byte() == labels.byte()).sum()) / (labels.shape[0] * patch_size ** 2)


# --------------------------------------------------------------------------
# full-scene tiling -- utils/inference.py:134-236
# --------------------------------------------------------------------------
def tile_scene(bands, p):
    """_get_patches: non-overlapping p x p tiles, then the last column strip,
    the last row strip and one corner tile, all anchored to the far edges.
    `bands` is [H,W,C].  Returns (tiles[N,p,p,C], hs, ws, lc, lr, H, W)."""
    import numpy as np
    h, w, _ = bands.shape
    hs, ws = h // p, w // p
    main = [bands[i * p:(i + 1) * p, j * p:(j + 1) * p] for i in range(hs) for j in range(ws)]
    col = [bands[i * p:(i + 1) * p, w - p:] for i in range(hs)]
    row = [bands[h - p:, j * p:(j + 1) * p] for j in range(ws)]
    corner = [bands[h - p:, w - p:]]
    return np.stack(main + col + row + corner), hs, ws, len(col), len(row), h, w


def stitch_scene(tiles, hs, ws, lc, lr, h, w, p):
    """_get_bands: inverse of tile_scene for [N,p,p] prediction tiles."""
    import numpy as np
    img = np.zeros((h, w))
    k = 0
    for i in range(hs):
        for j in range(ws):
            img[i * p:(i + 1) * p, j * p:(j + 1) * p] = tiles[k]
            k += 1
    for i in range(lc):
        img[i * p:(i + 1) * p, w - p:] = tiles[k]
        k += 1
    for i in range(lr):
        img[h - p:, i * p:(i + 1) * p] = tiles[k]
        k += 1
    img[h - p:, w - p:] = tiles[k]
    return img


# --------------------------------------------------------------------------
# stock-module assembly, used only to time the CPU baseline (bench.py)
# --------------------------------------------------------------------------
def build_torch_baseline(n_channels, n_classes):
    """The same graph as models/bidate_model.py + models/unet_parts.py built
    from stock torch.nn modules, for timing the reference CPU path on the GPU
    box's host cores (the reference files never travel).  State-dict keys are
    the reference's."""
    import torch.nn as nn

    class DoubleConv(nn.Module):
        def __init__(self, ci, co):
            super().__init__()
            self.conv = nn.Sequential(nn.Conv2d(ci, co, 3, padding=1), nn.BatchNorm2d(co), nn.ReLU(inplace=True),
                                      nn.Conv2d(co, co, 3, padding=1), nn.BatchNorm2d(co), nn.ReLU(inplace=True))

        def forward(self, x):
            return self.conv(x)

    class InConv(nn.Module):
        def __init__(self, ci, co):
            super().__init__()
            self.conv = DoubleConv(ci, co)

        def forward(self, x):
            return self.conv(x)

    class Down(nn.Module):
        def __init__(self, ci, co):
            super().__init__()
            self.mpconv = nn.Sequential(nn.MaxPool2d(2), DoubleConv(ci, co))

        def forward(self, x):
            return self.mpconv(x)

    class Up(nn.Module):
        def __init__(self, ci, co):
            super().__init__()
            self.up = nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)
            self.conv = DoubleConv(ci, co)

        def forward(self, x1, x2):
            x1 = pad_to(self.up(x1), x2)
            return self.conv(torch.cat([x2, x1], dim=1))

    class OutConv(nn.Module):
        def __init__(self, ci, co):
            super().__init__()
            self.conv = nn.Conv2d(ci, co, 1)

        def forward(self, x):
            return self.conv(x)

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.inc = InConv(n_channels, 64)
            self.down1, self.down2 = Down(64, 128), Down(128, 256)
            self.down3, self.down4 = Down(256, 512), Down(512, 512)
            self.up1, self.up2 = Up(1024, 256), Up(512, 128)
            self.up3, self.up4 = Up(256, 64), Up(128, 64)
            self.outc = OutConv(64, n_classes)

        def forward(self, a, b):
            fa = [self.inc(a)]
            for d in (self.down1, self.down2, self.down3, self.down4):
                fa.append(d(fa[-1]))
            fb = [self.inc(b)]
            for d in (self.down1, self.down2, self.down3, self.down4):
                fb.append(d(fb[-1]))
            f = [torch.relu(q * p) for p, q in zip(fa, fb)]
            x = self.up1(f[4], f[3])
            x = self.up2(x, f[2])
            x = self.up3(x, f[1])
            x = self.up4(x, f[0])
            return self.outc(x)

    return Net()
