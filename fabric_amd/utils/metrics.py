"""Losses and batch metrics: drop-in for the reference's utils/metrics.py.

TverskyLoss (the default criterion, metadata.json:42-44) is a fused HIP kernel
(softmax + the reference's TP/FP/FN sums + loss + d loss / d logits + argmax
TP/FP/FN counts for F1) behind ``bdn_overlap_loss``; dice_loss and jaccard_loss
are the same kernel with other coefficients, FocalLoss is ``bdn_focal``.  Both
label ranks of the reference are supported ([B,H,W] -> dims (0,2), [B,1,H,W] ->
dims (0,2,3)).  The sigmoid single-class branch (utils/metrics.py:65-72,
100-107, 149-157) is never reached by BiDateNet(13, 2) and is not built.
"""
import torch
import torch.nn as nn

from .. import _lib


def _check_labels(logits, labels):
    """-> (uint8 labels, reduce_w).  [B,H,W] labels make the reference reduce over dims (0,2) only; [B,1,H,W]
    labels over (0,2,3) (utils/metrics.py:80,115,164) -- two different loss values, both supported."""
    B, C, H, W = logits.shape
    lb = labels.detach()
    if lb.dim() == 4 and lb.shape == (B, 1, H, W):
        reduce_w = 1
    elif lb.shape == (B, H, W):
        reduce_w = 0
    else:
        raise RuntimeError(f'labels must be [B,H,W] or [B,1,H,W] for logits {tuple(logits.shape)}, got {tuple(lb.shape)}')
    return lb.to(torch.uint8).contiguous(), reduce_w


class _OverlapFunction(torch.autograd.Function):
    """TP / (TP + alpha FP + beta FN + eps) family: Tversky, Dice (0.5, 0.5, eps/2), Jaccard (1, 1, eps)."""

    @staticmethod
    def forward(ctx, logits, labels, alpha, beta, eps, holder):
        if not logits.is_cuda:
            raise RuntimeError('fabric_amd: the losses run only on a ROCm device -- there is no CPU path')
        B, C, H, W = logits.shape
        lg = logits.detach().contiguous().float()
        lb, reduce_w = _check_labels(logits, labels)
        ws = torch.empty(_lib.load().bdn_overlap_workspace_bytes(B, C, H, W, reduce_w) // 4, dtype=torch.float32, device=lg.device)
        loss = torch.empty((), dtype=torch.float32, device=lg.device)
        counts = torch.empty(4, dtype=torch.int32, device=lg.device)
        dl = torch.empty_like(lg)
        _lib.call('bdn_overlap_loss', lg.data_ptr(), lb.data_ptr(), float(alpha), float(beta), float(eps), reduce_w,
                  ws.data_ptr(), loss.data_ptr(), counts.data_ptr(), dl.data_ptr(), B, C, H, W, _lib.stream_ptr())
        ctx.save_for_backward(dl)
        if holder is not None:
            holder['counts'] = counts
        return loss

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None, None, None, None, None


_TverskyFunction = _OverlapFunction


def dice_loss(logits, true, eps=1e-7):
    """reference utils/metrics.py:51-83: 1 - mean(2I / (sum(p) + sum(t) + eps)) = Tversky(0.5, 0.5, eps/2)."""
    return _OverlapFunction.apply(logits, true, 0.5, 0.5, 0.5 * eps, None)


def jaccard_loss(logits, true, eps=1e-7):
    """reference utils/metrics.py:86-119: 1 - mean(I / (sum(p) + sum(t) - I + eps)) = Tversky(1, 1, eps)."""
    return _OverlapFunction.apply(logits, true, 1.0, 1.0, eps, None)


class _FocalFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, gamma, alpha, size_average, holder):
        if not logits.is_cuda:
            raise RuntimeError('fabric_amd: the losses run only on a ROCm device -- there is no CPU path')
        if logits.dim() != 4:
            raise RuntimeError('fabric_amd: FocalLoss is built for [B,C,H,W] logits (train.py:91)')
        B, C, H, W = logits.shape
        lg = logits.detach().contiguous().float()
        if target.numel() != B * H * W:
            raise RuntimeError(f'target must hold B*H*W={B * H * W} class indices, got {tuple(target.shape)}')
        lb = target.detach().reshape(B, H, W).to(torch.uint8).contiguous()
        ws = torch.empty(_lib.load().bdn_focal_workspace_bytes(), dtype=torch.uint8, device=lg.device)
        loss = torch.empty((), dtype=torch.float32, device=lg.device)
        counts = torch.empty(4, dtype=torch.int32, device=lg.device)
        dl = torch.empty_like(lg)
        a = alpha.to(device=lg.device, dtype=torch.float32).contiguous() if alpha is not None else None
        if a is not None and a.numel() < C:
            raise RuntimeError(f'alpha holds {a.numel()} class weights for {C} classes')
        _lib.call('bdn_focal', lg.data_ptr(), lb.data_ptr(), float(gamma), a.data_ptr() if a is not None else None,
                  1 if size_average else 0, ws.data_ptr(), loss.data_ptr(), counts.data_ptr(), dl.data_ptr(),
                  B, C, H, W, _lib.stream_ptr())
        ctx.save_for_backward(dl)
        holder['counts'] = counts
        return loss

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None, None, None, None, None


class FocalLoss(nn.Module):
    """reference utils/metrics.py:8-48 (same constructor; the modulating factor is detached like there)."""

    def __init__(self, gamma=0, alpha=None, size_average=True):
        super(FocalLoss, self).__init__()
        self.gamma = gamma
        self.alpha = alpha
        if isinstance(alpha, (float, int)):
            self.alpha = torch.Tensor([alpha, 1 - alpha])
        if isinstance(alpha, list):
            self.alpha = torch.Tensor(alpha)
        self.size_average = size_average
        self._holder = {}

    def forward(self, input, target):
        return _FocalFunction.apply(input, target, self.gamma, self.alpha, self.size_average, self._holder)

    @property
    def last_counts(self):
        return self._holder.get('counts')


class TverskyLoss(nn.Module):
    """reference utils/metrics.py:122-171"""

    def __init__(self, alpha=0.5, beta=0.5, eps=1e-7, size_average=True):
        super(TverskyLoss, self).__init__()
        self.alpha = alpha
        self.beta = beta
        self.size_average = size_average
        self.eps = eps
        self._holder = {}

    def forward(self, logits, true):
        return _TverskyFunction.apply(logits, true, self.alpha, self.beta, self.eps, self._holder)

    @property
    def last_counts(self):
        """int32[4] device tensor {TP, FP, FN, correct} of argmax(logits) vs labels for the last call."""
        return self._holder.get('counts')


def confusion_counts(logits, labels):
    """int32[4] device tensor {TP, FP, FN, correct} of argmax(logits, 1) vs labels (what train.py:96-106 hands to sklearn),
    for criteria that do not report it themselves: one pass of the overlap-loss kernel, loss and gradient discarded."""
    holder = {}
    with torch.no_grad():
        _OverlapFunction.apply(logits.detach(), labels, 0.5, 0.5, 1e-7, holder)
    return holder['counts']


def batch_prf_from_counts(counts):
    """sklearn precision_recall_fscore_support(average='binary', pos_label=1) as called at reference
    train.py:103-106, from on-device counts; zero division -> 0 like sklearn's default."""
    tp, fp, fn = [int(v) for v in counts[:3].tolist()]
    p = tp / (tp + fp) if tp + fp else 0.0
    r = tp / (tp + fn) if tp + fn else 0.0
    f = 2 * p * r / (p + r) if p + r else 0.0
    return p, r, f
