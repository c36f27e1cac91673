"""Losses and batch metrics: drop-in for the reference's utils/metrics.py.

TverskyLoss (the default criterion, metadata.json:42-44) is a fused HIP kernel
(softmax + the reference's (0,2)-dims TP/FP/FN sums + loss + d loss / d logits +
argmax TP/FP/FN counts for F1) behind ``bdn_tversky``.  The sigmoid single-class
branch of the reference (utils/metrics.py:149-157) is never reached by
BiDateNet(13, 2) and is not built.
"""
import torch
import torch.nn as nn

from .. import _lib


class _TverskyFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, alpha, beta, eps, holder):
        if not logits.is_cuda:
            raise RuntimeError('fabric_amd: TverskyLoss runs only on a ROCm device -- there is no CPU path')
        B, C, H, W = logits.shape
        lg = logits.detach().contiguous().float()
        lb = labels.detach()
        if lb.dim() == 4 and lb.shape[1] == 1:
            # [B,1,H,W] labels reduce over (0,2,3) in the reference (utils/metrics.py:164): different value.
            raise RuntimeError('fabric_amd: TverskyLoss is built for the [B,H,W] labels train.py:85 feeds '
                               '(reference dims == (0,2)); got [B,1,H,W]')
        if lb.shape != (B, H, W):
            raise RuntimeError(f'labels must be [B,H,W]={B, H, W}, got {tuple(lb.shape)}')
        lb = lb.to(torch.uint8).contiguous()
        ws = torch.empty(3 * C * W + 8, dtype=torch.float32, device=lg.device)
        loss = torch.empty((), dtype=torch.float32, device=lg.device)
        counts = torch.empty(4, dtype=torch.int32, device=lg.device)
        dl = torch.empty_like(lg)
        _lib.call('bdn_tversky', lg.data_ptr(), lb.data_ptr(), float(alpha), float(beta), float(eps),
                  ws.data_ptr(), loss.data_ptr(), counts.data_ptr(), dl.data_ptr(), B, C, H, W, _lib.stream_ptr())
        ctx.save_for_backward(dl)
        holder['counts'] = counts
        return loss

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None, None, None, None, None


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


def batch_prf_from_counts(counts):
    """sklearn precision_recall_fscore_support(average='binary', pos_label=1) as called at reference
    train.py:103-106, from on-device counts; zero division -> 0 like sklearn's default."""
    tp, fp, fn = [int(v) for v in counts[:3].tolist()]
    p = tp / (tp + fp) if tp + fp else 0.0
    r = tp / (tp + fn) if tp + fn else 0.0
    f = 2 * p * r / (p + r) if p + r else 0.0
    return p, r, f
