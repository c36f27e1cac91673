"""Glue with the reference's names and semantics (utils/helpers.py:24-89, 288-337) for the parts the
train step needs: running-mean metric dicts, criterion factory, model factory."""
import numpy as np

from ..models.bidate_model import BiDateNet
from .metrics import FocalLoss, TverskyLoss, dice_loss, jaccard_loss


def initialize_metrics():
    """reference utils/helpers.py:24-42"""
    return {'cd_losses': [], 'cd_corrects': [], 'cd_precisions': [], 'cd_recalls': [], 'cd_f1scores': []}


def get_mean_metrics(metric_dict):
    """reference utils/helpers.py:45-59: arithmetic mean of the per-batch values (val F1 = mean of per-batch F1)."""
    return {k: np.mean(v) for k, v in metric_dict.items()}


def set_metrics(metric_dict, cd_loss, cd_corrects, cd_report):
    """reference utils/helpers.py:62-89"""
    metric_dict['cd_losses'].append(float(cd_loss))
    metric_dict['cd_corrects'].append(float(cd_corrects))
    metric_dict['cd_precisions'].append(cd_report[0])
    metric_dict['cd_recalls'].append(cd_report[1])
    metric_dict['cd_f1scores'].append(cd_report[2])
    return metric_dict


def get_criterion(opt):
    """reference utils/helpers.py:288-314.  `focal` reads opt.focal_gamma exactly like the reference (absent from
    metadata.json there, so it raises AttributeError unless the caller adds it); `bce` cannot run on BiDateNet's
    [B,2,H,W] logits with [B,H,W] labels in the reference either (shape mismatch inside BCEWithLogitsLoss)."""
    if opt.loss_function == 'focal':
        return FocalLoss(opt.focal_gamma)
    if opt.loss_function == 'dice':
        return dice_loss
    if opt.loss_function == 'jaccard':
        return jaccard_loss
    if opt.loss_function == 'tversky':
        return TverskyLoss(alpha=opt.tversky_alpha, beta=opt.tversky_beta)
    raise NotImplementedError(f'fabric_amd: loss_function={opt.loss_function!r}: the reference\'s BCEWithLogitsLoss '
                              f'branch fails on [B,2,H,W] logits vs [B,H,W] labels and is not built')


def load_model(opt, device, precision=None):
    """reference utils/helpers.py:317-337 builds nn.DataParallel(BiDateNet(13, 2)); here one process drives one
    GPU and gradients are exchanged by fabric_amd.parallel (RCCL), so the bare module is returned."""
    return BiDateNet(13, 2, precision=precision).to(device)
