"""Glue with the reference's names and semantics (utils/helpers.py:24-89, 288-337) for the parts the
train step needs: running-mean metric dicts, criterion factory, model factory."""
import numpy as np

from ..models.bidate_model import BiDateNet
from .metrics import TverskyLoss


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
    """reference utils/helpers.py:288-314.  Only the default criterion (tversky, metadata.json:41) is on the
    built path; the reference's bce/focal options do not run as shipped."""
    if opt.loss_function == 'tversky':
        return TverskyLoss(alpha=opt.tversky_alpha, beta=opt.tversky_beta)
    raise NotImplementedError(f'fabric_amd: loss_function={opt.loss_function!r} is outside the built hot path '
                              f'(SURVEY.md section 8f, "next")')


def load_model(opt, device, precision=None):
    """reference utils/helpers.py:317-337 builds nn.DataParallel(BiDateNet(13, 2)); here one process drives one
    GPU and gradients are exchanged by fabric_amd.parallel (RCCL), so the bare module is returned."""
    return BiDateNet(13, 2, precision=precision).to(device)
