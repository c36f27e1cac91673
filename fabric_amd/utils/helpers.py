"""Glue with the reference's names and semantics (utils/helpers.py:24-89, 288-337) for the parts the
train step needs: running-mean metric dicts, criterion factory, model factory."""
import numpy as np
import torch

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


def get_loaders(opt):
    """reference utils/helpers.py:211-258, same signature and return value: (train_loader, val_loader) over OneraPreloader datasets
    built from `opt.dataset_dir` (OSCD layout), `opt.validation_cities`, `opt.patch_size`, `opt.stride`, `opt.augmentation`,
    `opt.batch_size`, `opt.num_workers` (+ `opt.band_ids / band_means / band_stds` for the ingest).  One process, no sharding: this is
    the reference's loader for the reference's loop; the data-parallel loop of fabric_amd.train uses make_loaders (per-rank shards)."""
    from . import ingest
    from .dataloaders import OneraPreloader
    train_samples, val_samples = ingest.get_train_val_metadata(opt.dataset_dir, opt.validation_cities, opt.patch_size, opt.stride)
    print('train samples : ', len(train_samples))
    print('val samples : ', len(val_samples))
    full_load = ingest.full_onera_loader(opt.dataset_dir, opt)
    train_dataset = OneraPreloader(opt.dataset_dir, train_samples, full_load, opt.patch_size, opt.augmentation)
    val_dataset = OneraPreloader(opt.dataset_dir, val_samples, full_load, opt.patch_size, False)
    train_loader = torch.utils.data.DataLoader(train_dataset, batch_size=opt.batch_size, shuffle=True, num_workers=opt.num_workers)
    val_loader = torch.utils.data.DataLoader(val_dataset, batch_size=opt.batch_size, shuffle=False, num_workers=opt.num_workers)
    return train_loader, val_loader


def load_model(opt, device, precision=None):
    """reference utils/helpers.py:317-337 builds nn.DataParallel(BiDateNet(13, 2)); here one process drives one
    GPU and gradients are exchanged by fabric_amd.parallel (RCCL), so the bare module is returned."""
    return BiDateNet(13, 2, precision=precision).to(device)


def strip_module_prefix(state_dict):
    """State dict saved from the reference's nn.DataParallel wrapper (utils/helpers.py:335): every key carries `module.`.
    Returns a dict with the prefix removed from the keys that have it (a bare-module state dict passes through unchanged)."""
    return {(k[len('module.'):] if k.startswith('module.') else k): v for k, v in state_dict.items()}


def load_checkpoint(src, device=None, precision=None, allow_pickle=True):
    """A usable BiDateNet from anything the reference (or fabric_amd.train) leaves behind.

    `src` is a path / file object for torch.load, or an object already loaded:
      * the whole-module pickle of train.py:222 -- nn.DataParallel(BiDateNet) or a bare BiDateNet (the class paths
        `models.bidate_model.BiDateNet`, `models.unet_parts.*` resolve through the repo-root shims; run with the repo root on
        sys.path).  The DataParallel wrapper is dropped: here one process drives one GPU (fabric_amd.parallel);
      * a state dict, with or without the `module.` prefix of a DataParallel save, optionally nested under 'state_dict' / 'model'.
    The result is always a fresh fabric_amd BiDateNet carrying the checkpoint's parameters AND BatchNorm buffers; the channel
    counts come from the tensors.  Mismatched / missing / unexpected keys raise (load_state_dict(strict=True)).

    Trust: a file is first read with torch.load(weights_only=True) -- tensors and plain containers only, no code runs; that covers
    every state-dict form (incl. the `*.state_dict.pt` fabric_amd.train writes for interchange).  Only when that loader refuses the
    file (a whole-module pickle, train.py:222) is it unpickled in full, which EXECUTES whatever the pickle holds: pass
    allow_pickle=False to refuse such files when the checkpoint does not come from a trusted source."""
    obj = src
    if isinstance(src, (str, bytes)) or hasattr(src, 'read') or hasattr(src, '__fspath__'):
        import pickle
        pos = src.tell() if hasattr(src, 'tell') else None
        try:
            obj = torch.load(src, map_location='cpu', weights_only=True)
        except (pickle.UnpicklingError, RuntimeError, AttributeError) as e:
            # a whole-module pickle needs the unpickler, not the tensors-only loader
            if not allow_pickle:
                raise RuntimeError('fabric_amd: this checkpoint is a whole-module pickle (train.py:222); loading it executes the pickle -- '
                                   'pass allow_pickle=True for files from a trusted source') from e
            if pos is not None:
                src.seek(pos)
            obj = torch.load(src, map_location='cpu', weights_only=False)
    if isinstance(obj, torch.nn.DataParallel) or (isinstance(obj, torch.nn.Module) and hasattr(obj, 'module')
                                                  and not isinstance(obj, BiDateNet)):
        obj = obj.module
    if isinstance(obj, torch.nn.Module):
        sd = obj.state_dict()
    elif isinstance(obj, dict):
        sd = obj
        for nest in ('state_dict', 'model'):
            if nest in sd and isinstance(sd[nest], dict):
                sd = sd[nest]
            elif nest in sd and isinstance(sd[nest], torch.nn.Module):
                return load_checkpoint(sd[nest], device, precision, allow_pickle)
    else:
        raise TypeError(f'fabric_amd: cannot load a checkpoint from {type(obj).__name__}')
    sd = strip_module_prefix(sd)
    try:
        n_channels = int(sd['inc.conv.conv.0.weight'].shape[1])
        n_classes = int(sd['outc.conv.weight'].shape[0])
    except KeyError as e:
        raise RuntimeError(f'fabric_amd: not a BiDateNet checkpoint (missing {e.args[0]!r})') from None
    model = BiDateNet(n_channels, n_classes, precision=precision)
    model.load_state_dict(sd, strict=True)
    return model.to(device) if device is not None else model
