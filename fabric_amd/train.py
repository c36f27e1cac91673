"""The reference's train.py loop (train.py:65-172) on the HIP path: epochs of fused train steps followed by a
validation pass that reports the reference's metrics -- per-batch accuracy and sklearn-style binary
precision / recall / F1, averaged over batches (utils/helpers.py:45-59).  Tracking SaaS clients (comet,
polyaxon), the GCS download and checkpoint upload of the reference are out of scope.

    python -m fabric_amd.train --synthetic --epochs 1          # needs an MI355X
"""
import argparse
import json
import os

import torch
import torch.utils.data

from .models.bidate_model import BiDateNet
from .train_step import TrainStep
from .utils.dataloaders import OneraPreloader, metadata_from_shapes, synthetic_onera
from .utils.helpers import get_mean_metrics, initialize_metrics, set_metrics
from .utils.metrics import TverskyLoss, batch_prf_from_counts

DEFAULTS = dict(patch_size=90, stride=180, augmentation=True, num_workers=2, epochs=1, batch_size=32,
                learning_rate=1e-3, loss_function='tversky', tversky_alpha=0.1, tversky_beta=0.9,
                validation_cities=['cupertino', 'rennes'])          # reference metadata.json:32-48


def make_loaders(full_load, val_cities, patch_size, stride, batch_size, augmentation, num_workers=0,
                 rank=0, world_size=1):
    """utils/helpers.py:211-258 (get_loaders) given an already loaded dataset dict; the training index list is
    sharded stride-by-rank for data-parallel runs."""
    from .parallel import shard_indices
    shapes = {c: d['labels'].shape for c, d in full_load.items()}
    train_meta, val_meta = metadata_from_shapes(shapes, val_cities, patch_size, stride)
    train_ds = OneraPreloader('', train_meta, full_load, patch_size, augmentation)
    val_ds = OneraPreloader('', val_meta, full_load, patch_size, False)
    if world_size > 1:
        train_ds = torch.utils.data.Subset(train_ds, shard_indices(len(train_ds), rank, world_size))
    kw = dict(batch_size=batch_size, num_workers=num_workers)
    return (torch.utils.data.DataLoader(train_ds, shuffle=True, drop_last=True, **kw),
            torch.utils.data.DataLoader(val_ds, shuffle=False, **kw))


def train_epoch(step, loader, dev, patch_size):
    """train.py:73-118 without the per-step host round trip: losses / counts are read back once per epoch."""
    step.model.train()
    recs = []
    for b1, b2, labels in loader:
        loss = step.step(b1.to(dev, non_blocking=True), b2.to(dev, non_blocking=True), labels.to(dev, non_blocking=True))
        recs.append((loss.clone(), step.last_counts.clone(), labels.shape[0]))
    metrics = initialize_metrics()
    for loss, counts, n in recs:
        c = counts.cpu()
        metrics = set_metrics(metrics, loss.item(), 100.0 * int(c[3]) / (n * patch_size ** 2), batch_prf_from_counts(c))
    return get_mean_metrics(metrics) if recs else {}


@torch.no_grad()
def validate(model, loader, dev, patch_size, alpha, beta):
    """train.py:125-172: eval-mode forward, Tversky loss, per-batch accuracy / P / R / F1, mean over batches."""
    model.eval()
    crit = TverskyLoss(alpha=alpha, beta=beta)
    metrics = initialize_metrics()
    for b1, b2, labels in loader:
        labels = labels.to(dev)
        logits = model(b1.to(dev), b2.to(dev))
        loss = crit(logits, labels.long())
        c = crit.last_counts.cpu()
        metrics = set_metrics(metrics, loss.item(), 100.0 * int(c[3]) / (labels.shape[0] * patch_size ** 2),
                              batch_prf_from_counts(c))
    return get_mean_metrics(metrics)


def main(argv=None):
    ap = argparse.ArgumentParser(description='Training change detection network (HIP path)')
    for k, v in DEFAULTS.items():
        if isinstance(v, bool):
            ap.add_argument(f'--{k}', type=lambda s: s.lower() in ('1', 'true', 'yes'), default=v)
        elif isinstance(v, list):
            ap.add_argument(f'--{k}', nargs='*', default=v)
        else:
            ap.add_argument(f'--{k}', type=type(v), default=v)
    ap.add_argument('--synthetic', action='store_true', help='use fabric_amd.utils.dataloaders.synthetic_onera()')
    ap.add_argument('--precision', default='bf16', choices=['bf16', 'fp32'])
    opt = ap.parse_args(argv)
    if not opt.synthetic:
        raise SystemExit('only --synthetic data is available here (GeoTIFF ingest needs rasterio/cv2; SURVEY.md 8f n3)')
    dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')))
    data = synthetic_onera(n_cities=6, bands=13, size=(360, 360))
    val_cities = ['city4', 'city5']
    train_loader, val_loader = make_loaders(data, val_cities, opt.patch_size, opt.stride // 2, opt.batch_size, opt.augmentation)
    model = BiDateNet(13, 2, precision=opt.precision).to(dev)
    step = TrainStep(model, lr=opt.learning_rate, tversky_alpha=opt.tversky_alpha, tversky_beta=opt.tversky_beta)
    for epoch in range(opt.epochs):
        tr = train_epoch(step, train_loader, dev, opt.patch_size)
        va = validate(model, val_loader, dev, opt.patch_size, opt.tversky_alpha, opt.tversky_beta)
        print(json.dumps({'epoch': epoch, **{'train_' + k: float(v) for k, v in tr.items()},
                          **{'validate_' + k: float(v) for k, v in va.items()}}))


if __name__ == '__main__':
    main()
