"""The reference's train.py loop (train.py:65-172) on the HIP path: epochs of fused train steps followed by a
validation pass that reports the reference's metrics -- per-batch accuracy and sklearn-style binary
precision / recall / F1, averaged over batches (utils/helpers.py:45-59).  Tracking SaaS clients (comet,
polyaxon), the GCS download and checkpoint upload of the reference are out of scope.

    python -m fabric_amd.train --synthetic --epochs 1                                   # needs an MI355X
    python -m fabric_amd.train --metadata metadata.json --dataset_dir ./onera/          # an OSCD directory tree

With real data the loop also does what train.py:182-205 does after validation: the full validation scenes are
predicted tile by tile (utils/inference.py) -- here on the device-resident city stacks -- and written as PNG masks.
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
from .utils.metrics import batch_prf_from_counts

DEFAULTS = dict(patch_size=90, stride=180, augmentation=True, num_workers=2, epochs=1, batch_size=32,
                learning_rate=1e-3, loss_function='tversky', tversky_alpha=0.1, tversky_beta=0.9,
                validation_cities=['cupertino', 'rennes'], dataset_dir='./onera/', log_dir='./log/')   # reference metadata.json:32-48


def make_loaders(full_load, val_cities, patch_size, stride, batch_size, augmentation, num_workers=0,
                 rank=0, world_size=1, seed=0):
    """utils/helpers.py:211-258 (get_loaders) given an already loaded dataset dict.  For data-parallel runs the training
    indices are sharded by a ShardSampler: every rank derives the same epoch permutation from (seed, epoch) and takes a
    disjoint stride-by-rank slice of it -- independent of the per-process `random` state that OneraPreloader's in-place
    shuffle (utils/dataloaders.py:171) consumes, and sorted first so that the reference's set-ordered city list
    (utils/dataloaders.py:55) cannot differ between ranks.  Call `train_loader.sampler.set_epoch(e)` every epoch."""
    from .parallel import ShardSampler
    shapes = {c: d['labels'].shape for c, d in full_load.items()}
    train_meta, val_meta = metadata_from_shapes(shapes, val_cities, patch_size, stride)
    train_meta = sorted(train_meta)                        # rank-independent base order; the sampler owns the shuffling
    train_ds = OneraPreloader('', train_meta, full_load, patch_size, augmentation)
    train_ds.imgs.sort()                                   # undo the constructor's process-local shuffle (same list object)
    val_ds = OneraPreloader('', val_meta, full_load, patch_size, False)
    sampler = ShardSampler(len(train_ds), rank, world_size, seed=seed)
    kw = dict(batch_size=batch_size, num_workers=num_workers, pin_memory=True)   # pinned batches: the copy stream DMAs them without staging
    # Augmentation draws (global `random`, utils/dataloaders.py:150-156) must differ between ranks.  With worker processes every worker
    # re-seeds `random` from base_seed + worker_id, and base_seed comes from the loader's generator: torch.manual_seed(seed) makes that
    # identical on every rank, so the loader gets a PER-RANK generator, and each worker additionally folds the rank into its seed.
    gen = torch.Generator()
    gen.manual_seed(seed * 7919 + rank)
    return (torch.utils.data.DataLoader(train_ds, sampler=sampler, drop_last=True, generator=gen,
                                        worker_init_fn=_RankWorkerSeed(seed, rank), **kw),
            torch.utils.data.DataLoader(val_ds, shuffle=False, **kw))


class _RankWorkerSeed:
    """worker_init_fn: Python's `random` of loader worker w on rank r is seeded from (seed, r, w, the worker's torch seed -- which the
    loader advances every epoch)."""

    def __init__(self, seed, rank):
        self.seed, self.rank = int(seed), int(rank)

    def __call__(self, worker_id):
        import random
        random.seed((self.seed * 7919 + self.rank) * 1000003 + worker_id * 65537 + torch.initial_seed() % 65521)


def train_epoch(step, loader, dev, patch_size, feeder=None):
    """train.py:73-118 without the per-step host round trip: losses / counts are read back once per epoch, and the
    host -> device copies of batch k+1 (train.py:83-85) run on a copy stream under the step of batch k."""
    from .input_pipeline import DeviceFeeder
    step.model.train()
    recs = []
    feeder = feeder or DeviceFeeder(dev)
    with torch.cuda.stream(step.stream()):                # the loop lives on the step's own stream: no joins per step
        for b1, b2, labels in feeder(loader):
            loss = step.step(b1, b2, labels)
            recs.append((loss, step.last_counts.clone(), labels.shape[0]))
    torch.cuda.current_stream(dev).wait_stream(step.stream())
    metrics = initialize_metrics()
    for loss, counts, n in recs:
        c = counts.cpu()
        metrics = set_metrics(metrics, loss.item(), 100.0 * int(c[3]) / (n * patch_size ** 2), batch_prf_from_counts(c))
    return get_mean_metrics(metrics) if recs else {}


@torch.no_grad()
def validate(model, loader, dev, patch_size, criterion, feeder=None):
    """train.py:125-172: eval-mode forward, the SAME criterion the run optimises (train.py:137), per-batch accuracy / P / R / F1,
    mean over batches.  Batches arrive through the feeder's copy stream when one is given."""
    from .utils.metrics import confusion_counts
    model.eval()
    metrics = initialize_metrics()
    batches = feeder(loader) if feeder is not None else ((b1.to(dev), b2.to(dev), lb.to(dev)) for b1, b2, lb in loader)
    for b1, b2, labels in batches:
        logits = model(b1, b2)
        loss = criterion(logits, labels.long())
        c = confusion_counts(logits, labels).cpu()
        metrics = set_metrics(metrics, loss.item(), 100.0 * int(c[3]) / (labels.shape[0] * patch_size ** 2),
                              batch_prf_from_counts(c))
    return get_mean_metrics(metrics)


def train_epoch_autograd(model, criterion, optimizer, loader, dev, patch_size, world=1, feeder=None):
    """The reference loop itself (train.py:83-101) for the criteria the fused step does not cover (dice / jaccard / focal):
    autograd through the one-node BiDateNet function, torch.optim.SGD, gradients averaged over the ranks after backward."""
    from .parallel import allreduce_mean_grads
    from .utils.metrics import batch_prf_from_counts, confusion_counts
    model.train()
    metrics = initialize_metrics()
    batches = feeder(loader) if feeder is not None else ((b1.to(dev), b2.to(dev), lb.to(dev)) for b1, b2, lb in loader)
    for b1, b2, labels in batches:
        optimizer.zero_grad()
        logits = model(b1, b2)
        loss = criterion(logits, labels.long())
        loss.backward()
        allreduce_mean_grads(model.parameters(), world)     # a few 16 MB buckets, not 74 blocking per-tensor calls
        optimizer.step()
        model.engine().invalidate_weights()
        c = confusion_counts(logits.detach(), labels).cpu()
        metrics = set_metrics(metrics, loss.item(), 100.0 * int(c[3]) / (labels.shape[0] * patch_size ** 2), batch_prf_from_counts(c))
        del logits, loss                                    # the graph (and its workspace lease) dies before the next forward
    return get_mean_metrics(metrics)


def save_if_better(model, mean_val_metrics, best_metrics, metadata, epoch, out_dir):
    """train.py:207-227: when validation precision, recall OR F1 improved, write `checkpoint_epoch_N.pt` (the pickled
    module, as the reference does with torch.save(model, ...)) and `metadata_epoch_N.json` (the run's metadata plus
    `validation_metrics`).  The upload to the outputs store / comet is out of scope.  Returns the new best metrics."""
    keys = ('cd_precisions', 'cd_recalls', 'cd_f1scores')
    if not any(mean_val_metrics[k] > best_metrics[k] for k in keys):
        return best_metrics
    os.makedirs(out_dir, exist_ok=True)
    metadata = dict(metadata)
    metadata['validation_metrics'] = {k: float(v) for k, v in mean_val_metrics.items()}
    with open(os.path.join(out_dir, f'metadata_epoch_{epoch}.json'), 'w') as fout:
        json.dump(metadata, fout)
    torch.save(model, os.path.join(out_dir, f'checkpoint_epoch_{epoch}.pt'))
    # ... and, for the way back, the parameters + BatchNorm buffers under the keys the REFERENCE's nn.DataParallel(BiDateNet) has
    # (utils/helpers.py:335): the pickle above names fabric_amd's classes, which a reference checkout cannot import; this file loads there
    # with model.load_state_dict(torch.load(path)) and here with fabric_amd.utils.helpers.load_checkpoint
    torch.save({'module.' + k: v.detach().cpu() for k, v in model.state_dict().items()},
               os.path.join(out_dir, f'checkpoint_epoch_{epoch}.state_dict.pt'))
    return mean_val_metrics


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
    ap.add_argument('--precision', default='bf16', choices=['bf16', 'bf16x3', 'bf16x3-fast', 'fp32'])
    ap.add_argument('--seed', type=int, default=0, help='seeds the shard permutation, the augmentation draws and the initial weights identically on every rank')
    ap.add_argument('--focal_gamma', type=float, default=None, help='required by --loss_function focal (utils/helpers.py:291)')
    ap.add_argument('--metadata', default=None, help="JSON in the reference's metadata.json schema (band_ids, band_means, "
                                                     "band_stds, ...): its entries become defaults like utils/parser.py:7-10")
    pre, _ = ap.parse_known_args(argv)
    meta = {}
    if pre.metadata:
        with open(pre.metadata) as fh:
            meta = json.load(fh)
        ap.set_defaults(**{k: v for k, v in meta.items() if k in DEFAULTS})
    opt = ap.parse_args(argv)
    for k in ('band_ids', 'band_means', 'band_stds'):
        setattr(opt, k, meta.get(k))
    if opt.loss_function not in ('tversky', 'dice', 'jaccard', 'focal'):
        raise SystemExit(f'--loss_function {opt.loss_function}: the reference offers bce / focal / dice / jaccard / tversky '
                         f"(utils/helpers.py:288-314); its bce branch cannot run on BiDateNet's logits and is not built")
    if opt.loss_function == 'focal' and opt.focal_gamma is None:
        raise SystemExit('--loss_function focal needs --focal_gamma')

    # one process per GPU (launched by torch.distributed.run): RANK / LOCAL_RANK / WORLD_SIZE from the environment
    world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')))
    torch.cuda.set_device(dev)                            # the library's stream / workspace helpers follow the current device
    if world > 1:
        import torch.distributed as dist
        from .parallel import init_rccl
        init_rccl(rank, world, dev)

    import random
    random.seed(opt.seed)                                  # same dataset order / initial weights on every rank; augmentation draws
    torch.manual_seed(opt.seed)                            # (global `random`, utils/dataloaders.py:150-156) are re-seeded per rank below
    scenes = None
    if opt.synthetic:
        data = synthetic_onera(n_cities=6, bands=13, size=(360, 360))
        val_cities = ['city4', 'city5']
    else:
        if not opt.band_ids:
            raise SystemExit('real data needs --metadata with band_ids / band_means / band_stds (the reference keeps them in '
                             'metadata.json); or use --synthetic')
        from .utils import ingest
        scenes = ingest.full_onera_loader(opt.dataset_dir, opt, device=dev)      # city stacks stay in HBM for the scene pass
        data = {c: {'images': d['images'].cpu().numpy(), 'labels': d['labels']} for c, d in scenes.items()}
        val_cities = [c for c in opt.validation_cities if c in data]
    train_loader, val_loader = make_loaders(data, val_cities, opt.patch_size, opt.stride // 2 if opt.synthetic else opt.stride,
                                            opt.batch_size, opt.augmentation, num_workers=opt.num_workers,
                                            rank=rank, world_size=world, seed=opt.seed)
    random.seed(opt.seed * 7919 + rank)                    # different augmentation draws per rank from here on
    model = BiDateNet(len(opt.band_ids) if opt.band_ids else 13, 2, precision=opt.precision).to(dev)
    fused = opt.loss_function == 'tversky'
    from .input_pipeline import DeviceFeeder
    from .utils.helpers import get_criterion
    feeder = DeviceFeeder(dev)                             # ONE feeder (copy stream, staging threads, device slots) for the whole run
    criterion = get_criterion(opt)                         # validation reports the criterion the run optimises (train.py:137)
    if fused:
        step = TrainStep(model, lr=opt.learning_rate, tversky_alpha=opt.tversky_alpha, tversky_beta=opt.tversky_beta)
        if world > 1:
            # measure (and, if it is the slow one, repair) the placement of RCCL's collective stream BEFORE the loop adopts the chain's stream
            rep = step.guard_collectives(opt.batch_size, opt.patch_size, opt.patch_size)
            if rank == 0:
                print(json.dumps({'collectives_guard': rep}), flush=True)
    else:
        optimizer = torch.optim.SGD(model.parameters(), lr=opt.learning_rate)      # train.py:55
        if world > 1:
            for p in model.parameters():
                dist.broadcast(p.data, src=0)
    best = {'cd_f1scores': -1, 'cd_recalls': -1, 'cd_precisions': -1}              # train.py:62
    run_meta = dict(meta, **{k: getattr(opt, k) for k in DEFAULTS}, precision=opt.precision, world_size=world)
    for epoch in range(opt.epochs):
        train_loader.sampler.set_epoch(epoch)
        if fused:
            tr = train_epoch(step, train_loader, dev, opt.patch_size, feeder)
        else:
            tr = train_epoch_autograd(model, criterion, optimizer, train_loader, dev, opt.patch_size, world, feeder)
        va = validate(model, val_loader, dev, opt.patch_size, criterion, feeder)
        if rank == 0:
            print(json.dumps({'epoch': epoch, **{'train_' + k: float(v) for k, v in tr.items()},
                              **{'validate_' + k: float(v) for k, v in va.items()}}), flush=True)
        if scenes is not None and rank == 0:                   # train.py:182-205: full validation images
            from .utils import ingest
            from .utils.inference import predict_scene
            os.makedirs(opt.log_dir, exist_ok=True)
            model.eval()
            for city in val_cities:
                st = scenes[city]['images']
                mask = predict_scene(model, st[0], st[1], patch_size=opt.patch_size, batch_size=opt.batch_size)
                ingest.write_png_gray(os.path.join(opt.log_dir, f'{city}_epoch_{epoch}.png'), (mask * 255).cpu().numpy())
        if rank == 0:                                          # replica 0's BatchNorm buffers, like DataParallel
            best = save_if_better(model, va, best, run_meta, epoch, opt.log_dir)
    feeder.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
