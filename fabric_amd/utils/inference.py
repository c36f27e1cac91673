"""Full-scene sliding-window inference: the reference's utils/inference.py + train.py:182-205.

Two ways in:

* the reference's own API -- ``_get_patches`` / ``_get_bands`` (host numpy, same tile order and paste order,
  pinned by fixture G7) and ``predict_patches`` (the ``train.py:190-203`` batch loop over a host patch stack);
* ``predict_scene`` -- the MI355X layout of the same computation: both dates of the scene are uploaded ONCE as
  band planes (a 10 000 x 10 000 x 13 x 2 float32 scene is 10.4 GB of the 288 GB HBM), every batch of tiles is
  gathered on the device straight into the packed NHWC encoder input (``bdn_gather_tiles``), and the class
  index of every pixel is written straight into the scene mask (``bdn_argmax_stitch``).  No host patch stack,
  no per-batch host<->device copy, no host sync inside the loop.

Tiles are independent: several GPUs take disjoint slices of the tile list (``shard=(rank, world)``); the only
exchange is the optional final merge of the uint8 masks.

``generate_patches`` (utils/inference.py:20-70) reads the band files through fabric_amd.utils.ingest; comet logging
(``log_full_image``) is outside this path.
"""
from os import path

import numpy as np
import torch

from .. import _lib
from .._lib import call, ptr


def get_path(path_list):
    """utils/inference.py:15-17."""
    return path.join(*[i.strip('/') for i in path_list])


def tile_origins(h, w, patch_dim):
    """(y0, x0) of every tile in the reference's order (utils/inference.py:156-184): hs*ws aligned tiles row by
    row, then hs tiles anchored on the right edge, then ws tiles anchored on the bottom edge, then the corner.
    Returns (int32 [n,2], hs, ws, lc, lr)."""
    p = patch_dim
    if h < p or w < p:
        raise ValueError(f'scene {h}x{w} is smaller than one {p}x{p} patch')
    hs, ws = h // p, w // p
    o = [(i * p, j * p) for i in range(hs) for j in range(ws)]
    o += [(i * p, w - p) for i in range(hs)]
    o += [(h - p, j * p) for j in range(ws)]
    o.append((h - p, w - p))
    return np.asarray(o, dtype=np.int32), hs, ws, hs, ws


def _get_patches(bands, patch_dim=64):
    """utils/inference.py:134-184: [H,W,C] scene -> (patches [n,p,p,C], hs, ws, lc, lr, H, W)."""
    h, w = bands.shape[:2]
    o, hs, ws, lc, lr = tile_origins(h, w, patch_dim)
    patches = np.stack([bands[y:y + patch_dim, x:x + patch_dim] for y, x in o])
    return patches, hs, ws, lc, lr, h, w


def _get_bands(patches, hs, ws, lc, lr, h, w, patch_size=64):
    """utils/inference.py:187-236: paste [n,p,p] prediction tiles back into an [h,w] float64 image; the edge
    tiles are pasted last (column, row, corner), overwriting the aligned tiles where they overlap."""
    o, hs2, ws2, _, _ = tile_origins(h, w, patch_size)
    if (hs2, ws2, hs2, ws2) != (hs, ws, lc, lr) or len(patches) != len(o):
        raise ValueError('tile counts do not belong to this scene size')
    img = np.zeros((h, w))
    for t, (y, x) in zip(patches, o):
        img[y:y + patch_size, x:x + patch_size] = t
    return img


def full_image_mask(out, hs, ws, lc, lr, h, w, patch_size):
    """The array part of log_full_image (utils/inference.py:72-104): list of per-batch predictions -> scene mask."""
    return _get_bands(np.vstack(out), hs, ws, lc, lr, h, w, patch_size=patch_size)


def generate_patches(opt, validation_city):
    """utils/inference.py:20-70: both dates of a city as patch stacks [n,C,p,p] + the reconstruction metadata.  Band files
    are decoded by fabric_amd.utils.ingest (own TIFF reader, device-side normalise + resize) instead of rasterio / cv2."""
    import glob
    from . import ingest
    city_dir = path.join(opt.dataset_dir, 'images', validation_city)      # (get_path would strip the root of an absolute dir)
    d1_bands = sorted(glob.glob(path.join(city_dir, 'imgs_1', '*')))
    template = ingest.read_tiff(d1_bands[2])                   # band 2 gives the 10 m grid (utils/inference.py:47)
    stack = ingest.city_loader([city_dir, template.shape[1], template.shape[0], opt])
    p1, hs, ws, lc, lr, h, w = _get_patches(stack[0].transpose(1, 2, 0), patch_dim=opt.patch_size)
    p2 = _get_patches(stack[1].transpose(1, 2, 0), patch_dim=opt.patch_size)[0]
    return p1.transpose(0, 3, 1, 2), p2.transpose(0, 3, 1, 2), hs, ws, lc, lr, h, w


def log_full_image(*_a, **_k):
    raise ImportError('fabric_amd: log_full_image (reference utils/inference.py:72-131) uploads figures through comet_ml and '
                      'cv2, which are outside this path; full_image_mask() returns the stitched array')


def _eval_params(model):
    if model.training:
        raise RuntimeError('full-scene inference runs on running BatchNorm statistics: call model.eval() first '
                           '(train.py:117)')
    return {k: v.detach() for k, v in model.state_dict(keep_vars=True).items()}


@torch.no_grad()
def predict_patches(model, patches1, patches2, batch_size, device='cuda'):
    """The loop of train.py:190-203 on host patch stacks [n,C,p,p]: returns the list `out` of per-batch int64
    numpy arrays [b,p,p] (= `torch.max(preds, 1)[1].cpu().numpy()`)."""
    P = _eval_params(model)
    eng = model.engine()
    out = []
    for i in range(0, patches1.shape[0], batch_size):
        b1 = torch.from_numpy(np.ascontiguousarray(patches1[i:i + batch_size])).to(device)
        b2 = torch.from_numpy(np.ascontiguousarray(patches2[i:i + batch_size])).to(device)
        logits, _ = eng.forward(b1, b2, P, training=False)
        b, ncls, hh, ww = logits.shape
        cd = torch.empty(b, hh, ww, dtype=torch.uint8, device=logits.device)
        call('bdn_argmax', ptr(logits), ptr(cd), b, ncls, hh, ww, _lib.stream_ptr())
        out.append(cd.cpu().numpy().astype(np.int64))
    return out


@torch.no_grad()
def predict_scene(model, scene_d1, scene_d2, patch_size=128, batch_size=64, shard=None, merge=True):
    """Change mask of a whole scene.

    scene_d1, scene_d2: [C,H,W] float32 tensors (what the reference's city_loader returns per date,
    utils/dataloaders.py:86-101); moved to the model's device if they are not there yet.
    Returns a uint8 [H,W] device tensor equal to ``_get_bands(argmax(model(tiles)))`` of the reference loop.
    shard=(rank, world): process only this rank's contiguous slice of the tile list; merge=True then combines
    the per-rank masks with one all-reduce(MAX) over the default process group (unwritten pixels are 0).
    batch_size: tiles per forward batch (256 is the fastest on MI355X, bench.py scene leg).  Every tensor a kernel addresses must
    stay below 4 GB (32-bit byte offsets); the widest one is the operand of a 64-channel full-resolution layer at 2 * batch_size
    images: 64 channels x 2 bytes in the bf16 setting (<= 1023 tiles of 128 x 128), 64 x 4 bytes in fp32 (<= 511) and the
    [hi | lo] split of the concatenated 128-channel decoder input, 256 x 2 bytes at batch_size images, in bf16x3 (<= 511)."""
    eng_ = model.engine()
    widest = max(2 * batch_size * 64 * eng_.esize, batch_size * 2 * 128 * 2 if eng_.x3 else 0)      # bytes per pixel position of the widest tensor
    if widest * patch_size * patch_size >= 1 << 32:
        raise ValueError(f'batch_size={batch_size} tiles of {patch_size} px make a 4 GB tensor in the {eng_.precision} setting; use a smaller batch')
    P = _eval_params(model)
    eng = model.engine()
    dev = next(model.parameters()).device
    d1 = torch.as_tensor(scene_d1).to(device=dev, dtype=torch.float32).contiguous()
    d2 = torch.as_tensor(scene_d2).to(device=dev, dtype=torch.float32).contiguous()
    if d1.dim() != 3 or d1.shape != d2.shape:
        raise RuntimeError(f'expected two [C,H,W] scenes of one shape, got {tuple(d1.shape)} and {tuple(d2.shape)}')
    _, h, w = d1.shape
    o_np, _, _, _, _ = tile_origins(h, w, patch_size)
    n = len(o_np)
    lo, hi = 0, n
    if shard is not None:
        rank, world = shard
        per = -(-n // world)
        lo, hi = min(n, rank * per), min(n, (rank + 1) * per)
    origins = torch.from_numpy(o_np).to(dev)
    mask = torch.zeros(h, w, dtype=torch.uint8, device=dev) if shard is not None \
        else torch.empty(h, w, dtype=torch.uint8, device=dev)
    seen = set()
    st = _lib.stream_ptr()
    for i in range(lo, hi, batch_size):
        o = origins[i:min(hi, i + batch_size)]
        nb = o.shape[0]
        logits, _ = eng.forward_tiles(d1, d2, o, P, patch_size, reuse_eval_bn=nb in seen)
        seen.add(nb)
        call('bdn_argmax_stitch', ptr(logits), ptr(o), ptr(mask), nb, logits.shape[1], patch_size, h, w, st)
    if shard is not None and merge and shard[1] > 1:
        import torch.distributed as dist
        dist.all_reduce(mask, op=dist.ReduceOp.MAX)
    return mask
