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
        cd, _ = eng.forward(b1, b2, P, training=False, class_map=True)       # uint8 [b,p,p] = torch.max(preds, 1)[1]
        out.append(cd.cpu().numpy().astype(np.int64))
    return out


@torch.no_grad()
def predict_scene(model, scene_d1, scene_d2, patch_size=128, batch_size=64, shard=None, merge=True, band_rows=None, two_streams=None):
    """Change mask of a whole scene.

    scene_d1, scene_d2: [C,H,W] float32 tensors (what the reference's city_loader returns per date,
    utils/dataloaders.py:86-101), resident on the model's device or in HOST memory.  Host scenes are streamed up in bands of
    `band_rows` rows (default 256) on the copy stream while the tiles of the bands that have arrived run (pin the tensors:
    pageable memory is staged through pinned buffers by host threads and is host-memcpy bound).
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
    s1, s2 = torch.as_tensor(scene_d1), torch.as_tensor(scene_d2)
    if s1.dim() != 3 or s1.shape != s2.shape:
        raise RuntimeError(f'expected two [C,H,W] scenes of one shape, got {tuple(s1.shape)} and {tuple(s2.shape)}')
    _, h, w = s1.shape
    o_np, _, _, _, _ = tile_origins(h, w, patch_size)
    n = len(o_np)
    lo, hi = 0, n
    if shard is not None:
        rank, world = shard
        per = -(-n // world)
        lo, hi = min(n, rank * per), min(n, (rank + 1) * per)
    feed = None
    if s1.is_cuda and s2.is_cuda:
        d1 = s1.to(device=dev, dtype=torch.float32).contiguous()
        d2 = s2.to(device=dev, dtype=torch.float32).contiguous()
    else:
        # scene in HOST memory (what the reference's city_loader returns): the planes go up in row bands on the copy stream while
        # the tiles of the bands that have arrived are already being predicted -- the upload (10.4 GB at 10 000^2 x 13 x 2 dates,
        # as long over PCIe as the whole forward takes) hides behind the compute instead of preceding it
        feed = _SceneFeeder(s1.cpu(), s2.cpu(), dev, band_rows or max(patch_size, 256))
        d1, d2 = feed.d1, feed.d2
    origins = torch.from_numpy(o_np).to(dev)
    mask = torch.zeros(h, w, dtype=torch.uint8, device=dev) if shard is not None \
        else torch.empty(h, w, dtype=torch.uint8, device=dev)
    # Tile batches are independent: they alternate between the caller's stream and the library's second stream (idle outside training),
    # each with its own workspace, so that one batch's HBM-bound stages (tile gather, pooling, upsampling, classifier, stitching) run
    # under the other's convolutions (+1.5 %).  The second lane needs a second full workspace (several GB at 256 tiles): it is created
    # under the CALLER's stream (its blocks belong to the caller's allocator pool) and dropped again on exit, so a training workspace
    # of the same process can take the memory over.  two_streams=None: only when the device has room for it; False: the single-stream loop.
    from .. import streams as _streams
    cur = torch.cuda.current_stream(dev)
    want_two = (hi - lo) > batch_size and two_streams is not False
    if want_two:
        nb0 = min(batch_size, hi - lo)
        a0 = torch.cuda.memory_allocated(dev)
        eng.workspace(nb0, patch_size, patch_size, dev, 0)
        first_ws = max(torch.cuda.memory_allocated(dev) - a0, 0)           # 0 when slot 0 existed already: then size it from the tensors
        if first_ws == 0:
            w0 = eng.workspace(nb0, patch_size, patch_size, dev, 0)
            first_ws = sum(t.numel() * t.element_size() for d_ in (w0.z, w0.pool, w0.f, w0.U) for t in d_.values()) + w0.x0.numel() * w0.x0.element_size()
        free_dev = torch.cuda.mem_get_info(dev)[0] + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
        if two_streams is None and free_dev < 1.5 * first_ws:
            want_two = False
    lanes = [cur, _streams.get('wgrad', dev)] if want_two else [cur]
    seen = [set() for _ in lanes]
    if len(lanes) > 1:
        eng._weights(eng.layers[0], P, False)            # the filter images are packed once, on the caller's stream, before the fork
        if eng._use_eval_schedule():
            eng.eval_tables(P)                           # and so are the folded BatchNorm tables both lanes read
            seen = [{min(batch_size, hi - lo), (hi - lo) % batch_size or batch_size} for _ in lanes]
        for nb_ in {min(batch_size, hi - lo), (hi - lo) % batch_size or batch_size}:
            eng.workspace(nb_, patch_size, patch_size, dev, 1)             # second lane's buffers: allocated under the caller's stream
        lanes[1].wait_stream(cur)
        for t in (d1, d2, mask, origins):
            t.record_stream(lanes[1])
    try:
        for it, i in enumerate(range(lo, hi, batch_size)):
            j = min(hi, i + batch_size)
            o = origins[i:j]
            nb = o.shape[0]
            k = it % len(lanes)
            with torch.cuda.stream(lanes[k]):
                if feed is not None:
                    feed.need_rows(int(o_np[i:j, 0].max()) + patch_size, lanes[k])     # this lane waits for the last band these tiles read
                eng.forward_tiles(d1, d2, o, P, patch_size, reuse_eval_bn=nb in seen[k], slot=k, scene_mask=mask)
                seen[k].add(nb)
    finally:
        for ln in lanes[1:]:
            cur.wait_stream(ln)
        if len(lanes) > 1:
            eng.drop_workspaces(slot=1)                  # back to the caller's pool (ordered behind the join above)
        if feed is not None:
            feed.close()          # also on an exception: the consumer stream joins every upload before the planes can be freed
    if shard is not None and merge and shard[1] > 1:
        import torch.distributed as dist
        dist.all_reduce(mask, op=dist.ReduceOp.MAX)
    return mask


class _SceneFeeder:
    """Uploads two [C,H,W] float32 host scenes into device planes band by band (rows [k R, (k+1) R) of every plane of both
    dates per band), date 1 on the process-wide copy stream and date 2 on a second one (two DMA engines keep PCIe busier beside
    the running forward than one: 0.209 -> 0.199 s for the 10 000^2 scene), one event per band and stream.  Pinned sources are DMA'd in place; pageable ones
    are first copied into two alternating pinned band buffers by a few host threads (slower: the host memcpy, not PCIe, is
    then the limit).  The reference copies every batch of host patches synchronously (train.py:194-197)."""

    def __init__(self, s1, s2, dev, band_rows):
        from concurrent.futures import ThreadPoolExecutor
        from .. import streams
        self.src = [s1.float().contiguous() if s1.dtype != torch.float32 or not s1.is_contiguous() else s1,
                    s2.float().contiguous() if s2.dtype != torch.float32 or not s2.is_contiguous() else s2]
        C, H, W = self.src[0].shape
        self.R, self.H = band_rows, H
        self.copy = streams.get('copy', dev)
        self.cur = torch.cuda.current_stream(dev)
        self.copy2 = streams.get('copy2', dev)            # one stream (= one DMA engine) per date: 49.8 -> 52.1 GB/s sustained beside the forward
        with torch.cuda.stream(self.copy):
            # allocated UNDER the copy stream (see input_pipeline.py: a block of the consumer stream's pool may still be in use
            # by kernels that stream has queued)
            self.d1 = torch.empty(C, H, W, dtype=torch.float32, device=dev)
            self.d2 = torch.empty(C, H, W, dtype=torch.float32, device=dev)
            alloc = torch.cuda.Event()
            alloc.record(self.copy)
        self.copy2.wait_event(alloc)          # d2 came from the copy stream's pool: whatever that stream still has queued on the block
        self.d2.record_stream(self.copy2)     # (a feeder that just closed) precedes copy2's writes, and the block is not re-used under them
        self.events, self.waited, self.issued = [], {}, 0
        self.nbands = -(-H // band_rows)
        self.pinned = all(t.is_pinned() for t in self.src)
        self.pool = None if self.pinned else ThreadPoolExecutor(max_workers=8)
        self.stage = None if self.pinned else [[torch.empty(C, min(band_rows, H), W, dtype=torch.float32, pin_memory=True)
                                                 for _ in range(2)] for _ in range(2)]
        self.stage_free = [None, None]                   # event after which a staging slot may be overwritten by the host
        self.d1.record_stream(self.cur)
        self.d2.record_stream(self.cur)
        self._issue(self.LOOKAHEAD)

    # pinned sources: one hipMemcpy2DAsync per band and date (bdn_upload_band) instead of one copy per plane (26 -> 2 copies per band).  Boxes come in two
    # kinds (same plain-copy rate, 57.6 GB/s): on one the per-plane form sustains 53 GB/s beside the forward and the 2-D form 52 (0.196 vs 0.200 s for
    # the 10 000^2 scene), on the other the per-plane form sustains 37.7 GB/s (0.276 s, 0.66 of max(compute, PCIe) -- the driver's box of rounds 5) and
    # the 2-D form 52.7 (0.197 s, 0.91).  False = the per-plane form (tools/bench_scene_hostfed.py, BAND2D=0)
    band_copy_2d = True
    LOOKAHEAD = 4        # bands enqueued beyond the one a batch waits for: the copy stream never runs dry, the host never runs far ahead

    def _issue(self, upto):
        """Enqueue the uploads of bands [issued, upto]."""
        C = self.src[0].shape[0]
        while self.issued <= min(upto, self.nbands - 1):
            k = self.issued
            r0, r1 = k * self.R, min(self.H, (k + 1) * self.R)
            if not self.pinned:
                slot = k % 2
                if self.stage_free[slot] is not None:
                    for e_ in self.stage_free[slot]:
                        e_.synchronize()
                jobs = [self.pool.submit(self.stage[d][slot][c, :r1 - r0].copy_, self.src[d][c, r0:r1]) for d in range(2) for c in range(C)]
                for j in jobs:
                    j.result()
            evs = []
            for d, (dst, cs) in enumerate(((self.d1, self.copy), (self.d2, self.copy2))):
                with torch.cuda.stream(cs):
                    if self.pinned and self.band_copy_2d:        # all planes' rows of the band in ONE 2-D copy (26 copies per band -> 2)
                        _lib.call('bdn_upload_band', dst.data_ptr(), self.src[d].data_ptr(), C, self.H, self.src[d].shape[2], r0, r1, cs.cuda_stream)
                    else:
                        for c in range(C):               # one contiguous [rows, W] block per plane
                            src = self.src[d][c, r0:r1] if self.pinned else self.stage[d][k % 2][c, :r1 - r0]
                            dst[c, r0:r1].copy_(src, non_blocking=True)
                    e_ = torch.cuda.Event()
                    e_.record(cs)
                    evs.append(e_)
            ev = evs
            self.events.append(ev)
            if not self.pinned:
                self.stage_free[k % 2] = ev
            self.issued += 1

    def need_rows(self, rows, stream=None):
        """Make the consumer stream (default: the one the feeder was created under) wait until scene rows [0, rows) have arrived, and keep
        the copy stream LOOKAHEAD bands ahead.  Every consumer stream keeps its own high-water mark."""
        k = min(self.nbands - 1, (min(rows, self.H) - 1) // self.R)
        self._issue(k + self.LOOKAHEAD)
        stream = stream or self.cur
        key = stream.cuda_stream
        if k > self.waited.get(key, -1):
            for e_ in self.events[k]:                    # bands are uploaded in order on each stream: band k implies 0..k
                stream.wait_event(e_)
            self.waited[key] = k

    def close(self):
        try:
            self.need_rows(self.H)
        finally:
            if self.pool is not None:
                self.pool.shutdown(wait=True)
                self.pool = None
