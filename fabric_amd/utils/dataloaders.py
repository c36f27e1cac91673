"""OSCD patch-pair dataset API: drop-in for the reference's utils/dataloaders.py:148-198 (the part of
that file that sits on the hot path: cropping / augmenting patch pairs out of pre-loaded city stacks).

Host-side numpy, like the reference.  The ingest half of that file (city_loader / full_onera_loader /
label_loader / get_train_val_metadata, reference utils/dataloaders.py:51-145) is fabric_amd.utils.ingest and is
re-exported from here under the reference's names; `synthetic_onera` builds a dataset of the same schema
for benchmarks and tests.
"""
import random

import numpy as np
import torch.utils.data as data


def onera_siamese_loader(dataset, city, x, y, size, aug):
    """reference utils/dataloaders.py:148-165: crop [2,C,x:x+size,y:y+size] and the label window; optional
    rot90(k in 0..3) / flip-H / flip-W drawn from the global `random` module in the reference's order."""
    out_img = np.copy(dataset[city]['images'][:, :, x:x + size, y:y + size])
    out_lbl = np.copy(dataset[city]['labels'][x:x + size, y:y + size])
    if aug:
        rot_deg = random.randint(0, 3)
        out_img = np.rot90(out_img, rot_deg, [2, 3]).copy()
        out_lbl = np.rot90(out_lbl, rot_deg, [0, 1]).copy()
        if random.random() > 0.5:
            out_img = np.flip(out_img, axis=2).copy()
            out_lbl = np.flip(out_lbl, axis=0).copy()
        if random.random() > 0.5:
            out_img = np.flip(out_img, axis=3).copy()
            out_lbl = np.flip(out_lbl, axis=1).copy()
    return out_img[0], out_img[1], out_lbl


class OneraPreloader(data.Dataset):
    """reference utils/dataloaders.py:168-198.  `metadata` = list of [city, i, j]; shuffled in place at
    construction like the reference (utils/dataloaders.py:171)."""

    def __init__(self, root, metadata, full_load, input_size, aug=False):
        random.shuffle(metadata)
        self.full_load = full_load
        self.root = root
        self.imgs = metadata
        self.loader = onera_siamese_loader
        self.aug = aug
        self.input_size = input_size

    def __getitem__(self, index):
        city, x, y = self.imgs[index]
        return self.loader(self.full_load, city, x, y, self.input_size, self.aug)

    def __len__(self):
        return len(self.imgs)


def patch_origins(height, width, patch_size, stride):
    """Enumeration rule of reference utils/dataloaders.py:63-67: origins on a `stride` grid whose patch
    fits entirely inside the label raster."""
    return [[i, j] for i in range(0, height, stride) for j in range(0, width, stride)
            if i + patch_size <= height and j + patch_size <= width]


def metadata_from_shapes(shapes, val_cities, patch_size, stride):
    """get_train_val_metadata (reference utils/dataloaders.py:51-78) given {city: (H, W)} instead of label
    PNGs on disk.  Cities are visited in sorted order (the reference iterates a set difference, i.e. in
    arbitrary order)."""
    train, val = [], []
    for city in sorted(shapes):
        h, w = shapes[city]
        dst = val if city in val_cities else train
        dst += [[city, i, j] for i, j in patch_origins(h, w, patch_size, stride)]
    return train, val


def synthetic_onera(n_cities=4, bands=13, size=(300, 260), seed=0, change_fraction=0.05):
    """A `full_load` dict of the reference's schema ({city: {'images': f32[2,C,H,W], 'labels': u8[H,W]}},
    utils/dataloaders.py:138-145) filled with z-scored noise; date 2 = date 1 + small noise + blobs of
    change where the label is 1."""
    r = np.random.default_rng(seed)
    out = {}
    for c in range(n_cities):
        h, w = size
        d1 = r.standard_normal((bands, h, w)).astype(np.float32)
        lbl = np.zeros((h, w), np.uint8)
        for _ in range(max(1, int(change_fraction * h * w / 400))):
            cy, cx = r.integers(0, h), r.integers(0, w)
            lbl[max(0, cy - 10):cy + 10, max(0, cx - 10):cx + 10] = 1
        d2 = d1 + 0.3 * r.standard_normal(d1.shape).astype(np.float32) + 1.5 * lbl[None].astype(np.float32)
        out[f'city{c}'] = {'images': np.stack([d1, d2]).astype(np.float32), 'labels': lbl}
    return out


def __getattr__(name):
    """city_loader / full_onera_loader / label_loader / get_train_val_metadata (reference utils/dataloaders.py:51-145) live in
    fabric_amd.utils.ingest (own TIFF / PNG decoding + device-side normalise-and-resize); resolved lazily because that
    module needs the HIP library."""
    if name in ('city_loader', 'full_onera_loader', 'label_loader', 'get_train_val_metadata'):
        from . import ingest
        return getattr(ingest, name)
    raise AttributeError(name)
