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


# The eight symmetries of a square patch as (transpose, reverse rows, reverse columns), applied in that order.
# np.rot90(m, k) over (rows, cols) is entry k; the reference's two optional flips toggle the reversals.
_ROT = ((False, False, False), (True, True, False), (False, True, True), (True, False, True))


def _draw_symmetry():
    """The augmentation draw of reference utils/dataloaders.py:154-163 -- randint(0, 3), then one random() per flip,
    always three draws from the global `random` module in this order -- folded into one element of the dihedral group."""
    k = random.randint(0, 3)
    flip_rows = random.random() > 0.5
    flip_cols = random.random() > 0.5
    t, rr, rc = _ROT[k]
    return t, rr ^ flip_rows, rc ^ flip_cols


def _apply_symmetry(window, sym):
    """`window`: array view whose LAST two axes are (rows, cols).  One strided view, no intermediate copies."""
    t, rr, rc = sym
    if t:
        window = window.swapaxes(-2, -1)
    return window[..., ::-1 if rr else 1, ::-1 if rc else 1]


def onera_siamese_loader(dataset, city, x, y, size, aug):
    """Drop-in for reference utils/dataloaders.py:148-165: the [C,size,size] windows of both dates and the label window
    at rows x.., columns y.., optionally under a random symmetry of the square (same draws, same result as the
    reference's rot90 / flip chain; pinned by fixture G7).  Returns fresh contiguous arrays (img_d1, img_d2, label)."""
    entry = dataset[city]
    rows, cols = slice(x, x + size), slice(y, y + size)
    sym = _draw_symmetry() if aug else (False, False, False)
    pair = np.ascontiguousarray(_apply_symmetry(entry['images'][:, :, rows, cols], sym))
    label = np.ascontiguousarray(_apply_symmetry(entry['labels'][rows, cols], sym))
    return pair[0], pair[1], label


class OneraPreloader(data.Dataset):
    """Drop-in for reference utils/dataloaders.py:168-198: a map-style dataset over `metadata` = list of [city, i, j]
    patch origins into the pre-loaded `full_load` dict.  The list is shuffled IN PLACE at construction with the global
    `random` module, as the reference does (utils/dataloaders.py:171) -- callers that seed `random` get the same order."""

    def __init__(self, root, metadata, full_load, input_size, aug=False):
        random.shuffle(metadata)
        self.root, self.full_load = root, full_load
        self.imgs = metadata                                 # attribute names are part of the reference's surface
        self.input_size, self.aug = input_size, aug
        self.loader = onera_siamese_loader

    def __len__(self):
        return len(self.imgs)

    def __getitem__(self, index):
        city, i, j = self.imgs[index]
        return self.loader(self.full_load, city, i, j, self.input_size, self.aug)


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
