"""OSCD ingest, host side: the TIFF / PNG decoders of fabric_amd.utils.ingest against files written by an
independent implementation (Pillow), and the oracle's restatement of cv2's INTER_LINEAR against Pillow's BILINEAR on
upscales (the two libraries sample identically there).  cv2 / rasterio themselves are not installed: parity with the
reference's own output is unpinned for this row (stated in oracle/ingest_oracle.py)."""
import importlib.util
import os
import sys
import types

import numpy as np
import pytest

from oracle import ingest_oracle as IO

PIL = pytest.importorskip('PIL.Image')


def _ingest_module():
    """fabric_amd.utils.ingest without loading the HIP library (decoders are pure host code)."""
    import fabric_amd.utils.ingest as ing
    return ing


@pytest.mark.parametrize('comp', ['raw', 'tiff_lzw', 'tiff_adobe_deflate', 'packbits'])
def test_read_tiff_matches_pillow_written_files(tmp_path, comp):
    ing = _ingest_module()
    r = np.random.default_rng(0)
    smooth = (np.add.outer(np.arange(217), np.arange(301)) * 13 % 4000 + r.integers(0, 50, (217, 301))).astype(np.uint16)
    noisy = r.integers(0, 65535, (64, 77)).astype(np.uint16)
    for i, arr in enumerate((smooth, noisy)):
        path = str(tmp_path / f'b{i}.tif')
        PIL.fromarray(arr).save(path, compression=None if comp == 'raw' else comp)
        got = ing.read_tiff(path)
        assert got.dtype == np.uint16 and np.array_equal(got, arr)
    f = r.standard_normal((33, 50)).astype(np.float32)
    path = str(tmp_path / 'f.tif')
    PIL.fromarray(f).save(path, compression=None if comp == 'raw' else comp)
    assert np.array_equal(ing.read_tiff(path), f)


def test_tiff_writer_round_trip(tmp_path):
    ing = _ingest_module()
    r = np.random.default_rng(1)
    for comp in ('none', 'deflate'):
        for shape in ((5, 7), (130, 129)):
            a = r.integers(0, 65535, shape).astype(np.uint16)
            p = str(tmp_path / f'{comp}_{shape[0]}.tif')
            ing.write_tiff(p, a, compression=comp)
            assert np.array_equal(ing.read_tiff(p), a)
            assert np.array_equal(np.asarray(PIL.open(p)), a)           # and Pillow reads what we write


def test_read_png_gray(tmp_path):
    ing = _ingest_module()
    r = np.random.default_rng(2)
    g = (r.uniform(0, 1, (40, 53)) < 0.2).astype(np.uint8) * 255
    p = str(tmp_path / 'g.png')
    PIL.fromarray(g).save(p)
    assert np.array_equal(ing.read_png_gray(p), g)
    grad = (np.add.outer(np.arange(40), np.arange(53)) * 3 % 256).astype(np.uint8)   # smooth: exercises sub/up/paeth filters
    PIL.fromarray(grad).save(p, optimize=True)
    assert np.array_equal(ing.read_png_gray(p), grad)
    rgb = r.integers(0, 256, (31, 20, 3)).astype(np.uint8)
    PIL.fromarray(rgb).save(p)
    assert np.array_equal(ing.read_png_gray(p), IO.gray_from_rgb(rgb))
    rgba = np.dstack([rgb, np.full((31, 20), 255, np.uint8)])
    PIL.fromarray(rgba).save(p)
    assert np.array_equal(ing.read_png_gray(p), IO.gray_from_rgb(rgb))
    pal = PIL.fromarray(g).convert('P')
    pal.save(p)
    assert np.array_equal(ing.read_png_gray(p), g)
    ing.write_png_gray(p, g)
    assert np.array_equal(np.asarray(PIL.open(p)), g) and np.array_equal(ing.read_png_gray(p), g)


@pytest.mark.parametrize('case', [(20, 30, 2.0), (11, 9, 6.0), (16, 16, 1.5), (7, 40, 3.0)])
def test_oracle_resize_matches_pillow_bilinear_on_upscales(case):
    h, w, s = case
    r = np.random.default_rng(3)
    img = r.standard_normal((h, w)).astype(np.float32)
    H, W = int(round(h * s)), int(round(w * s))
    ref = np.asarray(PIL.fromarray(img).resize((W, H), resample=PIL.BILINEAR))
    got = IO.resize_linear_f32(img, W, H)
    assert got.shape == (H, W) and np.abs(got - ref).max() < 3e-6 * np.abs(img).max()
    assert np.array_equal(IO.resize_linear_f32(img, w, h), img)                      # same size: a copy


def test_get_train_val_metadata_and_labels(tmp_path):
    ing = _ingest_module()
    root = str(tmp_path) + '/'
    shapes = {'cityA': (200, 310), 'cityB': (95, 90), 'cityC': (181, 270)}
    r = np.random.default_rng(4)
    for c, (h, w) in shapes.items():
        os.makedirs(root + f'labels/{c}/cm')
        ing.write_png_gray(root + f'labels/{c}/cm/cm.png', (r.uniform(0, 1, (h, w)) < 0.1).astype(np.uint8) * 255)
    os.makedirs(root + 'labels/.hidden')
    train, val = ing.get_train_val_metadata(root, ['cityB'], 90, 180)
    assert val == [['cityB', 0, 0]]
    assert train == [['cityA', 0, 0], ['cityA', 0, 180], ['cityC', 0, 0], ['cityC', 0, 180]]
    lab = ing.label_loader(root + 'labels/cityA')
    assert lab.dtype == np.float64 and set(np.unique(lab)) <= {0.0, 1.0} and lab.shape == shapes['cityA']
