"""-m gpu parity of the OSCD ingest arithmetic (reference utils/dataloaders.py:86-145, utils/inference.py:20-70):
bdn_ingest_band against the oracle's restatement of `(band - mean) / std` + cv2.resize, and the reference-named loaders on a
synthetic OSCD directory written with the module's own TIFF / PNG writers.  Tolerance 2e-6 of the plane's max magnitude
(float32 interpolation in a different association order).  Parity with cv2 itself is unpinned (oracle/ingest_oracle.py)."""
import os
import types

import numpy as np
import pytest
import torch

from fabric_amd import BiDateNet, _lib
from fabric_amd._lib import call, ptr
from fabric_amd.utils import ingest as ing
from fabric_amd.utils import inference as inf
from oracle import filler
from oracle import ingest_oracle as IO
from gpu_util import st

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('case', [(55, 60, 110, 120, 'u16'), (19, 21, 114, 126, 'u16'), (64, 48, 64, 48, 'u16'),
                                  (40, 40, 61, 77, 'f32'), (120, 90, 60, 45, 'u16'), (1, 1, 5, 7, 'f32')])
def test_ingest_band_matches_oracle(case):
    hs, ws, H, W, kind = case
    r = np.random.default_rng(5)
    if kind == 'u16':
        band = r.integers(0, 12000, (hs, ws)).astype(np.uint16)
        src = torch.from_numpy(band.view(np.int16)).cuda()
    else:
        band = (1500 + 400 * r.standard_normal((hs, ws))).astype(np.float32)
        src = torch.from_numpy(band).cuda()
    mean, std = 1422.37, 456.25
    out = torch.full((H, W), float('nan'), device='cuda')
    call('bdn_ingest_band', 0 if kind == 'u16' else 1, ptr(src), hs, ws, mean, std, ptr(out), H, W, st())
    ref = IO.ingest_band(band, mean, std, W, H)
    got = out.cpu().numpy()
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() <= 2e-6 * np.abs(ref).max() + 1e-6
    if (hs, ws) == (H, W):
        assert np.array_equal(got, ref)                        # same grid: pure normalisation, bit-exact


def _synthetic_oscd(root, cities, bands, seed=6):
    """Two dates per city, 13 band files each at 10 / 20 / 60 m, uint16 strip TIFFs; labels as grayscale PNGs."""
    r = np.random.default_rng(seed)
    res = {'B01': 6, 'B09': 6, 'B10': 6, 'B05': 2, 'B06': 2, 'B07': 2, 'B8A': 2, 'B11': 2, 'B12': 2}
    truth = {}
    for city, (h, w) in cities.items():
        os.makedirs(f'{root}labels/{city}/cm')
        lab = (r.uniform(0, 1, (h, w)) < 0.1).astype(np.uint8)
        ing.write_png_gray(f'{root}labels/{city}/cm/cm.png', lab * 255)
        truth[city] = {'label': lab, 'bands': {}}
        for d in (1, 2):
            os.makedirs(f'{root}images/{city}/imgs_{d}')
            for b in bands:
                k = res.get(b, 1)
                arr = r.integers(200, 6000, (-(-h // k), -(-w // k))).astype(np.uint16)
                ing.write_tiff(f'{root}images/{city}/imgs_{d}/S2A_{city}_{d}_{b}.tif', arr, compression='deflate' if d == 2 else 'none')
                truth[city]['bands'][(d, b)] = arr
    return truth


def test_full_onera_loader_and_generate_patches(tmp_path):
    root = str(tmp_path) + '/'
    bands = ['B01', 'B02', 'B03', 'B04', 'B05', 'B06', 'B07', 'B08', 'B8A', 'B09', 'B10', 'B11', 'B12']
    r = np.random.default_rng(7)
    opt = types.SimpleNamespace(band_ids=bands, band_means={b: float(r.uniform(900, 2500)) for b in bands},
                                band_stds={b: float(r.uniform(300, 900)) for b in bands}, dataset_dir=root, patch_size=32)
    cities = {'alpha': (96, 130), 'beta': (70, 64)}
    truth = _synthetic_oscd(root, cities, bands)
    ds = ing.full_onera_loader(root, opt)
    assert sorted(ds) == ['alpha', 'beta']
    for city, (h, w) in cities.items():
        img, lab = ds[city]['images'], ds[city]['labels']
        assert img.dtype == np.float32 and img.shape == (2, 13, h, w) and lab.dtype == np.uint8
        assert np.array_equal(lab, truth[city]['label'])
        for d in (1, 2):
            for c, b in enumerate(bands):
                ref = IO.ingest_band(truth[city]['bands'][(d, b)], opt.band_means[b], opt.band_stds[b], w, h)
                assert np.abs(img[d - 1, c] - ref).max() <= 2e-6 * np.abs(ref).max() + 1e-6, (city, d, b)
    # device-resident variant feeds the scene path directly
    dev = ing.full_onera_loader(root, opt, device='cuda')
    assert dev['alpha']['images'].is_cuda and torch.equal(dev['alpha']['images'].cpu(), torch.from_numpy(ds['alpha']['images']))
    model = filler.fill_module(BiDateNet(13, 2, precision='bf16')).cuda().eval()
    mask = inf.predict_scene(model, dev['alpha']['images'][0], dev['alpha']['images'][1], patch_size=32, batch_size=8)
    assert tuple(mask.shape) == cities['alpha'] and mask.dtype == torch.uint8
    # reference-named patch generator: same tiles as tiling the loaded stack
    p1, p2, hs, ws, lc, lr, h, w = inf.generate_patches(opt, 'alpha')
    t1 = inf._get_patches(ds['alpha']['images'][0].transpose(1, 2, 0), patch_dim=32)[0].transpose(0, 3, 1, 2)
    assert (h, w) == cities['alpha'] and p1.shape == t1.shape == (hs * ws + lc + lr + 1, 13, 32, 32) and np.array_equal(p1, t1)
    # the reference's dataloaders names resolve to the ingest module
    from fabric_amd.utils import dataloaders as dl
    assert dl.city_loader is ing.city_loader and dl.get_train_val_metadata is ing.get_train_val_metadata
    # the reference's get_loaders(opt) (utils/helpers.py:211-258), same signature: DataLoaders over the ingested cities
    from fabric_amd.utils.helpers import get_loaders
    opt.dataset_dir, opt.validation_cities, opt.patch_size, opt.stride = root, ['beta'], 16, 16
    opt.augmentation, opt.batch_size, opt.num_workers = False, 3, 0
    tr, va = get_loaders(opt)
    tm, vm = ing.get_train_val_metadata(root, ['beta'], 16, 16)
    assert len(tr.dataset) == len(tm) and len(va.dataset) == len(vm) and len(vm) > 0
    b1, b2, lb = next(iter(va))
    c0, i0, j0 = va.dataset.imgs[0]               # OneraPreloader shuffles its index list in place (utils/dataloaders.py:171)
    assert b1.shape[1:] == (13, 16, 16) and lb.dtype == torch.uint8
    assert np.array_equal(b1[0].numpy(), ds[c0]['images'][0][:, i0:i0 + 16, j0:j0 + 16])
    assert np.array_equal(lb[0].numpy(), ds[c0]['labels'][i0:i0 + 16, j0:j0 + 16])


def test_train_loop_on_an_oscd_directory(tmp_path, capsys):
    """python -m fabric_amd.train on a (synthetic) OSCD tree: ingest -> patch loaders -> fused steps -> validation ->
    full-scene masks written like train.py:182-205."""
    import json
    from fabric_amd import train as T
    root = str(tmp_path) + '/data/'
    bands = ['B01', 'B02', 'B03', 'B04', 'B05', 'B06', 'B07', 'B08', 'B8A', 'B09', 'B10', 'B11', 'B12']
    cities = {'aa': (128, 160), 'bb': (96, 96), 'cc': (100, 130)}
    _synthetic_oscd(root, cities, bands, seed=8)
    meta = {'band_ids': bands, 'band_means': {b: 3000.0 for b in bands}, 'band_stds': {b: 1500.0 for b in bands},
            'patch_size': 32, 'stride': 32, 'batch_size': 8, 'validation_cities': ['cc'], 'epochs': 1}
    mpath = str(tmp_path / 'metadata.json')
    json.dump(meta, open(mpath, 'w'))
    T.main(['--metadata', mpath, '--dataset_dir', root, '--log_dir', str(tmp_path / 'log'), '--augmentation', 'false'])
    out = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert out['epoch'] == 0 and np.isfinite(out['train_cd_losses']) and 0 <= out['validate_cd_corrects'] <= 100
    mask = ing.read_png_gray(str(tmp_path / 'log' / 'cc_epoch_0.png'))
    assert mask.shape == cities['cc'] and set(np.unique(mask)) <= {0, 255}
