"""-m gpu parity of full-scene sliding-window inference (reference train.py:182-205 and
utils/inference.py:134-236) against the oracle's tiler / eval-mode forward / stitcher.

Integer work (tile gather, argmax, stitching) is bit-exact.  The fp32 mask must equal the oracle's wherever the
oracle's own top-2 logit margin exceeds 1e-3 of the logit scale (the model-level tolerance of test_gpu_model.py);
the bf16 mask must agree on >= 97 % of the pixels.
"""
import numpy as np
import pytest
import torch

from fabric_amd import BiDateNet, _lib
from fabric_amd._lib import call, ptr
from fabric_amd.utils import inference as inf
from oracle import bidate_oracle as O
from oracle import filler
from gpu_util import DT, st, rnd

pytestmark = pytest.mark.gpu


def _scene(c, h, w, seed):
    r = np.random.default_rng(seed)
    d1 = r.standard_normal((c, h, w)).astype(np.float32)
    d2 = (d1 + 0.5 * r.standard_normal((c, h, w))).astype(np.float32)
    d2[:, h // 4:h // 2, w // 3:w // 2] += 2.0                    # a "changed" block
    return d1, d2


@pytest.mark.parametrize('prec', ['fp32', 'bf16'])
@pytest.mark.parametrize('shape', [(13, 150, 141, 32), (3, 96, 64, 32), (5, 40, 40, 40), (13, 33, 70, 16)])
def test_gather_tiles_matches_oracle_tiler(prec, shape):
    c, h, w, p = shape
    d1, d2 = _scene(c, h, w, 1)
    ref1, hs, ws, lc, lr, _, _ = O.tile_scene(d1.transpose(1, 2, 0), p)          # [n,p,p,C]
    ref2 = O.tile_scene(d2.transpose(1, 2, 0), p)[0]
    o, hs2, ws2, lc2, lr2 = inf.tile_origins(h, w, p)
    assert (hs, ws, lc, lr) == (hs2, ws2, lc2, lr2) and len(o) == ref1.shape[0]
    n, cp = len(o), 16
    dt, td = DT[prec]
    out = torch.full((2 * n, p, p, cp), 7.0, dtype=td, device='cuda')
    g1, g2, go = torch.from_numpy(d1).cuda(), torch.from_numpy(d2).cuda(), torch.from_numpy(o).cuda()
    call('bdn_gather_tiles', dt, ptr(g1), ptr(g2), ptr(go), ptr(out), n, c, h, w, p, cp, st())
    got = out.float().cpu()
    ref = rnd(prec, torch.from_numpy(np.concatenate([ref1, ref2])))
    assert torch.equal(got[..., :c], ref)
    assert (got[..., c:] == 0).all()


@pytest.mark.parametrize('shape', [(2, 150, 141, 32), (2, 96, 64, 32), (5, 64, 100, 64), (2, 40, 40, 40)])
def test_argmax_stitch_matches_oracle_stitcher(shape):
    ncls, h, w, p = shape
    o, hs, ws, lc, lr = inf.tile_origins(h, w, p)
    n = len(o)
    r = np.random.default_rng(2)
    # tiles deliberately disagree where they overlap, so the paste order is what is being tested; a few exact ties
    logits = r.integers(-3, 4, (n, ncls, p, p)).astype(np.float32)
    lt = torch.from_numpy(logits)
    pred = torch.max(lt, 1)[1].numpy()
    ref = O.stitch_scene(pred.astype(np.float64), hs, ws, lc, lr, h, w, p).astype(np.uint8)
    mask = torch.full((h, w), 255, dtype=torch.uint8, device='cuda')
    lg, go = lt.cuda(), torch.from_numpy(o).cuda()
    call('bdn_argmax_stitch', ptr(lg), ptr(go), ptr(mask), n, ncls, p, h, w, st())
    if h % p and w % p:
        assert np.array_equal(mask.cpu().numpy(), ref)
    else:
        # an edge tile that coincides with an aligned tile holds the same data in a real scene; with the
        # synthetic disagreeing tiles only the pixels owned by exactly one tile are comparable
        cover = np.zeros((h, w), np.int32)
        for y, x in o:
            same = [(yy, xx) for yy, xx in o if (yy, xx) == (y, x)]
            if len(same) > 1:
                cover[y:y + p, x:x + p] += 1
        ok = cover == 0
        assert np.array_equal(mask.cpu().numpy()[ok], ref[ok])
    dense = torch.empty(n, p, p, dtype=torch.uint8, device='cuda')
    call('bdn_argmax', ptr(lg), ptr(dense), n, ncls, p, p, st())
    assert np.array_equal(dense.cpu().numpy(), pred.astype(np.uint8))


def _calibrated_model(c, prec, d1, d2, p):
    """Filled model whose running statistics have seen the scene (a few train-mode forwards on its tiles), so the
    eval-mode mask has both classes; returns (model in eval mode, CPU state dict for the oracle)."""
    model = filler.fill_module(BiDateNet(c, 2, precision='fp32')).cuda().train()
    t1 = torch.from_numpy(np.ascontiguousarray(inf._get_patches(d1.transpose(1, 2, 0), p)[0].transpose(0, 3, 1, 2))).cuda()
    t2 = torch.from_numpy(np.ascontiguousarray(inf._get_patches(d2.transpose(1, 2, 0), p)[0].transpose(0, 3, 1, 2))).cuda()
    with torch.no_grad():
        for _ in range(25):
            model(t1, t2)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    model.precision = prec
    return model.eval(), sd


@pytest.mark.parametrize('prec', ['fp32', 'bf16'])
def test_predict_scene_matches_oracle(prec):
    c, h, w, p = 3, 88, 75, 32
    d1, d2 = _scene(c, h, w, 3)
    model, sd = _calibrated_model(c, prec, d1, d2, p)
    # oracle: tile -> eval forward -> argmax -> stitch (train.py:182-205)
    tiles1, hs, ws, lc, lr, _, _ = O.tile_scene(d1.transpose(1, 2, 0), p)
    tiles2 = O.tile_scene(d2.transpose(1, 2, 0), p)[0]
    x1 = torch.from_numpy(np.ascontiguousarray(tiles1.transpose(0, 3, 1, 2)))
    x2 = torch.from_numpy(np.ascontiguousarray(tiles2.transpose(0, 3, 1, 2)))
    with torch.no_grad():
        logits, _ = O.bidate_forward(sd, x1, x2, training=False)
    pred = torch.max(logits, 1)[1].numpy()
    ref = O.stitch_scene(pred.astype(np.float64), hs, ws, lc, lr, h, w, p).astype(np.uint8)
    margin = (logits[:, 1] - logits[:, 0]).abs().numpy()
    marg_img = O.stitch_scene(margin.astype(np.float64), hs, ws, lc, lr, h, w, p)
    scale = logits.abs().max().item()
    assert 0.02 < ref.mean() < 0.98, 'degenerate test scene: the oracle mask has a single class'

    for bs in (4, 5, 64):                                        # ragged last batch, one batch
        mask = inf.predict_scene(model, torch.from_numpy(d1), torch.from_numpy(d2), patch_size=p, batch_size=bs)
        assert mask.dtype == torch.uint8 and tuple(mask.shape) == (h, w)
        got = mask.cpu().numpy()
        diff = got != ref
        if prec == 'fp32':
            assert not (diff & (marg_img > 1e-3 * scale)).any(), f'{diff.sum()} pixels differ beyond the margin (bs={bs})'
        else:
            assert diff.mean() <= 0.03, f'bf16 mask agreement {1 - diff.mean():.4f} (bs={bs})'
    print(f'\n[scene {prec}] mask mean {ref.mean():.3f}, differing pixels {int(diff.sum())} of {h * w}')

    # the reference-loop API gives the same mask as the fused scene path, bit for bit
    p1 = np.ascontiguousarray(inf._get_patches(d1.transpose(1, 2, 0), p)[0].transpose(0, 3, 1, 2))
    p2 = np.ascontiguousarray(inf._get_patches(d2.transpose(1, 2, 0), p)[0].transpose(0, 3, 1, 2))
    out = inf.predict_patches(model, p1, p2, batch_size=4)
    assert out[0].dtype == np.int64
    img = inf.full_image_mask(out, hs, ws, lc, lr, h, w, p)
    assert np.array_equal(img.astype(np.uint8), inf.predict_scene(model, d1, d2, patch_size=p, batch_size=4).cpu().numpy())


@pytest.mark.parametrize('pinned', [True, False])
@pytest.mark.parametrize('band_rows', [8, 32, 50, 4096])
def test_host_fed_scene_equals_resident_scene(pinned, band_rows):
    """The scene in HOST memory (train.py:182-205's situation), uploaded band by band on the copy stream while tiles of the bands
    that arrived are predicted: the mask is the resident scene's mask bit for bit -- bands thinner than a tile, a ragged last
    band, one band for the whole scene; pinned sources (DMA in place) and pageable ones (staged through pinned buffers)."""
    c, h, w, p = 13, 300, 260, 64
    d1, d2 = _scene(c, h, w, 5)
    model, _ = _calibrated_model(c, 'bf16', d1, d2, p)
    t1, t2 = torch.from_numpy(d1), torch.from_numpy(d2)
    ref = inf.predict_scene(model, t1.cuda(), t2.cuda(), patch_size=p, batch_size=6)
    if pinned:
        t1, t2 = t1.pin_memory(), t2.pin_memory()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):                       # also from a non-default consumer stream
        got = inf.predict_scene(model, t1, t2, patch_size=p, batch_size=6, band_rows=band_rows)
    side.synchronize()
    assert torch.equal(got, ref)
    assert 0.02 < ref.float().mean().item() < 0.98


@pytest.mark.parametrize('host', [False, True])
def test_two_stream_scene_scan_equals_single_stream(host):
    """predict_scene alternates tile batches between the caller's stream and the library's second stream (each with its own workspace):
    the mask must equal the single-stream loop's bit for bit -- resident scenes and host scenes fed through the band feeder -- and
    the second lane's workspace must be gone from the engine afterwards (its memory returns to the caller's allocator pool)."""
    c, h, w, p = 13, 300, 260, 64
    d1, d2 = _scene(c, h, w, 6)
    model, _ = _calibrated_model(c, 'bf16', d1, d2, p)
    t1, t2 = torch.from_numpy(d1), torch.from_numpy(d2)
    if host:
        t1, t2 = t1.pin_memory(), t2.pin_memory()
    else:
        t1, t2 = t1.cuda(), t2.cuda()
    one = inf.predict_scene(model, t1, t2, patch_size=p, batch_size=6, two_streams=False, band_rows=32)
    for bs in (6, 7):                                          # equal batches, ragged last batch
        two = inf.predict_scene(model, t1, t2, patch_size=p, batch_size=bs, two_streams=True, band_rows=32)
        auto = inf.predict_scene(model, t1, t2, patch_size=p, batch_size=bs, band_rows=32)
        assert torch.equal(two, one) and torch.equal(auto, one)
    assert not [k for k in model.engine()._ws if k[4] == 1], 'second-lane workspaces must be dropped on exit'
    assert 0.02 < one.float().mean().item() < 0.98


def test_predict_scene_full_size_properties():
    """BASELINE config-5 shape at a bounded size (13 bands, 128-pixel tiles, 1000 x 900): the sharded scan equals
    the single scan, the scan is reproducible, and it equals the reference-style patch loop."""
    c, h, w, p = 13, 1000, 900, 128
    d1, d2 = _scene(c, h, w, 4)
    model = filler.fill_module(BiDateNet(c, 2, precision='bf16')).cuda().eval()
    t1, t2 = torch.from_numpy(d1).cuda(), torch.from_numpy(d2).cuda()
    a = inf.predict_scene(model, t1, t2, patch_size=p, batch_size=64)
    b = inf.predict_scene(model, t1, t2, patch_size=p, batch_size=64)
    assert torch.equal(a, b)
    parts = [inf.predict_scene(model, t1, t2, patch_size=p, batch_size=64, shard=(r, 3), merge=False) for r in range(3)]
    merged = torch.stack(parts).max(0)[0]
    assert torch.equal(merged, a)
    pa, hs, ws, lc, lr, _, _ = inf._get_patches(d1.transpose(1, 2, 0), p)
    pb = inf._get_patches(d2.transpose(1, 2, 0), p)[0]
    out = inf.predict_patches(model, np.ascontiguousarray(pa.transpose(0, 3, 1, 2)),
                              np.ascontiguousarray(pb.transpose(0, 3, 1, 2)), batch_size=64)
    assert np.array_equal(inf.full_image_mask(out, hs, ws, lc, lr, h, w, p).astype(np.uint8), a.cpu().numpy())


def test_predict_scene_at_the_full_baseline_size():
    """BASELINE configs[4] at its full size -- 13 bands, 10 000 x 10 000, 128-pixel tiles = 6 241 tiles, 10.4 GB of float32 scene
    planes resident in HBM -- through size-independent properties: the tile plan is the reference's (78 x 78 aligned tiles + 78 +
    78 edge-anchored + the corner), the scan is reproducible bit for bit, a 2-way sharded scan merges to the single scan, and the
    edge-anchored tiles cover the border (stitching an all-zero prediction over a poisoned mask leaves no pixel untouched)."""
    c, h, w, p = 13, 10000, 10000, 128
    g = torch.Generator(device='cuda').manual_seed(17)
    d1 = torch.randn(c, h, w, device='cuda', generator=g)
    d2 = d1 + 0.5 * torch.randn(c, h, w, device='cuda', generator=g)
    model = filler.fill_module(BiDateNet(c, 2, precision='bf16')).cuda().eval()
    o, hs, ws, lc, lr = inf.tile_origins(h, w, p)
    assert (hs, ws, lc, lr) == (78, 78, 78, 78) and len(o) == 6241
    a = inf.predict_scene(model, d1, d2, patch_size=p, batch_size=64)
    b = inf.predict_scene(model, d1, d2, patch_size=p, batch_size=64)
    assert a.shape == (h, w) and a.dtype == torch.uint8 and torch.equal(a, b)
    assert int(a.max()) <= 1
    parts = [inf.predict_scene(model, d1, d2, patch_size=p, batch_size=64, shard=(r, 2), merge=False) for r in range(2)]
    assert torch.equal(torch.maximum(parts[0], parts[1]), a)
    del parts, b
    full = torch.full((h, w), 255, dtype=torch.uint8, device='cuda')
    from fabric_amd._lib import call, ptr, stream_ptr
    origins = torch.from_numpy(o).cuda()
    logits = torch.zeros(64, 2, p, p, device='cuda')
    for i in range(0, len(o), 64):
        oo = origins[i:i + 64]
        call('bdn_argmax_stitch', ptr(logits), ptr(oo), ptr(full), oo.shape[0], 2, p, h, w, stream_ptr())
    assert int(full.max()) == 0                                   # every pixel of the scene is owned by some tile


def test_scene_errors():
    model = filler.fill_module(BiDateNet(3, 2, precision='fp32')).cuda()
    d = torch.zeros(3, 64, 64)
    with pytest.raises(RuntimeError, match='eval'):
        inf.predict_scene(model, d, d, patch_size=32)
    with pytest.raises(ValueError, match='smaller'):
        inf.predict_scene(model.eval(), d, d, patch_size=128)
    with pytest.raises(ImportError):
        inf.log_full_image(None)
