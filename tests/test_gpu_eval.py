"""-m gpu parity tests of the eval-shaped forward (round 6): the one-launch conv -> BatchNorm(eval) -> ReLU stage with its fused
consumers, through the C ABI against the CPU oracle, and the whole schedule against the round 1-5 eval path and the golden vectors.

Reference: model.eval() forward as validation and full-scene inference run it (train.py:125-172, 182-205; models/unet_parts.py:13-18,40;
models/bidate_model.py:35-38).  Tolerances as tests/test_gpu_kernels.py: fp32 2e-5 of the result's magnitude, bf16 1e-2 (output rounding).
"""
import os

import numpy as np
import pytest
import torch

from fabric_amd import BiDateNet, _lib
from fabric_amd.utils import inference as inf
from oracle import bidate_oracle as O
from oracle import filler
from tests.gpu_util import DT, assert_close, dev, from_nhwc, pack_w, rnd, st, to_nhwc

pytestmark = pytest.mark.gpu
TOL = {'fp32': 2e-5, 'bf16': 1e-2}
PRECS = ['fp32', 'bf16']


def _rand(shape, seed, scale=1.0):
    return torch.from_numpy((np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32))


def _fold(gamma, beta, rm, rv, bias, eps=1e-5):
    scale = gamma / torch.sqrt(rv + eps)
    return scale, bias * scale + (beta - rm * scale)


def test_bn_eval_fold_multi():
    import struct
    Cs = [64, 512, 128]
    recs, outs, refs, keep = b'', [], [], []
    for i, C in enumerate(Cs):
        g, b, rm, rv = _rand((C,), 10 + i).abs() + 0.5, _rand((C,), 20 + i), _rand((C,), 30 + i), _rand((C,), 40 + i).abs() + 0.1
        bias = _rand((C,), 50 + i) if i != 1 else None
        d = [dev(t) for t in (g, b, rm, rv)] + [dev(bias) if bias is not None else None]
        out = torch.full((2, C), float('nan'), device='cuda')
        keep += d + [out]
        recs += struct.pack('<QQQQQQii', *[t.data_ptr() if t is not None else 0 for t in d], out.data_ptr(), C, 0)
        sc, sh = _fold(g.double(), b.double(), rm.double(), rv.double(), bias.double() if bias is not None else torch.zeros(C).double())
        outs.append(out); refs.append((sc.float(), sh.float()))
    desc = torch.frombuffer(bytearray(recs), dtype=torch.uint8).cuda()
    _lib.call('bdn_bn_eval_fold_multi', desc.data_ptr(), len(Cs), max(Cs), 1e-5, st())
    torch.cuda.synchronize()
    for out, (sc, sh) in zip(outs, refs):
        assert_close('scale', out[0].cpu(), sc, 1e-6)
        assert_close('shift', out[1].cpu(), sh, 1e-6, 1e-6)


EVAL_CASES = [
    # N, H, W, C0real, C0, C1, Cout, mul, pool
    (4, 32, 32, 13, 16, 0, 64, False, False),     # first layer: 13 bands padded to 16
    (2, 32, 32, 64, 64, 0, 64, True, True),       # single-chunk 16x16 tiles: product + pooling (encoder level 1, date 2)
    (2, 16, 16, 64, 64, 0, 128, False, True),     # date 1 of a level: activation + pooled map
    (2, 24, 20, 64, 64, 64, 64, False, False),    # two-source K loop, ragged tiles
    (4, 8, 8, 128, 128, 0, 128, True, False),     # 8x8 maps: two images per tile (level 5: product only)
    (3, 8, 8, 128, 128, 0, 128, True, True),      # odd image count: one image per tile on an 8x8 map
    (1, 11, 45, 64, 64, 0, 192, True, True),      # odd sizes: floor pooling drops the last row / column, Cout not a multiple of 128
    (2, 22, 45, 128, 128, 0, 256, True, True),    # ragged 8x16 tiles, BN = 128 column tiles
    (8, 64, 64, 64, 64, 0, 64, True, True),       # 128 tiles
    (2, 6, 6, 256, 256, 0, 64, False, True),      # tiny map
]


@pytest.mark.parametrize('prec', PRECS)
@pytest.mark.parametrize('case', EVAL_CASES)
def test_conv3x3_eval_stage(prec, case):
    N, H, W, c0r, C0, C1, Cout, use_mul, use_pool = case
    dt, td = DT[prec]
    x0 = rnd(prec, _rand((N, C0, H, W), 1))
    x0[:, c0r:] = 0
    x1 = rnd(prec, _rand((N, C1, H, W), 2)) if C1 else None
    w = _rand((Cout, c0r + C1, 3, 3), 3, (2.0 / (9 * (c0r + C1))) ** 0.5)
    bias = _rand((Cout,), 4, 0.1)
    gamma, beta = _rand((Cout,), 6).abs() + 0.5, _rand((Cout,), 7, 0.3)
    rm, rv = _rand((Cout,), 8, 0.2), _rand((Cout,), 9).abs() + 0.5
    sc, sh = _fold(gamma, beta, rm, rv, bias)
    other = rnd(prec, _rand((N, Cout, H, W), 11).abs()) if use_mul else None
    # ---- oracle: conv -> BatchNorm(eval) -> ReLU -> storage rounding; product / pooling of the rounded activation
    a = torch.cat([x0[:, :c0r], x1], 1) if C1 else x0[:, :c0r]
    z = O.conv3x3(a, rnd(prec, w), None)
    act = rnd(prec, torch.relu(z * sc[None, :, None, None] + sh[None, :, None, None]))
    out_ref = rnd(prec, act * other) if use_mul else act
    pool_ref = O.maxpool2(act) if use_pool else None
    # ---- device
    wp = torch.zeros(Cout, C0 + C1, 3, 3)
    wp[:, :c0r] = w[:, :c0r]
    if C1:
        wp[:, C0:] = w[:, c0r:]
    wf, _ = pack_w(prec, wp, C0 + C1)
    d0, d1 = to_nhwc(prec, x0), (to_nhwc(prec, x1) if C1 else None)
    dm = to_nhwc(prec, other) if use_mul else None
    out = torch.full((N, H, W, Cout), float('nan'), dtype=td, device='cuda')
    pool = torch.full((N, H // 2, W // 2, Cout), float('nan'), dtype=td, device='cuda') if use_pool else None
    dsc, dsh = dev(sc), dev(sh)
    _lib.call('bdn_conv3x3_eval', dt, d0.data_ptr(), C0, d1.data_ptr() if C1 else None, C1, wf.data_ptr(), dsc.data_ptr(), dsh.data_ptr(),
              out.data_ptr(), dm.data_ptr() if use_mul else None, pool.data_ptr() if use_pool else None, N, H, W, Cout, st())
    torch.cuda.synchronize()
    assert_close('stage out', from_nhwc(out), out_ref, TOL[prec], 1e-6)
    if use_pool:
        assert_close('pooled', from_nhwc(pool), pool_ref, TOL[prec], 1e-6)
        if not use_mul:            # the pooled map is exactly the maximum of the stored activations (same rounded values)
            assert torch.equal(from_nhwc(pool), O.maxpool2(from_nhwc(out)))


PAIR_CASES = [
    # B, H, W, C, Cout, pool
    (2, 32, 32, 64, 64, True),        # encoder level 1: single-chunk 8x16x2 tiles
    (3, 22, 45, 64, 64, True),        # ragged, odd sizes
    (2, 16, 16, 128, 128, True),      # 8x8x2 tiles, 64-wide column tiles (small grid)
    (40, 16, 16, 128, 128, True),     # 8x8x2 tiles, 128-wide column tiles, 1x4 waves
    (3, 20, 11, 256, 256, True),      # ragged 8x8x2 tiles
    (2, 8, 8, 512, 512, False),       # encoder level 5: 8x8 maps, product only
    (70, 8, 8, 128, 128, False),      # 8x8 maps, 128-wide column tiles, 2x2 waves
    (1, 5, 5, 64, 64, True),          # tiny odd map
    (2, 24, 24, 128, 64, True),       # several chunks into 64 channels: 8x16x2 tiles, 4x1 waves
]


@pytest.mark.parametrize('prec', PRECS)
@pytest.mark.parametrize('case', PAIR_CASES)
def test_conv3x3_eval_pair(prec, case):
    """Date-paired stage against the oracle, and bit for bit against the per-date form (bdn_conv3x3_eval with mul / pool)."""
    B, H, W, C, Cout, use_pool = case
    dt, td = DT[prec]
    x = rnd(prec, _rand((2 * B, C, H, W), 1))
    w = _rand((Cout, C, 3, 3), 3, (2.0 / (9 * C)) ** 0.5)
    sc, sh = _rand((Cout,), 5).abs() + 0.5, _rand((Cout,), 6, 0.3)
    act = rnd(prec, torch.relu(O.conv3x3(x, rnd(prec, w), None) * sc[None, :, None, None] + sh[None, :, None, None]))
    f_ref = rnd(prec, act[:B] * act[B:])
    pool_ref = O.maxpool2(act) if use_pool else None
    wf, _ = pack_w(prec, w, C)
    dx, dsc, dsh = to_nhwc(prec, x), dev(sc), dev(sh)
    f = torch.full((B, H, W, Cout), float('nan'), dtype=td, device='cuda')
    pool = torch.full((2 * B, H // 2, W // 2, Cout), float('nan'), dtype=td, device='cuda') if use_pool else None
    _lib.call('bdn_conv3x3_eval_pair', dt, dx.data_ptr(), C, wf.data_ptr(), dsc.data_ptr(), dsh.data_ptr(), f.data_ptr(),
              pool.data_ptr() if use_pool else None, B, H, W, Cout, st())
    torch.cuda.synchronize()
    assert_close('skip product', from_nhwc(f), f_ref, TOL[prec], 1e-6)
    if use_pool:
        assert_close('pooled', from_nhwc(pool), pool_ref, TOL[prec], 1e-6)
    # per-date form: date 1 stores its activation (+ pool), date 2 multiplies with it in the copy-out
    a1 = torch.empty(B, H, W, Cout, dtype=td, device='cuda')
    f2 = torch.empty_like(f)
    pool2 = torch.empty_like(pool) if use_pool else None
    _lib.call('bdn_conv3x3_eval', dt, dx[:B].data_ptr(), C, None, 0, wf.data_ptr(), dsc.data_ptr(), dsh.data_ptr(), a1.data_ptr(), None,
              pool2[:B].data_ptr() if use_pool else None, B, H, W, Cout, st())
    _lib.call('bdn_conv3x3_eval', dt, dx[B:].data_ptr(), C, None, 0, wf.data_ptr(), dsc.data_ptr(), dsh.data_ptr(), f2.data_ptr(), a1.data_ptr(),
              pool2[B:].data_ptr() if use_pool else None, B, H, W, Cout, st())
    torch.cuda.synchronize()
    assert torch.equal(f2, f)                 # the same MFMA sequence per output pixel in both forms: the same bits
    if use_pool:
        assert torch.equal(pool2, pool)


@pytest.mark.parametrize('prec', PRECS)
@pytest.mark.parametrize('shape', [(3, 32, 32), (2, 20, 45), (5, 16, 16)])
@pytest.mark.parametrize('ncls', [2, 1])
def test_conv3x3_eval_classifier_epilogue(prec, shape, ncls):
    """Last decoder stage with the classifier in its epilogue: logits bit-identical to bdn_outc_fwd on the stored activation,
    class map = first maximum, both against the oracle's conv1x1."""
    N, H, W = shape
    C = 64
    dt, td = DT[prec]
    x = rnd(prec, _rand((N, C, H, W), 1))
    w = _rand((C, C, 3, 3), 3, (2.0 / (9 * C)) ** 0.5)
    sc, sh = _rand((C,), 5).abs() + 0.5, _rand((C,), 6, 0.3)
    cw, cb = _rand((ncls, C, 1, 1), 7, 0.2), _rand((ncls,), 8, 0.1)
    act_ref = rnd(prec, torch.relu(O.conv3x3(x, rnd(prec, w), None) * sc[None, :, None, None] + sh[None, :, None, None]))
    logit_ref = O.conv1x1(act_ref, cw, cb)
    wf, _ = pack_w(prec, w, C)
    d0, dsc, dsh, dcw, dcb = to_nhwc(prec, x), dev(sc), dev(sh), dev(cw.reshape(ncls, C)), dev(cb)
    act = torch.full((N, H, W, C), float('nan'), dtype=td, device='cuda')
    logits = torch.full((N, ncls, H, W), float('nan'), device='cuda')
    mask = torch.full((N, H, W), 255, dtype=torch.uint8, device='cuda')
    _lib.call('bdn_conv3x3_eval_cls', dt, d0.data_ptr(), C, wf.data_ptr(), dsc.data_ptr(), dsh.data_ptr(), act.data_ptr(),
              dcw.data_ptr(), dcb.data_ptr(), ncls, logits.data_ptr(), mask.data_ptr(), None, 0, 0, N, H, W, C, st())
    torch.cuda.synchronize()
    assert_close('activation', from_nhwc(act), act_ref, TOL[prec], 1e-6)
    assert_close('logits', logits.cpu(), logit_ref, 5e-5 if prec == 'fp32' else 1e-2, 1e-5)
    # the stand-alone classifier on the stored activation (identity BatchNorm table): the same bits
    ident = torch.zeros(1, 4, C, device='cuda'); ident[:, 1:3] = 1.0
    l2 = torch.empty_like(logits)
    _lib.call('bdn_outc_fwd', dt, act.data_ptr(), ident.data_ptr(), dcw.data_ptr(), dcb.data_ptr(), l2.data_ptr(), N, H, W, C, ncls, st())
    torch.cuda.synchronize()
    assert torch.equal(l2, logits)
    want = torch.max(logits, 1)[1].to(torch.uint8) if ncls > 1 else torch.zeros(N, H, W, dtype=torch.uint8, device='cuda')
    if ncls > 1:    # torch.max returns the first maximum on ties as well
        want = (logits[:, 1] > logits[:, 0]).to(torch.uint8)
    assert torch.equal(mask, want)
    # logits / activation are optional outputs
    mask2 = torch.full_like(mask, 255)
    _lib.call('bdn_conv3x3_eval_cls', dt, d0.data_ptr(), C, wf.data_ptr(), dsc.data_ptr(), dsh.data_ptr(), None,
              dcw.data_ptr(), dcb.data_ptr(), ncls, None, mask2.data_ptr(), None, 0, 0, N, H, W, C, st())
    torch.cuda.synchronize()
    assert torch.equal(mask2, mask)


def test_conv3x3_eval_classifier_stitches_like_argmax_stitch():
    """Scene stitching from the epilogue == bdn_argmax_stitch on the same logits (ownership rule of utils/inference.py:187-236)."""
    p, Hs, Ws, C = 32, 88, 75, 64
    o_np = inf.tile_origins(Hs, Ws, p)[0]
    n = len(o_np)
    x = rnd('bf16', _rand((n, C, p, p), 1))
    w = _rand((C, C, 3, 3), 3, (2.0 / (9 * C)) ** 0.5)
    sc, sh = _rand((C,), 5).abs() + 0.5, _rand((C,), 6, 0.3)
    cw, cb = _rand((2, C), 7, 0.2), _rand((2,), 8, 0.1)
    dt, td = DT['bf16']
    wf, _ = pack_w('bf16', w, C)
    d0, dsc, dsh, dcw, dcb = to_nhwc('bf16', x), dev(sc), dev(sh), dev(cw), dev(cb)
    origins = torch.from_numpy(o_np).cuda()
    logits = torch.empty(n, 2, p, p, device='cuda')
    m_fused = torch.full((Hs, Ws), 255, dtype=torch.uint8, device='cuda')
    _lib.call('bdn_conv3x3_eval_cls', dt, d0.data_ptr(), C, wf.data_ptr(), dsc.data_ptr(), dsh.data_ptr(), None,
              dcw.data_ptr(), dcb.data_ptr(), 2, logits.data_ptr(), m_fused.data_ptr(), origins.data_ptr(), Hs, Ws, n, p, p, C, st())
    m_ref = torch.full((Hs, Ws), 255, dtype=torch.uint8, device='cuda')
    _lib.call('bdn_argmax_stitch', logits.data_ptr(), origins.data_ptr(), m_ref.data_ptr(), n, 2, p, Hs, Ws, st())
    torch.cuda.synchronize()
    assert int((m_ref == 255).sum()) == 0
    assert torch.equal(m_fused, m_ref)


def test_eval_stage_argument_errors():
    t = torch.zeros(1, 8, 8, 64, dtype=torch.bfloat16, device='cuda')
    f = torch.zeros(64, device='cuda')
    with pytest.raises(RuntimeError, match='null pointer'):
        _lib.call('bdn_conv3x3_eval', 1, t.data_ptr(), 64, None, 0, t.data_ptr(), None, f.data_ptr(), t.data_ptr(), None, None, 1, 8, 8, 64, st())
    with pytest.raises(RuntimeError, match='multiple of 64'):
        _lib.call('bdn_conv3x3_eval', 1, t.data_ptr(), 64, None, 0, t.data_ptr(), f.data_ptr(), f.data_ptr(), t.data_ptr(), None, None, 1, 8, 8, 96, st())
    with pytest.raises(RuntimeError, match='bad dtype'):
        _lib.call('bdn_conv3x3_eval', 2, t.data_ptr(), 64, None, 0, t.data_ptr(), f.data_ptr(), f.data_ptr(), t.data_ptr(), None, None, 1, 8, 8, 64, st())
    with pytest.raises(RuntimeError, match='must be 64'):
        _lib.call('bdn_conv3x3_eval_cls', 1, t.data_ptr(), 64, t.data_ptr(), f.data_ptr(), f.data_ptr(), None, f.data_ptr(), f.data_ptr(), 2,
                  f.data_ptr(), None, None, 0, 0, 1, 8, 8, 128, st())
    with pytest.raises(RuntimeError, match='1 or 2'):
        _lib.call('bdn_conv3x3_eval_cls', 1, t.data_ptr(), 64, t.data_ptr(), f.data_ptr(), f.data_ptr(), None, f.data_ptr(), f.data_ptr(), 3,
                  f.data_ptr(), None, None, 0, 0, 1, 8, 8, 64, st())


# ------------------------------------------------------------------ the whole schedule
def _load(golden_dir, name):
    g = np.load(os.path.join(golden_dir, f'{name}.npz'))
    c, b, s, sw, dd = [int(v) for v in g['meta']]
    x1, x2, _ = filler.make_inputs(b, c, s, seed=0, different_dates=bool(dd), size_w=sw)
    return g, c, torch.from_numpy(x1).cuda(), torch.from_numpy(x2).cuda()


@pytest.mark.parametrize('name', ['g1_c3_b4_s32', 'g4_c13_b2_s90', 'g2_c13_b2_s128', 'g8_c13_b3_h40_w72'])
@pytest.mark.parametrize('prec', ['fp32', 'bf16'])
def test_eval_schedule_matches_the_training_kernel_eval_path(golden_dir, name, prec):
    """model.eval() logits of the eval-shaped schedule against the round 1-5 path (training kernels on a running-statistics table) and,
    where the fixture holds them, the reference's eval logits.  fp32: the two differ by the association of one FMA per layer."""
    g, c, x1, x2 = _load(golden_dir, name)
    model = filler.fill_module(BiDateNet(c, 2, precision=prec)).cuda().eval()
    eng = model.engine()
    with torch.no_grad():
        assert eng.eval_fused
        new = model(x1, x2).cpu()
        eng.eval_fused = False
        old = model(x1, x2).cpu()
        eng.eval_fused = True
        cd, _ = eng.forward(x1, x2, {k: v.detach() for k, v in model.state_dict(keep_vars=True).items()}, training=False, class_map=True)
        eng.eval_pair = ()
        per_date = model(x1, x2).cpu()
        eng.eval_pair = (1, 2, 3, 4, 5)
    assert torch.equal(per_date, new)            # date-paired tiles and the per-date launches compute the same bits
    with torch.no_grad():
        pass
    scale = old.abs().max().item()
    d = (new - old).abs().max().item()
    print(f'\n[{name} {prec}] eval-shaped vs training-kernel eval path: max|dlogit|={d:.3e} of scale {scale:.1f}')
    assert d <= {'fp32': 2e-5, 'bf16': 3e-2}[prec] * scale
    if 'eval_logits' in g:
        ref = torch.from_numpy(g['eval_logits'])
        dn, do = (new - ref).abs().max().item(), (old - ref).abs().max().item()
        print(f'    vs reference: new {dn:.3e}  old {do:.3e}')
        assert dn <= {'fp32': 2e-5, 'bf16': 3e-2}[prec] * ref.abs().max().item()
    # the class map out of the epilogue is the argmax of the logits the same schedule returns
    assert torch.equal(cd.cpu(), (new[:, 1] > new[:, 0]).to(torch.uint8))
    sd = model.state_dict()
    assert all(int(sd[k]) == 0 for k in sd if 'num_batches_tracked' in k)


@pytest.mark.parametrize('prec', ['fp32', 'bf16'])
def test_eval_schedule_with_a_wider_head(prec):
    """n_classes = 3 (outconv, models/unet_parts.py:83-90, wider than the fused classifier epilogue takes): the eval-shaped schedule stores the last
    activation and runs the stand-alone classifier on it; logits against the training-kernel eval path, class map = first maximum (train.py:199)."""
    torch.manual_seed(5)
    model = BiDateNet(3, 3, precision=prec).cuda()
    x1, x2 = torch.randn(3, 3, 40, 56, device='cuda'), torch.randn(3, 3, 40, 56, device='cuda')
    model.train()
    with torch.no_grad():
        for _ in range(2):
            model(x1, x2)                       # running statistics away from their initial values
    model.eval()
    eng = model.engine()
    with torch.no_grad():
        new = model(x1, x2).cpu()
        cd, _ = eng.forward(x1, x2, {k: v.detach() for k, v in model.state_dict(keep_vars=True).items()}, training=False, class_map=True)
        eng.eval_fused = False
        old = model(x1, x2).cpu()
        eng.eval_fused = True
    assert new.shape == (3, 3, 40, 56)
    assert (new - old).abs().max().item() <= {'fp32': 2e-5, 'bf16': 3e-2}[prec] * old.abs().max().item()
    assert torch.equal(cd.cpu(), new.argmax(1).to(torch.uint8))


@pytest.mark.parametrize('prec', ['fp32', 'bf16'])
def test_scene_masks_of_both_eval_paths(prec):
    """predict_scene on the eval-shaped schedule against the round 1-5 path on the oracle test scene: fp32 equal outside logit ties,
    bf16 equal up to near-tie pixels (one rounding fewer per layer in the new schedule)."""
    from tests.test_gpu_scene import _calibrated_model, _scene
    c, h, w, p = 3, 88, 75, 32
    d1, d2 = _scene(c, h, w, 3)
    model, sd = _calibrated_model(c, prec, d1, d2, p)
    eng = model.engine()
    t1, t2 = torch.from_numpy(d1).cuda(), torch.from_numpy(d2).cuda()
    for bs in (4, 5, 64):
        eng.eval_fused = True
        new = inf.predict_scene(model, t1, t2, patch_size=p, batch_size=bs)
        one = inf.predict_scene(model, t1, t2, patch_size=p, batch_size=bs, two_streams=False)
        eng.eval_fused = False
        old = inf.predict_scene(model, t1, t2, patch_size=p, batch_size=bs)
        eng.eval_fused = True
        assert torch.equal(new, one)
        diff = (new != old).float().mean().item()
        print(f'\n[scene {prec} bs={bs}] pixels differing between the two eval paths: {diff:.5f}')
        assert diff <= (1e-3 if prec == 'fp32' else 2e-2)
