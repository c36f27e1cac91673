"""CPU tests: the oracle (oracle/bidate_oracle.py) against the golden vectors captured from the reference
(tests/golden, oracle/make_golden.py).  This is what pins the oracle on any machine, GPU or not."""
import os

import numpy as np
import pytest
import torch

from oracle import bidate_oracle as O
from oracle import filler

torch.set_num_threads(8)


def _case(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    c, b, s, sw, dd = [int(v) for v in g['meta']]
    x1, x2, lbl = filler.make_inputs(b, c, s, seed=0, different_dates=bool(dd), size_w=sw)
    net = filler.fill_module(O.build_torch_baseline(c, 2))
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    return g, sd, torch.from_numpy(x1), torch.from_numpy(x2), torch.from_numpy(lbl)


@pytest.mark.parametrize('name', ['g1_c3_b4_s32', 'g6_c3_b4_s32_diffdates', 'g8_c13_b3_h40_w72'])
def test_train_step_matches_golden(golden_dir, name):
    g, sd, x1, x2, lbl = _case(golden_dir, name)
    o = O.train_step(sd, x1, x2, lbl, lr=1e-3, alpha=0.1, beta=0.9)
    assert np.abs(o['logits'].numpy() - g['logits']).max() < 5e-5
    assert abs(float(o['loss']) - float(g['loss'])) < 1e-6
    for k, grad in o['grads'].items():
        ref = g['gsamp/' + k]
        if float(g['gnorm/' + k]) < 1e-6:
            continue
        got = grad.reshape(-1)[torch.from_numpy(g['gidx/' + k])].numpy()
        assert np.linalg.norm(got - ref) <= 2e-2 * np.linalg.norm(ref) + 1e-9, k
    for k, v in o['new_sd'].items():
        if 'running_' in k:
            assert np.abs(v.numpy() - g['buf/' + k]).max() < 1e-5, k
        if 'num_batches_tracked' in k:       # the shared encoder BN modules run once per date
            assert int(v) == int(g['buf/' + k]) == (1 if k.startswith('up') else 2), k
    o2, _ = O.bidate_forward(o['new_sd'], x1, x2, training=True)
    assert np.abs(o2.detach().numpy() - g['logits_after_step']).max() < 1e-4


def test_eval_and_large_case_match_golden(golden_dir):
    for name in ('g1_c3_b4_s32', 'g4_c13_b2_s90', 'g2_c13_b2_s128'):
        g, sd, x1, x2, _ = _case(golden_dir, name)
        ev, _ = O.bidate_forward(sd, x1, x2, training=False)
        ref = g['eval_logits']
        assert np.abs(ev.numpy() - ref).max() <= 2e-6 * np.abs(ref).max() + 2e-5
        tr, _ = O.bidate_forward(sd, x1, x2, training=True)
        assert np.abs(tr.detach().numpy() - g['logits']).max() < 5e-5


def test_joint_batchnorm_would_be_wrong(golden_dir):
    """G6: dates drawn from different distributions.  Normalising both dates with ONE set of batch
    statistics (the tempting 2B-batch shortcut) moves the logits by O(1); the reference keeps per-date stats."""
    g, sd, x1, x2, _ = _case(golden_dir, 'g6_c3_b4_s32_diffdates')
    st = O.State(sd)
    both = torch.cat([x1, x2])
    z = O.conv3x3(both, st.p('inc.conv.conv.0.weight'), st.p('inc.conv.conv.0.bias'))
    m_joint, _ = O.bn_batch_stats(z)
    m_d1, _ = O.bn_batch_stats(z[:4])
    assert (m_joint - m_d1).abs().max() > 0.05


def test_losses_match_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g5_losses.npz'))
    logits, lbl3 = torch.from_numpy(g['logits']), torch.from_numpy(g['labels']).long()
    for rank, lbl in (('r3', lbl3), ('r4', lbl3[:, None])):
        assert abs(float(O.tversky_loss(logits, lbl, 0.1, 0.9)) - float(g[f'tversky_0.1_0.9_{rank}'])) < 1e-6
        assert abs(float(O.tversky_loss(logits, lbl, 0.5, 0.5)) - float(g[f'tversky_0.5_0.5_{rank}'])) < 1e-6
        assert abs(float(O.dice_loss(logits, lbl)) - float(g[f'dice_{rank}'])) < 1e-6
        assert abs(float(O.jaccard_loss(logits, lbl)) - float(g[f'jaccard_{rank}'])) < 1e-6
    # the (0,2)-dims quirk: [B,H,W] and [B,1,H,W] labels give different values
    assert abs(float(g['tversky_0.1_0.9_r3']) - float(g['tversky_0.1_0.9_r4'])) > 1e-4
    lg = logits.clone().requires_grad_(True)
    O.tversky_loss(lg, lbl3, 0.1, 0.9).backward()
    assert np.abs(lg.grad.numpy() - g['dlogits_tversky_r3']).max() < 1e-7


def test_tiling_matches_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g7_tiling.npz'))
    hs, ws, lc, lr, h, w = [int(v) for v in g['meta']]
    assert (hs, ws, lc, lr) == (2, 2, 2, 2)
    bands = np.random.default_rng(7)
    # regenerate the generator state exactly as make_golden did before the tiling case
    r = np.random.default_rng(7)
    r.standard_normal((2, 3, 40, 36)); r.uniform(0, 1, (40, 36)); r.standard_normal((2, 3, 30, 50)); r.uniform(0, 1, (30, 50))
    arr = r.standard_normal((300, 260, 13)).astype(np.float32)
    tiles, hs2, ws2, lc2, lr2, h2, w2 = O.tile_scene(arr, 128)
    assert (hs2, ws2, lc2, lr2, h2, w2) == (hs, ws, lc, lr, h, w) and tiles.shape[0] == 9
    assert np.allclose(tiles.reshape(9, -1).astype(np.float64).sum(1), g['tile_checksums'])
    img = O.stitch_scene(g['pred'].astype(np.float64), hs, ws, lc, lr, h, w, 128)
    assert np.array_equal(img.astype(np.uint8), g['stitched'])


def test_binary_prf_matches_sklearn():
    from sklearn.metrics import precision_recall_fscore_support as prfs
    r = np.random.default_rng(3)
    for p_pos in (0.3, 0.0):
        lab = (r.uniform(0, 1, 5000) < 0.2).astype(np.int64)
        pred = (r.uniform(0, 1, 5000) < p_pos).astype(np.int64)
        ref = prfs(lab, pred, average='binary', pos_label=1, zero_division=0)[:3]
        assert np.allclose(O.binary_prf(torch.from_numpy(lab), torch.from_numpy(pred)), ref)


def test_losses_more_match_golden(golden_dir):
    """G9: focal (all constructor forms), dice / jaccard / tversky values and gradients in both label ranks."""
    g = np.load(os.path.join(golden_dir, 'g9_losses_more.npz'))
    for tag in ('c2', 'c5'):
        logits = torch.from_numpy(g[f'{tag}/logits'])
        lbl3 = torch.from_numpy(g[f'{tag}/labels'].astype(np.int64))
        forms = [('g0', dict(gamma=0)), ('g2', dict(gamma=2)), ('g1.5_sum', dict(gamma=1.5, size_average=False))]
        forms += [('g2_a0.25', dict(gamma=2, alpha=0.25))] if tag == 'c2' else \
            [('g2_alist', dict(gamma=2, alpha=[0.1, 0.2, 0.3, 0.15, 0.25]))]
        for name, kw in forms:
            lg = logits.clone().requires_grad_(True)
            v = O.focal_loss(lg, lbl3, **kw)
            v.backward()
            ref = float(g[f'{tag}/focal_{name}'])
            assert abs(float(v) - ref) < 1e-5 * max(1.0, abs(ref))
            assert np.abs(lg.grad.numpy() - g[f'{tag}/dfocal_{name}']).max() < 1e-6 * max(1.0, np.abs(g[f'{tag}/dfocal_{name}']).max())
        for rank, lbl in (('r3', lbl3), ('r4', lbl3[:, None])):
            for name, fn in (('dice', O.dice_loss), ('jaccard', O.jaccard_loss),
                             ('tversky_0.3_0.7', lambda a, b: O.tversky_loss(a, b, 0.3, 0.7))):
                lg = logits.clone().requires_grad_(True)
                v = fn(lg, lbl)
                v.backward()
                assert abs(float(v) - float(g[f'{tag}/{name}_{rank}'])) < 1e-6
                assert np.abs(lg.grad.numpy() - g[f'{tag}/d{name}_{rank}']).max() < 1e-7


@pytest.mark.parametrize('name', ['g1_c3_b4_s32', 'g2_c13_b2_s128'])
def test_timed_cpu_baseline_graph_reproduces_the_reference_logits(golden_dir, name):
    """bench.py's cpu_baseline leg times oracle.build_torch_baseline (stock torch.nn modules assembled like
    models/bidate_model.py:22-40): its forward + Tversky + backward must be the reference's, not merely have its keys."""
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    c, b, s, sw, dd = [int(v) for v in g['meta']]
    x1, x2, lbl = filler.make_inputs(b, c, s, seed=0, different_dates=bool(dd), size_w=sw)
    net = filler.fill_module(O.build_torch_baseline(c, 2)).train()
    logits = net(torch.from_numpy(x1), torch.from_numpy(x2))
    assert np.abs(logits.detach().numpy() - g['logits']).max() < 5e-5
    loss = O.tversky_loss(logits, torch.from_numpy(lbl).long(), 0.1, 0.9)
    assert abs(float(loss) - float(g['loss'])) < 1e-6
    loss.backward()
    for k, p in net.named_parameters():
        if float(g['gnorm/' + k]) < 1e-6:
            continue
        got = p.grad.reshape(-1)[torch.from_numpy(g['gidx/' + k])].numpy()
        assert np.linalg.norm(got - g['gsamp/' + k]) <= 2e-2 * np.linalg.norm(g['gsamp/' + k]) + 1e-9, k
