"""-m gpu: the whole optimisation loop.  Several consecutive fused train steps (forward, Tversky, backward,
SGD, BatchNorm running statistics) must follow the CPU oracle's trajectory, and the train.py-style epoch /
validation helpers must report the reference's metric definitions (val F1 = mean of per-batch binary F1)."""
import numpy as np
import pytest
import torch

from fabric_amd import BiDateNet
from fabric_amd.train_step import TrainStep
from oracle import bidate_oracle as O
from oracle import filler

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('prec,tol_logit,tol_loss', [('fp32', 2e-3, 2e-5), ('bf16', 0.3, 5e-3)])
def test_five_steps_follow_the_oracle(prec, tol_logit, tol_loss):
    c, b, s, steps, lr = 3, 4, 32, 5, 0.05          # large lr so that a wrong update would be visible
    x1, x2, lbl = filler.make_inputs(b, c, s, seed=3)
    x1, x2, lbl = torch.from_numpy(x1), torch.from_numpy(x2), torch.from_numpy(lbl)
    model = filler.fill_module(BiDateNet(c, 2, precision=prec))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.cuda().train()
    ts = TrainStep(model, lr=lr, tversky_alpha=0.1, tversky_beta=0.9)
    dx1, dx2, dl = x1.cuda(), x2.cuda(), lbl.cuda()
    for i in range(steps):
        ref = O.train_step(sd, x1, x2, lbl, lr=lr, alpha=0.1, beta=0.9)
        loss = ts.step(dx1, dx2, dl)
        got_logits = ts.last_logits.cpu()
        assert abs(loss.item() - float(ref['loss'])) < tol_loss * (i + 1), (i, loss.item(), float(ref['loss']))
        assert (got_logits - ref['logits']).abs().max() < tol_logit * (i + 1), i
        p_ref = O.binary_prf(lbl, ref['preds'])
        from fabric_amd.utils.metrics import batch_prf_from_counts
        p_got = batch_prf_from_counts(ts.last_counts.cpu())
        assert np.allclose(p_got, p_ref, atol=2e-2 if prec == 'bf16' else 2e-3), (i, p_got, p_ref)
        sd = ref['new_sd']
    new = model.state_dict()
    wtol = 2e-4 if prec == 'fp32' else 3e-2
    for k, v in sd.items():
        if v.is_floating_point():
            assert (new[k].cpu() - v).abs().max() <= wtol * max(1.0, float(v.abs().max())), k
        else:
            assert int(new[k]) == int(v), k          # num_batches_tracked: 2 per step in the encoder, 1 in the decoder


def test_epoch_and_validation_helpers():
    from fabric_amd.train import make_loaders, train_epoch, validate
    from fabric_amd.utils.dataloaders import synthetic_onera
    torch.manual_seed(0)
    data = synthetic_onera(n_cities=3, bands=13, size=(150, 150), seed=1, change_fraction=0.2)
    tr, va = make_loaders(data, ['city2'], 64, 32, 4, augmentation=True)
    model = BiDateNet(13, 2, precision='bf16').cuda()
    step = TrainStep(model, lr=0.05, tversky_alpha=0.1, tversky_beta=0.9)
    dev = torch.device('cuda')
    m0 = train_epoch(step, tr, dev, 64)
    m1 = train_epoch(step, tr, dev, 64)
    assert set(m0) == {'cd_losses', 'cd_corrects', 'cd_precisions', 'cd_recalls', 'cd_f1scores'}
    assert m1['cd_losses'] < m0['cd_losses']                    # it learns the synthetic change blobs
    from fabric_amd.utils.metrics import TverskyLoss
    v = validate(model, va, dev, 64, TverskyLoss(alpha=0.1, beta=0.9))
    assert 0.0 <= v['cd_f1scores'] <= 1.0 and 0.0 <= v['cd_corrects'] <= 100.0
    sd = model.state_dict()
    assert int(sd['inc.conv.conv.1.num_batches_tracked']) == 2 * 2 * len(tr)   # eval passes leave the buffers alone


@pytest.mark.parametrize('loss_function', ['tversky', 'dice'])
def test_train_main_runs_an_epoch_and_writes_the_best_checkpoint(tmp_path, loss_function, capsys):
    """python -m fabric_amd.train on synthetic cities: one epoch (fused step for tversky, the autograd path for the other
    criteria), validation metrics in the reference's names, and train.py:207-227's artefacts -- the pickled module and the
    metadata JSON -- which reload into a model that reproduces the validation logits."""
    import json
    from fabric_amd import train as T
    T.main(['--synthetic', '--epochs', '1', '--batch_size', '8', '--patch_size', '64', '--stride', '128', '--num_workers', '0',
            '--learning_rate', '0.02', '--loss_function', loss_function, '--log_dir', str(tmp_path)])
    line = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert line['epoch'] == 0 and {'train_cd_losses', 'validate_cd_f1scores', 'validate_cd_corrects'} <= set(line)
    assert 0.0 <= line['validate_cd_f1scores'] <= 1.0
    meta = json.load(open(tmp_path / 'metadata_epoch_0.json'))
    assert meta['loss_function'] == loss_function and 'validation_metrics' in meta and meta['world_size'] == 1
    model = torch.load(tmp_path / 'checkpoint_epoch_0.pt', weights_only=False)
    assert isinstance(model, BiDateNet) and len(model.state_dict()) == 128
    model = model.cuda().eval()
    x = torch.randn(2, 13, 64, 64, device='cuda')
    with torch.no_grad():
        a = model(x, x.flip(0))
    ref = BiDateNet(13, 2, precision=model.precision).cuda().eval()
    ref.load_state_dict(model.state_dict())
    with torch.no_grad():
        b = ref(x, x.flip(0))
    assert torch.equal(a, b)
    with pytest.raises(SystemExit):
        T.main(['--synthetic', '--loss_function', 'bce'])


def test_three_train_steps_same_speed():
    """Every TrainStep of a process runs on the same library-created streams (fabric_amd/streams.py): the third instance is as
    fast as the first (round 2: successive instances took successive pool streams, some of which share a hardware queue --
    6.65 vs 7.2 ms).  Benchmark shape, bf16; medians of three interleaved rounds within 2 %."""
    import statistics
    from fabric_amd import streams
    B = 64
    x1 = torch.randn(B, 13, 128, 128, device='cuda'); x2 = torch.randn(B, 13, 128, 128, device='cuda')
    lbl = (torch.rand(B, 128, 128, device='cuda') < 0.1).to(torch.uint8)
    steps = []
    for i in range(3):
        torch.manual_seed(i)
        m = BiDateNet(13, 2, precision='bf16').cuda().train()
        steps.append(TrainStep(m, lr=1e-3))
        torch.cuda.Stream(); torch.cuda.Stream(priority=-1)     # an application that keeps taking pool streams in between
    assert len({s.stream().cuda_stream for s in steps}) == 1 and steps[0].stream().cuda_stream == streams.get('chain').cuda_stream
    assert len({s.model.engine()._side_stream(x1.device).cuda_stream for s in steps}) == 1
    assert streams.get('chain').cuda_stream != streams.get('wgrad').cuda_stream != streams.get('copy').cuda_stream
    times = [[] for _ in steps]
    for rnd in range(4):
        for i, s in enumerate(steps):
            with torch.cuda.stream(s.stream()):
                for _ in range(3): s.step(x1, x2, lbl)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): s.step(x1, x2, lbl)
                e1.record(); torch.cuda.synchronize()
            if rnd:                                                  # round 0 warms the workspaces up
                times[i].append(e0.elapsed_time(e1) / 10)
    med = [statistics.median(t) for t in times]
    assert max(med) <= 1.02 * min(med), med


@pytest.mark.parametrize('prec', ['bf16', 'fp32'])
@pytest.mark.parametrize('shape', [(13, 6, 128, 128), (3, 4, 40, 72), (13, 2, 90, 90)])
def test_two_chain_forward_is_bit_identical(prec, shape):
    """engine.fwd_chains = 2 runs the two dates of encoder levels 1-3 as two chains on two streams (reference order of the dates,
    models/bidate_model.py:23-33, is kept for the running statistics by events).  It must change NO bit: logits, loss, every BatchNorm
    table and buffer, every gradient and the parameters after three SGD steps equal the one-chain schedule's, for 1..4 split levels."""
    c, b, h, w = shape
    x1, x2, lbl = filler.make_inputs(b, c, h, seed=5, size_w=w)
    x1, x2, lbl = torch.from_numpy(x1).cuda(), torch.from_numpy(x2).cuda(), torch.from_numpy(lbl).cuda()

    def run(chains, levels):
        model = filler.fill_module(BiDateNet(c, 2, precision=prec)).cuda().train()
        eng = model.engine()
        eng.fwd_chains, eng.fwd_chain_levels = chains, levels
        ts = TrainStep(model, lr=0.05)
        out = []
        for _ in range(3):
            loss = ts.step(x1, x2, lbl)
            ws = eng.workspace(b, h, w, x1.device)
            out += [loss.clone(), ts.last_logits.clone(), ts.flat_grads.clone()] + [ws.bn[L.name].clone() for L in eng.layers]
            out += [ws.f[k].clone() for k in range(1, 6)] + [ws.pool[k].clone() for k in range(2, 6)]
        torch.cuda.synchronize()
        return out + [v.clone() for v in model.state_dict().values()]

    ref = run(1, 3)
    for levels in (3, 1, 2, 4):
        got = run(2, levels)
        assert len(got) == len(ref)
        for i, (a_, r_) in enumerate(zip(got, ref)):
            assert torch.equal(a_, r_), (levels, i)


def test_bn_backward_folded_into_the_data_gradient_conv_matches_the_separate_pass():
    """engine.fold_bn_bwd: for the listed 64-channel layers BatchNorm+ReLU backward is applied while the data-gradient conv stages its
    operand (bdn_conv3x3_dgrad_bb, dz = a g + b z + c) instead of by the bdn_bn_bwd_apply pass.  Same mathematics, different rounding
    points (bf16 dz differs by single steps in a few percent of the entries): loss identical, every gradient within 1 % relative L2 of
    the unfolded step and the whole gradient vector at cosine >= 0.99999 (autograd of models/unet_parts.py:14-15,17-18)."""
    c, b, s = 13, 4, 128
    x1, x2, lbl = filler.make_inputs(b, c, s, seed=9)
    x1, x2, lbl = torch.from_numpy(x1).cuda(), torch.from_numpy(x2).cuda(), torch.from_numpy(lbl).cuda()

    def run(fold):
        model = filler.fill_module(BiDateNet(c, 2, precision='bf16')).cuda().train()
        model.engine().fold_bn_bwd = fold
        ts = TrainStep(model, lr=0.0)
        loss = ts.step(x1, x2, lbl)
        torch.cuda.synchronize()
        return loss.item(), ts.flat_grads.clone(), {k: v.clone() for k, v in ts.grads.items()}

    l0, g0, d0 = run(())
    for fold in (('e1b',), ('e1b', 'd4a', 'd3a', 'd3b')):
        l1, g1, d1 = run(fold)
        assert l1 == l0
        cos = float((g0.double() * g1.double()).sum() / (g0.double().norm() * g1.double().norm()))
        worst = max(float((d1[k] - d0[k]).norm() / (d0[k].norm() + 1e-30)) for k in d0 if float(d0[k].norm()) > 1e-6)
        print(f'fold {fold}: gradient cosine {cos:.7f}, worst per-parameter relative L2 {worst:.2e}')
        assert cos >= 0.99999 and worst < 1e-2, (fold, cos, worst)
