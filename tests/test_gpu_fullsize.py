"""-m gpu: the BASELINE configuration itself -- BiDateNet(13, 2), batch 64, 13 x 128 x 128 patch pairs, bf16 -- checked through
properties that do not need a reference run of that size (the CPU oracle takes ~15 s per step there; the small golden cases
pin the numbers, these pin the full-size launch shapes: the 16384-tile statistics reductions, the 256-block split-K weight
gradients, the fused first / last layer paths, the side stream):

* backward is LINEAR in dlogits, and scaling by a power of two is exact in bf16 and f32: grads(4 * dlogits) == 4 * grads(dlogits)
  bit for bit, for every parameter the deterministic kernels produce;
* two identical runs of three steps give the same bits (fixed-order reductions everywhere, no races between the two streams);
* swapping the dates leaves the train-mode logits unchanged (shared encoder, commutative fusion, per-date BatchNorm groups);
* eval-mode images are independent: the batch of 64 equals its two halves run separately;
* SGD on a fixed batch lowers the Tversky loss, and the gradients are finite and non-trivial everywhere.
"""
import pytest
import torch

from fabric_amd import BiDateNet
from fabric_amd.train_step import TrainStep

pytestmark = pytest.mark.gpu
B, C, S = 64, 13, 128


def _inputs(seed=0):
    g = torch.Generator(device='cpu').manual_seed(seed)
    x1 = torch.randn(B, C, S, S, generator=g)
    x2 = x1 + 0.3 * torch.randn(B, C, S, S, generator=g)
    lbl = (torch.rand(B, S, S, generator=g) < 0.1).to(torch.uint8)
    return x1.cuda(), x2.cuda(), lbl.cuda()


def _model(seed=1234):
    torch.manual_seed(seed)
    return BiDateNet(C, 2, precision='bf16').cuda().train()


def _P(model):
    return {k: v.detach() for k, v in model.state_dict(keep_vars=True).items()}


def _grads_like(model):
    return {k: torch.full_like(p, float('nan')) for k, p in model.named_parameters()}


def test_backward_is_exactly_linear_in_dlogits():
    x1, x2, _ = _inputs()
    model = _model()
    eng, P = model.engine(), _P(model)
    logits, ws = eng.forward(x1, x2, P, training=True)
    dl = torch.randn(logits.shape, generator=torch.Generator(device='cpu').manual_seed(5)).cuda() * 1e-3
    g1, g4 = _grads_like(model), _grads_like(model)
    eng.backward(ws, dl, P, g1)
    eng.backward(ws, 4.0 * dl, P, g4)
    torch.cuda.synchronize()
    for k in g1:
        assert torch.isfinite(g1[k]).all(), k
        assert torch.equal(g4[k], 4 * g1[k]), k
        if not (k.endswith('.bias') and k.split('.')[-2] in ('0', '3')):     # conv biases in front of a BatchNorm: zero
            assert g1[k].abs().max() > 0, k


def test_two_identical_steps_give_identical_bits():
    x1, x2, lbl = _inputs()
    out = []
    for _ in range(2):
        model = _model()
        ts = TrainStep(model, lr=1e-3, tversky_alpha=0.1, tversky_beta=0.9)
        losses = [ts.step(x1, x2, lbl).clone() for _ in range(3)]
        torch.cuda.synchronize()
        out.append((torch.stack(losses).cpu(), ts.last_logits.cpu().clone(),
                    {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}))
    # every reduction on the training path has a fixed order (no float atomics): three steps, same bits
    assert torch.equal(out[0][0], out[1][0])
    assert torch.equal(out[0][1], out[1][1])
    for k in out[0][2]:
        assert torch.equal(out[0][2][k], out[1][2][k]), k
    assert out[0][0][2] < out[0][0][0]                  # three SGD steps on one batch: the loss goes down


def test_swapping_the_dates_leaves_the_logits_unchanged():
    x1, x2, _ = _inputs()
    a, b = _model(), _model()
    la, _ = a.engine().forward(x1, x2, _P(a), training=True)
    lb, _ = b.engine().forward(x2, x1, _P(b), training=True)
    torch.cuda.synchronize()
    assert torch.equal(la, lb)


def test_eval_batch_equals_its_halves():
    x1, x2, _ = _inputs()
    model = _model().eval()
    eng, P = model.engine(), _P(model)
    full, _ = eng.forward(x1, x2, P, training=False)
    full = full.clone()
    h = B // 2
    lo, _ = eng.forward(x1[:h].contiguous(), x2[:h].contiguous(), P, training=False)
    lo = lo.clone()
    hi, _ = eng.forward(x1[h:].contiguous(), x2[h:].contiguous(), P, training=False)
    torch.cuda.synchronize()
    assert torch.equal(full[:h], lo) and torch.equal(full[h:], hi)
