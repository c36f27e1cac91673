"""-m gpu: unusual problem shapes through the drop-in module surface in every numerics setting -- one pair, the smallest map the four poolings
allow (16 x 16), odd and very wide maps, odd batches, non-standard band counts -- in train mode (logits against the CPU oracle of
models/bidate_model.py:22-40, finite gradients) and in eval mode (the eval-shaped schedule against the training kernels on running
statistics, the class map of train.py:199 out of the last epilogue).  Round 6: a 3-class, non-square eval forward had found bdn_argmax refusing
H != W; this sweep keeps the corners of the shape space under test."""
import pytest
import torch

from fabric_amd import BiDateNet
from oracle import bidate_oracle as O
from oracle import filler

pytestmark = pytest.mark.gpu

SHAPES = [(1, 3, 16, 16), (1, 13, 17, 31), (5, 3, 33, 16), (2, 13, 16, 130), (3, 4, 48, 24)]


@pytest.mark.parametrize('shape', SHAPES)
def test_edge_shapes_train_and_eval(shape):
    B, C, H, W = shape
    x1, x2, lbl = filler.make_inputs(B, C, H, seed=1, size_w=W)
    x1, x2, lbl = torch.from_numpy(x1), torch.from_numpy(x2), torch.from_numpy(lbl)
    sd = {k: v.clone() for k, v in filler.fill_module(BiDateNet(C, 2, precision='fp32')).state_dict().items()}
    ref = O.train_step(sd, x1, x2, lbl, lr=1e-3, alpha=0.1, beta=0.9)
    for prec in ('fp32', 'bf16x3', 'bf16x3-fast', 'bf16'):
        m = filler.fill_module(BiDateNet(C, 2, precision=prec)).cuda().train()
        lg = m(x1.cuda(), x2.cuda())
        torch.nn.functional.cross_entropy(lg, lbl.cuda().long()).backward()
        d = (lg.detach().cpu() - ref['logits']).abs().max().item()
        assert d < (0.3 if prec == 'bf16' else 1e-3), (prec, d)
        assert all(torch.isfinite(p.grad).all() for p in m.parameters()), prec
        m.eval()
        eng = m.engine()
        with torch.no_grad():
            ev = m(x1.cuda(), x2.cuda())
            cd, _ = eng.forward(x1.cuda(), x2.cuda(), {k: v.detach() for k, v in m.state_dict(keep_vars=True).items()}, training=False, class_map=True)
            eng.eval_fused = not eng.eval_fused
            ev2 = m(x1.cuda(), x2.cuda())
        rel = (ev - ev2).abs().max().item() / max(1e-6, ev.abs().max().item())
        assert rel < (3e-2 if prec == 'bf16' else 1e-4), (prec, rel)       # (bf16x3 has one eval path: identical)
        assert tuple(cd.shape) == (B, H, W) and torch.equal(cd, (ev[:, 1] > ev[:, 0]).to(torch.uint8))
