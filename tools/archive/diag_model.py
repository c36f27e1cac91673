"""Diagnostic (test infrastructure): per-layer forward and per-parameter gradient errors of the HIP
engine vs the CPU oracle on one small configuration.  Usage: python tools/archive/diag_model.py [prec] [C B S]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fabric_amd import BiDateNet
from fabric_amd.engine import build_layers
from oracle import filler, bidate_oracle as O

prec = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
C, B, S = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (3, 4, 32)
x1, x2, lbl = filler.make_inputs(B, C, S, seed=0)
x1, x2, lbl = map(torch.from_numpy, (x1, x2, lbl))
model = filler.fill_module(BiDateNet(C, 2, precision=prec))
sd0 = {k: v.clone() for k, v in model.state_dict().items()}
ref = O.train_step(sd0, x1, x2, lbl)
# oracle intermediates: raw conv outputs per layer
st = O.State(sd0)
zs = {}
def dc(prefix, x, name):
    for ci, bi, tag in ((0, 1, 'a'), (3, 4, 'b')):
        z = O.conv3x3(x, st.p(f'{prefix}.{ci}.weight'), st.p(f'{prefix}.{ci}.bias'))
        zs[name + tag] = z
        m, v = O.bn_batch_stats(z)
        x = torch.relu(O.bn_apply(z, m, v, st.p(f'{prefix}.{bi}.weight'), st.p(f'{prefix}.{bi}.bias')))
    return x
def enc(x, d):
    f = [dc('inc.conv.conv', x, f'e1{d}')]
    for k in range(1, 5):
        f.append(dc(f'down{k}.mpconv.1.conv', O.maxpool2(f[-1]), f'e{k+1}{d}'))
    return f
f1, f2 = enc(x1, 'x'), enc(x2, 'y')
fused = [torch.relu(b * a) for a, b in zip(f1, f2)]
x = fused[4]
for j in range(1, 5):
    x = dc(f'up{j}.conv.conv', torch.cat([fused[4 - j], O.pad_to(O.upsample2x_align(x), fused[4 - j])], 1), f'd{j}')

model = model.cuda().train()
logits = model(x1.cuda(), x2.cuda())
ws = list(model.engine()._ws.values())[0]
def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
print(f'== {prec} C={C} B={B} S={S}')
for L in build_layers(C):
    z = ws.z[L.name].float().cpu().permute(0, 3, 1, 2)
    if L.enc:
        r = torch.cat([zs[L.name[:2] + 'x' + L.name[2]], zs[L.name[:2] + 'y' + L.name[2]]])
    else:
        r = zs[L.name]
    print(f'fwd z {L.name}: relL2 {rel(z, r):.3e} max|d| {float((z - r).abs().max()):.3e} (scale {float(r.abs().max()):.2f})')
print(f'logits: max|d| {float((logits.detach().cpu() - ref["logits"]).abs().max()):.3e}')
from oracle.bidate_oracle import tversky_loss
lg = logits
true = lbl.cuda().long()
one_hot = torch.eye(2, device='cuda')[true].permute(0, 3, 1, 2)
p = torch.softmax(lg, 1)
inter = (p * one_hot).sum((0, 2)); fps = (p * (1 - one_hot)).sum((0, 2)); fns = ((1 - p) * one_hot).sum((0, 2))
loss = 1 - (inter / (inter + 0.1 * fps + 0.9 * fns + 1e-7)).mean()
loss.backward()
print(f'loss {loss.item():.6f} ref {float(ref["loss"]):.6f}')
for k, prm in model.named_parameters():
    g, r = prm.grad.cpu(), ref['grads'][k]
    print(f'grad {k:34s} relL2 {rel(g, r):.3e}  |ref| {float(r.norm()):.3e} |got| {float(g.norm()):.3e}')
