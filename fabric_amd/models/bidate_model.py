"""BiDateNet: drop-in for the reference's models/bidate_model.py:7-40.

Same constructor, same ``forward(x_d1, x_d2)`` (two [B,C,H,W] float32 tensors ->
[B,n_classes,H,W] float32 logits that take part in autograd), same sub-module
tree / ``state_dict()`` schema, ``.train()/.eval()`` switch BatchNorm mode.  The
whole forward + backward runs on the gfx950 HIP library; there is no CPU path.
"""
import os

import torch
import torch.nn as nn

from .unet_parts import down, outconv, up, inconv
from ..engine import BiDateEngine


class _Lease:
    """Marks an engine workspace as owned by one autograd graph.  Released at the END of that graph's backward() -- not when
    the grad_fn node dies: in the usual loop `logits` / `loss` of the previous iteration are still bound when the next
    forward runs, and a lease held until then would make the loop ping-pong between two full workspaces -- or, without a
    backward, when the graph is freed.  A workspace carries a generation counter: a second backward through the same graph
    (retain_graph=True) after another forward reused the buffers raises instead of reading overwritten activations."""

    def __init__(self, ws):
        self.ws = ws
        ws.leased = True
        self.generation = ws.generation       # engine.forward bumped it when it filled the buffers

    def release(self):
        if self.ws is not None and self.ws.generation == self.generation:
            self.ws.leased = False
        self.ws_released = True

    def __del__(self):
        if not getattr(self, 'ws_released', False):
            self.release()


class _BiDateFunction(torch.autograd.Function):
    """One autograd node for the whole network: forward enqueues the fused HIP schedule, backward the
    hand-written backward schedule (no autograd tape through individual ops)."""

    @staticmethod
    def forward(ctx, module, x_d1, x_d2, *params):
        eng = module.engine()
        P = dict(module.state_dict(keep_vars=True))
        training = module.training
        logits, ws = eng.forward(x_d1.detach(), x_d2.detach(), {k: v.detach() for k, v in P.items()}, training)
        ctx.module, ctx.ws, ctx.training = module, ws, training
        ctx.n_params = len(params)
        if training and any(ctx.needs_input_grad):
            ctx.lease = _Lease(ws)            # nobody else may run a forward on these buffers while this graph lives
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        module = ctx.module
        if not ctx.training:
            raise RuntimeError('fabric_amd: backward through an eval-mode forward is not supported '
                               '(BatchNorm used running statistics); call model.train() first')
        eng = module.engine()
        named = list(module.named_parameters())
        P = {k: v.detach() for k, v in module.state_dict(keep_vars=True).items()}
        grads = {k: torch.empty_like(p, dtype=torch.float32) for k, p in named}
        lease = getattr(ctx, 'lease', None)
        if lease is not None and ctx.ws.generation != lease.generation:
            raise RuntimeError('fabric_amd: this graph\'s activations were overwritten by a later forward (second backward with '
                               'retain_graph=True after another forward of the same shape): re-run the forward')
        eng.backward(ctx.ws, dlogits, P, grads)
        if lease is not None:
            lease.release()                   # the activations are dead: the next forward of this shape reuses the workspace
        return (None, None, None) + tuple(grads[k] for k, _ in named)


class BiDateNet(nn.Module):
    def __init__(self, n_channels, n_classes, precision=None):
        super(BiDateNet, self).__init__()
        self.inc = inconv(n_channels, 64)
        self.down1 = down(64, 128)
        self.down2 = down(128, 256)
        self.down3 = down(256, 512)
        self.down4 = down(512, 512)

        self.up1 = up(1024, 256)
        self.up2 = up(512, 128)
        self.up3 = up(256, 64)
        self.up4 = up(128, 64)
        self.outc = outconv(64, n_classes)

        self.n_channels, self.n_classes = n_channels, n_classes
        # 'bf16' = throughput setting; 'bf16x3' = float32 tensors with split bf16 GEMM operands (logits within 1e-3 of the reference at
        # matrix-core speed); 'fp32' = exact f32 MFMA
        self.precision = precision or os.environ.get('BIDATE_PRECISION', 'bf16')
        self._engine = None

    def engine(self):
        if self._engine is None or self._engine.precision != self.precision:
            self._engine = BiDateEngine(self.n_channels, self.n_classes, self.precision)
        return self._engine

    def forward(self, x_d1, x_d2):
        params = [p for _, p in self.named_parameters()]
        return _BiDateFunction.apply(self, x_d1, x_d2, *params)

    # the packed bf16 / f32 GEMM images are derived from the master weights: anything that rewrites parameters behind
    # autograd's version counters has to drop them
    def load_state_dict(self, *args, **kw):
        out = super().load_state_dict(*args, **kw)
        if getattr(self, '_engine', None) is not None:
            self._engine.invalidate_weights()
        return out

    def _apply(self, fn, *args, **kw):
        out = super()._apply(fn, *args, **kw)
        if getattr(self, '_engine', None) is not None:
            self._engine.invalidate_weights()
        return out

    def train(self, mode=True):
        """nn.Module.train / eval; leaving training mode also drops the bf16x3 setting's per-layer operand-split buffers (about 1.5 GB
        per trained workspace at B=16, 128x128; re-grown on demand by the next training forward)."""
        out = super().train(mode)
        if not mode and getattr(self, '_engine', None) is not None and self._engine.x3:
            self._engine.release_split_buffers()
        return out

    # the engine and its workspaces are derived state: keep them out of pickles / deepcopies
    def __getstate__(self):
        d = self.__dict__.copy()
        d['_engine'] = None
        return d

    def __setstate__(self, state):
        """Unpickling.  The reference persists a model ONLY as a whole-module pickle (train.py:222 `torch.save(model, ...)` of
        nn.DataParallel(BiDateNet(13, 2)), utils/helpers.py:333-335).  Such a pickle resolves its class paths to this class through the
        root `models.*` shims, but its __dict__ is the reference's: sub-modules and parameters only.  Everything this class adds is
        derived state and is rebuilt here: the channel counts from the first / last convolution, the numerics setting from
        BIDATE_PRECISION (default 'bf16'), no engine yet."""
        super().__setstate__(state)
        d = self.__dict__
        if 'n_channels' not in d:
            d['n_channels'] = int(self.inc.conv.conv[0].weight.shape[1])
        if 'n_classes' not in d:
            d['n_classes'] = int(self.outc.conv.weight.shape[0])
        if d.get('precision') is None:
            d['precision'] = os.environ.get('BIDATE_PRECISION', 'bf16')
        d['_engine'] = None
