"""3x3x3 convolution for the multi-date stack (BASELINE.json configs[3]: "3D-UNet multi-date stack, 5 dates x 13 bands x
128 x 128 -- 3D-conv implicit-GEMM path").

There is NO reference source for that model: `UNetLSTM/` in the reference tree is an empty sub-module (reference README.md:5
points at an external repository), no `nn.Conv3d` exists anywhere in it.  Parity is therefore UNPINNED; what is built is the
operator such a network is made of, checked against `torch.nn.functional.conv3d` / `torch.nn.grad` (tests/test_gpu_conv3d.py):

    Conv3d3x3(cin, cout)            nn.Conv3d(cin, cout, 3, padding=1) on [N, D, H, W, C] device tensors (bf16 or f32)
        .forward(x)                 implicit GEMM, K = 27 Cin, on the 2-D MFMA kernel: three depth taps = three sources of one
                                    reduction, depth border = a block-uniform zero mask (csrc/conv3x3.hip, D3 instantiations)
        .dgrad(dz) / .wgrad(dz, x)  data gradient (same kernel, transposed filter view), weight gradient (three runs of the
                                    2-D split-K GEMM, one per depth tap, LDS-DMA kernel for >= 64 channels)

The layout keeps the depth slices of a sample consecutive ([N, D, H, W, C] = N*D NHWC images), so every HBM-bound stage kernel
of the 2-D path (BatchNorm statistics / backward, ReLU, spatial pooling and upsampling, which a 3-D U-Net over dates applies per
slice) runs on these tensors unchanged with N*D images.
"""
import torch

from . import _lib
from ._lib import BDN_BF16, BDN_BF16X3, BDN_F32, IN_BNRELU, IN_PLAIN, call, ptr


def _round_up(v, m):
    return (v + m - 1) // m * m


class Conv3d3x3:
    def __init__(self, weight_oidhw, bias=None, precision='bf16'):
        """weight_oidhw: float32 [Cout, Cin, 3, 3, 3] device tensor (nn.Conv3d layout); bias: float32 [Cout] or None."""
        if not weight_oidhw.is_cuda:
            raise RuntimeError('fabric_amd: Conv3d3x3 runs only on a ROCm device -- there is no CPU path')
        if precision not in ('bf16', 'fp32', 'bf16x3'):
            raise ValueError("precision must be 'bf16', 'bf16x3' or 'fp32'")
        # bf16x3 (as in the 2-D path): float32 tensors, every GEMM operand split into bf16 hi + lo, three bf16 MFMAs per product
        self.x3 = precision == 'bf16x3'
        self.dt = BDN_BF16 if precision == 'bf16' else BDN_F32          # storage type: what the HBM-bound kernels see
        self.td = torch.bfloat16 if precision == 'bf16' else torch.float32
        self.cout, self.cin = weight_oidhw.shape[:2]
        if self.cout % 64:
            raise RuntimeError(f'Cout={self.cout} must be a multiple of 64')
        self.cp = _round_up(self.cin, 16)
        self.bias = bias.float().contiguous() if bias is not None else None
        self.set_weight(weight_oidhw)

    def set_weight(self, w):
        """(Re)pack the two filter images from the float32 master weight."""
        w = w.detach().float()
        co, ci, cp, dev = self.cout, self.cin, self.cp, w.device
        wp = torch.zeros(co, cp, 3, 3, 3, device=dev)
        wp[:, :ci] = w
        if self.x3:
            # operand channels per depth tap are [a_hi | a_lo | a_hi]: the filter view holds [w_hi | w_hi | w_lo] per depth tap, packed as a
            # plain bf16 image (w_hi is exactly representable, w_lo is rounded by the pack)
            hi = wp.to(torch.bfloat16).float()
            lo = wp - hi
            w3 = torch.stack([hi, hi, lo], 0)                                   # [part, co, c, kd, kh, kw]
            fwd = w3.permute(1, 3, 0, 2, 4, 5).reshape(co, 9 * cp, 3, 3).contiguous()          # channel (kd, part, c)
            self.wf = torch.empty(co, 9, 9 * cp, dtype=torch.bfloat16, device=dev)
            call('bdn_pack_weights', BDN_BF16, ptr(fwd), ptr(self.wf), None, co, 9 * cp, 9 * cp, _lib.stream_ptr())
            self.wd = None
            if cp % 64 == 0:
                w3f = w3.flip(3, 4, 5)                                          # [part, co, c, 2-kd, 2-kh, 2-kw]
                back = w3f.permute(2, 3, 0, 1, 4, 5).reshape(cp, 9 * co, 3, 3).contiguous()    # [c][(s, part, co)]
                self.wd = torch.empty(cp, 9, 9 * co, dtype=torch.bfloat16, device=dev)
                call('bdn_pack_weights', BDN_BF16, ptr(back), ptr(self.wd), None, cp, 9 * co, 9 * co, _lib.stream_ptr())
            return
        # forward: OIHW view [Cout][3*cp][3][3], input channel kd*cp + c
        fwd = wp.permute(0, 2, 1, 3, 4).reshape(co, 3 * cp, 3, 3).contiguous()
        self.wf = torch.empty(co, 9, 3 * cp, dtype=self.td, device=dev)
        call('bdn_pack_weights', self.dt, ptr(fwd), ptr(self.wf), None, co, 3 * cp, 3 * cp, _lib.stream_ptr())
        # data gradient as a forward convolution of dz: [cp][3*Cout][3][3], channel s*Cout + co = w[co][c][2-s][2-kh][2-kw]
        self.wd = None
        if cp % 64 == 0:
            back = wp.flip(2, 3, 4).permute(1, 2, 0, 3, 4).reshape(cp, 3 * co, 3, 3).contiguous()
            self.wd = torch.empty(cp, 9, 3 * co, dtype=self.td, device=dev)
            call('bdn_pack_weights', self.dt, ptr(back), ptr(self.wd), None, cp, 3 * co, 3 * co, _lib.stream_ptr())

    def _check(self, x, c):
        if not x.is_cuda or x.dtype != self.td or x.dim() != 5 or x.shape[4] != c or not x.is_contiguous():
            raise RuntimeError(f'expected a contiguous [N,D,H,W,{c}] {self.td} device tensor, got {tuple(x.shape)} {x.dtype}')

    def split(self, x, c, in_bn=None, imgs_per_group=None):
        """bf16x3: the [hi | lo] operand split of a float32 [N,D,H,W,c] tensor (bdn_split_pack on the N*D slices; BatchNorm+ReLU of the
        producing layer applied first when in_bn is given).  bf16 [N,D,H,W,2c]: the weight-gradient GEMM takes it as it is."""
        n, d, h, w, _ = x.shape
        sp = torch.empty(n, d, h, w, 2 * c, dtype=torch.bfloat16, device=x.device)
        call('bdn_split_pack', ptr(x), c, None, 0, IN_BNRELU if in_bn is not None else IN_PLAIN, ptr(in_bn),
             (imgs_per_group or n) * d, ptr(sp), n * d, h, w, _lib.stream_ptr())
        return sp

    @staticmethod
    def _hi_lo_hi(sp, c):
        """[hi | lo] -> [hi | lo | hi]: the channel order the convolution's reduction walks (a layout copy, no arithmetic)."""
        return torch.cat([sp, sp[..., :c]], dim=-1)

    def forward(self, x, in_bn=None, imgs_per_group=None, stats=False, split=None):
        """x: [N,D,H,W,cp] (channels >= Cin zero).  in_bn: optional [G][4][cp] BatchNorm table applied (with ReLU) on load.
        Returns out [N,D,H,W,Cout] (and the per-tile statistics partials [rows][2][Cout] when stats=True).
        bf16x3: `split` may carry the operand already split by split() (kept by the caller for the weight gradient)."""
        self._check(x, self.cp)
        n, d, h, w, _ = x.shape
        out = torch.empty(n, d, h, w, self.cout, dtype=self.td, device=x.device)
        part = torch.empty(_lib.load().bdn_conv3d_num_mtiles(n, d, h, w), 2, self.cout, device=x.device) if stats else None
        if self.x3:
            sp = split if split is not None else self.split(x, self.cp, in_bn, imgs_per_group)
            op = self._hi_lo_hi(sp, self.cp)
            call('bdn_conv3d', BDN_BF16X3, ptr(op), 3 * self.cp, IN_PLAIN, None, imgs_per_group or n,
                 ptr(self.wf), ptr(self.bias), ptr(out), ptr(part), n, d, h, w, self.cout, _lib.stream_ptr())
            return (out, part) if stats else out
        call('bdn_conv3d', self.dt, ptr(x), self.cp, 1 if in_bn is not None else IN_PLAIN, ptr(in_bn), imgs_per_group or n,
             ptr(self.wf), ptr(self.bias), ptr(out), ptr(part), n, d, h, w, self.cout, _lib.stream_ptr())
        return (out, part) if stats else out

    def dgrad(self, dz, dz_split=None):
        """dz: [N,D,H,W,Cout] -> gradient wrt the input, [N,D,H,W,cp]."""
        if self.wd is None:
            raise RuntimeError('the data gradient needs a padded input width that is a multiple of 64 channels')
        self._check(dz, self.cout)
        n, d, h, w, _ = dz.shape
        out = torch.empty(n, d, h, w, self.cp, dtype=self.td, device=dz.device)
        if self.x3:
            sp = dz_split if dz_split is not None else self.split(dz, self.cout)
            call('bdn_conv3d', BDN_BF16X3, ptr(self._hi_lo_hi(sp, self.cout)), 3 * self.cout, IN_PLAIN, None, n, ptr(self.wd), None, ptr(out), None,
                 n, d, h, w, self.cp, _lib.stream_ptr())
            return out
        call('bdn_conv3d', self.dt, ptr(dz), self.cout, IN_PLAIN, None, n, ptr(self.wd), None, ptr(out), None,
             n, d, h, w, self.cp, _lib.stream_ptr())
        return out

    def wgrad(self, dz, x, dz_split=None, x_split=None):
        """float32 [Cout, Cin, 3, 3, 3] gradient wrt the weight from dz [N,D,H,W,Cout] and the (plain) input x [N,D,H,W,cp].
        bf16x3: the operands are the [hi | lo] splits (given, or made here); x may then be None."""
        n, d, h, w = dz.shape[:4]
        if self.x3:
            sd = dz_split if dz_split is not None else self.split(dz, self.cout)
            sx = x_split if x_split is not None else self.split(x, self.cp)
            lib = _lib.load()
            nb = lib.bdn_wgrad_workspace_bytes_ex(BDN_BF16, n * d, h, w, 2 * self.cout, 2 * self.cp, 0, 1, IN_PLAIN, 0) \
                + 4 * self.cout * self.cp * 27 * 4
            part = torch.empty(nb // 4, device=dz.device)
            dw = torch.empty(self.cout, self.cin, 3, 3, 3, device=dz.device)
            call('bdn_conv3d_wgrad', BDN_BF16X3, ptr(sd), self.cout, ptr(sx), self.cp, ptr(part), ptr(dw), self.cin, n, d, h, w, _lib.stream_ptr())
            return dw
        self._check(dz, self.cout)
        self._check(x, self.cp)
        nb = _lib.load().bdn_wgrad_workspace_bytes_ex(self.dt, n * d, h, w, self.cout, self.cp, 0, 1, IN_PLAIN, 0)
        part = torch.empty(nb // 4, device=x.device)
        dw = torch.empty(self.cout, self.cin, 3, 3, 3, device=x.device)
        call('bdn_conv3d_wgrad', self.dt, ptr(dz), self.cout, ptr(x), self.cp, ptr(part), ptr(dw), self.cin, n, d, h, w, _lib.stream_ptr())
        return dw


def to_ndhwc(x_ncdhw, cp, dtype):
    """[N,C,D,H,W] float32 -> [N,D,H,W,cp] device tensor of `dtype`, channels >= C zero (layout boundary helper)."""
    n, c, d, h, w = x_ncdhw.shape
    out = torch.zeros(n, d, h, w, cp, dtype=dtype, device=x_ncdhw.device)
    out[..., :c] = x_ncdhw.permute(0, 2, 3, 4, 1).to(dtype)
    return out


class DoubleConv3d:
    """(nn.Conv3d(ci, co, 3, padding=1) -> nn.BatchNorm3d(co) -> nn.ReLU) x 2 on [N, D, H, W, C] device tensors, forward and
    backward, training-mode statistics: the 3-D counterpart of the reference's `double_conv` (models/unet_parts.py:8-23), i.e.
    the block a 3D-UNet over the date axis would be made of (BASELINE configs[3]).  PARITY UNPINNED: the reference has no source
    for such a model; tests/test_gpu_conv3d.py checks this block against the same stack of stock torch.nn modules.

    Same fusion structure as the 2-D path: the convolutions emit per-tile BatchNorm statistics from their epilogue, relu(bn(z0))
    is never materialised for the second convolution (applied while it stages its input), and because the D slices of a sample
    are consecutive NHWC images every HBM-bound kernel of the 2-D path (BatchNorm finalize / backward, bdn_bnrelu) runs on these
    tensors unchanged with N*D images.  Differences to the 2-D training step, by design of what exists: the weight-gradient GEMM
    of the 3x3x3 convolution takes plain operands only, so backward materialises relu(bn(z0)) once (bdn_bnrelu), and the
    BatchNorm backward uses the stand-alone three-kernel path (bdn_bn_bwd)."""

    def __init__(self, cin, cout, precision='bf16', eps=1e-5, momentum=0.1, device='cuda'):
        if precision not in ('bf16', 'fp32', 'bf16x3'):
            raise ValueError("precision must be 'bf16', 'bf16x3' or 'fp32'")
        if cout % 64:
            raise RuntimeError(f'Cout={cout} must be a multiple of 64')
        self.cin, self.cout, self.precision, self.eps, self.momentum = cin, cout, precision, eps, momentum
        self.x3 = precision == 'bf16x3'      # float32 tensors, split bf16 GEMM operands: each operand is split ONCE (the forward's split of a
                                             # layer's input serves its weight gradient, dz is split once for data and weight gradient)
        self.dt = BDN_BF16 if precision == 'bf16' else BDN_F32
        self.td = torch.bfloat16 if precision == 'bf16' else torch.float32
        dev = torch.device(device)
        if dev.type != 'cuda':
            raise RuntimeError('fabric_amd: DoubleConv3d runs only on a ROCm device -- there is no CPU path')
        f = lambda *s: torch.zeros(*s, device=dev)
        # parameters / buffers under the names the torch.nn stack would give them (conv.0, bn 1, conv.3, bn 4)
        self.P = {'conv.0.weight': f(cout, cin, 3, 3, 3), 'conv.0.bias': f(cout), 'conv.3.weight': f(cout, cout, 3, 3, 3), 'conv.3.bias': f(cout)}
        for k in ('1', '4'):
            self.P[f'conv.{k}.weight'] = torch.ones(cout, device=dev)
            self.P[f'conv.{k}.bias'] = f(cout)
            self.P[f'conv.{k}.running_mean'] = f(cout)
            self.P[f'conv.{k}.running_var'] = torch.ones(cout, device=dev)
            self.P[f'conv.{k}.num_batches_tracked'] = torch.zeros((), dtype=torch.int64, device=dev)
        self._ops = None
        self._saved = None

    def load(self, state):
        """Copy a state dict of the equivalent nn.Sequential(Conv3d, BatchNorm3d, ReLU, Conv3d, BatchNorm3d, ReLU)."""
        for k, v in state.items():
            self.P[k].copy_(v.to(self.P[k].device))
        self._ops = None

    def _convs(self):
        if self._ops is None:
            self._ops = (Conv3d3x3(self.P['conv.0.weight'], self.P['conv.0.bias'], self.precision),
                         Conv3d3x3(self.P['conv.3.weight'], self.P['conv.3.bias'], self.precision))
        return self._ops

    def _finalize(self, part, key, count, dev):
        lib = _lib.load()
        rows, C = part.shape[0], self.cout
        bn = torch.empty(1, 4, C, device=dev)
        ws = torch.empty(max(lib.bdn_bn_finalize_workspace_bytes(rows, 1, C) // 8, 1), dtype=torch.float64, device=dev)
        call('bdn_bn_finalize', ptr(part), rows, 1, C, count, ptr(self.P[f'conv.{key}.weight']), ptr(self.P[f'conv.{key}.bias']),
             self.eps, self.momentum, ptr(self.P[f'conv.{key}.running_mean']), ptr(self.P[f'conv.{key}.running_var']),
             ptr(self.P[f'conv.{key}.num_batches_tracked']), ptr(bn), ptr(ws), _lib.stream_ptr())
        return bn

    def forward(self, x):
        """x: [N,D,H,W,cp] (cp = Cin rounded up to 16, channels >= Cin zero).  Returns relu(bn(conv(relu(bn(conv(x)))))) as
        [N,D,H,W,Cout]; keeps what backward needs."""
        op0, op1 = self._convs()
        n, d, h, w, _ = x.shape
        count = n * d * h * w
        sx = sa = None
        if self.x3:
            sx = op0.split(x, op0.cp)
            z0, part0 = op0.forward(x, stats=True, split=sx)
            bn0 = self._finalize(part0, '1', count, x.device)
            sa = op1.split(z0, self.cout, in_bn=bn0, imgs_per_group=n)          # relu(bn(z0)), split once: conv operand now, wgrad operand later
            z1, part1 = op1.forward(z0, stats=True, split=sa)
        else:
            z0, part0 = op0.forward(x, stats=True)
            bn0 = self._finalize(part0, '1', count, x.device)
            z1, part1 = op1.forward(z0, in_bn=bn0, imgs_per_group=n, stats=True)
        bn1 = self._finalize(part1, '4', count, x.device)
        out = torch.empty_like(z1)
        call('bdn_bnrelu', self.dt, ptr(z1), ptr(bn1), n * d, ptr(out), n * d, h, w, self.cout, _lib.stream_ptr())
        self._saved = (x, z0, bn0, z1, bn1, sx, sa)
        return out

    def _bn_bwd(self, dA, z, bn, key, grads):
        lib = _lib.load()
        nd, h, w, C = z.shape[0] * z.shape[1], z.shape[2], z.shape[3], self.cout
        ws = torch.empty(max(lib.bdn_bn_bwd_workspace_bytes(self.dt, nd, h, w, C, nd) // 4, 1), device=z.device)
        sums = torch.empty(1, 2, C, device=z.device)
        dz = torch.empty_like(z)
        grads[f'conv.{key}.weight'] = torch.empty(C, device=z.device)
        grads[f'conv.{key}.bias'] = torch.empty(C, device=z.device)
        call('bdn_bn_bwd', self.dt, ptr(dA), C, ptr(z), ptr(bn), nd, nd, h, w, C, ptr(ws), ptr(sums),
             ptr(grads[f'conv.{key}.weight']), ptr(grads[f'conv.{key}.bias']), ptr(dz), _lib.stream_ptr())
        return dz

    def backward(self, d_out):
        """d_out: gradient wrt forward()'s output.  Returns (dx or None, grads) -- dx [N,D,H,W,cp] when the padded input width is
        a multiple of 64 channels (a first layer with 13 bands has no data gradient), grads keyed like `P`."""
        if self._saved is None:
            raise RuntimeError('backward() needs a forward() first')
        x, z0, bn0, z1, bn1, sx, sa = self._saved
        op0, op1 = self._convs()
        n, d, h, w, _ = x.shape
        grads = {}
        dz1 = self._bn_bwd(d_out.contiguous(), z1, bn1, '4', grads)
        if self.x3:
            sd1 = op1.split(dz1, self.cout)
            grads['conv.3.weight'] = op1.wgrad(dz1, None, dz_split=sd1, x_split=sa)
            dA0 = op1.dgrad(dz1, dz_split=sd1)
            dz0 = self._bn_bwd(dA0, z0, bn0, '1', grads)
            sd0 = op0.split(dz0, self.cout)
            grads['conv.0.weight'] = op0.wgrad(dz0, None, dz_split=sd0, x_split=sx)
            grads['conv.0.bias'] = torch.zeros(self.cout, device=x.device)
            grads['conv.3.bias'] = torch.zeros(self.cout, device=x.device)
            dx = op0.dgrad(dz0, dz_split=sd0) if op0.wd is not None else None
            return dx, grads
        a0 = torch.empty_like(z0)                              # the weight-gradient GEMM of the 3x3x3 convolution takes plain operands
        call('bdn_bnrelu', self.dt, ptr(z0), ptr(bn0), n * d, ptr(a0), n * d, h, w, self.cout, _lib.stream_ptr())
        grads['conv.3.weight'] = op1.wgrad(dz1, a0)
        dA0 = op1.dgrad(dz1)
        dz0 = self._bn_bwd(dA0, z0, bn0, '1', grads)
        grads['conv.0.weight'] = op0.wgrad(dz0, x)
        grads['conv.0.bias'] = torch.zeros(self.cout, device=x.device)      # a bias in front of a BatchNorm has no gradient
        grads['conv.3.bias'] = torch.zeros(self.cout, device=x.device)
        dx = op0.dgrad(dz0) if op0.wd is not None else None
        return dx, grads
