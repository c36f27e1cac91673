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
from ._lib import BDN_BF16, BDN_F32, IN_PLAIN, call, ptr


def _round_up(v, m):
    return (v + m - 1) // m * m


class Conv3d3x3:
    def __init__(self, weight_oidhw, bias=None, precision='bf16'):
        """weight_oidhw: float32 [Cout, Cin, 3, 3, 3] device tensor (nn.Conv3d layout); bias: float32 [Cout] or None."""
        if not weight_oidhw.is_cuda:
            raise RuntimeError('fabric_amd: Conv3d3x3 runs only on a ROCm device -- there is no CPU path')
        if precision not in ('bf16', 'fp32'):
            raise ValueError("precision must be 'bf16' or 'fp32'")
        self.dt = BDN_BF16 if precision == 'bf16' else BDN_F32
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

    def forward(self, x, in_bn=None, imgs_per_group=None, stats=False):
        """x: [N,D,H,W,cp] (channels >= Cin zero).  in_bn: optional [G][4][cp] BatchNorm table applied (with ReLU) on load.
        Returns out [N,D,H,W,Cout] (and the per-tile statistics partials [rows][2][Cout] when stats=True)."""
        self._check(x, self.cp)
        n, d, h, w, _ = x.shape
        out = torch.empty(n, d, h, w, self.cout, dtype=self.td, device=x.device)
        part = torch.empty(_lib.load().bdn_conv3d_num_mtiles(n, d, h, w), 2, self.cout, device=x.device) if stats else None
        call('bdn_conv3d', self.dt, ptr(x), self.cp, 1 if in_bn is not None else IN_PLAIN, ptr(in_bn), imgs_per_group or n,
             ptr(self.wf), ptr(self.bias), ptr(out), ptr(part), n, d, h, w, self.cout, _lib.stream_ptr())
        return (out, part) if stats else out

    def dgrad(self, dz):
        """dz: [N,D,H,W,Cout] -> gradient wrt the input, [N,D,H,W,cp]."""
        if self.wd is None:
            raise RuntimeError('the data gradient needs a padded input width that is a multiple of 64 channels')
        self._check(dz, self.cout)
        n, d, h, w, _ = dz.shape
        out = torch.empty(n, d, h, w, self.cp, dtype=self.td, device=dz.device)
        call('bdn_conv3d', self.dt, ptr(dz), self.cout, IN_PLAIN, None, n, ptr(self.wd), None, ptr(out), None,
             n, d, h, w, self.cp, _lib.stream_ptr())
        return out

    def wgrad(self, dz, x):
        """float32 [Cout, Cin, 3, 3, 3] gradient wrt the weight from dz [N,D,H,W,Cout] and the (plain) input x [N,D,H,W,cp]."""
        self._check(dz, self.cout)
        self._check(x, self.cp)
        n, d, h, w, _ = x.shape
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
