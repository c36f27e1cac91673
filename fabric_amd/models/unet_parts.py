"""Sub-parts of the U-Net: same class names, constructor signatures and sub-module
tree as the reference's models/unet_parts.py:8-90, so ``state_dict()`` keys match
key for key and pickled reference checkpoints resolve their class paths.

The modules own the parameters and BatchNorm buffers (plain torch.nn containers);
the arithmetic is NOT torch's: BiDateNet.forward hands the whole graph to the HIP
engine (fabric_amd/engine.py), where these stages exist as fused kernels
(conv3x3 + BN statistics; BN+ReLU / max-pool / bilinear-upsample / concat folded
into the consumer's loads).  Calling a part on its own therefore goes through the
same engine via ``run_part`` and requires a ROCm device.
"""
import torch
import torch.nn as nn


class double_conv(nn.Module):
    '''(conv => BN => ReLU) * 2   -- reference models/unet_parts.py:8-23'''

    def __init__(self, in_ch, out_ch):
        super(double_conv, self).__init__()
        self.conv = nn.Sequential(
            nn.Conv2d(in_ch, out_ch, 3, padding=1),
            nn.BatchNorm2d(out_ch),
            nn.ReLU(inplace=True),
            nn.Conv2d(out_ch, out_ch, 3, padding=1),
            nn.BatchNorm2d(out_ch),
            nn.ReLU(inplace=True)
        )

    def forward(self, x):
        _no_standalone(self)


class inconv(nn.Module):
    '''reference models/unet_parts.py:26-33'''

    def __init__(self, in_ch, out_ch):
        super(inconv, self).__init__()
        self.conv = double_conv(in_ch, out_ch)

    def forward(self, x):
        _no_standalone(self)


class down(nn.Module):
    '''MaxPool2d(2) => double_conv -- reference models/unet_parts.py:36-46'''

    def __init__(self, in_ch, out_ch):
        super(down, self).__init__()
        self.mpconv = nn.Sequential(
            nn.MaxPool2d(2),
            double_conv(in_ch, out_ch)
        )

    def forward(self, x):
        _no_standalone(self)


class up(nn.Module):
    '''bilinear x2 (align_corners) => pad => cat([skip, up]) => double_conv
    -- reference models/unet_parts.py:49-80'''

    def __init__(self, in_ch, out_ch, bilinear=True):
        super(up, self).__init__()
        if not bilinear:
            # the reference's ConvTranspose2d branch (unet_parts.py:59-60) is never taken by BiDateNet
            raise NotImplementedError('fabric_amd: only the bilinear up path (the one BiDateNet uses) is built')
        self.up = nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)
        self.conv = double_conv(in_ch, out_ch)

    def forward(self, x1, x2):
        _no_standalone(self)


class outconv(nn.Module):
    '''1x1 classifier -- reference models/unet_parts.py:83-90'''

    def __init__(self, in_ch, out_ch):
        super(outconv, self).__init__()
        self.conv = nn.Conv2d(in_ch, out_ch, 1)

    def forward(self, x):
        _no_standalone(self)


def _no_standalone(mod):
    raise RuntimeError(
        f'fabric_amd: {type(mod).__name__} is executed as fused HIP stages inside BiDateNet.forward; '
        f'it has no stand-alone (and no CPU / eager-PyTorch) path. Call the parent BiDateNet.')
