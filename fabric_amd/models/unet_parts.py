"""Parameter containers of the U-Net stages.

Class names, constructor signatures and the sub-module tree are those of the reference's models/unet_parts.py:8-90
-- that is what makes ``state_dict()`` keys match key for key (``conv.conv.0.weight`` ...) and lets pickled
reference checkpoints resolve their class paths.  Nothing else is shared with it: the modules only OWN parameters and
BatchNorm buffers.  The arithmetic is not torch's -- BiDateNet.forward hands the whole graph to the HIP engine
(fabric_amd/engine.py), where these stages exist as fused kernels (conv3x3 + BN statistics; BN+ReLU / max-pool /
bilinear-upsample / concat folded into the consumer's loads) -- so a part cannot be called on its own.
"""
import torch.nn as nn


class _Part(nn.Module):
    """A stage that exists only inside BiDateNet's fused schedule."""

    def forward(self, *inputs):
        raise RuntimeError(
            f'fabric_amd: {type(self).__name__} is a parameter container; the U-Net parts run only as fused HIP stages '
            f'inside BiDateNet.forward (conv3x3+BN statistics, BN+ReLU on load, fused pool / upsample / concat)')


def _two_conv_bn_relu(in_ch, out_ch):
    """[Conv3x3, BN, ReLU] x 2 as one Sequential: indices 0,1 / 3,4 hold the parameters (reference unet_parts.py:13-18)."""
    stages = []
    for c_in in (in_ch, out_ch):
        stages += [nn.Conv2d(c_in, out_ch, kernel_size=3, padding=1), nn.BatchNorm2d(out_ch), nn.ReLU(inplace=True)]
    return nn.Sequential(*stages)


class double_conv(_Part):
    """reference models/unet_parts.py:8-23"""

    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.conv = _two_conv_bn_relu(in_ch, out_ch)


class inconv(_Part):
    """reference models/unet_parts.py:26-33"""

    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.conv = double_conv(in_ch, out_ch)


class down(_Part):
    """pool, then double_conv: the pool sits at index 0 so the parameters live under ``mpconv.1`` (unet_parts.py:36-46)"""

    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.mpconv = nn.Sequential(nn.MaxPool2d(2), double_conv(in_ch, out_ch))


class up(_Part):
    """bilinear x2 (align_corners) -> pad -> cat([skip, up]) -> double_conv (unet_parts.py:49-80).  Only the bilinear
    variant exists: BiDateNet never takes the reference's ConvTranspose2d branch (unet_parts.py:59-60)."""

    def __init__(self, in_ch, out_ch, bilinear=True):
        super().__init__()
        if not bilinear:
            raise NotImplementedError('fabric_amd: only the bilinear up path (the one BiDateNet uses) is built')
        self.up = nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)    # parameter-free; kept for the module tree
        self.conv = double_conv(in_ch, out_ch)


class outconv(_Part):
    """1x1 classifier (unet_parts.py:83-90)"""

    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.conv = nn.Conv2d(in_ch, out_ch, kernel_size=1)
