"""fabric_amd -- MI355X-native bi-date Siamese U-Net training path.

Drop-in for the hot path of granularai/fabric (models/bidate_model.py,
models/unet_parts.py, the train.py step and the utils/dataloaders patch-pair API),
executed by hand-written gfx950 HIP kernels behind a C ABI (include/bidate_hip.h).
"""
from .models.bidate_model import BiDateNet  # noqa: F401

__all__ = ['BiDateNet']
