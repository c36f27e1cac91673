"""Import-path shim for the reference's `utils.metrics` (see fabric_amd/utils/metrics.py)."""
from fabric_amd.utils.metrics import (FocalLoss, TverskyLoss, batch_prf_from_counts,  # noqa: F401
                                      dice_loss, jaccard_loss)
