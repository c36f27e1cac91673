from fabric_amd.utils.metrics import TverskyLoss, batch_prf_from_counts  # noqa: F401
