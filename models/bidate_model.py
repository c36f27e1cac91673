from fabric_amd.models.bidate_model import BiDateNet  # noqa: F401
