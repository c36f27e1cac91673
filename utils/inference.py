"""Import-path shim for the reference's `utils.inference` (see fabric_amd/utils/inference.py)."""
from fabric_amd.utils.inference import *  # noqa: F401,F403
from fabric_amd.utils.inference import _get_patches, _get_bands  # noqa: F401
