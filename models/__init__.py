"""Import-path shim: `from models.bidate_model import BiDateNet` (reference utils/helpers.py:16) and the
class paths inside pickled reference checkpoints (`models.unet_parts.*`) resolve to fabric_amd."""
