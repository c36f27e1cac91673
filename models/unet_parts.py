from fabric_amd.models.unet_parts import double_conv, inconv, down, up, outconv  # noqa: F401
