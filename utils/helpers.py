from fabric_amd.utils.helpers import *  # noqa: F401,F403
