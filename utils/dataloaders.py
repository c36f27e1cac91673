from fabric_amd.utils.dataloaders import *  # noqa: F401,F403
from fabric_amd.utils.dataloaders import OneraPreloader, onera_siamese_loader  # noqa: F401
