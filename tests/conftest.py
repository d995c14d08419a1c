import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "latent-diffusion-speech_amd")
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name)))
    return load


@pytest.fixture(scope="session")
def unet_weights():
    """Seeded UNet weights (build-owned initialiser, seed 0) shared across tests."""
    from lds import arch, init_weights
    cfg = arch.unet_config()
    return cfg, arch.unet_blocks(cfg), init_weights.init_state(arch.unet_param_shapes(cfg), 0)
