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


# ---- measured parity margins ------------------------------------------------------------------------------------------------
# Every parity test reports (measured relative error, tolerance) through `margin`; at the end of the session the table is written to
# gpurun_out/parity_margins.json (merged with what an earlier session left there) so that a GPU run leaves a record of HOW FAR inside
# the tolerance each comparison sits, not only that it passed.  profiles/rNN_parity_margins.json is a committed copy of one such run.
_MARGINS = {}


def margin(name, measured, tol):
    """record and assert: measured < tol"""
    _MARGINS[name] = {"relmax": float(measured), "tol": float(tol), "frac_of_tol": float(measured) / float(tol)}
    assert measured < tol, (name, measured, tol)


@pytest.fixture
def record_margin(request):
    def rec(measured, tol, tag=""):
        margin(request.node.name + (":" + tag if tag else ""), measured, tol)
    return rec


def pytest_sessionfinish(session, exitstatus):
    if not _MARGINS:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out", "parity_margins.json")
    try:
        os.makedirs(os.path.dirname(out), exist_ok=True)
        old = json.load(open(out)) if os.path.exists(out) else {}
        old.update(_MARGINS)
        json.dump(dict(sorted(old.items())), open(out, "w"), indent=1)
    except OSError:
        pass
