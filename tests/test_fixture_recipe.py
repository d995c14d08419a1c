"""The golden-fixture recipe must be runnable and must import the REFERENCE (the product has same-named packages):
regenerate everything into a temporary directory and require bit-identity with the committed files.  Build container only
(needs /root/reference; the GPU box never has it)."""
import importlib.util
import os

import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/diffusion"), reason="needs the reference checkout (build container only)")


def _verify_module():
    spec = importlib.util.spec_from_file_location("verify_fixtures", os.path.join(GOLDEN, "verify_fixtures.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_recipe_imports_the_reference_and_regenerates_bit_identically():
    origin, bad = _verify_module().verify()
    assert origin["GaussianDiffusion"] == "/root/reference/diffusion/diffusion.py"
    assert all(f.startswith("/root/reference/") for f in origin.values()), origin
    assert not bad, bad
