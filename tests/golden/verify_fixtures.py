"""Regenerate every golden fixture from the imported reference into a temporary directory and check that the result is
bit-identical to the committed files (every array of every .npz, every manifest).  Needs /root/reference, so it runs in
the build container only; tests/test_fixture_recipe.py calls `verify()` and is skipped elsewhere.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/verify_fixtures.py
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
FILES_NPZ = ["schedule.npz", "unet_fwd.npz", "solver_toy.npz", "sampler.npz", "sampler_shallow.npz", "vocoder.npz", "vocoder_rb2.npz", "units_align.npz", "roformer.npz"]
FILES_JSON = ["manifest_unet.json", "manifest_generator.json", "manifest_diffusion_buffers.json", "resume.json", "manifest_roformer.json"]


def regenerate(out_dir):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    env.pop("PYTHONPATH", None)          # nothing but the reference may be importable by package name
    r = subprocess.run([sys.executable, os.path.join(HERE, "make_fixtures.py"), "--out", out_dir], env=env, cwd=out_dir,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("make_fixtures.py failed:\n" + r.stdout[-4000:])
    return r.stdout


def origin_check():
    """The recipe's reference classes must resolve under /root/reference (not the product's same-named packages)."""
    code = ("import sys, json; sys.argv=['x']; import importlib.util as u; "
            f"s=u.spec_from_file_location('mf', {os.path.join(HERE, 'make_fixtures.py')!r}); m=u.module_from_spec(s); "
            "s.loader.exec_module(m); print(json.dumps(m.reference_origin()))")
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    env.pop("PYTHONPATH", None)
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=tempfile.gettempdir(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-4000:])
    origin = json.loads(r.stdout.strip().splitlines()[-1])
    for name, f in origin.items():
        assert f.startswith(REF + os.sep), (name, f)
    assert origin["GaussianDiffusion"] == os.path.join(REF, "diffusion", "diffusion.py"), origin
    return origin


def compare(new_dir, old_dir=HERE):
    """-> list of mismatch descriptions (empty = bit-identical)"""
    bad = []
    for f in FILES_NPZ:
        a, b = np.load(os.path.join(new_dir, f)), np.load(os.path.join(old_dir, f))
        if sorted(a.files) != sorted(b.files):
            bad.append(f"{f}: keys differ {sorted(set(a.files) ^ set(b.files))}")
            continue
        for k in a.files:
            x, y = a[k], b[k]
            if x.dtype != y.dtype or x.shape != y.shape or x.tobytes() != y.tobytes():
                bad.append(f"{f}[{k}]: dtype/shape/bytes differ")
    for f in FILES_JSON:
        if json.load(open(os.path.join(new_dir, f))) != json.load(open(os.path.join(old_dir, f))):
            bad.append(f"{f}: differs")
    return bad


def verify():
    origin = origin_check()
    with tempfile.TemporaryDirectory(prefix="lds_fixtures_") as tmp:
        regenerate(tmp)
        bad = compare(tmp)
    return origin, bad


if __name__ == "__main__":
    origin, bad = verify()
    for k, v in origin.items():
        print(f"{k:24s} <- {v}")
    if bad:
        print("MISMATCH:\n  " + "\n  ".join(bad))
        sys.exit(1)
    print(f"all {len(FILES_NPZ)} .npz files and {len(FILES_JSON)} manifests regenerate bit-identically")
