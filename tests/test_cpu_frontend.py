"""Host-side pieces in front of the hot path, no GPU: the k-means codebook loader, resume from the
highest-numbered checkpoint (against the behaviour recorded from the reference, tests/golden/resume.json) and the loud
failure of the frame resampler on host tensors."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN


def test_cluster_codebook_loader(tmp_path):
    """reference cluster/__init__.py:5-27: semantic_codebook.pt = dict poured into a scikit-learn KMeans"""
    import cluster
    rng = np.random.default_rng(0)
    centers = rng.standard_normal((32, 8)).astype(np.float32)
    p = tmp_path / "semantic_codebook.pt"
    torch.save({"n_features_in_": 8, "_n_threads": 4, "cluster_centers_": centers}, p)
    km = cluster.get_cluster_model(str(p))
    assert type(km).__name__ == "KMeans" and km.n_features_in_ == 8
    assert np.array_equal(km.cluster_centers_, centers)
    assert np.array_equal(cluster.get_center(km, np.array([3, 31, 0])), centers[[3, 31, 0]])
    x = centers[[5, 9, 9, 20]] + 1e-3
    assert np.array_equal(cluster.get_cluster_result(km, x), [5, 9, 9, 20])
    assert np.array_equal(cluster.get_cluster_center_result(km, x), centers[[5, 9, 9, 20]])
    t = cluster.codebook_to_device(km, "cpu")
    assert t.dtype == torch.float32 and t.is_contiguous() and np.array_equal(t.numpy(), centers)


def test_resume_from_highest_step_matches_reference(tmp_path):
    """tools/utils.py:load_model picks <name>_<largest step>.pt (non-numeric suffixes count as step 0); the expected outcomes
    were recorded by running the reference's own function (tests/golden/make_fixtures.py)"""
    from tools import utils
    want = json.load(open(os.path.join(GOLDEN, "resume.json")))
    for cname, rec in want.items():
        d = tmp_path / cname
        d.mkdir()
        for f in rec["files"]:
            if f.endswith(".pt"):
                stem = f[len("model_"):-3]
                torch.save({"global_step": int(stem) if stem.isdigit() else -1, "model": {"w": torch.tensor([float(len(stem))])}}, d / f)
            else:
                (d / f).write_text("x")
        m = torch.nn.Module()
        m.w = torch.nn.Parameter(torch.zeros(1))
        if "raises" in rec:
            with pytest.raises(FileNotFoundError):
                utils.load_model(str(d), m, None)
        else:
            step, m2, opt = utils.load_model(str(d), m, None)
            assert (int(step), float(m.w.item())) == (rec["global_step"], rec["w"]), cname
            assert m2 is m and opt is None


def test_traverse_dir_and_config(tmp_path):
    from tools import utils
    (tmp_path / "a").mkdir()
    for f in ("x.pt", "y.txt", "a/z.pt"):
        (tmp_path / f).write_text("1")
    got = utils.traverse_dir(str(tmp_path), ["pt"], is_pure=True, is_sort=True)
    assert got == ["a/z.pt", "x.pt"]
    assert utils.traverse_dir(str(tmp_path), ["pt"], is_pure=True, is_sort=True, is_ext=False) == ["a/z", "x"]
    assert utils.traverse_dir(str(tmp_path / "nope"), ["pt"]) == []
    (tmp_path / "config.yaml").write_text("data:\n  encoder: whisper_large_v3\n")
    assert utils.load_config(str(tmp_path / "config.yaml")).data.encoder == "whisper_large_v3"


def test_units_forced_alignment_has_no_cpu_fallback():
    from tools.tools import units_forced_alignment
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        units_forced_alignment(torch.zeros(1, 8, 4), scale_factor=1.5)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        units_forced_alignment(np.zeros((8, 4), dtype=np.float32), n_frames=12)
    with pytest.raises(AssertionError):
        units_forced_alignment(torch.zeros(1, 8, 4))
