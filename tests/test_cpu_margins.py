"""The margin bookkeeping of tests/conftest.py (CPU): the outlier rule flags round 3's committed table exactly where the red run came from."""
import json
import os

from conftest import ROOT, _order_key, margin_outliers


def test_outlier_rule_flags_round3_table():
    m = json.load(open(os.path.join(ROOT, "profiles", "r03_parity_margins.json")))
    out = margin_outliers(m)
    assert [k for k, _, _ in out] == ["test_latency_unet_other_sizes[split_f16-1-2050]"]


def test_outlier_rule_ignores_small_groups_and_tiny_errors():
    base = {"relmax": 1e-6, "tol": 2e-5, "frac_of_tol": 0.05}
    assert margin_outliers({"t[a]": base, "t[b]": dict(base, relmax=9e-5)}) == []                          # two cases: no siblings to speak of
    m = {"t[a]": dict(base, relmax=1e-9), "t[b]": dict(base, relmax=1e-9), "t[c]": dict(base, relmax=3e-8)}
    assert margin_outliers(m) == []                                                                         # far below the tolerance: noise
    m["t[c]"] = dict(base, relmax=1.5e-5)
    assert [k for k, _, _ in margin_outliers(m)] == ["t[c]"]


def test_collection_order_puts_hot_path_and_rccl_first():
    class It:
        def __init__(self, nid):
            self.nodeid = nid
    ids = ["tests/test_gpu_model.py::test_latency_unet_other_sizes[split_f16-1-2050]", "tests/test_gpu_zz_margins.py::test_margin_outliers",
           "tests/test_gpu_rccl_rehearsal.py::test_bench_single_rank_rccl_group[sampler]", "tests/test_gpu_model.py::test_unet_forward_vs_reference[f32-a]",
           "tests/test_gpu_model.py::test_unet_forward_vs_reference[split_f16-a]", "tests/test_gpu_kernels.py::test_conv"]
    got = [i.nodeid for i in sorted((It(i) for i in ids), key=_order_key)]
    assert got[0].endswith("[f32-a]") and got[1].endswith("test_conv") and "rccl" in got[2] and got[-1].endswith("test_margin_outliers")
    assert got.index(ids[0]) > got.index(ids[2])
