"""The steps in front of and around the hot path on a real MI355X: codebook row gather and unit-frame
alignment through the C ABI against reference-generated fixtures, and the 22_infer_tts.py counterpart (infer_tts.py) run
in-process: tokens -> units -> mel -> wav compared with the numpy oracle pipeline."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def relmax(a, b):
    return float(np.abs(a - b).max() / max(1e-30, np.abs(b).max()))


def test_gather_rows_codebook_lookup():
    from lds import init_weights, native
    cb = init_weights.uniform("gr.codebook", (4096, 1280), 5, -1.7, 1.7)
    tok = (np.arange(2 * 77).reshape(2, 77) * 131 % 4096).astype(np.int64)
    out = native.gather_rows(dev(cb), torch.from_numpy(tok).cuda()).cpu().numpy()
    assert out.shape == (2, 77, 1280) and np.array_equal(out, cb[tok])
    bad = native.gather_rows(dev(cb), torch.tensor([5, 4096, -1], device="cuda")).cpu().numpy()      # nn.Embedding raises; here: NaN rows
    assert np.array_equal(bad[0], cb[5]) and np.isnan(bad[1:]).all()


def test_units_forced_alignment_vs_reference(golden):
    """every reachable call pattern of reference tools/tools.py:193-223 + the scale-factor-only call 22_infer_tts.py makes"""
    from tools.tools import units_forced_alignment
    g = golden("units_align.npz")
    u, sf = dev(g["units"]), float(g["scale_factor"])
    assert np.array_equal(units_forced_alignment(u, n_frames=50).cpu().numpy(), g["nearest_size50"])
    assert np.array_equal(units_forced_alignment(u[0], n_frames=64).cpu().numpy(), g["nearest_size64_2d"])
    audio = torch.zeros(1, 512 * 45 + 17)
    assert np.array_equal(units_forced_alignment(u, audio=audio, sample_rate=44100, hop_size=512).cpu().numpy(), g["nearest_audio"])
    assert np.array_equal(units_forced_alignment(u[:1], n_frames=60, scale_factor=1.0 / sf, units_forced_mode="left").cpu().numpy(), g["left_60"])
    assert int(g["scale_only_raises"]) == 1      # the reference itself raises for this call (tools.py:195); the intent:
    assert np.array_equal(units_forced_alignment(u, scale_factor=sf).cpu().numpy(), g["nearest_sf"])
    assert np.array_equal(units_forced_alignment(u, scale_factor=1.0 / sf).cpu().numpy(), g["nearest_sf_down"])
    with pytest.raises(ValueError):
        units_forced_alignment(u, n_frames=50, scale_factor=sf)
    with pytest.raises(NotImplementedError):
        units_forced_alignment(u, n_frames=50, units_forced_mode="linear")


@pytest.mark.parametrize("scale", [None, 1.72265625])
def test_infer_tts_harness_vs_oracle(monkeypatch, scale):
    """infer_tts.py (counterpart of 22_infer_tts.py:100-114) in-process on synthetic weights: tokens -> codebook gather ->
    [forced alignment] -> Unit2Mel (10-step DPM-Solver++) -> HiFi-VAEGAN, against oracle.unit2mel o oracle.vocoder"""
    sys.path.insert(0, ROOT)
    import infer_tts
    from lds import arch, init_weights
    from oracle import schedule, unit2mel as o_u2m, vocoder as o_voc
    n_tok = 24
    svc, codebook, tokens = infer_tts.synthetic_pipeline("cuda", n_tok)
    T = n_tok if scale is None else int(np.floor(n_tok * scale))
    xT = init_weights.uniform("harness.xT", (1, 1, 80, T), 7, -1.7, 1.7)
    monkeypatch.setattr(torch, "randn", lambda *a, **k: dev(xT).clone())
    units, mel, wav = infer_tts.synthesize(svc, codebook, tokens, spk_id=9, speedup=100, method="dpm-solver", scale_factor=scale)
    assert units.shape == (1, T, 1280) and mel.shape == (1, T, 80) and wav.shape == (1, 1, T * 512)
    # oracle: the same pipeline in numpy
    cb, tok = codebook.cpu().numpy(), tokens.cpu().numpy()
    ru = cb[tok][None]
    if scale is not None:
        step = np.float32(1.0 / scale)
        idx = np.minimum(np.floor(np.arange(T, dtype=np.float32) * step).astype(np.int64), n_tok - 1)
        ru = ru[:, idx]
    assert np.array_equal(units.cpu().numpy(), ru)
    w = {k: v.detach().cpu().numpy() for k, v in svc.model.state_dict().items()}
    cfg = arch.unet_config()
    rmel = o_u2m.unit2mel(w, cfg, arch.unet_blocks(cfg), schedule.diffusion_buffers(), ru, np.array([[9]]), xT[:, 0], "dpm-solver", 100)
    assert relmax(mel.cpu().numpy(), rmel) < 1e-4, relmax(mel.cpu().numpy(), rmel)
    h = arch.SYNTHETIC_VOCODER_H
    wv = o_voc.fold_weight_norm(init_weights.init_state(arch.generator_param_shapes(h), 0))
    rwav = o_u2m.vocoder_infer(wv, h, rmel)
    assert relmax(wav.cpu().numpy(), rwav) < 2e-4, relmax(wav.cpu().numpy(), rwav)


def test_infer_tts_cli_synthetic(tmp_path):
    """the script itself, as a user runs it"""
    sys.path.insert(0, ROOT)
    import infer_tts
    out = tmp_path / "demo.npy"
    wav = infer_tts.main(["--synthetic", "--synthetic_tokens", "16", "-s", "250", "-o", str(out)])
    assert wav.shape == (16 * 512,) and np.isfinite(wav).all() and np.array_equal(np.load(out), wav)
    out2 = tmp_path / "demo.wav"
    infer_tts.main(["--synthetic", "--synthetic_tokens", "16", "-s", "250", "-o", str(out2)])
    assert os.path.getsize(out2) == 44 + 2 * 16 * 512
