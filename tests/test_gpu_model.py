"""Whole-path parity on a real MI355X through the reference's module API (Unit2Mel /
GaussianDiffusion / UNet1DConditionModel / Vocoder) against the committed golden fixtures
(outputs of the reference's own leaf modules) and the numpy oracle.

Stated tolerances (fp32 pipeline): UNet forward <= 2e-5 * absmax;
full sampler and vocoder <= 1e-4 * absmax (the reference's own fp32-vs-fp64 drift over a
50-step run is ~1e-6 relative)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


# the CPU oracle's result does not depend on the GEMM mode under test: computed once per case, reused by the other modes
_ORACLE_CACHE = {}


def oracle_once(key, fn):
    if key not in _ORACLE_CACHE:
        _ORACLE_CACHE[key] = fn()
    return _ORACLE_CACHE[key]


def relmax(a, b):
    return float(np.abs(a - b).max() / max(1e-30, np.abs(b).max()))


# every whole-path test runs in both GEMM modes at the SAME tolerances: "f32" = exact-fp32 MFMA (the default, the bench's `value`),
# "split_f16" = the opt-in mode with two fp16 terms per operand (csrc/conv_bf3.hip; UNet1DConditionModel.set_gemm_mode)
@pytest.fixture(scope="module", params=["f32", "split_f16"])
def unit2mel_gpu(request):
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from diffusion.unit2mel import Unit2Mel
    m = Unit2Mel(1280, 323, 80)
    m.to("cuda").eval()
    m.decoder.denoise_fn.set_gemm_mode(request.param)
    return m


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_unet_forward_vs_reference(golden, unit2mel_gpu, case, record_margin):
    g = golden("unet_fwd.npz")
    unet = unit2mel_gpu.decoder.denoise_fn
    t = g[f"{case}_t"]
    tt = dev(t) if t.dtype != np.int64 else torch.from_numpy(t).cuda()
    y = unet(dev(g[f"{case}_x"]), tt).sample.cpu().numpy()
    assert y.shape == g[f"{case}_y"].shape
    assert np.isfinite(y).all()
    record_margin(relmax(y, g[f"{case}_y"]), 2e-5)


@pytest.mark.parametrize("name,method,speedup,k_step,B", [
    ("dpm50", "dpm-solver", 20, 1000, 2), ("unipc20", "unipc", 50, 1000, 2),
    ("ddim10", "ddim", 100, 1000, 2), ("pndm10", "pndm", 100, 1000, 1), ("ddpm12", None, 1, 12, 2)])
def test_sampler_vs_reference(golden, unit2mel_gpu, monkeypatch, name, method, speedup, k_step, B, record_margin):
    """GaussianDiffusion.forward with the sampler RNG draws replaced by the recorded reference draws."""
    g = golden("sampler.npz")
    gd = unit2mel_gpu.decoder
    noise = g[name + "_noise"]                  # [n_draws, B, 1, M, T] in the reference's draw order
    draws = [dev(n) for n in noise]
    real = torch.randn

    def fake_randn(*a, **k):
        return draws.pop(0) if draws else real(*a, **k)
    monkeypatch.setattr(torch, "randn", fake_randn)
    gd.k_step = k_step
    try:
        y = gd(dev(g["cond"][:B]), infer=True, infer_speedup=speedup, method=method).cpu().numpy()
    finally:
        gd.k_step = 1000
    assert not draws
    assert y.shape == g[name + "_y"].shape
    record_margin(relmax(y, g[name + "_y"]), 1e-4)


def test_pndm_batch_gt1_raises_like_reference(unit2mel_gpu):
    gd = unit2mel_gpu.decoder
    with pytest.raises(RuntimeError):
        gd(torch.zeros(2, 16, 256, device="cuda"), infer=True, infer_speedup=100, method="pndm")
    with pytest.raises(NotImplementedError):
        gd(torch.zeros(1, 16, 256, device="cuda"), infer=True, infer_speedup=100, method="bogus")
    for method in ("dpm-solver", "unipc"):      # 1000 // 600 = 1 step < order 2: the reference's multistep solvers assert
        with pytest.raises(AssertionError):
            gd(torch.zeros(1, 16, 256, device="cuda"), infer=True, infer_speedup=600, method=method)


def test_vocoder_vs_reference(golden, record_margin):
    from diffusion.vocoder import Vocoder
    from encoder.hifi_vaegan.hifi_vaegan import Hifi_VAEGAN
    from lds import arch, init_weights
    g = golden("vocoder.npz")
    h = arch.SYNTHETIC_VOCODER_H
    state = init_weights.init_state(arch.generator_param_shapes(h), 0)
    voc = Vocoder.__new__(Vocoder)
    voc.vocoder = Hifi_VAEGAN(None, device="cuda", h=h, state=state)
    wav = voc.infer(dev(g["z"])).cpu().numpy()
    assert wav.shape == g["wav"].shape
    record_margin(relmax(wav, g["wav"]), 1e-4)


def test_unit2mel_end_to_end_vs_oracle(unit2mel_gpu, monkeypatch, record_margin):
    """units -> cond -> 20-step UniPC -> mel through Unit2Mel.forward vs the oracle pipeline."""
    from lds import arch, init_weights
    from oracle import schedule, unit2mel as o_u2m
    m = unit2mel_gpu
    B, T = 2, 24
    units = init_weights.uniform("e2e.units", (B, T, 1280), 21, -1.7, 1.7)
    spk = np.array([[3], [322]], dtype=np.int64)
    xT = init_weights.uniform("e2e.xT", (B, 1, 80, T), 21, -1.7, 1.7)
    monkeypatch.setattr(torch, "randn", lambda *a, **k: dev(xT))
    y = m(dev(units), None, spk_id=torch.from_numpy(spk).cuda(), infer=True, infer_speedup=50, method="unipc").cpu().numpy()
    w = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    cfg = arch.unet_config()
    ref = oracle_once("e2e", lambda: o_u2m.unit2mel(w, cfg, arch.unet_blocks(cfg), schedule.diffusion_buffers(), units, spk, xT[:, 0], "unipc", 50))
    assert y.shape == (B, T, 80)
    record_margin(relmax(y, ref), 1e-4)


def test_batch_shard_invariance_full_size(unit2mel_gpu):
    """BASELINE config sizes (T=512): kernels never reduce across the batch axis and the tile shapes (which fix the order of
    every reduction) are chosen from per-utterance sizes only, so an utterance's eps is bit-identical alone, inside a batch,
    at any position and next to any neighbours -- which is what makes the multi-GPU batch split exact."""
    from lds import init_weights
    unet = unit2mel_gpu.decoder.denoise_fn
    B, T = 4, 512
    x = dev(init_weights.uniform("inv.x", (B, 336, T), 31, -2, 2))
    t = dev(np.full((B,), 499.5, dtype=np.float32))
    full = unet(x, t).sample
    assert torch.isfinite(full).all()
    one = unet(x[2:3].contiguous(), t[2:3].contiguous()).sample      # alone
    assert torch.equal(full[2:3], one)
    perm = [2, 0, 3, 1]
    moved = unet(x[perm].contiguous(), t).sample                     # other positions / neighbours
    assert torch.equal(full[perm], moved)
    big = unet(torch.cat([x, x.flip(0), x], 0).contiguous(), torch.cat([t, t, t])).sample      # batch of 12
    assert torch.equal(big[:B], full) and torch.equal(big[2 * B:], full)


@pytest.mark.parametrize("B,T", [(2, 1000), (3, 77), (40, 512), (1, 2050)])
def test_unet_other_sizes(unit2mel_gpu, B, T):
    """lengths that are not multiples of 8 / 64 (size=-driven upsampling, ragged tiles everywhere), long utterances and a
    batch beyond the nominal one: finite, and utterance 0 bit-identical to its stand-alone evaluation"""
    from lds import init_weights
    unet = unit2mel_gpu.decoder.denoise_fn
    x = dev(init_weights.uniform(f"sz.{B}.{T}", (B, 336, T), 33, -2, 2))
    t = dev(np.full((B,), 250.25, dtype=np.float32))
    full = unet(x, t).sample
    assert full.shape == (B, 80, T) and torch.isfinite(full).all()
    one = unet(x[:1].contiguous(), t[:1].contiguous()).sample
    assert torch.equal(full[:1], one)
    assert float(full.abs().max()) > 1e-3


# ---- long utterances whose every level ends in a tile of one or two frames (2050 -> 1025 -> 513 -> 257; 1000 -> 500 -> 250 -> 125): the reference's
# forward_upsample_size path at scale (unet_1d_condition.py:789-797,1006-1009; resnet.py:157-160), every GEMM mode, default and latency mode,
# against the oracle itself -- so that when two HIP modes disagree the wrong one is known (VERDICT r3 missing #1).  ~15 s of CPU per size, once.
@pytest.mark.parametrize("B,T", [(1, 2050), (2, 1000)])
@pytest.mark.parametrize("latency", [False, True])
def test_unet_long_odd_lengths_vs_oracle(unit2mel_gpu, unet_weights, B, T, latency, record_margin):
    from lds import init_weights
    from oracle import unet1d
    cfg, blocks, w = unet_weights
    unet = unit2mel_gpu.decoder.denoise_fn
    x = init_weights.uniform(f"sz.{B}.{T}", (B, 336, T), 33, -2, 2)
    t = np.full((B,), 250.25, dtype=np.float32)
    ref = oracle_once(("long", B, T), lambda: unet1d.unet_forward(w, cfg, blocks, x, t))
    unet.set_latency_mode(latency)
    try:
        got = unet(dev(x), dev(t)).sample
        again = unet(dev(x), dev(t)).sample
    finally:
        unet.set_latency_mode(False)
    assert torch.equal(got, again)
    record_margin(relmax(got.cpu().numpy(), ref), 2e-5, "latency" if latency else "default")


def test_split_f16_range_check(unit2mel_gpu):
    """the split_f16 mode's precondition as a debug check (include/lds.h): every tensor stored as two fp16 planes must live at a scale of 2^-3 or
    more (below, the second term is subnormal: fewer than 22 bits, silently) and below 2^15.  The check returns the table of every stage's
    abs-max and raises naming the offenders: inputs scaled by 1e-3 push the first stages below the lower bound."""
    from lds import init_weights
    unet = unit2mel_gpu.decoder.denoise_fn
    x = dev(init_weights.uniform("rng.x", (1, 336, 96), 35, -2, 2))
    t = dev(np.array([300.5], dtype=np.float32))
    table = unet.check_split_f16_ranges(x, t, lo=0.0, hi=float("inf"))
    assert len(table) > 150 and all(np.isfinite(m) for _, m in table)
    tab = dict(table)
    assert "conv_in.out" in tab and "mid.tfm.att1" in tab and "out.gn" in tab and "up3.tfm2.ff1" in tab
    assert tab["conv_in.out"] > 2.0 ** -3
    with pytest.raises(ValueError, match="abs-max outside"):
        unet.check_split_f16_ranges(x, t, hi=0.5 * tab["conv_in.out"])      # an upper bound below what the first stage produces
    small = dict(unet.check_split_f16_ranges(x * 1e-3, t, lo=0.0, hi=float("inf")))
    assert small["conv_in.out"] < 2.0 ** -3
    with pytest.raises(ValueError, match="conv_in.out"):
        unet.check_split_f16_ranges(x * 1e-3, t)
    with pytest.raises(ValueError):
        unet.set_gemm_mode("split_bf16")      # the removed mode


def test_checkpoint_formats_and_facade(tmp_path, unit2mel_gpu, monkeypatch):
    """reference on-disk formats end to end: <dir>/config.yaml + {'global_step','model'} .pt (tools/saver.py:85-109) and
    decoder.pth = {'config': h, 'model': weight-norm state_dict} (hifi_vaegan.py:6-8,57-61), loaded through
    load_model_vocoder / DiffusionSVC and run units -> wav."""
    import os
    import shutil
    from conftest import ROOT
    from lds import arch, init_weights
    from tools.infer_tools import DiffusionSVC
    h = arch.SYNTHETIC_VOCODER_H
    vdir = tmp_path / "hifi-vaegan"
    vdir.mkdir()
    gstate = {k: torch.from_numpy(v) for k, v in init_weights.init_state(arch.generator_param_shapes(h), 0).items()}
    torch.save({"config": h, "model": gstate}, vdir / "decoder.pth")
    edir = tmp_path / "exp"
    edir.mkdir()
    cfg = open(os.path.join(ROOT, "tests", "golden", "config_like_reference.yaml")).read().replace("pretrain/hifi-vaegan", str(vdir))
    (edir / "config.yaml").write_text(cfg)
    torch.save({"global_step": 7, "model": {k: v.cpu() for k, v in unit2mel_gpu.state_dict().items()}}, edir / "model_7.pt")
    svc = DiffusionSVC(device="cuda")
    svc.load_model(str(edir / "model_7.pt"), f0_min=65, f0_max=800)
    svc.model.decoder.denoise_fn.set_gemm_mode(unit2mel_gpu.decoder.denoise_fn._gemm_mode)      # same GEMM mode as the fixture's model
    assert svc.vocoder.vocoder_hop_size == 512 and svc.vocoder.dimension == 80 and svc.vocoder.vocoder_sample_rate == 44100
    B, T = 1, 16
    units = dev(init_weights.uniform("facade.units", (B, T, 1280), 3, -1.7, 1.7))
    xT = dev(init_weights.uniform("facade.xT", (B, 1, 80, T), 3, -1.7, 1.7))
    monkeypatch.setattr(torch, "randn", lambda *a, **k: xT.clone())      # the sampler updates x in place
    wav = svc.infer(units, f0=None, volume=None, spk_id=5, infer_speedup=250, method="dpm-solver")
    assert wav.shape == (B, 1, T * 512) and bool(torch.isfinite(wav).all())
    mel = unit2mel_gpu(units, None, spk_id=torch.full((B, 1), 5, device="cuda"), infer=True, infer_speedup=250, method="dpm-solver")
    assert torch.equal(svc.vocoder.infer(mel), wav)            # same weights through both loaders -> identical result


@pytest.mark.parametrize("name,method,speedup,k_step,B", [
    ("dpm20", "dpm-solver", 10, 200, 2), ("unipc10", "unipc", 20, 200, 2), ("ddim8", "ddim", 25, 200, 2),
    ("pndm8", "pndm", 25, 200, 1), ("ddpm12", None, 1, 12, 2)])
def test_sampler_shallow_vs_reference(golden, unit2mel_gpu, monkeypatch, name, method, speedup, k_step, B, record_margin):
    """Shallow-diffusion entry of GaussianDiffusion.forward (reference diffusion.py:203-211): gt_spec + k_step ->
    x = q_sample(norm_spec(gt_spec), k_step - 1), every solver on the cut schedule betas[:k_step].  The reference's
    outputs stay O(1) here (|y| <= 3.5; the DDPM case never reaches the clamp), so 1e-4 * absmax is ~3e-4 absolute."""
    g = golden("sampler_shallow.npz")
    gd = unit2mel_gpu.decoder
    draws = [dev(n) for n in g[name + "_noise"]]      # draw 0 = q_sample's randn_like, then the DDPM per-step draws

    def fake(*a, **k):
        return draws.pop(0)
    monkeypatch.setattr(torch, "randn", fake)
    monkeypatch.setattr(torch, "randn_like", fake)
    y = gd(dev(g["cond"][:B]), gt_spec=dev(g["gt_spec"][:B]), infer=True, infer_speedup=speedup, method=method, k_step=k_step).cpu().numpy()
    assert not draws
    ref = g[name + "_y"]
    assert y.shape == ref.shape and np.abs(ref).max() < 4.0
    record_margin(relmax(y, ref), 1e-4)


def test_vocoder_resblock2_vs_reference(golden, record_margin):
    """Generator with resblock '2' (reference models.py:201-222,230): one dilated conv per residual step"""
    import json
    from encoder.hifi_vaegan.hifi_vaegan import Hifi_VAEGAN
    from lds import arch, init_weights
    g = golden("vocoder_rb2.npz")
    h = json.loads(bytes(g["h_json"]).decode())
    voc = Hifi_VAEGAN(None, device="cuda", h=h, state=init_weights.init_state(arch.generator_param_shapes(h), 0))
    wav = voc(dev(g["z"])).cpu().numpy()
    assert wav.shape == g["wav"].shape
    record_margin(relmax(wav, g["wav"]), 1e-4)


def test_unet_full_size_vs_oracle(unit2mel_gpu, unet_weights, record_margin):
    """BASELINE size: one T = 512 utterance-forward (every level's real tile shapes: 512/256/128/64 frames) against the
    numpy oracle, plus a second utterance in the same launch with another (fractional) timestep."""
    from lds import init_weights
    from oracle import unet1d
    cfg, blocks, w = unet_weights
    unet = unit2mel_gpu.decoder.denoise_fn
    x = init_weights.uniform("full.x", (2, 336, 512), 41, -2, 2)
    t = np.array([873.25, 40.5], dtype=np.float32)
    got = unet(dev(x), dev(t)).sample.cpu().numpy()
    ref = oracle_once("full0", lambda: unet1d.unet_forward(w, cfg, blocks, x[:1], t[:1]))
    record_margin(relmax(got[:1], ref), 2e-5, "utt0")
    ref1 = oracle_once("full1", lambda: unet1d.unet_forward(w, cfg, blocks, x[1:], t[1:]))
    record_margin(relmax(got[1:], ref1), 2e-5, "utt1")


@pytest.mark.parametrize("method,speedup", [("dpm-solver", 100), ("unipc", 100)])
def test_sampler_bench_size_vs_oracle(unit2mel_gpu, unet_weights, monkeypatch, record_margin, method, speedup):
    """Parity where the bench runs (VERDICT r2 #4a): one 512-frame utterance through a 10-NFE DPM-Solver++ / UniPC run -- every
    level's real tile shapes, ten evaluations of error growth at full width -- against oracle.solvers over the oracle UNet
    (reference dpm_solver_pytorch.py:1171-1213, uni_pc.py:590-658).  ~25 s of CPU per case."""
    from lds import arch, init_weights
    from oracle import schedule, solvers, unit2mel as o_u2m
    cfg, blocks, w = unet_weights
    gd = unit2mel_gpu.decoder
    T = 512
    cond = init_weights.uniform("bench512.cond", (1, T, 256), 51, -1, 1)                # [B, T, H] as GaussianDiffusion.forward takes it
    xT = init_weights.uniform("bench512.xT", (1, 1, 80, T), 52, -1.7, 1.7)
    monkeypatch.setattr(torch, "randn", lambda *a, **k: dev(xT))
    y = gd(dev(cond), infer=True, infer_speedup=speedup, method=method).cpu().numpy()       # [1, T, 80]
    f = o_u2m.make_eps_fn(w, cfg, blocks, np.ascontiguousarray(cond.transpose(0, 2, 1)))
    ref = oracle_once(("bench512", method), lambda: solvers.sample(f, schedule.diffusion_buffers(), xT[:, 0], method, speedup))         # [1, 80, T]
    ref = np.ascontiguousarray(ref.transpose(0, 2, 1))
    assert y.shape == ref.shape == (1, T, 80)
    record_margin(relmax(y, ref), 1e-4)


def test_vocoder_512_frames_vs_oracle(record_margin):
    """the bench's utterance length: 512 mel frames -> 262,144 samples against the numpy oracle (~20 s of CPU)"""
    from encoder.hifi_vaegan.hifi_vaegan import Hifi_VAEGAN
    from lds import arch, init_weights
    from oracle import vocoder as o_voc
    h = arch.SYNTHETIC_VOCODER_H
    state = init_weights.init_state(arch.generator_param_shapes(h), 0)
    voc = Hifi_VAEGAN(None, device="cuda", h=h, state=state)
    z = init_weights.uniform("full512.voc.z", (1, 512, 80), 44, -1.5, 1.5)
    wav = voc(dev(z)).cpu().numpy()
    ref = o_voc.generator_forward(o_voc.fold_weight_norm(state), h, np.ascontiguousarray(z.transpose(0, 2, 1)))
    assert wav.shape == ref.shape == (1, 1, 262144)
    record_margin(relmax(wav, ref), 1e-4)


def test_vocoder_full_size_vs_oracle(record_margin):
    """128 mel frames -> 65,536 samples: every upsampling stage runs its real tile shapes (256@1024 ... 16@65536 columns,
    the small-channel tail included) against the numpy oracle."""
    from encoder.hifi_vaegan.hifi_vaegan import Hifi_VAEGAN
    from lds import arch, init_weights
    from oracle import vocoder as o_voc
    h = arch.SYNTHETIC_VOCODER_H
    state = init_weights.init_state(arch.generator_param_shapes(h), 0)
    voc = Hifi_VAEGAN(None, device="cuda", h=h, state=state)
    z = init_weights.uniform("full.voc.z", (1, 128, 80), 42, -1.5, 1.5)
    wav = voc(dev(z)).cpu().numpy()
    ref = o_voc.generator_forward(o_voc.fold_weight_norm(state), h, np.ascontiguousarray(z.transpose(0, 2, 1)))
    assert wav.shape == ref.shape == (1, 1, 65536)
    record_margin(relmax(wav, ref), 1e-4)


def test_vocoder_ragged_batch_vs_oracle(record_margin):
    """2 utterances x 37 frames: every stage ends in a partial tile (37*8 = 296 columns ... 18,944 samples), the polyphase
    upsamplers scatter across utterance boundaries of the K4P tensors, and the batch index enters every address."""
    from encoder.hifi_vaegan.hifi_vaegan import Hifi_VAEGAN
    from lds import arch, init_weights
    from oracle import vocoder as o_voc
    h = arch.SYNTHETIC_VOCODER_H
    state = init_weights.init_state(arch.generator_param_shapes(h), 0)
    voc = Hifi_VAEGAN(None, device="cuda", h=h, state=state)
    z = init_weights.uniform("ragged.voc.z", (2, 37, 80), 43, -1.5, 1.5)
    wav = voc(dev(z)).cpu().numpy()
    ref = o_voc.generator_forward(o_voc.fold_weight_norm(state), h, np.ascontiguousarray(z.transpose(0, 2, 1)))
    assert wav.shape == ref.shape == (2, 1, 37 * 512)
    record_margin(relmax(wav, ref), 1e-4)
    one = voc(dev(z[1:2])).cpu().numpy()      # an utterance alone = inside the batch
    assert np.array_equal(one[0], wav[1])


def test_parent_load_state_dict_repacks_weights(unit2mel_gpu):
    """nn.Module.load_state_dict on a PARENT recurses through the children's _load_from_state_dict, never their
    load_state_dict: after a forward has packed the weights, loading other weights through Unit2Mel must drop the
    packed copies of the UNet and of the front end (reference behaviour: load_model_vocoder -> model.load_state_dict)."""
    from diffusion.unit2mel import Unit2Mel
    from lds import init_weights
    m = Unit2Mel(1280, 323, 80).to("cuda").eval()
    B, T = 1, 16
    units = dev(init_weights.uniform("reload.units", (B, T, 1280), 3, -1.7, 1.7))
    x = dev(init_weights.uniform("reload.x", (B, 336, T), 3, -2, 2))
    t = dev(np.array([100.0], dtype=np.float32))
    spk = torch.full((B, 1), 5, device="cuda")
    e0 = m.decoder.denoise_fn(x, t).sample.clone()
    c0 = m._native_embed().forward(units, spk).clone()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    sd["decoder.denoise_fn.conv_out.bias"] += 0.5
    sd["unit_embed.bias"] += 0.25
    m.load_state_dict(sd)
    e1 = m.decoder.denoise_fn(x, t).sample
    c1 = m._native_embed().forward(units, spk)
    assert torch.allclose(e1, e0 + 0.5, atol=1e-5) and not torch.equal(e1, e0)
    assert torch.allclose(c1, c0 + 0.25, atol=1e-5) and not torch.equal(c1, c0)
    # and loading through the diffusion wrapper alone
    sd2 = {k: v.clone() for k, v in m.decoder.state_dict().items()}
    sd2["denoise_fn.conv_out.bias"] -= 0.5
    m.decoder.load_state_dict(sd2)
    assert torch.allclose(m.decoder.denoise_fn(x, t).sample, e0, atol=1e-5)


# ---- latency mode (lds_unet_set_latency_mode): tile shapes and cluster split-K chosen from the ACTUAL batch ----------
# Same fixtures, same tolerances as the default mode; the oracle results are shared with the tests above (oracle_once).
@pytest.fixture(scope="module", params=["f32", "split_f16"])
def unit2mel_latency(request):
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from diffusion.unit2mel import Unit2Mel
    m = Unit2Mel(1280, 323, 80)
    m.to("cuda").eval()
    m.decoder.denoise_fn.set_gemm_mode(request.param)
    m.decoder.denoise_fn.set_latency_mode(True)
    assert m.decoder.denoise_fn.native().latency_mode() == 1
    return m


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_latency_unet_forward_vs_reference(golden, unit2mel_latency, case, record_margin):
    g = golden("unet_fwd.npz")
    unet = unit2mel_latency.decoder.denoise_fn
    t = g[f"{case}_t"]
    tt = dev(t) if t.dtype != np.int64 else torch.from_numpy(t).cuda()
    y = unet(dev(g[f"{case}_x"]), tt).sample.cpu().numpy()
    assert np.isfinite(y).all()
    record_margin(relmax(y, g[f"{case}_y"]), 2e-5)


def test_latency_unet_full_size_vs_oracle(unit2mel_latency, unet_weights, record_margin):
    """one and two T = 512 utterances (the sizes the mode is for: every deep reduction runs as a workgroup cluster) vs the oracle;
    a repeated call is bit-identical (the partial tiles are summed in a fixed order, whatever the order of arrival)"""
    from lds import init_weights
    from oracle import unet1d
    cfg, blocks, w = unet_weights
    unet = unit2mel_latency.decoder.denoise_fn
    x = init_weights.uniform("full.x", (2, 336, 512), 41, -2, 2)
    t = np.array([873.25, 40.5], dtype=np.float32)
    ref = oracle_once("full0", lambda: unet1d.unet_forward(w, cfg, blocks, x[:1], t[:1]))
    ref1 = oracle_once("full1", lambda: unet1d.unet_forward(w, cfg, blocks, x[1:], t[1:]))
    one = unet(dev(x[:1]), dev(t[:1])).sample
    record_margin(relmax(one.cpu().numpy(), ref), 2e-5, "B1")
    for _ in range(3):
        assert torch.equal(unet(dev(x[:1]), dev(t[:1])).sample, one)
    two = unet(dev(x), dev(t)).sample.cpu().numpy()
    record_margin(relmax(two[:1], ref), 2e-5, "B2.utt0")
    record_margin(relmax(two[1:], ref1), 2e-5, "B2.utt1")


@pytest.mark.parametrize("B,T", [(1, 77), (2, 1000), (1, 2050), (5, 512)])
def test_latency_unet_other_sizes(unit2mel_latency, B, T, record_margin):
    """ragged lengths and other batches: the latency mode against the default mode of the same model (both within 2e-5 of the oracle at
    the sizes checked above; here 4e-5 between them)"""
    from lds import init_weights
    unet = unit2mel_latency.decoder.denoise_fn
    x = dev(init_weights.uniform(f"sz.{B}.{T}", (B, 336, T), 33, -2, 2))
    t = dev(np.full((B,), 250.25, dtype=np.float32))
    lat = unet(x, t).sample
    unet.set_latency_mode(False)
    try:
        base = unet(x, t).sample
    finally:
        unet.set_latency_mode(True)
    assert torch.isfinite(lat).all()
    record_margin(relmax(lat.cpu().numpy(), base.cpu().numpy()), 4e-5)


@pytest.mark.parametrize("name,method,speedup,k_step,B", [
    ("dpm50", "dpm-solver", 20, 1000, 2), ("unipc20", "unipc", 50, 1000, 2), ("ddpm12", None, 1, 12, 2)])
def test_latency_sampler_vs_reference(golden, unit2mel_latency, monkeypatch, name, method, speedup, k_step, B, record_margin):
    g = golden("sampler.npz")
    gd = unit2mel_latency.decoder
    draws = [dev(n) for n in g[name + "_noise"]]
    real = torch.randn
    monkeypatch.setattr(torch, "randn", lambda *a, **k: draws.pop(0) if draws else real(*a, **k))
    gd.k_step = k_step
    try:
        y = gd(dev(g["cond"][:B]), infer=True, infer_speedup=speedup, method=method).cpu().numpy()
    finally:
        gd.k_step = 1000
    assert not draws
    record_margin(relmax(y, g[name + "_y"]), 1e-4)


def test_latency_sampler_bench_size_vs_oracle(unit2mel_latency, unet_weights, monkeypatch, record_margin):
    """the caller the mode is for: one 512-frame utterance, 10-NFE DPM-Solver++, against oracle.solvers over the oracle UNet"""
    from lds import init_weights
    from oracle import schedule, solvers, unit2mel as o_u2m
    cfg, blocks, w = unet_weights
    gd = unit2mel_latency.decoder
    T = 512
    cond = init_weights.uniform("bench512.cond", (1, T, 256), 51, -1, 1)
    xT = init_weights.uniform("bench512.xT", (1, 1, 80, T), 52, -1.7, 1.7)
    monkeypatch.setattr(torch, "randn", lambda *a, **k: dev(xT))
    y = gd(dev(cond), infer=True, infer_speedup=100, method="dpm-solver").cpu().numpy()
    f = o_u2m.make_eps_fn(w, cfg, blocks, np.ascontiguousarray(cond.transpose(0, 2, 1)))
    ref = oracle_once(("bench512", "dpm-solver"), lambda: solvers.sample(f, schedule.diffusion_buffers(), xT[:, 0], "dpm-solver", 100))
    record_margin(relmax(y, np.ascontiguousarray(ref.transpose(0, 2, 1))), 1e-4)


# ---- ragged batches in one call (lds_unet_forward_ragged / lds_sampler_run_ragged): per-utterance lengths inside buffers of T frames ----
@pytest.mark.parametrize("T,lens,latency", [(77, [77, 50, 33], False), (512, [512, 300, 272, 401], False), (130, [64, 130, 7, 129, 2], False),
                                            (512, [301, 512], True), (200, [200] * 3, False)])
def test_ragged_unet_forward_vs_alone(unit2mel_gpu, unet_weights, T, lens, latency, record_margin):
    """every utterance of a padded batch against the same utterance evaluated alone at its own length (odd lengths at every level: the
    decoder resamples each utterance to its own skip lengths; lengths 2 and 7: levels of one frame); zeros beyond each length"""
    from lds import init_weights
    unet = unit2mel_gpu.decoder.denoise_fn
    unet.set_latency_mode(latency)
    try:
        B = len(lens)
        x = init_weights.uniform(f"rag.{T}.{B}", (B, 336, T), 61, -2, 2)
        t = np.linspace(40.5, 873.25, B).astype(np.float32)
        dx, dt_ = dev(x), dev(t)
        got = unet.native().forward(dx[:, :80].contiguous(), dx[:, 80:].contiguous(), dt_, lengths=lens)
        assert torch.isfinite(got).all()
        worst = 0.0
        for b, n in enumerate(lens):
            alone = unet(dx[b:b + 1, :, :n].contiguous(), dt_[b:b + 1]).sample
            worst = max(worst, relmax(got[b:b + 1, :, :n].cpu().numpy(), alone.cpu().numpy()))
            assert float(got[b, :, n:].abs().max()) == 0.0 if n < T else True
        record_margin(worst, 2e-5)
        if T == 77:      # and against the oracle itself for one utterance
            from oracle import unet1d
            cfg, blocks, w = unet_weights
            ref = unet1d.unet_forward(w, cfg, blocks, x[1:2, :, :50], t[1:2])
            record_margin(relmax(got[1:2, :, :50].cpu().numpy(), ref), 2e-5, "oracle")
    finally:
        unet.set_latency_mode(False)


def test_ragged_ddpm_vs_alone(unit2mel_gpu, monkeypatch, record_margin):
    """the ancestral sampler on a ragged batch (one noise draw per step, padded like the state): 12 steps, every utterance against its own run"""
    from lds import init_weights
    gd = unit2mel_gpu.decoder
    T, lens, K = 64, [64, 37], 12
    B = len(lens)
    cond = init_weights.uniform("rag.d.cond", (B, T, 256), 73, -1, 1)
    draws = init_weights.uniform("rag.d.noise", (K + 1, B, 1, 80, T), 74, -1.7, 1.7)
    gd.k_step = K
    try:
        q = [dev(d) for d in draws]
        monkeypatch.setattr(torch, "randn", lambda *a, **k: q.pop(0))
        y = gd.forward_ragged(dev(cond), lens, infer_speedup=1, method=None)
        worst = 0.0
        for b, n in enumerate(lens):
            q = [dev(np.ascontiguousarray(d[b:b + 1, :, :, :n])) for d in draws]
            monkeypatch.setattr(torch, "randn", lambda *a, **k: q.pop(0))
            alone = gd(dev(np.ascontiguousarray(cond[b:b + 1, :n])), infer=True, infer_speedup=1, method=None)
            worst = max(worst, relmax(y[b:b + 1, :n].cpu().numpy(), alone.cpu().numpy()))
    finally:
        gd.k_step = 1000
    record_margin(worst, 1e-4)


@pytest.mark.parametrize("method,speedup", [("dpm-solver", 250), ("unipc", 250), ("ddim", 250)])
def test_ragged_sampler_vs_alone(unit2mel_gpu, monkeypatch, method, speedup, record_margin):
    """GaussianDiffusion.forward_ragged: a padded batch of three lengths through a 4-step run, every utterance against its own run alone"""
    from lds import init_weights
    gd = unit2mel_gpu.decoder
    T, lens = 96, [96, 61, 40]
    B = len(lens)
    cond = init_weights.uniform("rag.s.cond", (B, T, 256), 71, -1, 1)
    xT = init_weights.uniform("rag.s.xT", (B, 1, 80, T), 72, -1.7, 1.7)
    monkeypatch.setattr(torch, "randn", lambda *a, **k: dev(xT))
    y = gd.forward_ragged(dev(cond), lens, infer_speedup=speedup, method=method)
    assert y.shape == (B, T, 80) and torch.isfinite(y).all()
    worst = 0.0
    for b, n in enumerate(lens):
        monkeypatch.setattr(torch, "randn", lambda *a, **k: dev(np.ascontiguousarray(xT[b:b + 1, :, :, :n])))
        alone = gd(dev(np.ascontiguousarray(cond[b:b + 1, :n])), infer=True, infer_speedup=speedup, method=method)
        worst = max(worst, relmax(y[b:b + 1, :n].cpu().numpy(), alone.cpu().numpy()))
        assert n == T or float(y[b, n:].abs().max()) == 0.0
    record_margin(worst, 1e-4)


@pytest.mark.parametrize("T,lens", [(40, [40, 24, 33]), (96, [17, 96, 60, 1])])
def test_ragged_vocoder_vs_alone(T, lens, record_margin):
    """lds_vocoder_forward_ragged: a padded batch of latents with garbage beyond each length; every utterance's samples against the
    utterance decoded alone, zeros beyond"""
    from encoder.hifi_vaegan.hifi_vaegan import Hifi_VAEGAN
    from lds import arch, init_weights
    h = arch.SYNTHETIC_VOCODER_H
    voc = Hifi_VAEGAN(None, device="cuda", h=h, state=init_weights.init_state(arch.generator_param_shapes(h), 0))
    B = len(lens)
    z = dev(init_weights.uniform(f"ragv.{T}", (B, T, h["inter_channels"]), 81, -1.5, 1.5))
    wav = voc.forward_ragged(z, lens)
    hop = h["hop_size"]
    assert wav.shape == (B, 1, T * hop) and torch.isfinite(wav).all()
    worst = 0.0
    for b, n in enumerate(lens):
        alone = voc(z[b:b + 1, :n].contiguous())
        worst = max(worst, relmax(wav[b:b + 1, :, :n * hop].cpu().numpy(), alone.cpu().numpy()))
        assert n == T or float(wav[b, :, n * hop:].abs().max()) == 0.0
    record_margin(worst, 1e-4)
