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


# ---- text2semantic RoFormer ----
@pytest.fixture(scope="module")
def lm_gpu():
    import yaml
    from text2semantic.utils import get_language_model
    args = yaml.safe_load(open(os.path.join(ROOT, "tests", "golden", "config_lm_like_reference.yaml")))
    m = get_language_model(**args).to("cuda").eval()
    return m


def test_roformer_state_dict_and_encoder(golden, lm_gpu):
    import json
    g = golden("roformer.npz")
    man = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest_roformer.json")))
    sd = lm_gpu.state_dict()
    assert set(sd) == set(man) and all(list(sd[k].shape) == man[k] for k in man)
    assert sd["semantic_decoder.cls.predictions.decoder.weight"].data_ptr() == sd["semantic_decoder.roformer.embeddings.word_embeddings.weight"].data_ptr()
    enc = lm_gpu.encode(torch.from_numpy(g["phone"]).cuda(), torch.from_numpy(g["tone"]).cuda(), torch.from_numpy(g["spk_id"]).cuda()).cpu().numpy()
    assert relmax(enc, g["enc"]) < 2e-5, relmax(enc, g["enc"])


@pytest.mark.parametrize("tag,do_sample,max_length", [("greedy", False, 24), ("sample", True, 40), ("eos", True, 40)])
def test_roformer_generate_vs_reference(golden, lm_gpu, monkeypatch, tag, do_sample, max_length):
    """Roformer.generate with the call 22_infer_tts.py:83-98 makes (top_k 5, top_p 1, temperature 1, no repetition penalty) against
    the reference's own token sequences and per-step logits; the sampler's uniforms are the ones recorded from the reference run"""
    g = golden("roformer.npz")
    m = lm_gpu
    bias = m.semantic_decoder.cls.predictions.bias
    if tag == "eos":
        with torch.no_grad():
            bias[m.semantic_eos_token_id] += float(g["eos_bias"])
        m._native = None
    try:
        if do_sample:
            u = g[tag + "_uniforms"]
            full = np.zeros((max_length - 1, u.shape[1]), dtype=np.float32)
            full[: u.shape[0]] = u
            monkeypatch.setattr(torch, "rand", lambda *a, **k: dev(full))
        toks, logits = m.generate(torch.from_numpy(g["phone"]).cuda(), torch.from_numpy(g["tone"]).cuda(), attention_mask=None, use_cache=None,
                                  max_length=max_length, do_sample=do_sample, temperature=1.0, top_k=5, top_p=1.0, repetition_penalty=1.0, num_beams=1,
                                  no_repeat_ngram_size=0, early_stopping=True, spk_id=torch.from_numpy(g["spk_id"]).cuda(), end_gate_threshold=None,
                                  return_logits=True)
    finally:
        if tag == "eos":
            with torch.no_grad():
                bias[m.semantic_eos_token_id] -= float(g["eos_bias"])
            m._native = None
    want = g[tag + "_tokens"]
    assert toks.shape == want.shape, (toks.shape, want.shape)
    assert np.array_equal(toks.cpu().numpy(), want)
    assert relmax(logits.cpu().numpy(), g[tag + "_logits"]) < 2e-5


def test_roformer_ragged_batch_vs_reference(golden, lm_gpu, monkeypatch):
    """two phone lengths in one (right-padded) batch: attention_mask through the encoder's self-attention and the decoder's
    cross-attention (reference roformer.py:182,209-214,229-236) against the reference's own tokens / logits / encoder states; and the
    padded row generates exactly what it generates alone, un-padded"""
    g = golden("roformer.npz")
    m = lm_gpu
    lens, mask = g["ragged_len"], g["ragged_mask"]
    phone, tone, spk = [torch.from_numpy(g[k]).cuda() for k in ("phone", "tone", "spk_id")]
    enc = m.encode(phone, tone, spk, attention_mask=torch.from_numpy(mask).cuda()).cpu().numpy()
    for b, n in enumerate(lens):
        assert relmax(enc[b, :n], g["ragged_enc"][b, :n]) < 2e-5
    u = g["ragged_uniforms"]
    full = np.zeros((31, u.shape[1]), dtype=np.float32)
    full[: u.shape[0]] = u
    monkeypatch.setattr(torch, "rand", lambda *a, **k: dev(full))
    kw = dict(use_cache=None, max_length=32, do_sample=True, temperature=1.0, top_k=5, top_p=1.0, repetition_penalty=1.0, num_beams=1,
              no_repeat_ngram_size=0, early_stopping=True, end_gate_threshold=None, return_logits=True)
    toks, logits = m.generate(phone, tone, attention_mask=torch.from_numpy(mask).cuda(), spk_id=spk, **kw)
    assert np.array_equal(toks.cpu().numpy(), g["ragged_tokens"])
    assert relmax(logits.cpu().numpy(), g["ragged_logits"]) < 2e-5
    # the short utterance alone, without padding: same tokens, same logits bit for bit (no kernel reduces across rows or reads the pad)
    n1 = int(lens[1])
    monkeypatch.setattr(torch, "rand", lambda *a, **k: dev(np.ascontiguousarray(full[:, 1:2])))
    t1, l1 = m.generate(phone[1:2, :n1].contiguous(), tone[1:2, :n1].contiguous(), attention_mask=None, spk_id=spk[1:2, :n1].contiguous(), **kw)
    assert torch.equal(t1[0], toks[1]) and torch.equal(l1[:, 0], logits[:, 1])


def test_ragged_batch_phones_to_wav_equals_utterances_alone():
    """BASELINE configs[4] in small: a batch of sentences with different phone lengths and different token lengths (one row stops at an
    early EOS) -> per-row tokens -> sampler + vocoder by length bucket; every utterance's mel and waveform equal, bit for bit, what the
    utterance gives when it runs alone (un-padded, batch of one)."""
    import infer_tts
    from lds import init_weights
    svc, codebook, _ = infer_tts.synthetic_pipeline("cuda", 8)
    lm = infer_tts.synthetic_lm("cuda")
    B, L = 4, 12
    lens = [12, 7, 12, 9]
    idx = np.arange(B * L).reshape(B, L)
    phones = torch.from_numpy((idx * 7 % 107 + 1).astype(np.int64)).cuda()
    tones = torch.from_numpy((idx * 5 % 12).astype(np.int64)).cuda()
    # one row ends early: EOS bias makes sampling hit EOS after a few tokens for some rows only with these uniforms
    bias = lm.semantic_decoder.cls.predictions.bias
    with torch.no_grad():
        bias[lm.semantic_eos_token_id] += 19.0
    lm._native = None
    try:
        torch.manual_seed(7)
        rows = infer_tts.text2semantic_rows(lm, phones, tones, 1, 24, phone_lengths=lens)
        alone = []
        for b in range(B):
            torch.manual_seed(7)
            # the same uniforms column: generate draws rand(max_length - 1, B) -- take row b's column by replaying the full draw
            u = torch.rand(23, B, device="cuda")[:, b:b + 1].contiguous()
            real = torch.rand
            torch.rand = lambda *a, **k: u
            try:
                alone.append(infer_tts.text2semantic_rows(lm, phones[b:b + 1, :lens[b]].contiguous(), tones[b:b + 1, :lens[b]].contiguous(), 1, 24)[0])
            finally:
                torch.rand = real
    finally:
        with torch.no_grad():
            bias[lm.semantic_eos_token_id] -= 19.0
        lm._native = None
    assert len({int(r.numel()) for r in rows}) >= 2, [int(r.numel()) for r in rows]      # really ragged
    assert all(int(r.max()) < lm.semantic_eos_token_id for r in rows if r.numel())
    for b in range(B):      # padded-batch tokens = the sentence alone (prefix up to the batch's common stop)
        n = min(rows[b].numel(), alone[b].numel())
        assert n > 0 and torch.equal(rows[b][:n], alone[b][:n])

    def noise(idx_list, T):      # per-utterance start noise, a function of the utterance only
        return torch.stack([torch.from_numpy(init_weights.uniform(f"ragged.xT.{i}", (1, 80, T), 3, -1.7, 1.7)) for i in idx_list]).cuda()
    got = infer_tts.synthesize_ragged(svc, codebook, rows, 1, 250, "dpm-solver", noise_fn=noise)
    for i, r in enumerate(rows):
        one = infer_tts.synthesize_ragged(svc, codebook, [r], 1, 250, "dpm-solver", noise_fn=lambda idx_list, T, i=i: noise([i], T))
        assert got[i][0].shape == (r.numel(), 80) and got[i][1].shape == (r.numel() * 512,)
        assert torch.equal(got[i][0], one[0][0]) and torch.equal(got[i][1], one[0][1])


@pytest.mark.parametrize("threads", [False, True])
def test_ragged_buckets_on_streams_equal_sequential(threads):
    """synthesize_ragged(streams=3): the length buckets overlap on three HIP streams, each with its own workspace (and, without injected
    noise, on three host threads); every utterance's mel and waveform are the tensors of the sequential run, bit for bit"""
    import infer_tts
    from lds import init_weights
    svc, codebook, _ = infer_tts.synthetic_pipeline("cuda", 8)
    lens = [40, 24, 40, 33, 17, 24, 56]
    rows = [torch.from_numpy((np.arange(n) * (7 + i) % 4096).astype(np.int64)).cuda() for i, n in enumerate(lens)]
    if threads:      # the start noise comes from torch's generator: seeded runs draw the same x_T only if the draws happen in the same order,
        # so compare the deterministic part -- a run whose noise is injected -- on one host thread, and here only check completion + shapes
        got = infer_tts.synthesize_ragged(svc, codebook, rows, 1, 250, "dpm-solver", streams=3)
        torch.cuda.synchronize()
        for (mel, wav), n in zip(got, lens):
            assert mel.shape == (n, 80) and wav.shape == (n * 512,) and bool(torch.isfinite(mel).all()) and bool(torch.isfinite(wav).all())
        return

    def noise(idx_list, T):
        return torch.stack([torch.from_numpy(init_weights.uniform(f"streams.xT.{i}", (1, 80, T), 3, -1.7, 1.7)) for i in idx_list]).cuda()
    seq = infer_tts.synthesize_ragged(svc, codebook, rows, 1, 250, "dpm-solver", noise_fn=noise)
    par = infer_tts.synthesize_ragged(svc, codebook, rows, 1, 250, "dpm-solver", noise_fn=noise, streams=3)
    torch.cuda.synchronize()
    for a, b in zip(seq, par):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    # the same in latency mode: the cluster split-K's partial tiles and arrival counters live in the per-stream workspaces, and the
    # fixed-order sum makes a bucket's result independent of what runs beside it
    unet = svc.model.decoder.denoise_fn
    unet.set_latency_mode(True)
    try:
        seq = infer_tts.synthesize_ragged(svc, codebook, rows, 1, 250, "dpm-solver", noise_fn=noise)
        par = infer_tts.synthesize_ragged(svc, codebook, rows, 1, 250, "dpm-solver", noise_fn=noise, streams=3)
        torch.cuda.synchronize()
    finally:
        unet.set_latency_mode(False)
    for a, b in zip(seq, par):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_ragged_masked_batch_equals_utterances_alone():
    """synthesize_ragged_masked: the sampler runs ONE padded batch with per-utterance lengths inside the kernels; every utterance's mel and
    waveform against the bucketed run (= each utterance alone), within the whole-path tolerance"""
    import infer_tts
    from lds import init_weights
    svc, codebook, _ = infer_tts.synthetic_pipeline("cuda", 8)
    lens = [40, 24, 40, 33, 17, 24, 56]
    rows = [torch.from_numpy((np.arange(n) * (7 + i) % 4096).astype(np.int64)).cuda() for i, n in enumerate(lens)]

    def noise(idx_list, T):
        return torch.stack([torch.from_numpy(init_weights.uniform(f"masked.xT.{i}", (1, 80, T), 3, -1.7, 1.7)) for i in idx_list]).cuda()
    seq = infer_tts.synthesize_ragged(svc, codebook, rows, 1, 250, "dpm-solver", noise_fn=noise)
    for ragged_voc in (True, False):
        msk = infer_tts.synthesize_ragged_masked(svc, codebook, rows, 1, 250, "dpm-solver", noise_fn=noise, streams=2, max_batch=4, ragged_vocoder=ragged_voc)
        torch.cuda.synchronize()
        for (m0, w0), (m1, w1), n in zip(seq, msk, lens):
            assert m1.shape == (n, 80) and w1.shape == (n * 512,)
            assert relmax(m1.cpu().numpy(), m0.cpu().numpy()) < 1e-4 and relmax(w1.cpu().numpy(), w0.cpu().numpy()) < 1e-4


def test_roformer_generate_bench_size_vs_oracle(lm_gpu):
    """Parity where the bench runs (VERDICT r2 #4b): 8 utterances x 64 phones -> 512 SAMPLED tokens (max_length 513: KV length, rotary
    positions and the key-split decode attention over the whole range) token-exact against oracle.roformer.generate, per-step logits 2e-5."""
    from oracle import roformer as R
    m = lm_gpu
    cfg = m.cfg
    w = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    B, L, ML = 8, 64, 513
    idx = np.arange(B * L).reshape(B, L)
    phone, tone = (idx * 7 % 107 + 1).astype(np.int64), (idx * 5 % 12).astype(np.int64)
    spk = np.repeat((np.arange(B) * 37 % 323 + 1).astype(np.int64)[:, None], L, axis=1)
    u = np.random.default_rng(11).random((ML - 1, B)).astype(np.float32)
    enc = m.encode(torch.from_numpy(phone).cuda(), torch.from_numpy(tone).cuda(), torch.from_numpy(spk).cuda())
    toks, logits = m.native().generate(enc, ML, True, 5, 1.0, 1.0, 1.0, dev(u), True)
    ref_enc = R.encoder_forward(w, cfg, phone, tone, spk)
    assert relmax(enc.cpu().numpy(), ref_enc) < 2e-5
    rt, rl = R.generate(w, cfg, enc.cpu().numpy(), ML, True, 5, u)
    assert toks.shape == rt.shape == (B, ML)                     # seeded random weights never emit EOS: the full length
    assert np.array_equal(toks.cpu().numpy(), rt)
    assert relmax(logits.cpu().numpy(), rl) < 2e-5, relmax(logits.cpu().numpy(), rl)


def test_roformer_sampling_controls_vs_oracle(lm_gpu):
    """top_p < 1, temperature != 1 and a repetition penalty (the defaults of Roformer.generate's signature) against the numpy
    restatement of the HF logits processors, B = 3, 1-token encoder edge case included via L = 1"""
    from lds import arch
    from oracle import roformer as R
    m = lm_gpu
    cfg = m.cfg
    w = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    B, L, ML = 3, 1, 20
    phone = torch.tensor([[5], [17], [60]], device="cuda")
    tone = torch.tensor([[1], [0], [7]], device="cuda")
    spk = torch.tensor([[2], [9], [300]], device="cuda")
    enc = m.encode(phone, tone, spk)
    assert relmax(enc.cpu().numpy(), R.encoder_forward(w, cfg, phone.cpu().numpy(), tone.cpu().numpy(), spk.cpu().numpy())) < 2e-5
    rng = np.random.default_rng(3)
    u = rng.random((ML - 1, B)).astype(np.float32)
    toks, _ = m.native().generate(enc, ML, True, 8, 0.7, 1.3, 1.2, dev(u), False)
    # oracle with the same processors
    kv = R.cross_kv(w, cfg, enc.cpu().numpy())
    caches = [dict() for _ in range(cfg["dec_layers"])]
    seq = np.full((B, 1), cfg["sem_bos"], dtype=np.int64)
    for step in range(ML - 1):
        lg = R.decoder_step(w, cfg, seq[:, -1], step, caches, kv)
        nxt = []
        for b in range(B):
            s = lg[b].copy()
            for t in set(seq[b].tolist()):
                s[t] = s[t] * 1.2 if s[t] < 0 else s[t] / 1.2
            s = (s / np.float32(1.3)).astype(np.float32)
            order = np.argsort(-s, kind="stable")[:8]
            p = np.exp(s[order] - s[order][0]).astype(np.float32)
            p = p / p.sum()
            keep = 8
            tail = 0.0
            for j in range(7, 0, -1):
                tail += p[j]
                if tail <= 1.0 - 0.7:
                    keep = j
                else:
                    break
            ids, pp = order[:keep], p[:keep] / p[:keep].sum()
            o = np.argsort(ids)
            c = np.cumsum(pp[o].astype(np.float32), dtype=np.float32)
            j = min(int(np.searchsorted(c, u[step, b], side="right")), keep - 1)
            nxt.append(int(ids[o][j]))
        seq = np.concatenate([seq, np.array(nxt)[:, None]], axis=1)
    assert np.array_equal(toks.cpu().numpy(), seq[:, : toks.shape[1]])


def test_full_tts_path_phones_to_wav(tmp_path):
    """BASELINE config 5's path in-process: phone / tone ids -> RoFormer sampling -> codebook gather -> Unit2Mel -> vocoder"""
    sys.path.insert(0, ROOT)
    import infer_tts
    ph = np.stack([(np.arange(12) * 7 % 107 + 1), (np.arange(12) * 5 % 12)]).astype(np.int64)
    np.save(tmp_path / "phones.npy", ph)
    wav = infer_tts.main(["--synthetic", "--phones", str(tmp_path / "phones.npy"), "--max_length", "17", "-s", "250", "-o", str(tmp_path / "o.npy")])
    assert wav.shape == (16 * 512,) and np.isfinite(wav).all()      # 17 - BOS tokens (random weights never emit EOS)


@pytest.mark.parametrize("top_k,top_p,temp,pen", [(5, 1.0, 1.0, 1.0), (5, 0.8, 1.0, 1.2), (0, 1.0, 1.0, 1.0), (0, 0.8, 1.3, 1.2), (0, 0.5, 0.7, 1.0), (3, 1.0, 1.0, 1.0)])
def test_token_choice_hf_semantics(top_k, top_p, temp, pen):
    """the token choice of lds_lm_generate against HF's logits processors restated in numpy (oracle.roformer.pick_token_hf): top_k = 0 (HF: None / 0)
    applies no top-k filter -- the nucleus cut and the draw run over the whole 4099-entry vocabulary --, and scores tied with the k-th largest
    survive the top-k filter (rows 2 and 3 below carry such ties; reference roformer.py:216-227 forwards whatever the caller passes)"""
    import ctypes as ct
    from lds import native
    from oracle import roformer as R
    rng = np.random.default_rng(17)
    B, V, NH = 6, 4099, 5
    lg = (rng.standard_normal((B, V)) * 2.5).astype(np.float32)
    lg[2, [7, 900, 4000]] = lg[2].max() + 1.0                    # three-way tie at the top: top_k = 3 keeps exactly them, top_k = 5 two more
    k5 = np.sort(lg[3])[-5]
    lg[3, [11, 12, 13]] = k5                                      # ties WITH the 5th largest: HF keeps all of them
    hist = rng.integers(0, V, size=(B, NH)).astype(np.int64)
    hist[1, 1] = hist[1, 3]                                       # a token met twice is penalised once
    for draw in range(4):
        u = rng.random(B).astype(np.float32)
        out = torch.empty(B, dtype=torch.int64, device="cuda")
        dl, du, dh = dev(lg), dev(u), torch.from_numpy(hist).cuda()
        native.check(native.lib().lds_test_lm_sample(ct.c_void_p(dl.data_ptr()), B, V, 1, top_k, ct.c_float(top_p), ct.c_float(temp), ct.c_float(pen),
                                                     ct.c_void_p(du.data_ptr()), ct.c_void_p(dh.data_ptr()), NH, ct.c_void_p(out.data_ptr()),
                                                     ct.c_void_p(torch.cuda.current_stream().cuda_stream)))
        want = [R.pick_token_hf(lg[b], hist[b], top_k, top_p, temp, pen, u[b]) for b in range(B)]
        assert out.cpu().tolist() == want, (draw, out.cpu().tolist(), want)
