"""Harness counterpart of the reference's 22_infer_tts.py (22_infer_tts.py:27-114) from the phone / tone ids on:
[phones, tones -> text2semantic RoFormer generate ->] semantic tokens -> unit embeddings (k-means codebook row gather in liblds)
[-> forced alignment] -> DiffusionSVC.infer (Unit2Mel sampler + HiFi-VAEGAN vocoder) -> 44.1 kHz wav.
The grapheme-to-phoneme front end (text/cleaner.py: pypinyin / jieba / g2p tables) is outside this build's scope.

    python infer_tts.py -dm exp/diffusion/model_300000.pt -cb pretrain/semantic_codebook.pt -t tokens.npy -o out.wav
    python infer_tts.py --synthetic -o /tmp/demo.npy        # seeded random weights, synthetic tokens (no checkpoints exist)
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "latent-diffusion-speech_amd")
if PKG not in sys.path:
    sys.path.insert(0, PKG)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("-dm", "--diffusion_model")
    ap.add_argument("-cb", "--codebook", help="semantic_codebook.pt: the reference's KMeans dict (cluster/__init__.py:5-11) or a plain [n_codes, dim] tensor")
    ap.add_argument("-t", "--tokens", help=".npy int array [T] of semantic token ids")
    ap.add_argument("-lm", "--language_model", help="exp/lm/model_<step>.pt ({'model': Roformer.state_dict()}) with config.yaml next to it")
    ap.add_argument("-p", "--phones", help=".npy int array [2, L]: phone ids and tone ids (text_to_sequence output); tokens come from the LM")
    ap.add_argument("--max_length", type=int, default=1024)
    ap.add_argument("-o", "--output", default="output.npy")
    ap.add_argument("-id", "--spk_id", type=int, default=1)
    ap.add_argument("-s", "--speedup", type=int, default=10)
    ap.add_argument("-me", "--method", default="dpm-solver")
    ap.add_argument("--scale_factor", type=float, default=None,
                    help="nearest-resample the unit frames by this factor first (22_infer_tts.py:108-110, units_forced_alignment)")
    ap.add_argument("--latency_mode", action="store_true",
                    help="one sentence per call: tile shapes / reduction splits from the actual batch (include/lds.h lds_unet_set_latency_mode)")
    ap.add_argument("--gemm_mode", default="f32", choices=["f32", "split_f16"], help="the UNet's GEMMs (include/lds.h lds_unet_set_gemm_mode)")
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--synthetic_tokens", type=int, default=256)
    return ap.parse_args(argv)


def synthetic_pipeline(dev, n_tokens):
    """seeded random-init model, vocoder, codebook and tokens (no checkpoints ship with the reference)"""
    from diffusion.unit2mel import Unit2Mel
    from diffusion.vocoder import Vocoder
    from encoder.hifi_vaegan.hifi_vaegan import Hifi_VAEGAN
    from lds import arch, init_weights
    from tools.infer_tools import DiffusionSVC
    h = arch.SYNTHETIC_VOCODER_H
    voc = Vocoder.__new__(Vocoder)
    voc.vocoder = Hifi_VAEGAN(None, device=dev, h=h, state=init_weights.init_state(arch.generator_param_shapes(h), 0))
    voc.vocoder_hop_size, voc.vocoder_sample_rate, voc.dimension, voc.device = h["hop_size"], h["sampling_rate"], h["inter_channels"], dev
    svc = DiffusionSVC(device=dev)
    svc.model, svc.vocoder = Unit2Mel(1280, 323, 80).to(dev).eval(), voc
    codebook = torch.from_numpy(init_weights.uniform("synthetic.codebook", (4096, 1280), 5, -1.7, 1.7)).to(dev)
    tokens = torch.from_numpy((np.arange(n_tokens) * 131 % 4096).astype(np.int64)).to(dev)
    return svc, codebook, tokens


def load_codebook(path, dev):
    from cluster import codebook_to_device, get_cluster_model
    try:
        return codebook_to_device(get_cluster_model(path), dev)              # the reference's on-disk format
    except (KeyError, TypeError, IndexError):
        return torch.load(path, map_location=dev).float().contiguous()     # a bare centre matrix


def synthetic_lm(dev):
    """seeded random-init text2semantic model in the reference's phone-mode configuration"""
    from lds import arch
    from text2semantic.roformer.roformer import Roformer
    c = arch.roformer_config()
    hf = dict(hidden_size=c["hidden"], num_attention_heads=c["heads"], intermediate_size=c["inter"], hidden_act="gelu",
              max_position_embeddings=c["max_pos"], layer_norm_eps=c["eps"])
    return Roformer(dict(hf, num_hidden_layers=c["enc_layers"]), dict(hf, num_hidden_layers=c["dec_layers"]), mode="phone",
                    semantic_kmeans_num=c["semantic_kmeans_num"], codebook_path="", n_spk=c["n_spk"]).to(dev).eval()


def load_lm(path, dev):
    import yaml
    from text2semantic.utils import get_language_model
    args = yaml.safe_load(open(os.path.join(os.path.split(path)[0], "config.yaml")))
    lm = get_language_model(**args).to(dev)
    lm.load_state_dict(torch.load(path, map_location=torch.device(dev))["model"])
    return lm.eval()


def text2semantic(lm, phones, tones, spk_id=1, max_length=1024):
    """22_infer_tts.py:76-104: sample the semantic tokens (top-k 5, temperature 1) and strip BOS / EOS.  phones, tones [B,L] int64."""
    spk = torch.ones_like(phones) * spk_id
    tok = lm.generate(phones, tones, attention_mask=None, use_cache=None, max_length=max_length, do_sample=True, temperature=1.0, top_k=5, top_p=1.0,
                      repetition_penalty=1.0, num_beams=1, no_repeat_ngram_size=0, early_stopping=True, spk_id=spk, end_gate_threshold=None)
    # reference 22_infer_tts.py:100-103 (`if semantic_token[:, -1] == eos`: written for one utterance; on a batch that line itself raises).
    # The batched counterpart: every row ending with its EOS at the same step is stripped like the single row; rows that all ran to max_length
    # come back whole; rows that ended at DIFFERENT steps (EOS / PAD ids inside the result) cannot be one rectangular tensor.
    eos = lm.semantic_eos_token_id
    body_has_stop = bool((tok[:, 1:-1] >= eos).any()) if tok.shape[1] > 2 else False
    if not body_has_stop and bool((tok[:, -1] == eos).all()):
        return tok[:, 1:-1]
    if body_has_stop or bool((tok[:, -1] >= eos).any()):
        raise ValueError("rows of a batch ended at different lengths (EOS / PAD ids in the result): use text2semantic_rows, which cuts every "
                         "row at its own EOS, and synthesize_ragged")
    return tok[:, 1:]


def text2semantic_rows(lm, phones, tones, spk_id=1, max_length=1024, phone_lengths=None):
    """A batch of sentences of DIFFERENT lengths (BASELINE configs[4]: 64 sentences per call).  phones / tones [B,L] right-padded,
    phone_lengths [B] (None = all L): the padding mask goes through the encoder and the cross-attention (reference roformer.py:209-236).
    Returns one 1-D token tensor per row: BOS stripped, cut before the row's first EOS (rows that finish early are padded by generate;
    the ids EOS = kmeans_num + 1 and PAD = kmeans_num + 2 have no codebook row and must never reach the unit lookup)."""
    B, L = phones.shape
    mask = None
    if phone_lengths is not None:
        mask = (torch.arange(L, device=phones.device)[None] < torch.as_tensor(phone_lengths, device=phones.device)[:, None]).to(torch.int64)
    spk = torch.ones_like(phones) * spk_id
    tok = lm.generate(phones, tones, attention_mask=mask, use_cache=None, max_length=max_length, do_sample=True, temperature=1.0, top_k=5, top_p=1.0,
                      repetition_penalty=1.0, num_beams=1, no_repeat_ngram_size=0, early_stopping=True, spk_id=spk, end_gate_threshold=None)
    tok = tok[:, 1:].cpu()
    rows = []
    for b in range(B):
        stop = (tok[b] >= lm.semantic_eos_token_id).nonzero()
        n = int(stop[0]) if stop.numel() else tok.shape[1]
        rows.append(tok[b, :n].to(phones.device))
    return rows


_STREAM_POOLS = {}


def _stream_pool(dev, n):
    """n side streams per device, created once: every (handle, stream) pair owns a workspace (lds/native.py Workspace), so the streams
    are reused across calls instead of drawn anew"""
    key = str(torch.device(dev))
    pool = _STREAM_POOLS.setdefault(key, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(dev))
    return pool[:n]


def synthesize_ragged_masked(svc, codebook, token_rows, spk_id=1, speedup=10, method="dpm-solver", noise_fn=None, streams=1, max_batch=16,
                             ragged_vocoder=True):
    """Utterances of different lengths with the SAMPLER run as one padded batch per `max_batch` rows and per-utterance lengths inside the
    kernels (Unit2Mel.forward_ragged: every utterance as if it ran alone -- statistics, attention and resampling stop at its own length;
    within the parity tolerances of the stand-alone run, not bit for bit), then the vocoder on the same padded batch (ragged_vocoder, the
    default: lds_vocoder_forward_ragged) or per length bucket on `streams` HIP streams.
    Rows are sorted by length so that a padded batch wastes few frames.  Returns [(mel [T,M], wav [T*hop])] in the order of `token_rows`."""
    from lds import native
    order = sorted((i for i, r in enumerate(token_rows) if r.numel() > 0), key=lambda i: -int(token_rows[i].numel()))
    mels = [None] * len(token_rows)
    wavs = [None] * len(token_rows)
    for c0 in range(0, len(order), max_batch):
        idx = order[c0:c0 + max_batch]
        lens = [int(token_rows[i].numel()) for i in idx]
        T = max(lens)
        tok = torch.zeros(len(idx), T, dtype=torch.int64, device=codebook.device)
        for j, i in enumerate(idx):
            tok[j, :lens[j]] = token_rows[i]
        units = native.gather_rows(codebook, tok)
        xT = None
        if noise_fn is not None:      # the start noise handed down explicitly (GaussianDiffusion.forward's x_T), every row at its own length
            xT = torch.zeros(len(idx), 1, svc.model.decoder.out_dims, T, device=codebook.device)
            for j, i in enumerate(idx):
                xT[j:j + 1, :, :, :lens[j]] = noise_fn([i], lens[j])
        mel = svc.call_ragged(units, lens, spk_id=spk_id, infer_speedup=speedup, method=method, x_T=xT)
        if ragged_vocoder:      # ... and the vocoder on the same padded batch, every stage masked at the utterance's up-sampled length
            wav = svc.vocoder.infer_ragged(mel, lens)
            hop = wav.shape[-1] // T
            for j, i in enumerate(idx):
                wavs[i] = wav[j, 0, :lens[j] * hop].contiguous()
        for j, i in enumerate(idx):
            mels[i] = mel[j, :lens[j]].contiguous()
    if ragged_vocoder:
        return [(mels[i], wavs[i]) if token_rows[i].numel() else (torch.empty(0, svc.model.decoder.out_dims), torch.empty(0)) for i in range(len(token_rows))]
    # ragged_vocoder = False: the vocoder per length bucket (bit-identical with each mel decoded alone)
    out = [None] * len(token_rows)
    by_len = {}
    for i, r in enumerate(token_rows):
        by_len.setdefault(int(r.numel()), []).append(i)

    def voc_bucket(T, idx):
        if T == 0:
            for i in idx:
                out[i] = (torch.empty(0, svc.model.decoder.out_dims), torch.empty(0))
            return
        wav = svc.mel2wav(torch.stack([mels[i] for i in idx]), None)
        for j, i in enumerate(idx):
            out[i] = (mels[i], wav[j, 0])
    buckets = sorted(by_len.items())
    if streams <= 1 or len(buckets) <= 1:
        for T, idx in buckets:
            voc_bucket(T, idx)
        return out
    dev = codebook.device
    cur = torch.cuda.current_stream(dev)
    pool = _stream_pool(dev, min(streams, len(buckets)))
    for k, s_ in enumerate(pool):
        s_.wait_stream(cur)
        with torch.cuda.stream(s_):
            for T, idx in buckets[k::len(pool)]:
                voc_bucket(T, idx)
    for s_ in pool:
        cur.wait_stream(s_)
    for mel_wav in out:
        for t in mel_wav:
            if t.is_cuda:
                t.record_stream(cur)
    return out


def synthesize_ragged(svc, codebook, token_rows, spk_id=1, speedup=10, method="dpm-solver", noise_fn=None, streams=1):
    """Utterances of different lengths through the sampler and the vocoder: rows of equal length run as one batch, and because no kernel
    reduces across the batch axis (batch invariance) every utterance's mel / waveform is bit-identical to running it alone.
    token_rows: list of 1-D int64 tensors.  noise_fn(n_utt, T) -> x_T [n_utt,1,M,T] injects the start noise (tests); returns a list of
    (mel [T,M], wav [T*hop]) in the order of `token_rows`.
    streams > 1: the length buckets are spread over that many HIP streams (each with its own workspace, lds/native.py Workspace) and, unless
    noise_fn is given, over as many host threads -- ctypes releases the GIL while a sampler call enqueues its ~12 k launches.  A bucket of
    one or two utterances leaves most of the chip idle, so buckets overlap on the GPU; the results are the same tensors either way."""
    from lds import native
    out = [None] * len(token_rows)
    by_len = {}
    for i, r in enumerate(token_rows):
        by_len.setdefault(int(r.numel()), []).append(i)

    def run_bucket(T, idx):
        if T == 0:
            for i in idx:
                out[i] = (torch.empty(0, svc.model.decoder.out_dims), torch.empty(0))
            return
        tok = torch.stack([token_rows[i] for i in idx])
        units = native.gather_rows(codebook, tok)
        xT = noise_fn(idx, T) if noise_fn is not None else None
        mel = svc(units, f0=None, volume=None, spk_id=spk_id, infer_speedup=speedup, method=method, x_T=xT)
        wav = svc.mel2wav(mel, None)
        for j, i in enumerate(idx):
            out[i] = (mel[j], wav[j, 0])

    buckets = sorted(by_len.items(), key=lambda kv: -kv[0] * len(kv[1]))      # largest first
    if streams <= 1 or len(buckets) <= 1:
        for T, idx in sorted(by_len.items()):
            run_bucket(T, idx)
        return out
    dev = codebook.device
    cur = torch.cuda.current_stream(dev)
    pool = _stream_pool(dev, min(streams, len(buckets)))
    for s_ in pool:
        s_.wait_stream(cur)      # the inputs were produced on the caller's stream

    with torch.cuda.stream(pool[0]):      # the first bucket on this thread: it creates whatever native handle does not exist yet
        run_bucket(*buckets[0])
    buckets = buckets[1:]

    def worker(k):
        with torch.cuda.stream(pool[k]):
            for T, idx in buckets[k::len(pool)]:
                run_bucket(T, idx)
    import threading
    ths = [threading.Thread(target=worker, args=(k,)) for k in range(len(pool))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for s_ in pool:
        cur.wait_stream(s_)
    for mel_wav in out:      # the caller's stream owns the results from here on
        for t in mel_wav:
            if t.is_cuda:
                t.record_stream(cur)
    return out


def synthesize(svc, codebook, tokens, spk_id=1, speedup=10, method="dpm-solver", scale_factor=None):
    """tokens [T] (or [B,T]) int64 on the device -> (units [B,T',C], mel [B,T',M], wav [B,1,T'*hop])"""
    from lds import native
    from tools.tools import units_forced_alignment
    tok = tokens if tokens.dim() == 2 else tokens[None]
    units = native.gather_rows(codebook, tok)                                  # semantic_embedding(semantic_token), 22_infer_tts.py:106
    if scale_factor is not None:
        units = units_forced_alignment(units, scale_factor=scale_factor)
    mel = svc(units, f0=None, volume=None, spk_id=spk_id, infer_speedup=speedup, method=method)
    wav = svc.mel2wav(mel, None)
    return units, mel, wav


def main(argv=None):
    a = parse_args(argv)
    dev = "cuda"
    lm = None
    if a.synthetic:
        svc, codebook, tokens = synthetic_pipeline(dev, a.synthetic_tokens)
        if a.phones:
            lm = synthetic_lm(dev)
    else:
        from tools.infer_tools import DiffusionSVC
        svc = DiffusionSVC(device=dev)
        svc.load_model(a.diffusion_model)
        codebook = load_codebook(a.codebook, dev)
        tokens = torch.from_numpy(np.load(a.tokens).astype(np.int64)).to(dev) if a.tokens else None
        if a.language_model:
            lm = load_lm(a.language_model, dev)
    svc.model.decoder.denoise_fn.set_gemm_mode(a.gemm_mode)
    svc.model.decoder.denoise_fn.set_latency_mode(a.latency_mode)
    if a.phones:
        if lm is None:
            raise SystemExit("--phones needs --language_model (or --synthetic)")
        pt = torch.from_numpy(np.load(a.phones).astype(np.int64)).to(dev)
        tokens = text2semantic(lm, pt[0:1], pt[1:2], a.spk_id, a.max_length)
        tokens = tokens.clamp(max=codebook.shape[0] - 1)      # pad ids of finished rows (batch > 1) have no codebook row
    _, _, wav = synthesize(svc, codebook, tokens, a.spk_id, a.speedup, a.method, a.scale_factor)
    wav = wav[0, 0].cpu().numpy()
    if a.output.endswith(".wav"):
        import wave
        with wave.open(a.output, "wb") as f:
            f.setnchannels(1); f.setsampwidth(2); f.setframerate(44100)
            f.writeframes((np.clip(wav, -1, 1) * 32767).astype("<i2").tobytes())
    else:
        np.save(a.output, wav)
    print(f"wrote {a.output}: {wav.shape[0]} samples ({wav.shape[0] / 44100:.2f} s)")
    return wav


if __name__ == "__main__":
    main()
