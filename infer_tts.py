"""Harness counterpart of the reference's 22_infer_tts.py from the point where semantic tokens exist
(22_infer_tts.py:42-52,100-114): tokens -> unit embeddings (k-means codebook row gather in liblds) [-> forced alignment]
-> DiffusionSVC.infer (Unit2Mel sampler + HiFi-VAEGAN vocoder) -> 44.1 kHz wav.
The text front end and the RoFormer LM that produce the tokens are outside this build's scope.

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
    ap.add_argument("-o", "--output", default="output.npy")
    ap.add_argument("-id", "--spk_id", type=int, default=1)
    ap.add_argument("-s", "--speedup", type=int, default=10)
    ap.add_argument("-me", "--method", default="dpm-solver")
    ap.add_argument("--scale_factor", type=float, default=None,
                    help="nearest-resample the unit frames by this factor first (22_infer_tts.py:108-110, units_forced_alignment)")
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
    if a.synthetic:
        svc, codebook, tokens = synthetic_pipeline(dev, a.synthetic_tokens)
    else:
        from tools.infer_tools import DiffusionSVC
        svc = DiffusionSVC(device=dev)
        svc.load_model(a.diffusion_model)
        codebook = load_codebook(a.codebook, dev)
        tokens = torch.from_numpy(np.load(a.tokens).astype(np.int64)).to(dev)
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
