"""Harness counterpart of the reference's 22_infer_tts.py from the point where semantic tokens exist
(22_infer_tts.py:100-114): tokens -> unit embeddings (k-means codebook gather) -> DiffusionSVC.infer -> 44.1 kHz wav.
The text front end and the RoFormer LM that produce the tokens are outside this build's scope.

    python infer_tts.py -dm exp/diffusion/model_300000.pt -cb pretrain/semantic_codebook.pt -t tokens.npy -o out.wav
    python infer_tts.py --synthetic -o /tmp/demo.npy        # seeded random weights, synthetic tokens (no checkpoints exist)
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion-speech_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-dm", "--diffusion_model")
    ap.add_argument("-cb", "--codebook", help="torch file holding KMeans cluster centers [n_codes, 1280] (22_infer_tts.py:43-52)")
    ap.add_argument("-t", "--tokens", help=".npy int array [T] of semantic token ids")
    ap.add_argument("-o", "--output", default="output.npy")
    ap.add_argument("-id", "--spk_id", type=int, default=1)
    ap.add_argument("-s", "--speedup", type=int, default=10)
    ap.add_argument("-me", "--method", default="dpm-solver")
    ap.add_argument("--synthetic", action="store_true")
    a = ap.parse_args()
    dev = "cuda"
    from tools.infer_tools import DiffusionSVC
    svc = DiffusionSVC(device=dev)
    if a.synthetic:
        from diffusion.unit2mel import Unit2Mel
        from diffusion.vocoder import Vocoder
        from encoder.hifi_vaegan.hifi_vaegan import Hifi_VAEGAN
        from lds import arch, init_weights
        h = arch.SYNTHETIC_VOCODER_H
        voc = Vocoder.__new__(Vocoder)
        voc.vocoder = Hifi_VAEGAN(None, device=dev, h=h, state=init_weights.init_state(arch.generator_param_shapes(h), 0))
        voc.vocoder_hop_size, voc.vocoder_sample_rate, voc.dimension, voc.device = h["hop_size"], h["sampling_rate"], h["inter_channels"], dev
        svc.model, svc.vocoder = Unit2Mel(1280, 323, 80).to(dev).eval(), voc
        codebook = torch.from_numpy(init_weights.uniform("synthetic.codebook", (4096, 1280), 5, -1.7, 1.7)).to(dev)
        tokens = torch.from_numpy((np.arange(256) * 131 % 4096).astype(np.int64)).to(dev)
    else:
        svc.load_model(a.diffusion_model)
        codebook = torch.load(a.codebook, map_location=dev).float()
        tokens = torch.from_numpy(np.load(a.tokens).astype(np.int64)).to(dev)
    units = torch.nn.functional.embedding(tokens[None], codebook)          # [1, T, 1280]  (22_infer_tts.py:106)
    wav = svc.infer(units, f0=None, volume=None, spk_id=a.spk_id, infer_speedup=a.speedup, method=a.method)
    wav = wav[0, 0].cpu().numpy()
    if a.output.endswith(".wav"):
        import wave
        with wave.open(a.output, "wb") as f:
            f.setnchannels(1); f.setsampwidth(2); f.setframerate(44100)
            f.writeframes((np.clip(wav, -1, 1) * 32767).astype("<i2").tobytes())
    else:
        np.save(a.output, wav)
    print(f"wrote {a.output}: {wav.shape[0]} samples ({wav.shape[0] / 44100:.2f} s)")


if __name__ == "__main__":
    main()
