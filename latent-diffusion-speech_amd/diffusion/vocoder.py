"""`Vocoder` wrapper with the reference's constructor, attributes and `infer` (reference
diffusion/vocoder.py:5-33).  `extract` (audio -> latent, the VAE encoder + torchaudio resampler)
is preprocessing and out of scope."""
import torch

from encoder.hifi_vaegan.hifi_vaegan import Hifi_VAEGAN


class Vocoder:
    def __init__(self, vocoder_type, vocoder_ckpt, device=None):
        if device is None:
            device = "cuda" if torch.cuda.is_available() else "cpu"
        self.device = device
        self.vocoder_type = vocoder_type
        if vocoder_type == "hifi-vaegan":
            self.vocoder = Hifi_VAEGAN(vocoder_ckpt, device=device)
        else:
            raise ValueError(f" [x] Unknown vocoder: {vocoder_type}")
        self.resample_kernel = {}
        self.vocoder_sample_rate = self.vocoder.sample_rate()
        self.vocoder_hop_size = self.vocoder.hop_size()
        self.dimension = self.vocoder.dimension()

    def extract(self, audio, sample_rate, keyshift=0, **kwargs):
        raise NotImplementedError("Vocoder.extract (audio -> latent encoder) is preprocessing, outside the sampler hot path")

    def infer(self, mel):
        return self.vocoder(mel)

    def infer_ragged(self, mel, lengths):
        """Extension (not in the reference): a padded ragged batch of mels [B,T,M] + per-utterance frame counts -> wav [B,1,T*hop]"""
        return self.vocoder.forward_ragged(mel, lengths)
