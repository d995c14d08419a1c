"""`Hifi_VAEGAN` decoder wrapper (reference encoder/hifi_vaegan/hifi_vaegan.py:10-65): reads
`<model_path>/decoder.pth` = {'config': h, 'model': state_dict with weight-norm pairs}, lazily
builds the native Generator on first call and maps z [B,T,C] -> wav [B,1,T*hop]."""
import os

import torch

from lds import native


def load_config(model_path):
    h = torch.load(os.path.join(model_path, "decoder.pth"), map_location="cpu", weights_only=False)["config"]
    return h


class Hifi_VAEGAN(torch.nn.Module):
    def __init__(self, model_path, device=None, h=None, state=None):
        """`h`/`state` (optional, not in the reference) inject a config + Generator state_dict directly,
        for synthetic-weight runs where no decoder.pth exists."""
        super().__init__()
        if device is None:
            device = "cuda" if torch.cuda.is_available() else "cpu"
        self.device = device
        self.model_path = model_path
        self.encoder_model = None
        self.decoder_model = None
        self._state = state
        self.h = h if h is not None else load_config(model_path)

    def sample_rate(self):
        return self.h["sampling_rate"]

    def hop_size(self):
        return self.h["hop_size"]

    def dimension(self):
        return self.h["inter_channels"]

    def extract(self, audio, only_z=False, only_mean=False):
        raise NotImplementedError("the VAE encoder is preprocessing, outside the sampler hot path")

    @torch.no_grad()
    def forward_ragged(self, z, lengths):
        """Extension (not in the reference): z [B,T,C] padded to the longest utterance + per-utterance frame counts -> wav [B,1,T*hop] with
        zeros beyond each utterance's samples; every utterance as if it were decoded alone (include/lds.h lds_vocoder_forward_ragged)"""
        return self._decode(z, lengths)

    @torch.no_grad()
    def forward(self, z):
        return self._decode(z, None)

    def _decode(self, z, _lengths):
        if not z.is_cuda:
            raise RuntimeError("Hifi_VAEGAN.forward needs tensors on a HIP device (no CPU fallback)")
        if self.decoder_model is None:
            state = self._state
            if state is None:
                print("| Load Vaegan:", self.model_path)
                state = torch.load(os.path.join(self.model_path, "decoder.pth"), map_location="cpu", weights_only=False)["model"]
            self.decoder_model = native.Generator(self.h, state)     # folds weight norm like remove_weight_norm()
            self._state = None
        zt = native.transpose(z.contiguous().float())                # z.transpose(-1,-2) -> [B,C,T]
        return self.decoder_model.forward(zt, _lengths)
