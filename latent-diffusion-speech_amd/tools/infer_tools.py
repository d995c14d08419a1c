"""`DiffusionSVC` inference facade for the TTS path: the reference's tools/infer_tools.py:9-81 with call signatures that
are consistent with `Unit2Mel.forward` / `Vocoder.infer` (the reference's own versions raise TypeError before any compute,
SURVEY.md 3.1).  Only what 22_infer_tts.py uses is kept: load_model, __call__, infer, mel2wav.  The speech encoders,
volume extractor and long-audio slicing are preprocessing / SVC features outside the sampler hot path."""
import numpy as np
import torch

from diffusion.unit2mel import load_model_vocoder


class DiffusionSVC:
    def __init__(self, device=None):
        self.device = device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu")
        self.model_path = None
        self.model = None
        self.vocoder = None
        self.args = None
        self.units_encoder = None
        self.volume_extractor = None

    def load_model(self, model_path, loaded_vocoder=None, **_ignored):
        """reference infer_tools.py:28-30 (22_infer_tts.py passes extra f0_min/f0_max keywords that the reference's own
        method does not accept; they are accepted and ignored here)"""
        self.model_path = model_path
        self.model, self.vocoder, self.args = load_model_vocoder(model_path, device=self.device, loaded_vocoder=loaded_vocoder)

    def encode_units(self, audio, sr=44100, padding_mask=None):
        raise NotImplementedError("speech->units encoders are preprocessing, outside the sampler hot path")

    @torch.no_grad()
    def mel2wav(self, mel, f0=None, start_frame=0):
        """reference infer_tools.py:60-67; the vocoder takes the mel only (reference vocoder.py:32)"""
        if start_frame == 0:
            return self.vocoder.infer(mel)
        out_wav = self.vocoder.infer(mel[:, start_frame:, :].contiguous())
        return torch.nn.functional.pad(out_wav, (start_frame * self.vocoder.vocoder_hop_size, 0))

    @torch.no_grad()
    def __call__(self, units, f0=None, volume=None, spk_id=1, aug_shift=0, gt_spec=None, infer_speedup=10, method="unipc", use_tqdm=True, x_T=None):
        """reference infer_tools.py:70-74: units [B,T,C] -> mel [B,T,M]; f0 is unused by the TTS model and must be None"""
        if f0 is not None:
            raise NotImplementedError("the TTS Unit2Mel has no f0 input (22_infer_tts.py passes f0=None)")
        B = units.shape[0]
        if torch.is_tensor(spk_id):
            sid = spk_id.to(self.device).long().reshape(B, -1)
        else:
            sid = torch.LongTensor(np.full((B, 1), int(spk_id))).to(self.device)
        return self.model(units.to(self.device), volume, spk_id=sid, aug_shift=None, gt_spec=gt_spec, infer=True,
                          infer_speedup=infer_speedup, method=method, use_tqdm=use_tqdm, x_T=x_T)

    @torch.no_grad()
    def call_ragged(self, units, lengths, spk_id=1, infer_speedup=10, method="unipc", x_T=None):
        """Extension: a padded ragged batch of units [B,T,C] + per-utterance frame counts -> mel [B,T,M] (Unit2Mel.forward_ragged)"""
        B = units.shape[0]
        sid = spk_id.to(self.device).long().reshape(B, -1) if torch.is_tensor(spk_id) else torch.LongTensor(np.full((B, 1), int(spk_id))).to(self.device)
        return self.model.forward_ragged(units.to(self.device), lengths, spk_id=sid, infer_speedup=infer_speedup, method=method, x_T=x_T)

    @torch.no_grad()
    def infer(self, units, f0=None, volume=None, gt_spec=None, spk_id=1, aug_shift=0, infer_speedup=10, method="unipc", use_tqdm=True, x_T=None):
        """reference infer_tools.py:77-81: units -> waveform [B,1,T*hop]"""
        out_mel = self.__call__(units, f0, volume, spk_id=spk_id, aug_shift=aug_shift, gt_spec=None, infer_speedup=infer_speedup,
                                method=method, use_tqdm=use_tqdm, x_T=x_T)
        return self.mel2wav(out_mel, f0)
