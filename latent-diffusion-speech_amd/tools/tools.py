"""The pieces of reference tools/tools.py that sit on the TTS inference path: the encoder-width table and
`units_forced_alignment` (the speech encoders, volume extractor and schedulers there are preprocessing / training and out
of scope)."""
import math

import numpy as np
import torch

from lds import native
from lds.arch import get_encoder_out_channels


def get_encdoer_out_channels(encoder):  # sic: the reference's spelling (tools/tools.py:257-264)
    return get_encoder_out_channels(encoder)


def units_forced_alignment(units, audio=None, sample_rate=None, hop_size=None, n_frames=None, scale_factor=None,
                           units_forced_mode="nearest", device="cpu"):
    """Resample unit frames [B,T,C] (or [T,C]) along time (reference tools/tools.py:193-223).  'nearest' (and the two 'rfa*'
    aliases) = F.interpolate(mode='nearest'): out[i] = units[min(floor(i * s), T-1)] with s = 1/scale_factor (or T/n_frames
    when the size is given) in fp32; 'left' = units[min(round(scale_factor * i), T-1)].  The frame gather runs in liblds, so
    the units must live on a HIP device (22_infer_tts.py:108-110 passes the device tensor).  Other interpolate modes are not
    used by the TTS path."""
    assert (audio is not None and sample_rate is not None and hop_size is not None) or n_frames is not None or scale_factor is not None
    n_frames = int(audio.size(-1) // hop_size + 1) if (n_frames is None and audio is not None) else n_frames
    if isinstance(units, np.ndarray) or not units.is_cuda:
        raise RuntimeError("units_forced_alignment needs the units on a HIP device (no CPU fallback for the hot path)")
    squeeze = units.dim() == 2
    u = (units.unsqueeze(0) if squeeze else units).contiguous().float()
    T = u.shape[1]
    if units_forced_mode == "left":
        assert scale_factor is not None and n_frames is not None
        idx = torch.clamp(torch.round(scale_factor * torch.arange(n_frames, device=u.device)).long(), max=T - 1)
        out = native.gather_rows(u.reshape(-1, u.shape[-1]), (idx[None, :] + T * torch.arange(u.shape[0], device=u.device)[:, None]))
    elif units_forced_mode in ("nearest", "rfa441to512", "rfa512to441"):
        if n_frames is not None and scale_factor is not None:
            raise ValueError("only one of size or scale_factor should be defined")      # F.interpolate's own check
        if n_frames is not None:
            n_out, step = int(n_frames), np.float32(T) / np.float32(n_frames)
        else:
            n_out, step = int(math.floor(float(T) * float(scale_factor))), np.float32(1.0 / float(scale_factor))
        out = native.resample_frames(u, n_out, float(step))
    else:
        raise NotImplementedError(f"units_forced_mode {units_forced_mode!r} is not used on the TTS path")
    return out.squeeze(0) if squeeze else out
