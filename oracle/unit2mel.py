"""Unit2Mel front end and the end-to-end unit -> mel -> wav pipeline (reference
diffusion/unit2mel.py:73-88, diffusion/diffusion.py:189-343, diffusion/vocoder.py:32-33,
encoder/hifi_vaegan/hifi_vaegan.py:52-65).  The reference's own wrappers cannot be imported in
the build container (absent torchaudio/librosa), so these few affine lines are
restated from the text and pinned only through the pieces they call."""
import numpy as np

from . import solvers, unet1d, vocoder

f32 = np.float32


def condition(w, units, spk_id):
    """x = unit_embed(units) + spk_embed(spk_id - 1)   (volume/aug_shift embeds are None)."""
    x = unet1d.linear(units.astype(f32), w["unit_embed.weight"], w["unit_embed.bias"])
    if "spk_embed.weight" in w:
        x = x + w["spk_embed.weight"][np.asarray(spk_id).reshape(units.shape[0], -1)[:, :1] - 1]
    return x.astype(f32)  # [B,T,H]


def make_eps_fn(w_unet, cfg, blocks, cond_bht):
    def eps_fn(x, t):
        return unet1d.unet_forward(w_unet, cfg, blocks, np.concatenate([x, cond_bht], axis=1), t)
    return eps_fn


def unit2mel(w, cfg, blocks, bufs, units, spk_id, x_T, method, infer_speedup, k_step=1000, noise=None,
             acoustic_scale=1.0):
    """Unit2Mel.forward(infer=True) with x_T (and DDPM noise) injected. Returns mel [B,T,M]."""
    cond = condition(w, units, spk_id).transpose(0, 2, 1)
    wu = {k[len("decoder.denoise_fn."):]: v for k, v in w.items() if k.startswith("decoder.denoise_fn.")}
    x = solvers.sample(make_eps_fn(wu, cfg, blocks, np.ascontiguousarray(cond)), bufs, x_T, method,
                       infer_speedup, k_step, noise)
    return (x.transpose(0, 2, 1) / f32(acoustic_scale)).astype(f32)


def vocoder_infer(w_folded, h, mel):
    """Vocoder.infer(mel [B,T,C]) -> wav [B,1,T*hop]."""
    return vocoder.generator_forward(w_folded, h, np.ascontiguousarray(mel.transpose(0, 2, 1)))
