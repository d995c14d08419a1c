"""`Unit2Mel`, `load_model_vocoder`, `load_svc_model` with the reference's names, signatures,
attributes and state_dict keys (reference diffusion/unit2mel.py:1-88)."""
import os

import torch
import torch.nn as nn
import yaml

from lds import native
from tools.tools import get_encdoer_out_channels

from .diffusion import GaussianDiffusion
from .unet1d.unet_1d_condition import UNet1DConditionModel
from .vocoder import Vocoder


class DotDict(dict):
    def __getattr__(*args):
        val = dict.get(*args)
        return DotDict(val) if type(val) is dict else val

    __setattr__ = dict.__setitem__
    __delattr__ = dict.__delitem__


def load_model_vocoder(model_path, device="cpu", loaded_vocoder=None):
    config_file = os.path.join(os.path.split(model_path)[0], "config.yaml")
    with open(config_file, "r") as config:
        args = yaml.safe_load(config)
    args = DotDict(args)
    if loaded_vocoder is None:
        vocoder = Vocoder(args["common"]["vocoder"]["type"], args["common"]["vocoder"]["ckpt"], device=device)
    else:
        vocoder = loaded_vocoder
    model = load_svc_model(args=args, vocoder_dimension=vocoder.dimension)
    ckpt = torch.load(model_path, map_location=torch.device(device))
    model.to(device)
    model.load_state_dict(ckpt["model"])
    model.eval()
    return model, vocoder, args


def load_svc_model(args, vocoder_dimension):
    # The reference passes a stray `use_pitch_aug` positional here (unit2mel.py:38-48) and so cannot run
    # (SURVEY.md 3.1); this is the same call with the arguments Unit2Mel.__init__ actually declares.
    return Unit2Mel(
        get_encdoer_out_channels(args["data"]["encoder"]),
        args["common"]["n_spk"],
        vocoder_dimension,
        args["diffusion"]["model"]["n_layers"],
        args["diffusion"]["model"]["block_out_channels"],
        args["diffusion"]["model"]["n_heads"],
        args["diffusion"]["model"]["n_hidden"],
        args["data"]["acoustic_scale"])


class Unit2Mel(nn.Module):
    def __init__(self, input_channel, n_spk, out_dims=128, n_layers=2, block_out_channels=(256, 384, 512, 512), n_heads=8,
                 n_hidden=256, acoustic_scale=1.0):
        super().__init__()
        from lds import init_weights
        self.unit_embed = nn.Linear(input_channel, n_hidden)
        self.aug_shift_embed = None
        self.volume_embed = None
        self.n_spk = n_spk
        if n_spk is not None and n_spk > 1:
            self.spk_embed = nn.Embedding(n_spk, n_hidden)
        with torch.no_grad():   # build-owned seeded init (no pretrained weights exist)
            for k, p in list(self.named_parameters()):
                p.copy_(torch.from_numpy(init_weights.init_tensor(k, tuple(p.shape), 0)))
                p.requires_grad_(False)
        self.decoder = GaussianDiffusion(UNet1DConditionModel(
            in_channels=out_dims + n_hidden,
            out_channels=out_dims,
            block_out_channels=block_out_channels,
            norm_num_groups=8,
            cross_attention_dim=block_out_channels,
            attention_head_dim=n_heads,
            only_cross_attention=True,
            layers_per_block=n_layers,
            resnet_time_scale_shift="scale_shift"), out_dims=out_dims, acoustic_scale=acoustic_scale)
        self._embed = None

    def _apply(self, fn, *a, **k):
        self._embed = None
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):      # reached from load_state_dict of this module or of any parent
        self._embed = None
        return super()._load_from_state_dict(*a, **k)

    def _native_embed(self):
        if self._embed is None:
            spk = self.spk_embed.weight.detach().cpu().numpy() if (self.n_spk is not None and self.n_spk > 1) else None
            self._embed = native.Embed(self.unit_embed.weight.detach().cpu().numpy(), self.unit_embed.bias.detach().cpu().numpy(), spk)
        return self._embed

    def forward(self, units, volume, spk_id=None, aug_shift=None, gt_spec=None, infer=True, infer_speedup=10, method="unipc",
                use_tqdm=False, *, x_T=None):
        """reference unit2mel.py:73 plus the optional x_T of GaussianDiffusion.forward (the start noise, given instead of drawn)"""
        return self._forward(units, volume, spk_id, aug_shift, gt_spec, infer, infer_speedup, method, None, x_T)

    def forward_ragged(self, units, lengths, spk_id=None, infer_speedup=10, method="unipc", x_T=None):
        """Extension (not in the reference): units [B, T, C] padded to the longest utterance + the utterances' own frame counts -> mel
        [B, T, M] with zeros beyond each length; every utterance as if it ran alone (GaussianDiffusion.forward_ragged)."""
        return self._forward(units, None, spk_id, None, None, True, infer_speedup, method, lengths, x_T)

    def _forward(self, units, volume, spk_id, aug_shift, gt_spec, infer, infer_speedup, method, lengths, x_T=None):
        # reference unit2mel.py:74-77: volume_embed is None, so a non-None volume cannot be embedded there either
        if volume is not None:
            raise NotImplementedError("volume_embed is None in the reference (unit2mel.py:55); pass volume=None")
        if self.aug_shift_embed is not None and aug_shift is not None:
            raise NotImplementedError("aug_shift_embed is None in the reference (unit2mel.py:54)")
        if self.n_spk is not None and self.n_spk > 1 and spk_id is None:
            raise TypeError("spk_id is required when n_spk > 1 (reference unit2mel.py:81-82)")
        if self.n_spk is not None and self.n_spk > 1:
            # nn.Embedding(spk_id - 1) raises IndexError for ids outside [1, n_spk] (reference unit2mel.py:82).  Ids that
            # arrive on the host are checked here; ids already on the device are checked by the gather kernel, which writes
            # NaN rows for them (no device->host synchronisation on the hot path).
            if not torch.is_tensor(spk_id):
                spk_id = torch.as_tensor(spk_id, dtype=torch.int64).reshape(units.shape[0], -1)
            if not spk_id.is_cuda:
                if bool(((spk_id < 1) | (spk_id > self.n_spk)).any()):
                    raise IndexError("index out of range in self")
                spk_id = spk_id.to(units.device)
        # x = unit_embed(units) + spk_embed(spk_id - 1), produced channel-major by liblds
        cond = self._native_embed().forward(units.contiguous().float(), spk_id)        # [B,H,T]
        x = native.transpose(cond)                                                     # [B,T,H] as the reference hands over
        if lengths is not None:
            return self.decoder.forward_ragged(x, lengths, gt_spec=gt_spec, infer_speedup=infer_speedup, method=method, x_T=x_T)
        return self.decoder(x, gt_spec=gt_spec, infer=infer, infer_speedup=infer_speedup, method=method, use_tqdm=False, x_T=x_T)
