"""`UNet1DConditionModel` with the reference's constructor keywords, state_dict keys and
`.forward(sample, timestep).sample` contract (reference diffusion/unet1d/unet_1d_condition.py:61,
743-1036), executed by liblds (hand-written gfx950 kernels).  Only the configuration Unit2Mel
uses is supported (reference diffusion/unit2mel.py:62-71); anything else raises."""
from dataclasses import dataclass

import torch
from torch import nn

from lds import arch, native
from lds.paramtree import ParamTree


@dataclass
class UNet1DConditionOutput:
    """reference unet_1d_condition.py:48-58 (a BaseOutput with a `.sample` field)"""
    sample: torch.FloatTensor = None


class UNet1DConditionModel(ParamTree):
    def __init__(self, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), norm_num_groups=32,
                 cross_attention_dim=1280, attention_head_dim=8, only_cross_attention=False, layers_per_block=2,
                 resnet_time_scale_shift="default", cond_channels=None, **unsupported):
        if unsupported:
            raise NotImplementedError(f"UNet1DConditionModel options not on the hot path: {sorted(unsupported)}")
        if resnet_time_scale_shift != "scale_shift" or not only_cross_attention:
            raise NotImplementedError("only the Unit2Mel configuration (scale_shift, only_cross_attention) is built")
        boc = tuple(int(c) for c in block_out_channels)
        if tuple(cross_attention_dim) != boc:
            raise NotImplementedError("cross_attention_dim must equal block_out_channels (self-attention, SURVEY F7)")
        if cond_channels is None:
            cond_channels = in_channels - out_channels
        cfg = arch.unet_config(out_dims=out_channels, n_hidden=cond_channels, block_out_channels=boc,
                               n_layers=layers_per_block, n_heads=attention_head_dim, norm_groups=norm_num_groups)
        super().__init__(arch.unet_param_shapes(cfg), seed=0)
        self.cfg = cfg
        self.in_channels, self.out_channels = in_channels, out_channels
        self._native = None
        self._gemm_mode = "f32"
        self._latency_mode = False

    # Any change of the parameters drops the packed copy: .to()/.float() go through _apply; load_state_dict -- called on
    # this module OR on any parent (Unit2Mel / GaussianDiffusion: nn.Module.load_state_dict recurses through the children's
    # _load_from_state_dict, never their load_state_dict) -- goes through _load_from_state_dict of this root node.
    def _apply(self, fn, *a, **k):
        self._native = None
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):
        self._native = None
        return super()._load_from_state_dict(*a, **k)

    def native(self):
        if self._native is None:
            self._native = native.UNet(self.cfg, {k: v for k, v in self.state_dict().items()})
            if self._gemm_mode != "f32":
                self._native.set_gemm_mode(self._gemm_mode)
            if self._latency_mode:
                self._native.set_latency_mode(True)
        return self._native

    def set_latency_mode(self, on):
        """Opt-in for one or two utterances per call (the reference's 22_infer_tts.py loop): tile shapes and K splits are chosen from
        the actual batch so that a single utterance spreads over the chip.  Same tolerances against the reference, but not
        bit-identical with the default mode, whose results never depend on how a batch is split (include/lds.h).  Not part of the
        reference's API."""
        self._latency_mode = bool(on)
        if self._native is not None:
            self._native.set_latency_mode(self._latency_mode)

    def set_gemm_mode(self, mode):
        """"f32" (default): every conv / linear on the exact-fp32 MFMA.  "split_f16" (opt-in, experimental): the same layers with two fp16
        terms per operand on the fp16 matrix pipe (22-bit operands, fp32 accumulate; csrc/conv_bf3.hip; precondition on the activations'
        range: include/lds.h, check_split_f16_ranges) -- not part of the reference's API.  ("split_bf16" was removed in round 4.)"""
        if mode == "split_bf16":
            raise ValueError("the split_bf16 mode was removed (lossless, but no faster than exact fp32: DESIGN.md 10.1)")
        if mode not in ("f32", "split_f16"):
            raise ValueError(mode)
        self._gemm_mode = mode
        if self._native is not None:
            self._native.set_gemm_mode(mode)

    def check_split_f16_ranges(self, sample, timestep, lo=2.0 ** -3, hi=2.0 ** 15):
        """Debug aid for the split_f16 mode's precondition (include/lds.h): one traced forward of these inputs (include/lds_test.h
        lds_debug_trace: synchronises after every stage; not for the timed path), the abs-max of every tensor the mode stores as two fp16 planes.
        Returns [(stage, absmax)]; raises ValueError naming the stages whose abs-max lies outside [lo, hi] -- above: fp16 overflow ahead; below:
        the second fp16 term goes subnormal and the tensor keeps fewer than 22 bits.  Works in either GEMM mode (the ranges are the model's)."""
        B = sample.shape[0]
        native.debug_trace(True)
        try:
            self.forward(sample, timestep)
        finally:
            native.debug_trace(False)
        ranges = []
        for name, raw in native.debug_trace_records():
            if name.count("|") != 3:
                continue      # fp32 tensors of every mode: q / k / v, eps
            nm, arr = native.debug_trace_decode(name, raw, B)
            ranges.append((nm, float(abs(arr).max())))
        bad = [(nm, m) for nm, m in ranges if not (lo <= m <= hi)]
        if bad:
            raise ValueError("split_f16 precondition: tensors with an abs-max outside [%g, %g]: %s" % (lo, hi, ", ".join("%s %.3g" % b for b in bad[:12])
                                                                                                     + (" ... (%d in all)" % len(bad) if len(bad) > 12 else "")))
        return ranges

    def forward(self, sample, timestep, **kwargs):
        """sample [B, M+H, T] (x stacked on cond, reference diffusion.py:105), timestep [B] or scalar."""
        if any(v is not None for v in kwargs.values()):
            raise NotImplementedError(f"unsupported forward arguments {sorted(kwargs)}")
        B = sample.shape[0]
        M = self.out_channels
        if not torch.is_tensor(timestep):
            timestep = torch.tensor([timestep], device=sample.device)
        t = timestep.reshape(-1).to(device=sample.device, dtype=torch.float32).expand(B).contiguous()
        x = sample[:, :M].contiguous()
        cond = sample[:, M:].contiguous()
        return UNet1DConditionOutput(sample=self.native().forward(x, cond, t))
