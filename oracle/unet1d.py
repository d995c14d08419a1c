"""UNet1DConditionModel forward, numpy fp32 (reference diffusion/unet1d/unet_1d_condition.py:743-1036
with the Unit2Mel configuration, diffusion/unit2mel.py:62-71).  `w` is a dict of the reference's
state_dict keys -> float32 arrays; `cfg` comes from lds.arch.unet_config (passed in by the caller so
this package does not import the product)."""
import math

import numpy as np
from scipy.special import erf

f32 = np.float32


def silu(x):
    return (x / (f32(1.0) + np.exp(-x))).astype(f32)


def conv1d(x, w, b=None, stride=1, pad=0, dil=1):
    """F.conv1d restated as a sum over taps of [Co,Ci]@[Ci,T] products. x [B,Ci,T], w [Co,Ci,K]."""
    B, Ci, T = x.shape
    Co, _, K = w.shape
    xp = np.pad(x, ((0, 0), (0, 0), (pad, pad))) if pad else x
    To = (T + 2 * pad - dil * (K - 1) - 1) // stride + 1
    out = np.zeros((B, Co, To), dtype=f32)
    for k in range(K):
        seg = xp[:, :, k * dil: k * dil + (To - 1) * stride + 1: stride]
        out += np.matmul(w[None, :, :, k], seg)
    if b is not None:
        out += b[None, :, None]
    return out


def linear(x, w, b=None):
    y = np.matmul(x, w.T)
    if b is not None:
        y = y + b
    return y.astype(f32)


def group_norm(x, g, b, groups, eps):
    """nn.GroupNorm over [B,C,T] (biased variance)."""
    B, C, T = x.shape
    xr = x.reshape(B, groups, -1).astype(np.float64)
    mu = xr.mean(-1, keepdims=True)
    var = xr.var(-1, keepdims=True)
    y = ((xr - mu) / np.sqrt(var + eps)).reshape(B, C, T)
    return (y * g[None, :, None] + b[None, :, None]).astype(f32)


def layer_norm(x, g, b, eps=1e-5):
    xr = x.astype(np.float64)
    mu = xr.mean(-1, keepdims=True)
    var = xr.var(-1, keepdims=True)
    return (((xr - mu) / np.sqrt(var + eps)) * g + b).astype(f32)


def timestep_embedding(t, dim=256):
    """Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0) (embeddings.py:24-64):
    [cos | sin](t * exp(-ln(1e4) * i / half))."""
    half = dim // 2
    expo = (f32(-math.log(10000)) * np.arange(half, dtype=f32)) / f32(half)
    freq = np.exp(expo).astype(f32)
    arg = np.asarray(t).astype(f32)[:, None] * freq[None, :]
    return np.concatenate([np.cos(arg), np.sin(arg)], axis=-1).astype(f32)


def time_mlp(w, temb_in):
    """TimestepEmbedding (embeddings.py:157-201): Linear -> SiLU -> Linear."""
    h = linear(temb_in, w["time_embedding.linear_1.weight"], w["time_embedding.linear_1.bias"])
    return linear(silu(h), w["time_embedding.linear_2.weight"], w["time_embedding.linear_2.bias"])


def resnet(w, p, x, emb, groups):
    """ResnetBlock2D.forward, scale_shift (resnet.py:591-641)."""
    h = silu(group_norm(x, w[p + "norm1.weight"], w[p + "norm1.bias"], groups, 1e-5))
    h = conv1d(h, w[p + "conv1.weight"], w[p + "conv1.bias"], pad=1)
    te = linear(silu(emb), w[p + "time_emb_proj.weight"], w[p + "time_emb_proj.bias"])[:, :, None]
    h = group_norm(h, w[p + "norm2.weight"], w[p + "norm2.bias"], groups, 1e-5)
    scale, shift = np.split(te, 2, axis=1)
    h = silu((h * (f32(1.0) + scale) + shift).astype(f32))
    h = conv1d(h, w[p + "conv2.weight"], w[p + "conv2.bias"], pad=1)
    if p + "conv_shortcut.weight" in w:
        x = conv1d(x, w[p + "conv_shortcut.weight"], w[p + "conv_shortcut.bias"])
    return (x + h).astype(f32)


def attention(w, p, x, heads):
    """Attention + AttnProcessor2_0 as self-attention (attention_processor.py:980-1052)."""
    B, T, C = x.shape
    d = C // heads
    q = linear(x, w[p + "to_q.weight"]).reshape(B, T, heads, d).transpose(0, 2, 1, 3)
    k = linear(x, w[p + "to_k.weight"]).reshape(B, T, heads, d).transpose(0, 2, 1, 3)
    v = linear(x, w[p + "to_v.weight"]).reshape(B, T, heads, d).transpose(0, 2, 1, 3)
    s = np.matmul(q, k.transpose(0, 1, 3, 2)) * f32(1.0 / math.sqrt(d))
    s = s - s.max(-1, keepdims=True)
    e = np.exp(s)
    pr = (e / e.sum(-1, keepdims=True)).astype(f32)
    o = np.matmul(pr, v).transpose(0, 2, 1, 3).reshape(B, T, C)
    return linear(o, w[p + "to_out.0.weight"], w[p + "to_out.0.bias"])


def transformer(w, p, x, heads, groups):
    """Transformer2DModel (transformer_1d.py:256-295) + BasicTransformerBlock (attention.py:130-203)
    + GEGLU feed-forward (attention.py:229-247, 299-301)."""
    res = x
    h = group_norm(x, w[p + "norm.weight"], w[p + "norm.bias"], groups, 1e-6)
    h = conv1d(h, w[p + "proj_in.weight"], w[p + "proj_in.bias"]).transpose(0, 2, 1)
    b = p + "transformer_blocks.0."
    h = attention(w, b + "attn1.", layer_norm(h, w[b + "norm1.weight"], w[b + "norm1.bias"]), heads) + h
    h = attention(w, b + "attn2.", layer_norm(h, w[b + "norm2.weight"], w[b + "norm2.bias"]), heads) + h
    n = layer_norm(h, w[b + "norm3.weight"], w[b + "norm3.bias"])
    pj = linear(n, w[b + "ff.net.0.proj.weight"], w[b + "ff.net.0.proj.bias"])
    a, gate = np.split(pj, 2, axis=-1)
    gelu = (f32(0.5) * gate * (f32(1.0) + erf(gate / f32(math.sqrt(2.0))))).astype(f32)
    h = linear((a * gelu).astype(f32), w[b + "ff.net.2.weight"], w[b + "ff.net.2.bias"]) + h
    h = conv1d(np.ascontiguousarray(h.transpose(0, 2, 1)), w[p + "proj_out.weight"], w[p + "proj_out.bias"])
    return (h + res).astype(f32)


def upsample_nearest(x, size=None):
    """F.interpolate(mode='nearest'): scale_factor=2 or explicit size (resnet.py:157-160)."""
    T = x.shape[-1]
    if size is None:
        return np.repeat(x, 2, axis=-1)
    idx = np.minimum((np.arange(size) * (T / size)).astype(np.float32).astype(np.int64), T - 1)
    return x[..., idx]


def unet_forward(w, cfg, blocks, sample, timestep, taps=None):
    """sample [B, Cx+Ccond, T], timestep [B] (float or int) -> [B, Cout, T]."""
    down, mid, up = blocks
    G, H = cfg["groups"], cfg["heads"]
    x = sample.astype(f32)
    T = x.shape[-1]
    n_up = sum(1 for b in up if b["upsample"])
    fwd_size = (T % (2 ** n_up)) != 0
    emb = time_mlp(w, timestep_embedding(timestep, cfg["time_proj_dim"]))
    if taps is not None:
        taps["time_embedding"] = emb
    x = conv1d(x, w["conv_in.weight"], w["conv_in.bias"], pad=1)
    if taps is not None:
        taps["conv_in"] = x
    skips = [x]
    for blk in down:
        p = f"down_blocks.{blk['idx']}."
        for j in range(len(blk["resnets"])):
            x = resnet(w, p + f"resnets.{j}.", x, emb, G)
            if taps is not None:
                taps[p + f"resnets.{j}"] = x
            if blk["attn"]:
                x = transformer(w, p + f"attentions.{j}.", x, H, G)
                if taps is not None:
                    taps[p + f"attentions.{j}"] = x
            skips.append(x)
        if blk["downsample"]:
            x = conv1d(x, w[p + "downsamplers.0.conv.weight"], w[p + "downsamplers.0.conv.bias"], stride=2, pad=1)
            skips.append(x)
        if taps is not None:
            taps[p[:-1]] = x
    x = resnet(w, "mid_block.resnets.0.", x, emb, G)
    x = transformer(w, "mid_block.attentions.0.", x, H, G)
    x = resnet(w, "mid_block.resnets.1.", x, emb, G)
    if taps is not None:
        taps["mid_block"] = x
    for blk in up:
        p = f"up_blocks.{blk['idx']}."
        n = len(blk["resnets"])
        res = skips[-n:]
        skips = skips[:-n]
        for j in range(n):
            x = np.concatenate([x, res.pop()], axis=1)
            x = resnet(w, p + f"resnets.{j}.", x, emb, G)
            if taps is not None:
                taps[p + f"resnets.{j}"] = x
            if blk["attn"]:
                x = transformer(w, p + f"attentions.{j}.", x, H, G)
        if blk["upsample"]:
            size = skips[-1].shape[-1] if fwd_size else None
            x = upsample_nearest(x, size)
            x = conv1d(x, w[p + "upsamplers.0.conv.weight"], w[p + "upsamplers.0.conv.bias"], pad=1)
        if taps is not None:
            taps[p[:-1]] = x
    x = silu(group_norm(x, w["conv_norm_out.weight"], w["conv_norm_out.bias"], G, 1e-5))
    return conv1d(x, w["conv_out.weight"], w["conv_out.bias"], pad=1)
