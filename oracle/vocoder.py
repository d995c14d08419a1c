"""HiFi-VAEGAN Generator forward, numpy fp32 (reference encoder/hifi_vaegan/modules/models.py:161-272,
commons.py:13-14; weight-norm folding as remove_weight_norm in hifi_vaegan.py:57-61)."""
import numpy as np

from .unet1d import conv1d

f32 = np.float32
LRELU_SLOPE = 0.1


def fold_weight_norm(w):
    """weight = g * v / ||v|| with the norm over every dim but 0 (torch weight_norm dim=0)."""
    out = {}
    for k, v in w.items():
        if k.endswith("weight_v"):
            g = w[k[:-1] + "g"]
            nrm = np.sqrt((v.astype(np.float64) ** 2).sum(axis=tuple(range(1, v.ndim)), keepdims=True))
            out[k[:-2]] = (v * (g / nrm)).astype(f32)
        elif not k.endswith("weight_g"):
            out[k] = v
    return out


def lrelu(x, slope=LRELU_SLOPE):
    return np.where(x >= 0, x, x * f32(slope)).astype(f32)


def conv_transpose1d(x, w, b, stride, pad):
    """F.conv_transpose1d: x [B,Ci,T], w [Ci,Co,K] -> [B,Co,(T-1)*stride - 2*pad + K]."""
    B, Ci, T = x.shape
    _, Co, K = w.shape
    full = np.zeros((B, Co, (T - 1) * stride + K), dtype=f32)
    for k in range(K):
        full[:, :, k: k + (T - 1) * stride + 1: stride] += np.matmul(w[:, :, k].T[None], x)
    out = full[:, :, pad: full.shape[-1] - pad]
    return (out + b[None, :, None]).astype(f32)


def get_padding(k, d=1):
    return int((k * d - d) / 2)


def resblock1(w, p, x, k, dil):
    for m, d in enumerate(dil):
        xt = conv1d(lrelu(x), w[p + f"convs1.{m}.weight"], w[p + f"convs1.{m}.bias"], pad=get_padding(k, d), dil=d)
        xt = conv1d(lrelu(xt), w[p + f"convs2.{m}.weight"], w[p + f"convs2.{m}.bias"], pad=get_padding(k, 1))
        x = (xt + x).astype(f32)
    return x


def resblock2(w, p, x, k, dil):
    for m, d in enumerate(dil):
        xt = conv1d(lrelu(x), w[p + f"convs.{m}.weight"], w[p + f"convs.{m}.bias"], pad=get_padding(k, d), dil=d)
        x = (xt + x).astype(f32)
    return x


def generator_forward(w_folded, h, z):
    """z [B, C, T] -> wav [B, 1, T*prod(upsample_rates)]."""
    w = w_folded
    x = conv1d(z.astype(f32), w["conv_pre.weight"], w["conv_pre.bias"], pad=3)
    nk = len(h["resblock_kernel_sizes"])
    rb = resblock1 if h["resblock"] == "1" else resblock2
    for i, (u, k) in enumerate(zip(h["upsample_rates"], h["upsample_kernel_sizes"])):
        x = conv_transpose1d(lrelu(x), w[f"ups.{i}.weight"], w[f"ups.{i}.bias"], u, (k - u + 1) // 2)
        xs = None
        for j, (kk, dil) in enumerate(zip(h["resblock_kernel_sizes"], h["resblock_dilation_sizes"])):
            y = rb(w, f"resblocks.{i * nk + j}.", x, kk, dil)
            xs = y if xs is None else (xs + y).astype(f32)
        x = (xs / f32(nk)).astype(f32)
    x = lrelu(x, 0.01)  # F.leaky_relu default slope (models.py:260)
    x = conv1d(x, w["conv_post.weight"], w["conv_post.bias"], pad=3)
    return np.tanh(x).astype(f32)
