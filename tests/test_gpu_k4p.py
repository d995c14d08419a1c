"""Parity of the UNet's K4P kernels (k-interleaved padded layout, DMA-fed VALU-free GEMM, materialised GroupNorm /
LayerNorm, vector-operand attention) against the numpy oracle, each through its C-ABI test entry point."""
import ctypes as ct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def relmax(a, b):
    return float(np.abs(a - b).max() / max(1e-30, np.abs(b).max()))


def U(name, shape, lo=-1.0, hi=1.0):
    from lds import init_weights
    return init_weights.uniform("k4." + name, shape, 9, lo, hi)


def stream():
    return ct.c_void_p(torch.cuda.current_stream().cuda_stream)


def run_dconv(x1, w, bias=None, x2=None, stride=1, pad=0, ups=0, res=None, epi=0, plain_out=0, v_split=0, cfg=0, want_ln=False):
    from lds import native
    B, C1, T = x1.shape
    C2 = 0 if x2 is None else x2.shape[1]
    Co, _, K = w.shape
    a = native.DConvTest()
    dx1, dx2 = dev(x1), (dev(x2) if x2 is not None else None)
    keep = [np.ascontiguousarray(w, dtype=np.float32)]
    a.x1, a.x2 = dx1.data_ptr(), (dx2.data_ptr() if dx2 is not None else None)
    a.C1, a.C2, a.T = C1, C2, T
    a.w = keep[0].ctypes.data
    if bias is not None:
        keep.append(np.ascontiguousarray(bias, dtype=np.float32))
        a.bias = keep[-1].ctypes.data
    a.Co, a.K, a.stride, a.pad, a.ups = Co, K, stride, pad, ups
    Tin = 2 * T if ups else T
    To = (Tin + 2 * pad - (K - 1) - 1) // stride + 1
    Cout = Co // 2 if epi == 1 else Co
    dres = dev(res) if res is not None else None
    a.res = dres.data_ptr() if dres is not None else None
    a.epilogue, a.plain_out, a.v_split, a.cfg = epi, plain_out, v_split, cfg
    out = torch.full((B, Cout, To), float("nan"), dtype=torch.float32, device="cuda")
    ln = torch.full((B, Cout // 32, To, 2), float("nan"), dtype=torch.float32, device="cuda") if want_ln else None
    native.check(native.lib().lds_test_dconv(ct.byref(a), ct.c_void_p(out.data_ptr()), ct.c_void_p(ln.data_ptr()) if want_ln else None, B,
                                             stream()))
    torch.cuda.synchronize()
    return (out.cpu().numpy(), ln.cpu().numpy()) if want_ln else out.cpu().numpy()


def ref_dconv(x1, w, bias=None, x2=None, stride=1, pad=0, ups=0, res=None, epi=0):
    from oracle import unet1d
    from scipy.special import erf
    x = x1 if x2 is None else np.concatenate([x1, x2], axis=1)
    if ups:
        x = np.repeat(x, 2, axis=-1)
    y = unet1d.conv1d(np.ascontiguousarray(x), w, bias, stride=stride, pad=pad)
    if epi == 1:
        a, g = np.split(y, 2, axis=1)
        y = (a * (0.5 * g * (1 + erf(g / np.sqrt(2.0))))).astype(np.float32)
    if res is not None:
        y = y + res
    return y.astype(np.float32)


DCASES = {
    # name: (B, C1, C2, T, Co, K, kw)
    "1x1_64": (2, 64, 0, 64, 64, 1, {}),
    "1x1_res_bias": (2, 128, 0, 96, 192, 1, dict(bias=True, res=True)),
    "1x1_ragged_T": (2, 64, 0, 50, 128, 1, dict(bias=True)),
    "1x1_concat": (2, 64, 64, 72, 128, 1, dict(bias=True)),
    "1x1_bk32": (1, 96, 0, 64, 64, 1, dict(bias=True)),                    # Ci % 64 != 0 -> BK 32
    "1x1_bk16": (1, 80, 0, 64, 64, 1, dict(bias=True)),                    # Ci % 32 != 0 -> BK 16
    "k3": (2, 64, 0, 64, 64, 3, dict(pad=1, bias=True)),
    "k3_ragged": (2, 96, 0, 37, 128, 3, dict(pad=1, bias=True, res=True)),
    "k3_336_bk16": (1, 80, 256, 40, 256, 3, dict(pad=1, bias=True)),       # conv_in geometry: x (80 ch) ++ cond (256 ch)
    "k3_s2": (2, 64, 0, 64, 64, 3, dict(pad=1, stride=2, bias=True)),
    "k3_s2_odd": (2, 64, 0, 45, 64, 3, dict(pad=1, stride=2)),
    "k3_ups": (2, 64, 0, 40, 64, 3, dict(pad=1, ups=1, bias=True)),
    "k3_ups_odd": (1, 64, 0, 33, 64, 3, dict(pad=1, ups=1)),
    "k3_plain_out_co80": (2, 64, 0, 64, 80, 3, dict(pad=1, bias=True, plain_out=1)),
    "geglu": (2, 64, 0, 64, 512, 1, dict(epi=1, bias=True)),
    "geglu_128128": (1, 64, 0, 256, 512, 1, dict(epi=1, bias=True, cfg=128128322)),
    "qkv_split": (2, 64, 0, 72, 192, 1, dict(v_split=1)),
    "t128128_k3": (1, 128, 0, 256, 128, 3, dict(pad=1, bias=True, cfg=128128162)),
    "t128064": (1, 64, 0, 128, 128, 1, dict(bias=True, cfg=128064322)),
    "nst3_k3": (2, 64, 0, 100, 64, 3, dict(pad=1, bias=True, res=True, cfg=64064323)),
    "nst3_1x1_bk16": (1, 256, 0, 64, 64, 1, dict(cfg=64064163)),
    "nst2_1x1_deepk": (1, 1024, 0, 64, 64, 1, dict(cfg=64064642)),
    # split-K inside the workgroup (32 x 64 tile, two wave pairs each take half of every K-step)
    "split_1x1": (2, 128, 0, 50, 96, 1, dict(bias=True, res=True, cfg=32064322)),
    "split_1x1_bk64": (1, 256, 0, 64, 64, 1, dict(bias=True, cfg=32064642)),
    "split_k3": (2, 64, 0, 37, 128, 3, dict(pad=1, bias=True, res=True, cfg=32064322)),
    "split_k3_concat": (1, 64, 96, 64, 64, 3, dict(pad=1, cfg=32064322)),
}


@pytest.mark.parametrize("name", list(DCASES))
def test_conv_dma(name):
    B, C1, C2, T, Co, K, kw = DCASES[name]
    kw = dict(kw)
    x1 = U(name + ".x1", (B, C1, T), -2, 2)
    x2 = U(name + ".x2", (B, C2, T), -2, 2) if C2 else None
    Ci = C1 + C2
    w = U(name + ".w", (Co, Ci, K)) / np.float32(np.sqrt(Ci * K))
    args = dict(x2=x2)
    if kw.pop("bias", False):
        args["bias"] = U(name + ".b", (Co,))
    for k in ("stride", "pad", "ups", "epi"):
        if k in kw:
            args[k] = kw[k]
    ref0 = ref_dconv(x1, w, **args)
    if kw.pop("res", False):
        args["res"] = U(name + ".res", ref0.shape, -1, 1)
    ref = ref_dconv(x1, w, **args)
    out = run_dconv(x1, w, plain_out=kw.get("plain_out", 0), v_split=kw.get("v_split", 0), cfg=kw.get("cfg", 0), **args)
    assert out.shape == ref.shape
    assert np.isfinite(out).all()
    assert relmax(out, ref) < 2e-5, relmax(out, ref)


@pytest.mark.parametrize("C,heads,T", [(64, 2, 72), (96, 2, 37), (128, 2, 50)])      # head dim 32 / 48 / 64; T % 4 = 0, 1, 2
def test_conv_dma_value_layout(C, heads, T):
    """QKV convolution: q, k leave in K4P, v in attention's VT layout [B][heads][ceil(T/4)][D][4] with a zeroed key tail"""
    from lds import native
    B, D = 2, C // heads
    x = U(f"vt{C}.x", (B, 64, T), -2, 2)
    w = U(f"vt{C}.w", (3 * C, 64, 1)) / np.float32(8.0)
    ref = ref_dconv(x, w)
    T4 = (T + 3) // 4 * 4
    a = native.DConvTest()
    dx = dev(x)
    wk = np.ascontiguousarray(w, dtype=np.float32)
    a.x1, a.x2, a.C1, a.C2, a.T = dx.data_ptr(), None, 64, 0, T
    a.w, a.bias, a.Co, a.K, a.stride, a.pad, a.ups = wk.ctypes.data, None, 3 * C, 1, 1, 0, 0
    a.res, a.epilogue, a.plain_out, a.v_split, a.cfg = None, 0, 0, D, 0
    out = torch.full((B * 2 * C * T + B * C * T4,), float("nan"), dtype=torch.float32, device="cuda")
    native.check(native.lib().lds_test_dconv(ct.byref(a), ct.c_void_p(out.data_ptr()), None, B, stream()))
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    qk = o[:B * 2 * C * T].reshape(B, 2 * C, T)
    vt = o[B * 2 * C * T:].reshape(B, heads, T4 // 4, D, 4)
    assert relmax(qk, ref[:, :2 * C]) < 2e-5
    v = np.zeros((B, C, T4), dtype=np.float32)
    v[:, :, :T] = ref[:, 2 * C:]
    want = v.reshape(B, heads, D, T4 // 4, 4).transpose(0, 1, 3, 2, 4)
    assert np.isfinite(vt).all()
    assert relmax(vt, want) < 2e-5
    assert (vt.transpose(0, 1, 3, 2, 4).reshape(B, C, T4)[:, :, T:] == 0).all()


@pytest.mark.parametrize("cfg", [0, 64064322, 32064322])
def test_conv_dma_layernorm_partials(cfg):
    """the epilogue's per-frame (mean, M2) over each 32-channel tile (also from the split-K tile)"""
    B, C, T = 2, 128, 70
    x = U("lnp.x", (B, 64, T), -2, 2)
    w = U("lnp.w", (C, 64, 1)) / np.float32(8.0)
    out, ln = run_dconv(x, w, want_ln=True, cfg=cfg)
    t = out.reshape(B, C // 32, 32, T).astype(np.float64)
    assert np.abs(ln[..., 0] - t.mean(2)).max() < 1e-5
    assert np.abs(ln[..., 1] - ((t - t.mean(2, keepdims=True)) ** 2).sum(2)).max() < 1e-3


@pytest.mark.parametrize("C1,C2,T,silu,ss", [(128, 0, 64, 1, False), (80, 176, 50, 1, False), (128, 0, 64, 1, True), (256, 0, 37, 0, False),
                                             (512, 384, 128, 1, False),     # 112-channel groups straddle the two sources
                                             (256, 256, 512, 1, True),      # largest model shape: 64 partials per group
                                             (512, 0, 700, 1, False),       # more than one chunk per workgroup, ragged last frame block
                                             (64, 64, 3000, 1, True)])      # 94 partials per group, six chunks
def test_gn_apply(C1, C2, T, silu, ss):
    """streaming GroupNorm (gn_stream) fed by stand-alone partial statistics (gn_partials)"""
    from lds import native
    from oracle import unet1d
    B, C = 2, C1 + C2
    x1 = U(f"gn{C}.x1", (B, C1, T), -2, 2) + np.float32(0.7)
    x2 = U(f"gn{C}.x2", (B, C2, T), -3, 1) if C2 else None
    g, be = U(f"gn{C}.g", (C,), 0.5, 1.5), U(f"gn{C}.b", (C,), -0.5, 0.5)
    sst = U(f"gn{C}.ss", (B, 2 * C), -0.5, 0.5) if ss else None
    out = torch.full((B, C, T), float("nan"), dtype=torch.float32, device="cuda")
    d1, d2, dss = dev(x1), (dev(x2) if C2 else None), (dev(sst) if ss else None)
    native.check(native.lib().lds_test_gn_apply(ct.c_void_p(d1.data_ptr()), ct.c_void_p(d2.data_ptr()) if C2 else None, C1, C2, T, 8,
                                                ct.c_float(1e-5), ct.c_void_p(g.ctypes.data), ct.c_void_p(be.ctypes.data),
                                                ct.c_void_p(dss.data_ptr()) if ss else None, silu, ct.c_void_p(out.data_ptr()), B, stream()))
    torch.cuda.synchronize()
    x = x1 if x2 is None else np.concatenate([x1, x2], axis=1)
    ref = unet1d.group_norm(x, g, be, 8, 1e-5)
    if ss:
        ref = (ref * (1 + sst[:, :C, None]) + sst[:, C:, None]).astype(np.float32)
    if silu:
        ref = unet1d.silu(ref)
    assert relmax(out.cpu().numpy(), ref) < 1e-5, relmax(out.cpu().numpy(), ref)


@pytest.mark.parametrize("C,Co,T,B,cfg,silu", [(64, 256, 70, 2, 0, 1), (128, 512, 37, 1, 32064322, 0), (64, 384, 100, 2, 128064322, 1),
                                               (64, 256, 512, 2, 128128162, 1), (256, 384, 256, 1, 64064163, 1), (64, 128, 33, 3, 0, 1)])
def test_groupnorm_chain_k4p(C, Co, T, B, cfg, silu):
    """the UNet's GroupNorm statistics path: the producing convolution's epilogue writes (mean, M2) of every (16 channels x
    32 frames) block -- from every tile shape, the split-K one included -- and gn_stream combines them"""
    from lds import native
    from oracle import unet1d
    x = U(f"gnc{C}.{T}.x", (B, C, T), -2, 2)
    w1 = (U(f"gnc{C}.{T}.w1", (Co, C)) / np.float32(np.sqrt(C))).astype(np.float32)
    b1 = U(f"gnc{C}.{T}.b1", (Co,), 0.5, 1.5)                                      # a mean well away from zero
    g, be = U(f"gnc{C}.{T}.g", (Co,), 0.5, 1.5), U(f"gnc{C}.{T}.b", (Co,), -0.5, 0.5)
    mid = torch.full((B, Co, T), float("nan"), dtype=torch.float32, device="cuda")
    out = torch.full((B, Co, T), float("nan"), dtype=torch.float32, device="cuda")
    dx = dev(x)
    native.check(native.lib().lds_test_gn_chain_k4p(ct.c_void_p(dx.data_ptr()), ct.c_void_p(w1.ctypes.data), ct.c_void_p(b1.ctypes.data),
                                                    ct.c_void_p(g.ctypes.data), ct.c_void_p(be.ctypes.data), ct.c_float(1e-5), 8, silu,
                                                    ct.c_void_p(mid.data_ptr()), ct.c_void_p(out.data_ptr()), B, C, Co, T, cfg, stream()))
    torch.cuda.synchronize()
    rmid = unet1d.conv1d(x, w1[:, :, None], b1)
    ref = unet1d.group_norm(rmid, g, be, 8, 1e-5)
    if silu:
        ref = unet1d.silu(ref)
    assert relmax(mid.cpu().numpy(), rmid) < 2e-5
    assert relmax(out.cpu().numpy(), ref) < 2e-5, relmax(out.cpu().numpy(), ref)


# (Cm = the normalised tensor's channels: group sizes 32 / 48 / 64; T, B pick the tile: 64 x 64 / 128 x 64 at full grids, the 32 x 64 split-K
#  tile at small ones; tile_batch > 0 = the latency mode's choice with a workgroup cluster per tile)
@pytest.mark.parametrize("Cm,T,B,tile_batch", [(256, 512, 16, 0), (256, 512, 2, 0), (384, 256, 16, 0), (384, 256, 2, 0), (384, 100, 1, 0), (512, 128, 16, 0),
                                               (512, 64, 3, 0), (512, 37, 1, 0), (256, 512, 1, 1), (384, 256, 1, 1), (512, 128, 2, 2), (512, 64, 1, 1),
                                               (128, 70, 2, 0)])
def test_groupnorm_fold_k4p(Cm, T, B, tile_batch, record_margin):
    """the transformer blocks' `norm` (GroupNorm, affine, no activation) folded into proj_in (reference transformer_1d.py:256-266): the
    1x1 convolution reads the un-normalised tensor, rescales its accumulators between groups and adds a per-row constant; statistics
    from the producer's epilogue partials.  The normalised tensor's mean is well away from zero (the fold subtracts mean * sum(W))."""
    from lds import native
    from oracle import unet1d
    C, Co = 64, Cm
    x = U(f"gnf{Cm}.{T}.x", (B, C, T), -2, 2)
    w1 = (U(f"gnf{Cm}.{T}.w1", (Cm, C)) / np.float32(np.sqrt(C))).astype(np.float32)
    b1 = U(f"gnf{Cm}.{T}.b1", (Cm,), 0.5, 1.5)
    g, be = U(f"gnf{Cm}.{T}.g", (Cm,), 0.5, 1.5), U(f"gnf{Cm}.{T}.b", (Cm,), -0.5, 0.5)
    w2 = (U(f"gnf{Cm}.{T}.w2", (Co, Cm)) / np.float32(np.sqrt(Cm))).astype(np.float32)
    b2 = U(f"gnf{Cm}.{T}.b2", (Co,), -0.5, 0.5)
    mid = torch.full((B, Cm, T), float("nan"), dtype=torch.float32, device="cuda")
    out = torch.full((B, Co, T), float("nan"), dtype=torch.float32, device="cuda")
    dx = dev(x)
    P = lambda a: ct.c_void_p(a.ctypes.data)
    native.check(native.lib().lds_test_gn_fold_k4p(ct.c_void_p(dx.data_ptr()), P(w1), P(b1), P(g), P(be), ct.c_float(1e-6), 8, P(w2), P(b2),
                                                   ct.c_void_p(mid.data_ptr()), ct.c_void_p(out.data_ptr()), B, C, Cm, Co, T, 0, tile_batch, stream()))
    torch.cuda.synchronize()
    rmid = unet1d.conv1d(x, w1[:, :, None], b1)
    ref = unet1d.conv1d(unet1d.group_norm(rmid, g, be, 8, 1e-6), w2[:, :, None], b2)
    assert relmax(mid.cpu().numpy(), rmid) < 2e-5
    record_margin(relmax(out.cpu().numpy(), ref), 2e-5)


@pytest.mark.parametrize("C,Co,T,B", [(256, 768, 64, 2), (384, 384, 100, 1), (512, 1536, 37, 2), (1024, 256, 70, 1), (128, 128, 40, 1)])
def test_layernorm_chain_k4p(C, Co, T, B):
    """... 1024 channels: 32 partials per column, the two-read form of gn_chan.h ln_column_stats (the UNet's widest level has 16)"""
    from lds import native
    from oracle import unet1d
    x = U(f"lnc{C}.x", (B, C, T), -2, 2)
    w1 = (U(f"lnc{C}.w1", (C, C)) / np.float32(np.sqrt(C))).astype(np.float32)
    w2 = (U(f"lnc{C}.w2", (Co, C)) / np.float32(np.sqrt(C))).astype(np.float32)
    g, be = U(f"lnc{C}.g", (C,), 0.5, 1.5), U(f"lnc{C}.b", (C,), -0.5, 0.5)
    mid = torch.full((B, C, T), float("nan"), dtype=torch.float32, device="cuda")
    out = torch.full((B, Co, T), float("nan"), dtype=torch.float32, device="cuda")
    dx = dev(x)
    native.check(native.lib().lds_test_ln_chain_k4p(ct.c_void_p(dx.data_ptr()), ct.c_void_p(w1.ctypes.data), ct.c_void_p(w2.ctypes.data),
                                                    ct.c_void_p(g.ctypes.data), ct.c_void_p(be.ctypes.data), ct.c_float(1e-5),
                                                    ct.c_void_p(mid.data_ptr()), ct.c_void_p(out.data_ptr()), B, C, Co, T, stream()))
    torch.cuda.synchronize()
    rmid = unet1d.conv1d(x, w1[:, :, None])
    rn = unet1d.layer_norm(rmid.transpose(0, 2, 1), g, be, 1e-5).transpose(0, 2, 1)
    ref = unet1d.conv1d(np.ascontiguousarray(rn), w2[:, :, None])
    assert relmax(mid.cpu().numpy(), rmid) < 2e-5
    assert relmax(out.cpu().numpy(), ref) < 2e-5, relmax(out.cpu().numpy(), ref)


# B * heads * ceil(T/128) >= 256 selects 128-query workgroups, T <= 32 single-wave ones, the rest 64-query ones
@pytest.mark.parametrize("C,T,B", [(256, 64, 2), (256, 512, 1), (384, 256, 1), (384, 100, 2), (512, 128, 2), (512, 37, 1), (256, 130, 1),
                                   (256, 512, 8), (384, 256, 16), (512, 128, 32), (256, 20, 2), (384, 32, 1)])
@pytest.mark.parametrize("math", ["f32", "f16x2", "f32-lat", "f16x2-lat"])
def test_attention_k4p(C, T, B, math):
    """math "f16x2": the same kernel with Q K^T and P V on the fp16 matrix pipe, operands split in registers into two fp16 terms (the
    attention of the split-fp16 GEMM mode) -- same tolerance"""
    from lds import native
    heads = 8
    d = C // heads
    qkv = U(f"att{C}.{T}", (B, 3 * C, T), -1.5, 1.5)
    if T == 130:
        qkv = qkv.copy()
        qkv[:, C:2 * C, 100] *= 10.0          # a dominating key late in the sequence forces the online-softmax rescale
    out = torch.full((B, C, T), float("nan"), dtype=torch.float32, device="cuda")
    dq = dev(qkv)
    if math.endswith("-lat"):      # the latency mode's choice: 32-query workgroups, key chunks round-robin over four waves
        native.check(native.lib().lds_test_attention_latency(ct.c_void_p(dq.data_ptr()), ct.c_void_p(out.data_ptr()), B, C, T, heads,
                                                             1 if math.startswith("f16") else 0, stream()))
    else:
        fn = native.lib().lds_test_attention_k4p if math == "f32" else native.lib().lds_test_attention_f16math
        native.check(fn(ct.c_void_p(dq.data_ptr()), ct.c_void_p(out.data_ptr()), B, C, T, heads, stream()))
    torch.cuda.synchronize()
    q, k, v = [qkv[:, i * C:(i + 1) * C].reshape(B, heads, d, T).astype(np.float64) for i in range(3)]
    s = np.einsum("bhdq,bhdk->bhqk", q, k) / np.sqrt(d)
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s)
    p /= p.sum(-1, keepdims=True)
    ref = np.einsum("bhqk,bhdk->bhdq", p, v).reshape(B, C, T)
    got = out.cpu().numpy()
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() < 3e-5 * np.abs(ref).max() + 1e-6
