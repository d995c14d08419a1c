"""Parity of the fp32-equivalent split-bf16 path (csrc/conv_bf3.hip, k8b3.h: three bf16 terms per fp32 operand, six bf16 products
per fp32 product, fp32 accumulate) against the numpy oracle, each kernel through its C-ABI test entry point and at the SAME
tolerances as the exact-fp32 kernels (tests/test_gpu_k4p.py): 2e-5 relative to the output's abs-max."""
import ctypes as ct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from test_gpu_k4p import U, dev, ref_dconv, relmax, stream  # noqa: E402


FMTS = [0, 1]      # csrc/k8b3.h: 0 = three bf16 planes (lossless, 6 products), 1 = two fp16 planes (22-bit operands, 3 products)
FMT_IDS = ["bf16x3", "f16x2"]


def run_dconv_bf3(x1, w, bias=None, x2=None, stride=1, pad=0, ups=0, res=None, epi=0, plain_out=0, v_split=0, cfg=0, want_ln=False, nprod=0, fmt=0):
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
    native.check(native.lib().lds_test_dconv_split(ct.byref(a), ct.c_void_p(out.data_ptr()), ct.c_void_p(ln.data_ptr()) if want_ln else None, B,
                                                   nprod, fmt, stream()))
    torch.cuda.synchronize()
    return (out.cpu().numpy(), ln.cpu().numpy()) if want_ln else out.cpu().numpy()


def test_k8h2_roundtrip_error_bound():
    """the fp16 pair is NOT lossless: |v - (v1 + v2)| <= 2^-22 |v| in fp16's normal range, and an absolute 2^-25 below 2^-3 (second term
    subnormal); magnitudes beyond 65504 overflow (documented precondition of the format)"""
    from lds import native
    B, C, T = 2, 64, 300
    x = U("rth", (B, C, T), -3, 3)
    x.reshape(-1)[:8] = np.array([0.0, 1.0, -1.0, 0.1, 1 / 3, 1000.5, 6.0e4, 1e-3], dtype=np.float32)
    out = torch.full((B, C, T), float("nan"), dtype=torch.float32, device="cuda")
    dx = dev(x)
    native.check(native.lib().lds_test_split_roundtrip(ct.c_void_p(dx.data_ptr()), ct.c_void_p(out.data_ptr()), B, C, T, 1, stream()))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    err = np.abs(got.astype(np.float64) - x)
    assert (err <= np.maximum(np.abs(x) * 2.0 ** -22, 2.0 ** -25)).all(), float((err / np.maximum(np.abs(x), 2.0 ** -3)).max())
    assert (err > 0).any()


@pytest.mark.parametrize("shape", [(2, 64, 70), (1, 8, 1), (3, 336, 513)])
def test_k8b3_roundtrip_is_lossless(shape):
    """fp32 -> three bf16 planes -> fp32 returns every value bit for bit (normal range, tiny and huge magnitudes); the one exception
    is the sign of a zero (-0.0 comes back as +0.0: its residual terms are +0), which no consumer on the path can observe"""
    from lds import native
    B, C, T = shape
    x = U(f"rt{C}.{T}", shape, -3, 3)
    flat = x.reshape(-1)
    special = np.array([0.0, -0.0, 1.0, -1.0, 1e-30, -3e-25, 6.5e4, 3.0e38, 1.17549435e-38 * 8192, np.float32(1) + np.float32(2 ** -23),
                        np.float32(1) - np.float32(2 ** -24), 0.1, 1 / 3, 255.99998], dtype=np.float32)
    flat[: min(len(special), flat.size)] = special[: flat.size]
    out = torch.full(shape, float("nan"), dtype=torch.float32, device="cuda")
    dx = dev(x)
    native.check(native.lib().lds_test_k8b3_roundtrip(ct.c_void_p(dx.data_ptr()), ct.c_void_p(out.data_ptr()), B, C, T, stream()))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    nz = x != 0
    assert np.array_equal(got.view(np.uint32)[nz], x.view(np.uint32)[nz])
    assert (got[~nz] == 0).all()


BCASES = {
    # name: (B, C1, C2, T, Co, K, kw)
    "1x1_64": (2, 64, 0, 64, 64, 1, {}),
    "1x1_res_bias": (2, 128, 0, 96, 192, 1, dict(bias=True, res=True)),
    "1x1_ragged_T": (2, 64, 0, 50, 128, 1, dict(bias=True)),
    "1x1_concat": (2, 64, 64, 72, 128, 1, dict(bias=True)),
    "1x1_bk32": (1, 96, 0, 64, 64, 1, dict(bias=True)),                    # Ci % 64 != 0 -> BK 32
    "1x1_bk16": (1, 80, 0, 64, 64, 1, dict(bias=True)),                    # Ci % 32 != 0 -> BK 16 (odd group count per K-step)
    "k3": (2, 64, 0, 64, 64, 3, dict(pad=1, bias=True)),
    "k3_ragged": (2, 96, 0, 37, 128, 3, dict(pad=1, bias=True, res=True)),
    "k3_80_bk16": (1, 80, 0, 40, 256, 3, dict(pad=1, bias=True, res=True)),         # conv_in over the sample's channels + condition half
    "k3_336_bk16": (1, 80, 256, 40, 256, 3, dict(pad=1, bias=True)),
    "k3_s2": (2, 64, 0, 64, 64, 3, dict(pad=1, stride=2, bias=True)),
    "k3_s2_odd": (2, 64, 0, 45, 64, 3, dict(pad=1, stride=2)),
    "k3_ups": (2, 64, 0, 40, 64, 3, dict(pad=1, ups=1, bias=True)),
    "k3_ups_odd": (1, 64, 0, 33, 64, 3, dict(pad=1, ups=1)),
    "k3_plain_out_co80": (2, 64, 0, 64, 80, 3, dict(pad=1, bias=True, plain_out=1)),
    "geglu": (2, 64, 0, 64, 512, 1, dict(epi=1, bias=True)),
    "geglu_128128": (1, 64, 0, 256, 512, 1, dict(epi=1, bias=True, cfg=128128322)),
    "qkv_split": (2, 64, 0, 72, 192, 1, dict(v_split=1)),
    "t128128_1x1": (1, 128, 0, 256, 128, 1, dict(bias=True, res=True, cfg=128128322)),
    "t128128_1x1_bk16": (1, 80, 0, 256, 128, 1, dict(bias=True, cfg=128128163)),
    "t128128_k3": (1, 128, 0, 256, 128, 3, dict(pad=1, bias=True, cfg=128128162)),
    "t128064": (1, 64, 0, 128, 128, 1, dict(bias=True, cfg=128064322)),
    "t064128_k3": (1, 64, 0, 200, 64, 3, dict(pad=1, bias=True, res=True, cfg=64128162)),
    "nst3_k3": (2, 64, 0, 100, 64, 3, dict(pad=1, bias=True, res=True, cfg=64064163)),
    "nst4_1x1": (1, 256, 0, 64, 64, 1, dict(cfg=64064324)),
    "nst2_1x1_deepk": (1, 1024, 0, 64, 64, 1, dict(cfg=64064642)),
    "split_1x1": (2, 128, 0, 50, 96, 1, dict(bias=True, res=True, cfg=32064322)),
    "split_1x1_bk64": (1, 256, 0, 64, 64, 1, dict(bias=True, cfg=32064642)),
    "split_k3": (2, 64, 0, 37, 128, 3, dict(pad=1, bias=True, res=True, cfg=32064322)),
    "split_k3_concat": (1, 64, 96, 64, 64, 3, dict(pad=1, cfg=32064322)),
}


@pytest.mark.parametrize("fmt", FMTS, ids=FMT_IDS)
@pytest.mark.parametrize("name", list(BCASES))
def test_conv_bf3(name, fmt):
    B, C1, C2, T, Co, K, kw = BCASES[name]
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
    out = run_dconv_bf3(x1, w, plain_out=kw.get("plain_out", 0), v_split=kw.get("v_split", 0), cfg=kw.get("cfg", 0), fmt=fmt, **args)
    assert out.shape == ref.shape
    assert np.isfinite(out).all()
    assert relmax(out, ref) < 2e-5, relmax(out, ref)


@pytest.mark.parametrize("fmt", FMTS, ids=FMT_IDS)
@pytest.mark.parametrize("C,heads,T", [(64, 2, 72), (96, 2, 37), (128, 2, 50)])
def test_conv_bf3_value_layout(C, heads, T, fmt):
    """QKV projection of the split-bf16 path: q, k leave in fp32 K4P (the attention kernel's input), v in its VT layout"""
    from lds import native
    B, D = 2, C // heads
    x = U(f"bvt{C}.x", (B, 64, T), -2, 2)
    w = U(f"bvt{C}.w", (3 * C, 64, 1)) / np.float32(8.0)
    ref = ref_dconv(x, w)
    T4 = (T + 3) // 4 * 4
    a = native.DConvTest()
    dx = dev(x)
    wk = np.ascontiguousarray(w, dtype=np.float32)
    a.x1, a.x2, a.C1, a.C2, a.T = dx.data_ptr(), None, 64, 0, T
    a.w, a.bias, a.Co, a.K, a.stride, a.pad, a.ups = wk.ctypes.data, None, 3 * C, 1, 1, 0, 0
    a.res, a.epilogue, a.plain_out, a.v_split, a.cfg = None, 0, 0, D, 0
    out = torch.full((B * 2 * C * T + B * C * T4,), float("nan"), dtype=torch.float32, device="cuda")
    native.check(native.lib().lds_test_dconv_split(ct.byref(a), ct.c_void_p(out.data_ptr()), None, B, 0, fmt, stream()))
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


@pytest.mark.parametrize("fmt", FMTS, ids=FMT_IDS)
@pytest.mark.parametrize("cfg", [0, 64064322, 32064322])
def test_conv_bf3_layernorm_partials(cfg, fmt):
    B, C, T = 2, 128, 70
    x = U("blnp.x", (B, 64, T), -2, 2)
    w = U("blnp.w", (C, 64, 1)) / np.float32(8.0)
    out, ln = run_dconv_bf3(x, w, want_ln=True, cfg=cfg, fmt=fmt)
    t = out.reshape(B, C // 32, 32, T).astype(np.float64)
    assert np.abs(ln[..., 0] - t.mean(2)).max() < 1e-5
    assert np.abs(ln[..., 1] - ((t - t.mean(2, keepdims=True)) ** 2).sum(2)).max() < 1e-3


@pytest.mark.parametrize("C1,C2,T,silu,ss", [(128, 0, 64, 1, False), (80, 176, 50, 1, False), (128, 0, 64, 1, True), (256, 0, 37, 0, False),
                                             (512, 384, 128, 1, False), (256, 256, 512, 1, True), (512, 0, 700, 1, False), (64, 64, 3000, 1, True)])
@pytest.mark.parametrize("fmt", FMTS, ids=FMT_IDS)
def test_gn_apply_bf3(C1, C2, T, silu, ss, fmt):
    """streaming GroupNorm over K8B3 tensors, statistics from the stand-alone partials pass"""
    from lds import native
    from oracle import unet1d
    B, C = 2, C1 + C2
    x1 = U(f"bgn{C}.x1", (B, C1, T), -2, 2) + np.float32(0.7)
    x2 = U(f"bgn{C}.x2", (B, C2, T), -3, 1) if C2 else None
    g, be = U(f"bgn{C}.g", (C,), 0.5, 1.5), U(f"bgn{C}.b", (C,), -0.5, 0.5)
    sst = U(f"bgn{C}.ss", (B, 2 * C), -0.5, 0.5) if ss else None
    out = torch.full((B, C, T), float("nan"), dtype=torch.float32, device="cuda")
    d1, d2, dss = dev(x1), (dev(x2) if C2 else None), (dev(sst) if ss else None)
    native.check(native.lib().lds_test_gn_apply_split(ct.c_void_p(d1.data_ptr()), ct.c_void_p(d2.data_ptr()) if C2 else None, C1, C2, T, 8,
                                                      ct.c_float(1e-5), ct.c_void_p(g.ctypes.data), ct.c_void_p(be.ctypes.data),
                                                      ct.c_void_p(dss.data_ptr()) if ss else None, silu, ct.c_void_p(out.data_ptr()), B, fmt, stream()))
    torch.cuda.synchronize()
    x = x1 if x2 is None else np.concatenate([x1, x2], axis=1)
    ref = unet1d.group_norm(x, g, be, 8, 1e-5)
    if ss:
        ref = (ref * (1 + sst[:, :C, None]) + sst[:, C:, None]).astype(np.float32)
    if silu:
        ref = unet1d.silu(ref)
    assert relmax(out.cpu().numpy(), ref) < 1e-5, relmax(out.cpu().numpy(), ref)


@pytest.mark.parametrize("scale", [1e-3, 1.0, 1e3])
def test_split_formats_vs_activation_scale(scale):
    """What each format promises when a whole activation tensor lives at another scale (fp64 reference): bf16x3 is lossless, so its error
    does not depend on the scale; fp16x2 keeps fp32-like relative accuracy from ~0.1 up to 65504 and degrades below (second term
    subnormal under 2^-3: absolute floor 2^-25) -- the documented precondition of the split-fp16 mode."""
    B, C, T, Co = 2, 256, 128, 128
    x = (U("sc.x", (B, C, T), -2, 2) * np.float32(scale)).astype(np.float32)
    w = (U("sc.w", (Co, C, 1)) / np.float32(16.0)).astype(np.float32)
    ref = np.einsum("oc,bct->bot", w[:, :, 0].astype(np.float64), x.astype(np.float64))
    rms = float(np.sqrt((ref ** 2).mean()))

    def err(out):
        return float(np.sqrt(((out - ref) ** 2).mean())) / rms
    e_b = err(run_dconv_bf3(x, w, fmt=0))
    e_h = err(run_dconv_bf3(x, w, fmt=1))
    assert e_b < 1e-6, e_b                                   # the same fp32-level error at every scale
    if scale >= 1.0:
        assert e_h < 1e-6, e_h
    else:
        assert 1e-6 < e_h < 2e-4, e_h                        # the floor shows: ~2^-25 / 1e-3 per operand
