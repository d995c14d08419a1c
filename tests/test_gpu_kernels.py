"""Per-kernel parity on a real MI355X: the generic convolution (vocoder / front-end path), ConvTranspose, attention's
rescale path and the small utility kernels of liblds against the numpy oracle
(itself pinned to the reference's leaf modules by test_oracle_vs_golden.py).  All calls go
through the C ABI (ctypes).  Tolerances: the kernels compute in exact fp32 (v_mfma_f32_32x32x2
is a k-ordered fmaf chain), so differences against the oracle are summation-order only."""
import ctypes as ct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def relmax(a, b):
    return float(np.abs(a - b).max() / max(1e-30, np.abs(b).max()))


def U(name, shape, lo=-1.0, hi=1.0):
    from lds import init_weights
    return init_weights.uniform("t." + name, shape, 7, lo, hi)


def run_conv(x1, w, bias=None, x2=None, pad=0, dil=1, act=0, slope=0.0, res=None, epi=0, tile=0):
    from lds import native
    L = native.lib()
    B, C1, T = x1.shape
    C2 = 0 if x2 is None else x2.shape[1]
    Co, _, K = w.shape
    a = native.ConvTest()
    dx1 = dev(x1)
    dx2 = dev(x2) if x2 is not None else None
    keep = [np.ascontiguousarray(w, dtype=np.float32)]
    a.x1 = dx1.data_ptr()
    a.x2 = dx2.data_ptr() if dx2 is not None else None
    a.C1, a.C2, a.Tsrc = C1, C2, T
    a.w = keep[0].ctypes.data
    if bias is not None:
        keep.append(np.ascontiguousarray(bias, dtype=np.float32))
        a.bias = keep[-1].ctypes.data
    a.Co, a.K, a.pad, a.dil = Co, K, pad, dil
    a.act_in, a.slope = act, slope
    To = T + 2 * pad - dil * (K - 1)
    dres = dev(res) if res is not None else None
    a.res = dres.data_ptr() if dres is not None else None
    a.epilogue, a.tile = epi, tile
    out = torch.full((B, Co, To), float("nan"), dtype=torch.float32, device="cuda")
    native.check(L.lds_test_conv(ct.byref(a), ct.c_void_p(out.data_ptr()), B, ct.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    return out.cpu().numpy()


def ref_conv(x1, w, bias=None, x2=None, pad=0, dil=1, act=0, slope=0.0, res=None, epi=0):
    from oracle import unet1d
    x = x1 if x2 is None else np.concatenate([x1, x2], axis=1)
    if act == 2:
        x = np.where(x >= 0, x, x * np.float32(slope)).astype(np.float32)
    y = unet1d.conv1d(np.ascontiguousarray(x), w, bias, pad=pad, dil=dil)
    if res is not None:
        y = y + res
    if epi == 2:
        y = np.tanh(y)
    return y.astype(np.float32)


CONV_CASES = {
    # name: (B, C1, C2, T, Co, K, kwargs)
    "1x1_small": (2, 64, 0, 64, 64, 1, {}),
    "1x1_t128": (1, 256, 0, 128, 256, 1, dict(tile=128128)),
    "1x1_t12864": (2, 128, 0, 96, 128, 1, dict(tile=128064)),
    "1x1_res_bias": (2, 96, 0, 80, 192, 1, dict(use_bias=True, use_res=True)),
    "k3_pad": (2, 64, 0, 64, 64, 3, dict(pad=1, use_bias=True)),
    "k3_ragged_T": (2, 80, 0, 50, 128, 3, dict(pad=1, use_bias=True)),
    "k3_odd_T": (1, 32, 0, 37, 64, 3, dict(pad=1)),
    "k3_t128": (1, 128, 0, 256, 128, 3, dict(pad=1, tile=128128, use_bias=True)),
    "k3_concat": (2, 64, 48, 72, 128, 3, dict(pad=1, use_bias=True)),
    "co80_k3": (2, 64, 0, 64, 80, 3, dict(pad=1, use_bias=True)),
    "voc_k7": (1, 80, 0, 40, 128, 7, dict(pad=3, use_bias=True)),
    "voc_k3_d3": (1, 64, 0, 200, 64, 3, dict(pad=3, dil=3, act=2, slope=0.1, use_bias=True)),
    "voc_k7_d5": (1, 64, 0, 300, 64, 7, dict(pad=15, dil=5, act=2, slope=0.1, use_bias=True, use_res=True)),
    "voc_k11_d5": (1, 32, 0, 300, 32, 11, dict(pad=25, dil=5, act=2, slope=0.1, use_bias=True, use_res=True)),
    "voc_k11_d1_c16": (2, 16, 0, 500, 16, 11, dict(pad=5, act=2, slope=0.1, use_bias=True)),
    "voc_k11_64128": (1, 64, 0, 300, 64, 11, dict(pad=15, dil=3, act=2, slope=0.1, tile=64128)),
    "voc_post_tanh": (2, 16, 0, 300, 1, 7, dict(pad=3, act=2, slope=0.01, epi=2, use_bias=True)),
    "voc_post_tanh_odd": (3, 16, 0, 1027, 1, 7, dict(pad=3, act=2, slope=0.01, epi=2, use_bias=True)),      # conv_mono: scalar tail path
    "voc_post_32ch_tiny": (1, 32, 0, 8, 1, 7, dict(pad=3, act=2, slope=0.01, epi=2, use_bias=True)),
    "voc_post_long": (1, 16, 0, 5000, 1, 7, dict(pad=3, act=2, slope=0.01, epi=2)),
    # conv_small (16 channels, v_mfma_f32_16x16x4_f32, register-resident weights): every tap count / dilation, residual, several
    # 512-frame blocks, a length that is not a multiple of 4 (scalar staging path), a short one
    "small_k3_d1": (2, 16, 0, 1300, 16, 3, dict(pad=1, act=2, slope=0.1, use_bias=True, use_res=True)),
    "small_k3_d5": (1, 16, 0, 700, 16, 3, dict(pad=5, dil=5, act=2, slope=0.1, use_bias=True)),
    "small_k7_d3": (2, 16, 0, 1024, 16, 7, dict(pad=9, dil=3, act=2, slope=0.1, use_bias=True, use_res=True)),
    "small_k7_d1_odd": (1, 16, 0, 501, 16, 7, dict(pad=3, act=2, slope=0.1, use_bias=True)),
    "small_k11_d5": (3, 16, 0, 2100, 16, 11, dict(pad=25, dil=5, act=2, slope=0.1, use_bias=True, use_res=True)),
    "small_k11_d3_short": (1, 16, 0, 40, 16, 11, dict(pad=15, dil=3, use_bias=True)),
}


@pytest.mark.parametrize("name", list(CONV_CASES))
def test_conv_gemm(name):
    _need_gpu()
    B, C1, C2, T, Co, K, kw = CONV_CASES[name]
    kw = dict(kw)
    x1 = U(name + ".x1", (B, C1, T), -2, 2)
    x2 = U(name + ".x2", (B, C2, T), -2, 2) if C2 else None
    Ci = C1 + C2
    w = U(name + ".w", (Co, Ci, K), -1, 1) / np.float32(np.sqrt(Ci * K))
    args = dict(x2=x2)
    if kw.pop("use_bias", False):
        args["bias"] = U(name + ".b", (Co,))
    for k in ("pad", "dil", "act", "slope", "epi"):
        if k in kw:
            args[k] = kw[k]
    ref0 = ref_conv(x1, w, **args)
    if kw.pop("use_res", False):
        args["res"] = U(name + ".res", ref0.shape, -1, 1)
    ref = ref_conv(x1, w, **args)
    out = run_conv(x1, w, tile=kw.get("tile", 0), **args)
    assert out.shape == ref.shape
    assert np.isfinite(out).all()
    assert relmax(out, ref) < 2e-5, relmax(out, ref)


def test_attention_large_logits():
    """forces the online-softmax rescale branch: one key dominates late in the sequence"""
    _need_gpu()
    from lds import native
    import ctypes
    B, C, T, heads = 1, 256, 256, 8
    d = C // heads
    qkv = U("attspike", (B, 3 * C, T), -1, 1).copy()
    qkv[:, C:2 * C, 200] *= 12.0      # spike one key column
    qkv[:, :C, 5] *= 9.0              # and one query
    out = torch.full((B, C, T), float("nan"), dtype=torch.float32, device="cuda")
    dq = dev(qkv)
    native.check(native.lib().lds_test_attention_k4p(ctypes.c_void_p(dq.data_ptr()), ctypes.c_void_p(out.data_ptr()), B, C, T, heads,
                                                 ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    q, k, v = [qkv[:, i * C:(i + 1) * C].reshape(B, heads, d, T).astype(np.float64) for i in range(3)]
    s = np.einsum("bhdq,bhdk->bhqk", q, k) / np.sqrt(d)
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s)
    p /= p.sum(-1, keepdims=True)
    ref = np.einsum("bhqk,bhdk->bhdq", p, v).reshape(B, C, T)
    got = out.cpu().numpy()
    assert np.abs(got - ref).max() < 3e-5 * np.abs(ref).max()


@pytest.mark.parametrize("Ci,Co,T,K,s", [(64, 32, 20, 16, 8), (32, 16, 33, 4, 2), (128, 64, 7, 16, 8), (512, 256, 12, 16, 8)])
def test_conv_transpose(Ci, Co, T, K, s):
    _need_gpu()
    import ctypes
    from lds import native
    from oracle import vocoder
    B = 2
    pad = (K - s + 1) // 2
    x = U(f"ct{Ci}.x", (B, Ci, T), -2, 2)
    w = (U(f"ct{Ci}.w", (Ci, Co, K)) / np.float32(np.sqrt(Ci * K / s))).astype(np.float32)
    b = U(f"ct{Ci}.b", (Co,))
    ref = vocoder.conv_transpose1d(vocoder.lrelu(x, 0.1), w, b, s, pad)
    out = torch.full(ref.shape, float("nan"), dtype=torch.float32, device="cuda")
    dx = dev(x)
    native.check(native.lib().lds_test_conv_transpose(ctypes.c_void_p(dx.data_ptr()), ctypes.c_void_p(w.ctypes.data),
                                                      ctypes.c_void_p(b.ctypes.data), ctypes.c_void_p(out.data_ptr()), B, Ci, Co, T,
                                                      K, s, pad, ctypes.c_float(0.1),
                                                      ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.isfinite(got).all()
    assert relmax(got, ref) < 2e-5


def test_transpose_and_axpby():
    _need_gpu()
    from lds import native
    x = U("tr", (3, 45, 70))
    y = native.transpose(dev(x), 2.0).cpu().numpy()
    assert np.array_equal(y, (x.transpose(0, 2, 1) / np.float32(2.0)))
    a, b = U("ax.a", (1000,)), U("ax.b", (1000,))
    z = native.axpby(dev(a), dev(b), 0.25, 3.0).cpu().numpy()
    assert np.allclose(z, 0.25 * a + 3.0 * b, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("C,T,K,dil,mode,B", [(64, 300, 3, 1, 0, 2), (64, 300, 3, 5, 1, 1), (128, 200, 7, 3, 0, 1), (64, 1000, 7, 5, 2, 1), (128, 129, 11, 1, 1, 2),
                                             (64, 700, 11, 5, 0, 1), (256, 96, 11, 3, 2, 1)])
def test_vocoder_resblock_step_dma(C, T, K, dil, mode, B):
    """one residual step of ResBlock1 on the K4P / LDS-DMA vocoder family (dilated k 3 / 7 / 11 windows, 32-frame pads, LeakyReLU /
    dual-store / running-sum epilogues, plain store), ragged last tiles included, against the numpy oracle"""
    _need_gpu()
    from lds import native
    from oracle import unet1d
    x = U(f"vs{C}.{T}.{K}.x", (B, C, T), -2, 2)
    w1 = U(f"vs{C}.{T}.{K}.w1", (C, C, K)) / np.float32(np.sqrt(C * K))
    w2 = U(f"vs{C}.{T}.{K}.w2", (C, C, K)) / np.float32(np.sqrt(C * K))
    b1, b2 = U(f"vs{C}.{K}.b1", (C,), -0.3, 0.3), U(f"vs{C}.{K}.b2", (C,), -0.3, 0.3)
    acc = U(f"vs{C}.{T}.acc", (B, C, T), -1, 1)

    def lrelu(a):
        return np.where(a >= 0, a, a * np.float32(0.1)).astype(np.float32)
    xt = unet1d.conv1d(lrelu(x), w1, b1, pad=(K * dil - dil) // 2, dil=dil)
    y = (unet1d.conv1d(lrelu(xt), w2, b2, pad=(K - 1) // 2) + x).astype(np.float32)
    out = torch.full((B, C, T), float("nan"), dtype=torch.float32, device="cuda")
    out_act = torch.full((B, C, T), float("nan"), dtype=torch.float32, device="cuda")
    dx, dacc = dev(x), dev(acc)
    native.check(native.lib().lds_test_voc_step(ct.c_void_p(dx.data_ptr()), ct.c_void_p(w1.ctypes.data), ct.c_void_p(b1.ctypes.data), ct.c_void_p(w2.ctypes.data),
                                                ct.c_void_p(b2.ctypes.data), C, T, K, dil, mode, ct.c_void_p(dacc.data_ptr()), ct.c_float(3.0),
                                                ct.c_void_p(out.data_ptr()), ct.c_void_p(out_act.data_ptr()), B,
                                                ct.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    if mode == 2:
        y = ((acc + y) / np.float32(3.0)).astype(np.float32)
    assert relmax(out.cpu().numpy(), y) < 2e-5, relmax(out.cpu().numpy(), y)
    if mode == 1:
        assert relmax(out_act.cpu().numpy(), lrelu(y)) < 2e-5


@pytest.mark.parametrize("C,T,K,dil,B,acc,ragged", [(16, 1500, 3, 1, 2, False, False), (16, 1037, 7, 3, 1, True, False), (16, 2048, 11, 5, 2, True, True),
                                                   (16, 130, 11, 1, 1, False, False), (32, 1000, 3, 5, 2, True, True), (32, 777, 7, 5, 1, False, False),
                                                   (32, 1024, 7, 1, 2, True, False), (32, 1200, 11, 3, 2, False, True), (32, 40, 11, 5, 1, True, False),
                                                   (32, 733, 11, 1, 3, True, True), (16, 4, 3, 1, 1, False, False), (32, 8, 11, 5, 2, True, True),
                                                   (16, 5, 7, 3, 1, False, False), (32, 252, 3, 1, 1, False, False), (16, 508, 3, 5, 2, True, False)])
def test_vocoder_resblock_step_fused(C, T, K, dil, B, acc, ragged):
    """one residual step of ResBlock1 at the vocoder's 16- / 32-channel stages as ONE launch (csrc/voc_pair.hip: c1 -> LDS -> c2, frames as the
    MFMA rows) against the numpy oracle: several tiles per utterance, last tiles ragged, T not a multiple of 4 (the scalar-access instantiation),
    T shorter than one tile, the running-sum epilogue, and per-utterance lengths (the intermediate is ZERO beyond an utterance's end, the output too)"""
    _need_gpu()
    from lds import native
    from oracle import unet1d
    tag = f"vp{C}.{T}.{K}.{dil}"
    x = U(tag + ".x", (B, C, T), -2, 2)
    w1 = U(tag + ".w1", (C, C, K)) / np.float32(np.sqrt(C * K))
    w2 = U(tag + ".w2", (C, C, K)) / np.float32(np.sqrt(C * K))
    b1, b2 = U(tag + ".b1", (C,), -0.3, 0.3), U(tag + ".b2", (C,), -0.3, 0.3)
    a = U(tag + ".acc", (B, C, T), -1, 1)
    lens = np.array([T if b == 0 else max(1, (T * (3 + b)) // (5 + b)) for b in range(B)], dtype=np.int32) if ragged else None

    def lrelu(v):
        return np.where(v >= 0, v, v * np.float32(0.1)).astype(np.float32)
    ref = np.zeros_like(x)
    for b in range(B):
        n = int(lens[b]) if ragged else T
        xb = x[b:b + 1, :, :n].copy()
        xt = unet1d.conv1d(lrelu(xb), w1, b1, pad=(K * dil - dil) // 2, dil=dil)
        y = (unet1d.conv1d(lrelu(xt), w2, b2, pad=(K - 1) // 2) + xb).astype(np.float32)
        if acc:
            y = ((a[b:b + 1, :, :n] + y) / np.float32(3.0)).astype(np.float32)
        ref[b, :, :n] = y[0]
    xin = x.copy()
    if ragged:
        for b in range(B):
            xin[b, :, lens[b]:] = 0          # (the producer of a ragged batch writes zeros beyond each utterance)
    out = torch.full((B, C, T), float("nan"), dtype=torch.float32, device="cuda")
    dx, dacc = dev(xin), dev(a)
    native.check(native.lib().lds_test_voc_pair(ct.c_void_p(dx.data_ptr()), ct.c_void_p(w1.ctypes.data), ct.c_void_p(b1.ctypes.data), ct.c_void_p(w2.ctypes.data),
                                                ct.c_void_p(b2.ctypes.data), C, T, K, dil, ct.c_void_p(dacc.data_ptr() if acc else None),
                                                ct.c_float(3.0 if acc else 1.0), ct.c_void_p(lens.ctypes.data if ragged else None),
                                                ct.c_void_p(out.data_ptr()), B, ct.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.isfinite(got).all()
    if ragged:
        for b in range(B):
            assert not got[b, :, lens[b]:].any(), "frames beyond an utterance's length must be written as zeros"
    assert relmax(got, ref) < 2e-5, relmax(got, ref)


def test_vocoder_fused_steps_equal_two_launch_steps():
    """the whole decode with the narrow stages' residual steps fused equals the decode with two launches per step (same summation order at
    16 channels: bit-equal there; 32 channels used conv_gemm's order: 1e-6)"""
    _need_gpu()
    from encoder.hifi_vaegan.hifi_vaegan import Hifi_VAEGAN
    from lds import arch, init_weights, native
    h = arch.SYNTHETIC_VOCODER_H
    state = init_weights.init_state(arch.generator_param_shapes(h), 0)
    voc = Hifi_VAEGAN(None, device="cuda", h=h, state=state)
    mel = torch.from_numpy(U("vf.mel", (2, 37, 80), -4, 1)).cuda()
    try:
        native.check(native.lib().lds_debug_set_voc_pair(0))
        two = voc(mel).cpu().numpy()
    finally:
        native.check(native.lib().lds_debug_set_voc_pair(1))
    one = voc(mel).cpu().numpy()
    assert np.isfinite(one).all() and relmax(one, two) < 2e-6, relmax(one, two)
