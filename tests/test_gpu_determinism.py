"""The result of a call is a function of its inputs: not of what the caller's workspace held before, not of which workgroups happen to share a CU.

Round 3's driver run went red on `test_latency_unet_other_sizes[split_f16-1-2050]` with an error that differed between two boxes.  The cause
 was the GroupNorm fold's group statistics coming out differently, in one kernel instantiation, in workgroups that shared
their CU with another workgroup.  These tests keep both ways of seeing it: every GroupNorm-fold kernel launched many times on the same inputs
with more workgroups than CUs (all repetitions bit-identical, and right), and every UNet mode run on differently poisoned workspaces."""
import ctypes as ct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

FMT = {"f32": -1, "split_bf16": 0, "split_f16": 1}


def relmax(a, b):
    return float(np.abs(a - b).max() / max(1e-30, np.abs(b).max()))


# (normalised channels, frames, batch, tile_batch): grids of 264 .. 504 workgroups of the 32 x 64 tile (two or three per CU on some CUs) for each
# K-step the fold's launcher can pick (group sizes 32 / 48 / 64), the nominal-batch tiles (tile_batch 0), and a one-workgroup-per-CU control
SHAPES = [(256, 2114, 1, 1), (256, 1000, 3, 3), (256, 4000, 1, 1), (384, 2000, 1, 1), (512, 700, 2, 2), (256, 512, 16, 0), (384, 256, 16, 0), (256, 2048, 1, 1)]


@pytest.mark.parametrize("Cm,T,B,tile_batch", SHAPES)
@pytest.mark.parametrize("fmt", ["f32", "split_bf16", "split_f16"])
def test_gn_fold_repeatable_and_right(fmt, Cm, T, B, tile_batch, record_margin):
    from lds import init_weights, native
    from oracle import unet1d
    U = lambda n, s, lo=-1.0, hi=1.0: init_weights.uniform(f"det.{Cm}.{T}.{n}", s, 7, lo, hi)
    C, Co, reps = 64, Cm, 24
    x = U("x", (B, C, T), -2, 2)
    w1 = (U("w1", (Cm, C)) / np.float32(np.sqrt(C))).astype(np.float32)
    b1 = U("b1", (Cm,), 0.5, 1.5)
    g, be = U("g", (Cm,), 0.5, 1.5), U("b", (Cm,), -0.5, 0.5)
    w2 = (U("w2", (Co, Cm)) / np.float32(np.sqrt(Cm))).astype(np.float32)
    b2 = U("b2", (Co,), -0.5, 0.5)
    dx = torch.from_numpy(x).cuda()
    mid = torch.full((B, Cm, T), float("nan"), dtype=torch.float32, device="cuda")
    out = torch.full((reps, B, Co, T), float("nan"), dtype=torch.float32, device="cuda")
    P = lambda v: ct.c_void_p(v.ctypes.data)
    native.check(native.lib().lds_test_gn_fold_split(ct.c_void_p(dx.data_ptr()), P(w1), P(b1), P(g), P(be), ct.c_float(1e-6), 8, P(w2), P(b2),
                                                     ct.c_void_p(mid.data_ptr()), ct.c_void_p(out.data_ptr()), B, C, Cm, Co, T, 0, tile_batch, FMT[fmt], reps,
                                                     ct.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    differing = [r for r in range(1, reps) if not torch.equal(out[r], out[0])]
    assert not differing, f"repetitions {differing} of {reps} differ from the first on identical inputs"
    rmid = unet1d.conv1d(x, w1[:, :, None], b1)
    ref = unet1d.conv1d(unet1d.group_norm(rmid, g, be, 8, 1e-6), w2[:, :, None], b2)
    record_margin(relmax(out[0].cpu().numpy(), ref), 2e-5)


# (Ci, Co, K, T, B): the latency mode's deep reductions -- T/8 and T/4 level shapes of one and two 512-frame utterances (S = 16 .. 2), a k 3
# resnet convolution, the 2560-deep ff.net.2 + proj_out
CLUSTER = [(512, 512, 1, 64, 1), (512, 512, 3, 64, 1), (2560, 512, 1, 64, 1), (512, 512, 1, 128, 2), (384, 384, 3, 256, 1), (1024, 512, 3, 64, 2), (256, 256, 1, 512, 1)]


@pytest.mark.parametrize("Ci,Co,K,T,B", CLUSTER)
@pytest.mark.parametrize("fmt", ["f32", "split_f16", "split_bf16"])
def test_cluster_join_under_uneven_load(fmt, Ci, Co, K, T, B):
    """the cluster split-K hand-off the way MI355X_MICROARCH.md asks for hand-offs to be tested: the chip unevenly loaded (a second stream streams
    HBM meanwhile), the consumer L1-warm (back-to-back launches reuse the same scratch slots, alternating between two inputs so that a stale
    partial is the OTHER input's), every word of every joined tile compared -- with the one-workgroup-per-tile result of the same tile shape, and bit
    for bit with the first launch on the same input"""
    from lds import init_weights, native
    U = lambda n, s, lo=-1.0, hi=1.0: init_weights.uniform(f"cj.{Ci}.{K}.{T}.{n}", s, 9, lo, hi)
    reps = 60
    xa, xb = U("xa", (B, Ci, T), -2, 2), U("xb", (B, Ci, T), -2, 2)
    w = (U("w", (Co, Ci, K)) / np.float32(np.sqrt(Ci * K))).astype(np.float32)
    bias = U("b", (Co,), -0.5, 0.5)
    da, db = torch.from_numpy(xa).cuda(), torch.from_numpy(xb).cuda()
    out = torch.full((reps, B, Co, T), float("nan"), dtype=torch.float32, device="cuda")
    ra, rb = torch.empty((B, Co, T), dtype=torch.float32, device="cuda"), torch.empty((B, Co, T), dtype=torch.float32, device="cuda")
    big = torch.zeros(256 << 20, dtype=torch.float32, device="cuda")      # 1 GiB, streamed on a second stream by a second host thread while the launches run
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    import threading
    stop, passes = threading.Event(), [0]

    def load():
        with torch.cuda.stream(side):
            while not stop.is_set() and passes[0] < 20000:
                for _ in range(4):
                    big.add_(1.0)
                side.synchronize()      # (bounds the queue: at most four passes, ~2 ms, are ever pending)
                passes[0] += 4

    th = threading.Thread(target=load)
    th.start()
    while passes[0] < 8:      # the load is running before the entry point starts
        pass
    before = passes[0]
    cfg = ct.create_string_buffer(128)
    P = lambda v: ct.c_void_p(v.ctypes.data)
    try:
        native.check(native.lib().lds_test_cluster_join(ct.c_void_p(da.data_ptr()), ct.c_void_p(db.data_ptr()), P(w), P(bias), Ci, Co, K, T, B, FMT[fmt], reps,
                                                        ct.c_void_p(out.data_ptr()), ct.c_void_p(ra.data_ptr()), ct.c_void_p(rb.data_ptr()), cfg, ct.c_size_t(len(cfg)),
                                                        ct.c_void_p(torch.cuda.current_stream().cuda_stream)))
    finally:
        during = passes[0] - before
        stop.set()
        th.join()
    torch.cuda.synchronize()
    assert " KS" in cfg.value.decode(), f"no cluster split was chosen: {cfg.value.decode()}"
    assert during >= 4, "the second stream did not stream while the launches under test ran"
    refs = (ra.cpu().numpy(), rb.cpu().numpy())
    o = out.cpu().numpy()
    for r in range(reps):
        assert relmax(o[r], refs[r & 1]) < 2e-5, (r, cfg.value.decode(), relmax(o[r], refs[r & 1]))
        assert np.array_equal(o[r], o[r & 1]), f"launch {r} differs from launch {r & 1} on the same input ({cfg.value.decode()})"


PATTERNS = {"zeros": 0x00000000, "nan": 0x7FC07FC0, "ones": 0x3F803C00, "big": 0x7B007B00}


@pytest.fixture(scope="module")
def unet_any():
    from diffusion.unit2mel import Unit2Mel
    m = Unit2Mel(1280, 323, 80).to("cuda").eval()
    return m.decoder.denoise_fn


@pytest.mark.parametrize("B,T", [(1, 2050), (2, 1000), (1, 77), (5, 512)])
@pytest.mark.parametrize("latency", [False, True])
@pytest.mark.parametrize("mode", ["f32", "split_f16"])
def test_unet_poisoned_workspace(unet_any, mode, latency, B, T):
    """the whole caller workspace filled with zeros / a NaN pattern (fp32, fp16 and bf16 alike) / finite patterns before the call: finite and
    bit-identical results -- a kernel that reads a slot no kernel of the call wrote, or whose result depends on timing, shows here"""
    from lds import init_weights, native
    unet = unet_any
    unet.set_gemm_mode(mode)
    unet.set_latency_mode(latency)
    try:
        x = torch.from_numpy(init_weights.uniform(f"sz.{B}.{T}", (B, 336, T), 33, -2, 2)).cuda()
        t = torch.from_numpy(np.full((B,), 250.25, dtype=np.float32)).cuda()
        outs = {}
        for name, pat in PATTERNS.items():
            native.debug_fill(unet.native().workspace_tensor(B, T, x.device), pat)
            outs[name] = unet(x, t).sample.clone()
        for name, y in outs.items():
            assert torch.isfinite(y).all(), f"workspace fill {name}: non-finite output"
            assert torch.equal(y, outs["zeros"]), f"workspace fill {name}: result differs from the zero-filled run by {relmax(y.cpu().numpy(), outs['zeros'].cpu().numpy()):.2e}"
    finally:
        unet.set_latency_mode(False)
        unet.set_gemm_mode("f32")


def test_unet_poisoned_workspace_bench_size(unet_any):
    """the bench's own shape (16 utterances x 512 frames, exact fp32, default mode): every CU holds several workgroups of every launch"""
    from lds import init_weights, native
    unet = unet_any
    B, T = 16, 512
    x = torch.from_numpy(init_weights.uniform("poison.bench", (B, 336, T), 34, -2, 2)).cuda()
    t = torch.from_numpy(np.linspace(3.5, 990.25, B).astype(np.float32)).cuda()
    outs = []
    for pat in (0, 0x7FC07FC0, 0x3F803C00, 0):
        native.debug_fill(unet.native().workspace_tensor(B, T, x.device), pat)
        outs.append(unet(x, t).sample.clone())
    assert all(torch.isfinite(o).all() for o in outs)
    assert all(torch.equal(o, outs[0]) for o in outs[1:])


@pytest.mark.parametrize("method,speedup", [("dpm-solver", 250), ("unipc", 250)])
def test_sampler_poisoned_workspace(method, speedup):
    """the sampler's own scratch (history, predicted points, time embeddings) under the same treatment, latency mode x split-fp16 and the default"""
    from diffusion.unit2mel import Unit2Mel
    from lds import init_weights, native
    m = Unit2Mel(1280, 323, 80).to("cuda").eval()
    gd = m.decoder
    B, T = 1, 130
    cond = torch.from_numpy(init_weights.uniform("poison.cond", (B, T, 256), 5, -1, 1)).cuda()
    xT = torch.from_numpy(init_weights.uniform("poison.xT", (B, 1, 80, T), 5, -1.7, 1.7)).cuda()
    real = torch.randn
    for mode, lat in (("f32", False), ("split_f16", True)):
        gd.denoise_fn.set_gemm_mode(mode)
        gd.denoise_fn.set_latency_mode(lat)
        outs = []
        for pat in (0, 0x7FC07FC0, 0x3F803C00):
            native.debug_fill(gd.denoise_fn.native().workspace_tensor(B, T, cond.device, sampler=True), pat)
            torch.randn = lambda *a, **k: xT.clone()
            try:
                outs.append(gd(cond, infer=True, infer_speedup=speedup, method=method).clone())
            finally:
                torch.randn = real
        assert all(torch.isfinite(o).all() for o in outs)
        assert torch.equal(outs[1], outs[0]) and torch.equal(outs[2], outs[0])


def test_mode_switch_refused_while_a_call_is_in_progress():
    """include/lds.h: the two mode switches are the only mutable state of a denoiser handle; while another thread is inside a sampler call of the
    handle they return LDS_EBUSY instead of changing the plan under it (and work again afterwards)"""
    import threading
    from diffusion.unit2mel import Unit2Mel
    from lds import init_weights
    m = Unit2Mel(1280, 323, 80).to("cuda").eval()
    gd = m.decoder
    cond = torch.from_numpy(init_weights.uniform("busy.cond", (1, 256, 256), 5, -1, 1)).cuda()
    gd(cond, infer=True, infer_speedup=500, method="dpm-solver")      # creates and packs the native handle
    torch.cuda.synchronize()
    nat = gd.denoise_fn.native()
    refused, done = [], threading.Event()

    def run():
        try:
            for _ in range(3):
                gd(cond, infer=True, infer_speedup=10, method="dpm-solver")      # 100 evaluations: ~25 k launches enqueued per call
        finally:
            done.set()

    th = threading.Thread(target=run)
    th.start()
    while not done.is_set():
        try:
            nat.set_latency_mode(False)      # (the value it already has: a refused or an accepted call changes nothing)
        except RuntimeError as e:
            refused.append(str(e))
    th.join()
    torch.cuda.synchronize()
    assert refused and "in progress" in refused[0], "no switch was refused while the sampler call was enqueueing"
    nat.set_latency_mode(True)
    nat.set_latency_mode(False)


@pytest.mark.gpu
def test_vocoder_decode_repeatable():
    """the decode with the fused residual steps of the narrow stages (csrc/voc_pair.hip: persistent workgroups, two LDS tiles handed from
    phase to phase and from tile to tile behind workgroup barriers) is a function of its inputs: five decodes of one batch, bit for bit,
    with a second stream streaming HBM beside the last three"""
    import threading
    import numpy as np
    import torch
    from encoder.hifi_vaegan.hifi_vaegan import Hifi_VAEGAN
    from lds import arch, init_weights
    h = arch.SYNTHETIC_VOCODER_H
    voc = Hifi_VAEGAN(None, device="cuda", h=h, state=init_weights.init_state(arch.generator_param_shapes(h), 0))
    mel = torch.from_numpy(init_weights.uniform("det.voc.mel", (3, 70, 80), 9, -4, 1)).cuda()
    ref = voc(mel).cpu().numpy()
    assert np.isfinite(ref).all()
    for _ in range(2):
        assert np.array_equal(voc(mel).cpu().numpy(), ref)
    stop = threading.Event()
    side = torch.cuda.Stream()
    big = torch.empty(64 << 20, dtype=torch.float32, device="cuda")

    def load():
        with torch.cuda.stream(side):
            while not stop.is_set():
                big.add_(1.0)
                side.synchronize()

    th = threading.Thread(target=load)
    th.start()
    try:
        for _ in range(3):
            assert np.array_equal(voc(mel).cpu().numpy(), ref)
    finally:
        stop.set()
        th.join()


def test_experiment_switches_do_not_change_results():
    """the experiment switch of include/lds_test.h that only adds launches -- a convolution's weights read by a small launch before it
    (lds_debug_set_touch_weights) -- leaves a UNet forward bit for bit as it was, and restores cleanly"""
    from diffusion.unit2mel import Unit2Mel
    from lds import init_weights, native
    m = Unit2Mel(1280, 323, 80).to("cuda").eval()
    unet = m.decoder.denoise_fn
    x = torch.from_numpy(init_weights.uniform("sw.x", (2, 336, 96), 33, -2, 2)).cuda()
    t = torch.from_numpy(np.array([250.25, 40.5], dtype=np.float32)).cuda()
    ref = unet(x, t).sample.clone()
    try:
        native.check(native.lib().lds_debug_set_touch_weights(1))
        assert torch.equal(unet(x, t).sample, ref)
    finally:
        native.check(native.lib().lds_debug_set_touch_weights(0))
    assert torch.equal(unet(x, t).sample, ref)
