"""A/B on one box, one process: the sampler step with a library switch on (default) and off, in the exact-fp32 and the split-fp16 GEMM
modes, default and latency mode.  Switch: gn_fold (the transformer blocks' GroupNorm folded into proj_in vs its own pass).

    python tools/ab_gn_fold.py [gn_fold]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion-speech_amd"))
import torch  # noqa: E402

from diffusion.unit2mel import Unit2Mel  # noqa: E402
from lds import init_weights, native  # noqa: E402

SWITCH = sys.argv[1] if len(sys.argv) > 1 else "gn_fold"
setter = {"gn_fold": native.lib().lds_debug_set_gn_fold}[SWITCH]
T = 512
m = Unit2Mel(1280, 323, 80).to("cuda").eval()
unet = m.decoder.denoise_fn


def timeit(B, reps=3):
    units = torch.from_numpy(init_weights.uniform("bench.units", (B, T, 1280), 1, -1.7, 1.7)).cuda()
    spk = torch.ones(B, 1, dtype=torch.int64, device="cuda")

    def call():
        return m(units, None, spk_id=spk, infer=True, infer_speedup=20, method="dpm-solver")
    call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for mode in ("f32", "split_f16"):
    unet.set_gemm_mode(mode)
    for B, lat in ((16, False), (1, True)):
        unet.set_latency_mode(lat)
        res = {}
        for fold in (1, 0, 1, 0):
            setter(fold)
            res.setdefault(fold, []).append(timeit(B))
        setter(1)
        print(f"{SWITCH} {mode:10s} B={B:2d} latency_mode={int(lat)}: on {min(res[1]):7.2f} ms, off {min(res[0]):7.2f} ms  ({res})", flush=True)
