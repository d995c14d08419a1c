"""What does a launch cost when the previous launch ran OTHER code?  The same 1x1 convolution (o_256@512: 13 us of work) launched 240 times
back to back, the launches rotating through n = 1, 2, 4, 8 instantiations of conv_dma (tile shape / K-step / ring depth variants of the
same operator: 14-26 KB of code each; the instruction cache is 64 KB per CU pair):

    python tools/bench_icache.py [--B 16]
"""
import argparse
import ctypes as ct
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion-speech_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from lds import native  # noqa: E402

CFGS = [64064322, 64064163, 64064323, 64064162, 128064322, 128064163, 64064642, 128064162]      # BM*1e6 + BN*1e3 + BK*10 + NST


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=16)
    ap.add_argument("--iters", type=int, default=240)
    args = ap.parse_args()
    L = native.lib()
    B, C, T = args.B, 256, 512
    rng = np.random.default_rng(0)
    x1 = torch.randn(B, C, T, device="cuda")
    w = (rng.standard_normal((C, C, 1)) / np.sqrt(C)).astype(np.float32)
    bias = rng.standard_normal(C).astype(np.float32)
    out = torch.empty(B, C, T, device="cuda")
    res = torch.randn(B, C, T, device="cuda")
    a = native.DConvTest()
    a.x1, a.x2 = x1.data_ptr(), None
    a.C1, a.C2, a.T = C, 0, T
    a.w, a.bias = w.ctypes.data, bias.ctypes.data
    a.Co, a.K, a.stride, a.pad, a.ups = C, 1, 1, 0, 0
    a.res = res.data_ptr()
    a.epilogue, a.plain_out, a.v_split, a.cfg = 0, 0, 0, CFGS[0]
    st = ct.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run(cfgs):
        arr = (ct.c_int * len(cfgs))(*cfgs)
        ms = ct.c_float()
        native.check(L.lds_bench_dconv_alt(ct.byref(a), ct.c_void_p(out.data_ptr()), B, args.iters, arr, len(cfgs), ct.byref(ms), st))
        return ms.value * 1e3
    single = {c: run([c]) for c in CFGS}
    print("each instantiation alone (us per launch):", {c: round(v, 2) for c, v in single.items()})
    for n in (1, 2, 4, 8):
        cfgs = CFGS[:n]
        alone = sum(single[c] for c in cfgs) / n
        print(f"rotating through {n}: {run(cfgs):6.2f} us per launch (mean of the same {n} run alone: {alone:6.2f})", flush=True)


if __name__ == "__main__":
    main()
