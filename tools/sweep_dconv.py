"""Fixed-cost decomposition of conv_dma: time of a 1x1 conv (Co=256, T=512) against the reduction length Ci and the batch B.
The intercept at Ci -> 0 is the per-launch cost that is not main loop (dispatch, first-tile latency, epilogue, drain)."""
import ctypes as ct
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion-speech_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from lds import native  # noqa: E402


def run(L, B, Ci, Co, T, K, cfg, res, iters=30):
    rng = np.random.default_rng(0)
    x1 = torch.randn(B, Ci, T, device="cuda")
    w = (rng.standard_normal((Co, Ci, K)) / np.sqrt(Ci * K)).astype(np.float32)
    bias = rng.standard_normal(Co).astype(np.float32)
    out = torch.empty(B, Co, T, device="cuda")
    r = torch.randn(B, Co, T, device="cuda") if res else None
    a = native.DConvTest()
    a.x1, a.x2, a.C1, a.C2, a.T = x1.data_ptr(), None, Ci, 0, T
    a.w, a.bias = w.ctypes.data, bias.ctypes.data
    a.Co, a.K, a.stride, a.pad, a.ups = Co, K, 1, K // 2, 0
    a.res = r.data_ptr() if res else None
    a.epilogue, a.plain_out, a.v_split, a.cfg = 0, 0, 0, cfg
    ms = ct.c_float()
    cs = ct.create_string_buffer(128)
    rc = L.lds_bench_dconv(ct.byref(a), ct.c_void_p(out.data_ptr()), B, iters, ct.byref(ms), cs, 128,
                           ct.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, L.lds_last_error()
    return ms.value * 1e3, cs.value.decode()


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfgs", default="64064642,64064323")
    ap.add_argument("--Bs", default="4,8,16,32,64")
    ap.add_argument("--Co", type=int, default=256)
    ap.add_argument("--K", type=int, default=1)
    args = ap.parse_args()
    L = native.lib()
    cis = (64, 128, 256, 512, 1024, 2048)
    for cfg in [int(c) for c in args.cfgs.split(",")]:
        print("cfg", cfg)
        for B in [int(b) for b in args.Bs.split(",")]:
            row = [run(L, B, Ci, args.Co, 512, args.K, cfg, False)[0] for Ci in cis]
            fl = 2.0 * B * 512 * args.Co * args.K * (cis[-1] - cis[-2])
            slope_tf = fl / ((row[-1] - row[-2]) * 1e-6) / 1e12
            print(f"  B={B:3d}  " + "  ".join(f"Ci{c}:{u:6.1f}us" for c, u in zip(cis, row)) + f"   main-loop slope {slope_tf:6.1f} TF", flush=True)


if __name__ == "__main__":
    main()
