"""Micro-benchmark of conv_dma (K4P, DMA-fed) on the UNet's layer shapes at B=16 / T=512:
    python tools/bench_dconv.py [--cfgs 0,64064642,64064323,...] [--filter qkv]
cfg = BM*1000000 + BN*1000 + BK*10 + NST (0 = the launcher's own choice)."""
import argparse
import ctypes as ct
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion-speech_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from lds import native  # noqa: E402

PEAK = 157.3
SHAPES = {
    "qkv_256@512": (256, 0, 512, 768, 1, dict(v_split=1)),
    "o_256@512": (256, 0, 512, 256, 1, dict(res=True)),
    "ff1_256@512": (256, 0, 512, 2048, 1, dict(epi=1)),
    "ff2_256@512": (1024, 0, 512, 256, 1, dict(res=True)),
    "conv3_256@512": (256, 0, 512, 256, 3, dict(pad=1)),
    "conv3_512->256@512": (512, 0, 512, 256, 3, dict(pad=1)),
    "sc_256+256->256@512": (256, 256, 512, 256, 1, dict()),
    "qkv_384@256": (384, 0, 256, 1152, 1, dict(v_split=1)),
    "ff1_384@256": (384, 0, 256, 3072, 1, dict(epi=1)),
    "ff2_384@256": (1536, 0, 256, 384, 1, dict(res=True)),
    "conv3_384@256": (384, 0, 256, 384, 3, dict(pad=1)),
    "conv3_768->384@256": (768, 0, 256, 384, 3, dict(pad=1)),
    "qkv_512@128": (512, 0, 128, 1536, 1, dict(v_split=1)),
    "o_512@128": (512, 0, 128, 512, 1, dict(res=True)),
    "ff1_512@128": (512, 0, 128, 4096, 1, dict(epi=1)),
    "ff2_512@128": (2048, 0, 128, 512, 1, dict(res=True)),
    "conv3_512@128": (512, 0, 128, 512, 3, dict(pad=1)),
    "conv3_1024->512@128": (1024, 0, 128, 512, 3, dict(pad=1)),
    "conv3_512@64": (512, 0, 64, 512, 3, dict(pad=1)),
    "conv3_1024->512@64": (1024, 0, 64, 512, 3, dict(pad=1)),
    "sc_512+512->512@64": (512, 512, 64, 512, 1, dict()),
    "qkv_512@64": (512, 0, 64, 1536, 1, dict(v_split=1)),
    "o_512@64": (512, 0, 64, 512, 1, dict(res=True)),
    "ff2_512@64": (2560, 0, 64, 512, 1, dict(res=True)),
    "ff2c_512@128": (2560, 0, 128, 512, 1, dict(res=True)),
    "ff2c_384@256": (1920, 0, 256, 384, 1, dict(res=True)),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfgs", default="0")
    ap.add_argument("--filter", default="")
    ap.add_argument("--B", type=int, default=16)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    L = native.lib()
    B = args.B
    rng = np.random.default_rng(0)
    for name, (C1, C2, T, Co, K, kw) in SHAPES.items():
        if args.filter and args.filter not in name:
            continue
        Ci = C1 + C2
        x1 = torch.randn(B, C1, T, device="cuda")
        x2 = torch.randn(B, C2, T, device="cuda") if C2 else None
        w = (rng.standard_normal((Co, Ci, K)) / np.sqrt(Ci * K)).astype(np.float32)
        bias = rng.standard_normal(Co).astype(np.float32)
        Cout = Co // 2 if kw.get("epi") == 1 else Co
        out = torch.empty(B, Cout, T, device="cuda")
        Ck = (Cout // 3) * 2 if kw.get("v_split") else Cout
        res = torch.randn(B, Ck, T, device="cuda") if kw.get("res") else None
        for cfg in [int(t) for t in args.cfgs.split(",")]:
            a = native.DConvTest()
            a.x1, a.x2 = x1.data_ptr(), (x2.data_ptr() if x2 is not None else None)
            a.C1, a.C2, a.T = C1, C2, T
            a.w, a.bias = w.ctypes.data, bias.ctypes.data
            a.Co, a.K, a.stride, a.pad, a.ups = Co, K, 1, kw.get("pad", 0), 0
            a.res = res.data_ptr() if res is not None else None
            a.epilogue, a.plain_out, a.v_split, a.cfg = kw.get("epi", 0), 0, kw.get("v_split", 0), cfg
            ms = ct.c_float()
            cs = ct.create_string_buffer(128)
            rc = L.lds_bench_dconv(ct.byref(a), ct.c_void_p(out.data_ptr()), B, args.iters, ct.byref(ms), cs, 128,
                                   ct.c_void_p(torch.cuda.current_stream().cuda_stream))
            if rc != 0:
                print(f"{name:24s} cfg {cfg}: {L.lds_last_error().decode()[:80]}")
                continue
            fl = 2.0 * B * T * Co * Ci * K
            tf = fl / (ms.value * 1e-3) / 1e12
            print(f"{name:24s} {cs.value.decode():52s} {ms.value * 1e3:8.1f} us {tf:6.1f} TF {100 * tf / PEAK:5.1f}%", flush=True)


if __name__ == "__main__":
    main()
