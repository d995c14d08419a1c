"""Does a UNet forward read workspace memory that no kernel of the call wrote?  Runs every (GEMM mode, latency mode, size) on a workspace
filled with zeros, with a NaN pattern and with a finite non-zero pattern: the three results must be finite and bit-identical.

    python tools/diag_poison.py [--modes f32,split_bf16,split_f16] [--sizes 1x2050,2x1000,1x77,5x512]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "latent-diffusion-speech_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

PATTERNS = {"zeros": 0x00000000, "nan": 0x7FC07FC0, "ones": 0x3F803C00, "big": 0x7B007B00}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--modes", default="f32,split_bf16,split_f16")
    ap.add_argument("--sizes", default="1x2050,2x1000,1x77,5x512")
    a = ap.parse_args()
    import torch
    from diffusion.unit2mel import Unit2Mel
    from lds import init_weights, native
    m = Unit2Mel(1280, 323, 80).to("cuda").eval()
    unet = m.decoder.denoise_fn
    bad = 0
    for mode in a.modes.split(","):
        unet.set_gemm_mode(mode)
        for lat in (True, False):
            unet.set_latency_mode(lat)
            for sz in a.sizes.split(","):
                B, T = (int(v) for v in sz.split("x"))
                x = torch.from_numpy(init_weights.uniform(f"sz.{B}.{T}", (B, 336, T), 33, -2, 2)).cuda()
                t = torch.from_numpy(np.full((B,), 250.25, dtype=np.float32)).cuda()
                outs = {}
                for name, pat in PATTERNS.items():
                    ws = unet.native().workspace_tensor(B, T, x.device)
                    native.debug_fill(ws, pat)
                    outs[name] = unet(x, t).sample.cpu().numpy()
                ref = outs["zeros"]
                line = f"{mode:10s} latency {int(lat)} B {B} T {T:5d}:"
                for name in PATTERNS:
                    y = outs[name]
                    fin = bool(np.isfinite(y).all())
                    same = bool(np.array_equal(y, ref))
                    d = float(np.abs(np.nan_to_num(y) - ref).max() / np.abs(ref).max())
                    line += f"  {name} finite={fin} same={same} rel={d:.2e}"
                    if not fin or not same:
                        bad += 1
                print(line, flush=True)
    print("cases with a dependence on the workspace's prior contents:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
