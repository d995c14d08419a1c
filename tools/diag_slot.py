"""Which workspace slot does a forward read before writing?  Fills the whole workspace with zeros, then ONE slot of the plan with a finite
non-zero pattern, and compares the result with the all-zero run (include/lds_test.h lds_debug_unet_plan / lds_debug_fill_u32).

    python tools/diag_slot.py --mode split_f16 --latency 1 --B 1 --T 2050"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "latent-diffusion-speech_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="split_f16")
    ap.add_argument("--latency", type=int, default=1)
    ap.add_argument("--B", type=int, default=1)
    ap.add_argument("--T", type=int, default=2050)
    ap.add_argument("--pattern", default="3f803c00")
    a = ap.parse_args()
    import torch
    from diffusion.unit2mel import Unit2Mel
    from lds import init_weights, native
    B, T = a.B, a.T
    m = Unit2Mel(1280, 323, 80).to("cuda").eval()
    unet = m.decoder.denoise_fn
    unet.set_gemm_mode(a.mode)
    unet.set_latency_mode(bool(a.latency))
    x = torch.from_numpy(init_weights.uniform(f"sz.{B}.{T}", (B, 336, T), 33, -2, 2)).cuda()
    t = torch.from_numpy(np.full((B,), 250.25, dtype=np.float32)).cuda()
    nat = unet.native()
    ws = nat.workspace_tensor(B, T, x.device)
    plan = nat.plan(B, T)
    pat = int(a.pattern, 16)
    native.debug_fill(ws, 0)
    ref = unet(x, t).sample.clone()
    hits = []
    for name, off, nb in plan:
        native.debug_fill(ws, 0)
        native.debug_fill(ws[off:off + nb], pat)
        y = unet(x, t).sample
        if not torch.equal(y, ref):
            d = float((y - ref).abs().max() / ref.abs().max())
            hits.append(name)
            print(f"slot {name:16s} offset {off:12d} bytes {nb:10d}: result changes by {d:.3e}", flush=True)
    # second level: halves of the first hit
    for name, off, nb in plan:
        if name not in hits:
            continue
        lo, hi = off, off + nb
        while hi - lo > 4096:
            mid = (lo + (hi - lo) // 2) & ~255
            native.debug_fill(ws, 0)
            native.debug_fill(ws[lo:mid], pat)
            if not torch.equal(unet(x, t).sample, ref):
                hi = mid
            else:
                native.debug_fill(ws, 0)
                native.debug_fill(ws[mid:hi], pat)
                if not torch.equal(unet(x, t).sample, ref):
                    lo = mid
                else:
                    print(f"  {name}: neither half of [{lo - off}, {hi - off}) alone changes the result")
                    break
        print(f"  {name}: a sensitive range is bytes [{lo - off}, {hi - off}) of the slot ({nb} bytes)")
    print("slots the result depends on:", hits)


if __name__ == "__main__":
    main()
