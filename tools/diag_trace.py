"""Stage-by-stage comparison of two UNet forwards of the same inputs (include/lds_test.h lds_debug_trace): default mode against latency
mode, or one mode against itself on differently filled workspaces.  Names the first stage whose outputs differ by more than a threshold.

    python tools/diag_trace.py --mode split_f16 --B 1 --T 2050            # latency vs default, both vs the oracle
    python tools/diag_trace.py --mode f32 --B 2 --T 1000 --no-oracle

Reference op set: /root/reference/diffusion/unet1d/unet_1d_condition.py:743-1036 (what the stages are)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "latent-diffusion-speech_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def decode(name, raw, B):
    from lds import native
    return native.debug_trace_decode(name, raw, B)


def traced_forward(unet, x, t, B, fill=None):
    from lds import native
    if fill is not None:      # the call's workspace starts from this 32-bit pattern
        native.debug_fill(unet.native().workspace_tensor(B, x.shape[2], x.device), fill)
    native.debug_trace(True)
    try:
        y = unet(x, t).sample
    finally:
        native.debug_trace(False)
    return y, [decode(n, r, B) for n, r in native.debug_trace_records()]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="split_f16")
    ap.add_argument("--B", type=int, default=1)
    ap.add_argument("--T", type=int, default=2050)
    ap.add_argument("--no-oracle", action="store_true")
    ap.add_argument("--thresh", type=float, default=2e-5)
    ap.add_argument("--fills", default="", help="two hex patterns a,b: compare ONE mode (--latency 0/1) on workspaces filled with a and with b")
    ap.add_argument("--latency", type=int, default=1)
    a = ap.parse_args()
    import torch
    from diffusion.unit2mel import Unit2Mel
    from lds import arch, init_weights, native
    B, T = a.B, a.T
    m = Unit2Mel(1280, 323, 80).to("cuda").eval()
    unet = m.decoder.denoise_fn
    unet.set_gemm_mode(a.mode)
    x = torch.from_numpy(init_weights.uniform(f"sz.{B}.{T}", (B, 336, T), 33, -2, 2)).cuda()
    t = torch.from_numpy(np.full((B,), 250.25, dtype=np.float32)).cuda()

    def rel(p, q):
        return float(np.abs(p - q).max() / max(1e-30, np.abs(q).max()))

    if a.fills:
        fa, fb = (int(v, 16) for v in a.fills.split(","))
        unet.set_latency_mode(bool(a.latency))
        ya, ra = traced_forward(unet, x, t, B, fa)
        yb, rb = traced_forward(unet, x, t, B, fb)
        print(f"mode {a.mode} latency {a.latency} B {B} T {T}: fill {fa:#x} vs fill {fb:#x}: {rel(ya.cpu().numpy(), yb.cpu().numpy()):.3e}")
        first = None
        for (n, p_), (_, q_) in zip(ra, rb):
            same = np.array_equal(p_, q_, equal_nan=True)
            if not same:
                bad = np.argwhere(~((p_ == q_) | (np.isnan(p_) & np.isnan(q_))))
                first = first or n
                print(f"{n:28s} DIFFERS at {len(bad)} of {p_.size} elements; first {bad[:4].tolist()} last {bad[-2:].tolist()}  max|a-b| {np.nanmax(np.abs(p_ - q_)):.3e} absmax {np.nanmax(np.abs(q_)):.3e} nan {int(np.isnan(p_).sum())}/{int(np.isnan(q_).sum())}")
                if first == n and p_.ndim == 3:
                    ch = np.unique(bad[:, 1]); fr = np.unique(bad[:, 2])
                    print("    channels", ch[:16].tolist(), "... frames", fr[:16].tolist(), "...", fr[-4:].tolist())
        print("first stage that depends on the fill:", first)
        return
    unet.set_latency_mode(True)
    lat, rec_lat = traced_forward(unet, x, t, B)
    lat2, rec_lat2 = traced_forward(unet, x, t, B)
    unet.set_latency_mode(False)
    base, rec_base = traced_forward(unet, x, t, B)
    lat, lat2, base = lat.cpu().numpy(), lat2.cpu().numpy(), base.cpu().numpy()
    print(f"mode {a.mode} B {B} T {T}: latency vs default {rel(lat, base):.3e}; latency run 1 vs run 2 {rel(lat, lat2):.3e} "
          f"(bit-equal {np.array_equal(lat, lat2)}); finite {np.isfinite(lat).all()} {np.isfinite(base).all()}")
    if not a.no_oracle:
        from oracle import unet1d
        cfg = arch.unet_config()
        w = init_weights.init_state(arch.unet_param_shapes(cfg), 0)
        ref = unet1d.unet_forward(w, cfg, arch.unet_blocks(cfg), x.cpu().numpy(), t.cpu().numpy())
        print(f"vs oracle: latency {rel(lat, ref):.3e}  default {rel(base, ref):.3e}")
    assert [n for n, _ in rec_lat] == [n for n, _ in rec_base], "the two modes trace different stages"
    first = None
    for (n, p), (_, q), (_, p2) in zip(rec_lat, rec_base, rec_lat2):
        if p.shape != q.shape:
            print(f"{n:28s} shapes differ {p.shape} {q.shape}")
            continue
        fin = np.isfinite(p).all() and np.isfinite(q).all()
        r = rel(p, q) if fin else float("nan")
        r2 = rel(p, p2) if fin else float("nan")
        flag = ""
        if not fin or r > a.thresh:
            flag = " <==" if first is None else " <"
            if first is None:
                first = n
        if r2 != 0.0:
            flag += " [latency runs differ %.2e]" % r2
        print(f"{n:28s} lat-vs-base {r:.3e}  absmax {np.abs(q).max():.3e}{flag}")
    print("first stage over the threshold:", first)


if __name__ == "__main__":
    main()
