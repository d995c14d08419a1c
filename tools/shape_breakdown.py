"""Per-shape time / TFLOP/s of every convolution launch inside one real sampler run (HIP-event profiler, level 2 = names carry shapes),
in the exact-fp32 mode, the split-bf16 mode, or both side by side:

    python tools/shape_breakdown.py [B] [--mode f32|split_bf16|both] [--json out.json]
"""
import argparse
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion-speech_amd"))
import torch  # noqa: E402

from diffusion.unit2mel import Unit2Mel  # noqa: E402
from lds import init_weights, native  # noqa: E402


def profile(m, units, spk, mode):
    m.decoder.denoise_fn.set_gemm_mode(mode)
    m(units, None, spk_id=spk, infer=True, infer_speedup=100, method="dpm-solver")
    torch.cuda.synchronize()
    native.prof_enable(2)
    m(units, None, spk_id=spk, infer=True, infer_speedup=100, method="dpm-solver")     # 10 NFE
    torch.cuda.synchronize()
    prof = native.prof_summary()
    native.prof_enable(0)
    return prof


def shape_key(name):
    m = re.search(r"> (Ci[\d+]+ Co\d+ K[\d+]+ To\d+.*)$", name)
    return m.group(1) if m else name.split("<")[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("B", nargs="?", type=int, default=16)
    ap.add_argument("--mode", default="f32")
    ap.add_argument("--json", default=None)
    ap.add_argument("--latency", action="store_true", help="latency mode (tile / split choices from the actual batch)")
    a = ap.parse_args()
    B, T = a.B, 512
    m = Unit2Mel(1280, 323, 80).to("cuda").eval()
    m.decoder.denoise_fn.set_latency_mode(a.latency)
    units = torch.from_numpy(init_weights.uniform("bench.units", (B, T, 1280), 1, -1.7, 1.7)).cuda()
    spk = torch.ones(B, 1, dtype=torch.int64, device="cuda")
    modes = ["f32", "split_bf16"] if a.mode == "both" else [a.mode]
    profs = {md: profile(m, units, spk, md) for md in modes}
    for md in modes:
        prof = sorted(profs[md], key=lambda r: -r["ms"])
        tot = sum(r["ms"] for r in prof)
        print(f"== {md}: total {tot:.1f} ms for 10 NFE")
        if a.mode != "both":
            for r in prof[:90]:
                tf = r["flops"] / (r["ms"] * 1e-3) / 1e12 if r["flops"] else 0
                print(f"{r['name']:86s} n={r['count']:4d} {r['ms']:8.2f} ms {100 * r['ms'] / tot:5.1f}%  {1e3 * r['ms'] / r['count']:7.1f} us/launch {tf:6.1f} TF")
    if a.mode == "both":
        agg = {}
        for md in modes:
            for r in profs[md]:
                e = agg.setdefault(shape_key(r["name"]), {})
                d = e.setdefault(md, {"ms": 0.0, "n": 0, "flops": 0.0, "cfg": set()})
                d["ms"] += r["ms"]; d["n"] += r["count"]; d["flops"] += r["flops"]; d["cfg"].add(r["name"].split(">")[0] + ">")
        rows = []
        for k, e in agg.items():
            if "f32" in e and "split_bf16" in e:
                f, b = e["f32"], e["split_bf16"]
                rows.append((f["ms"], k, f, b))
        rows.sort(reverse=True)
        print(f"{'shape':44s} {'n':>4s} {'f32 us':>8s} {'bf3 us':>8s} {'ratio':>6s}  {'f32 TF':>7s} {'bf3 TF':>7s}  tiles")
        for _, k, f, b in rows:
            fu, bu = 1e3 * f["ms"] / f["n"], 1e3 * b["ms"] / b["n"]
            ft = f["flops"] / (f["ms"] * 1e-3) / 1e12 if f["flops"] else 0
            bt = b["flops"] / (b["ms"] * 1e-3) / 1e12 if b["flops"] else 0
            print(f"{k:44s} {f['n']:4d} {fu:8.1f} {bu:8.1f} {fu / bu:6.2f}  {ft:7.1f} {bt:7.1f}  {' '.join(sorted(f['cfg']))} | {' '.join(sorted(b['cfg']))}")
        if a.json:
            json.dump({k: {md: {"us_per_launch": 1e3 * e[md]["ms"] / e[md]["n"], "launches": e[md]["n"], "cfg": sorted(e[md]["cfg"])} for md in e}
                       for k, e in agg.items()}, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
