"""Go / no-go probe for the fp32-equivalent split-bf16 GEMM (csrc/conv_bf3.hip) against the exact-fp32 MFMA kernel
(csrc/conv_dma.hip) on the UNet's heaviest layer shapes at B = 16:

    python tools/split_bf16_probe.py [--out gpurun_out/split_bf16_probe.json] [--iters 30]

For every shape: the same random operands go through both kernels (C-ABI single-op entry points); the reference is the same
convolution in fp64 (torch on the device, checked against a numpy fp64 product on a sample).  Reported per kernel: launch time,
TFLOP/s, and max / RMS error relative to the RMS of the exact result.  A K sweep at fixed M, N separates the K loop's rate from
the per-launch fixed cost (slope of time over K).  Gates: error <= 2x the exact-fp32 kernel's on every shape,
>= 1.5x on the K loop.
"""
import argparse
import ctypes as ct
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion-speech_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from lds import native  # noqa: E402

# name: (C1, C2, T, Co, K, pad, bf3 tile codes to sweep (0 = the launcher's choice))
SHAPES = {
    "ff1_256@512": (256, 0, 512, 2048, 1, 0, [0, 128128322, 128128323, 128128163, 128128164, 128064322, 128064642]),
    "ff2_proj_out_256@512": (1024, 256, 512, 256, 1, 0, [0, 64064642, 64064643, 64064323, 64064324, 128064322, 128064642, 128128322]),
    "conv3_512@128": (512, 0, 128, 512, 3, 1, [0, 64064322, 64064163, 64064164, 64128162, 64128163, 32064322, 32064323, 128064162]),
    "o_512@128 (split-K 1x1 family)": (512, 0, 128, 512, 1, 0, [0, 32064642, 32064643, 32064322, 32064323, 64064642, 64064323]),
    "qkv_512@128": (512, 0, 128, 1536, 1, 0, [0, 64064642, 128064642, 128064322, 128128322]),
    "conv3_256@512": (256, 0, 512, 256, 3, 1, [0, 64064322, 64128162, 64128163, 128128162, 128064162]),
}
KSWEEP = {"M": 256, "T": 512, "Ks": [256, 512, 1024, 2048]}


def stream():
    return ct.c_void_p(torch.cuda.current_stream().cuda_stream)


def make_args(x1, x2, w, K, pad, cfg):
    a = native.DConvTest()
    a.x1, a.x2 = x1.data_ptr(), (x2.data_ptr() if x2 is not None else None)
    a.C1, a.C2, a.T = x1.shape[1], (x2.shape[1] if x2 is not None else 0), x1.shape[2]
    a.w, a.bias = w.ctypes.data, None
    a.Co, a.K, a.stride, a.pad, a.ups = w.shape[0], K, 1, pad, 0
    a.res, a.epilogue, a.plain_out, a.v_split, a.cfg = None, 0, 0, 0, cfg
    return a


def run(kind, x1, x2, w, K, pad, cfg, nprod, iters, fmt=0):
    """-> (out tensor, ms per launch, configuration string)"""
    L = native.lib()
    B, _, T = x1.shape
    out = torch.full((B, w.shape[0], T), float("nan"), dtype=torch.float32, device="cuda")
    a = make_args(x1, x2, w, K, pad, cfg)
    ms, cs = ct.c_float(), ct.create_string_buffer(160)
    if kind == "f32":
        rc = L.lds_bench_dconv(ct.byref(a), ct.c_void_p(out.data_ptr()), B, iters, ct.byref(ms), cs, 160, stream())
    else:
        rc = L.lds_bench_dconv_split(ct.byref(a), ct.c_void_p(out.data_ptr()), B, iters, nprod, fmt, ct.byref(ms), cs, 160, stream())
    if rc != 0:
        return None, None, L.lds_last_error().decode()
    torch.cuda.synchronize()
    return out, ms.value, cs.value.decode()


def errors(out, ref64):
    d = (out.double() - ref64)
    rms_ref = float(ref64.pow(2).mean().sqrt())
    return {"max_rel_rms": float(d.abs().max()) / rms_ref, "rms_rel_rms": float(d.pow(2).mean().sqrt()) / rms_ref,
            "max_rel_absmax": float(d.abs().max()) / float(ref64.abs().max())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "split_bf16_probe.json"))
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--B", type=int, default=16)
    args = ap.parse_args()
    B = args.B
    g = torch.Generator(device="cuda").manual_seed(1234)
    rng = np.random.default_rng(7)
    res = {"device": torch.cuda.get_device_name(0), "batch": B, "shapes": {}, "notes": [
        "errors are relative to the RMS of the fp64 result; operands: activations N(0,1), weights N(0,1)/sqrt(Ci*K)",
        "bf3 = three bf16 terms per operand; P6 = six bf16 products per fp32 product (product path), P9 all nine, P3 three"]}
    for name, (C1, C2, T, Co, K, pad, cfgs) in SHAPES.items():
        Ci = C1 + C2
        x1 = torch.randn(B, C1, T, device="cuda", generator=g)
        x2 = torch.randn(B, C2, T, device="cuda", generator=g) if C2 else None
        w = (rng.standard_normal((Co, Ci, K)) / np.sqrt(Ci * K)).astype(np.float32)
        x = x1 if x2 is None else torch.cat([x1, x2], 1)
        ref64 = torch.nn.functional.conv1d(x.double(), torch.from_numpy(w).cuda().double(), padding=pad)
        # independent check of the fp64 reference on one output row (numpy)
        xs = torch.nn.functional.pad(x[0].double(), (pad, pad)).cpu().numpy()
        row = sum(w[5, :, k].astype(np.float64) @ xs[:, k:k + T] for k in range(K))
        assert np.abs(row - ref64[0, 5].cpu().numpy()).max() < 1e-9 * max(1.0, np.abs(row).max())
        flops = 2.0 * B * T * Co * Ci * K
        ent = {"M": Co, "K_total": Ci * K, "N": B * T, "gflop": flops / 1e9, "kernels": {}}
        o, ms, cs = run("f32", x1, x2, w, K, pad, 0, 0, args.iters)
        ent["kernels"]["exact_f32"] = {"cfg": cs, "us": ms * 1e3, "tflops": flops / (ms * 1e-3) / 1e12, **errors(o, ref64)}
        print(f"{name:34s} f32  {cs:56s} {ms * 1e3:8.1f} us {flops / (ms * 1e-3) / 1e12:6.1f} TF  max {ent['kernels']['exact_f32']['max_rel_rms']:.2e} "
              f"rms {ent['kernels']['exact_f32']['rms_rel_rms']:.2e}", flush=True)
        best, best_h = None, None
        for fmt, tag, key in ((0, "bf3 ", "split_bf16_P6"), (1, "f16 ", "split_f16_H3")):
            for cfg in cfgs:
                o, ms, cs = run("bf3", x1, x2, w, K, pad, cfg, 0, args.iters, fmt)
                if o is None:
                    print(f"{name:34s} {tag} cfg {cfg}: {cs[:90]}", flush=True)
                    continue
                e = errors(o, ref64)
                rec = {"cfg": cs, "auto": cfg == 0, "us": ms * 1e3, "tflops": flops / (ms * 1e-3) / 1e12, **e}
                ent["kernels"].setdefault(key, []).append(rec)
                print(f"{name:34s} {tag} {cs:56s} {ms * 1e3:8.1f} us {rec['tflops']:6.1f} TF  max {e['max_rel_rms']:.2e} rms {e['rms_rel_rms']:.2e}", flush=True)
                if fmt == 0 and (best is None or rec["us"] < best["us"]):
                    best = rec
                if fmt == 1 and (best_h is None or rec["us"] < best_h["us"]):
                    best_h = rec
        if K == 1 and Co % 128 == 0:
            o, ms, cs = run("bf3", x1, x2, w, K, pad, 128128322, 4, args.iters, 1)      # fp16 planes with the fourth product (a2 b2)
            if o is not None:
                e = errors(o, ref64)
                ent["kernels"]["split_f16_H4"] = {"cfg": cs, "us": ms * 1e3, "tflops": flops / (ms * 1e-3) / 1e12, **e}
                print(f"{name:34s} H4   {cs:56s} {ms * 1e3:8.1f} us  max {e['max_rel_rms']:.2e} rms {e['rms_rel_rms']:.2e}", flush=True)
        if K == 1 and Co % 128 == 0:      # the product-count study on the one tile that has the three variants
            for nprod in (3, 9):
                o, ms, cs = run("bf3", x1, x2, w, K, pad, 128128322, nprod, args.iters)
                if o is not None:
                    e = errors(o, ref64)
                    ent["kernels"][f"split_bf16_P{nprod}"] = {"cfg": cs, "us": ms * 1e3, "tflops": flops / (ms * 1e-3) / 1e12, **e}
                    print(f"{name:34s} P{nprod}   {cs:56s} {ms * 1e3:8.1f} us  max {e['max_rel_rms']:.2e} rms {e['rms_rel_rms']:.2e}", flush=True)
        f32 = ent["kernels"]["exact_f32"]
        ent["best_split_bf16"] = best
        ent["speedup_launch"] = f32["us"] / best["us"]
        ent["error_ratio_max"] = best["max_rel_rms"] / f32["max_rel_rms"]
        ent["error_ratio_rms"] = best["rms_rel_rms"] / f32["rms_rel_rms"]
        ent["best_split_f16"] = best_h
        ent["f16_speedup_launch"] = f32["us"] / best_h["us"]
        ent["f16_error_ratio_max"] = best_h["max_rel_rms"] / f32["max_rel_rms"]
        ent["f16_error_ratio_rms"] = best_h["rms_rel_rms"] / f32["rms_rel_rms"]
        res["shapes"][name] = ent
    # ---- K sweep: time = fixed + K * slope at M = 256, N = B * 512 ----
    M, T = KSWEEP["M"], KSWEEP["T"]
    sweep = {"M": M, "N": B * T, "points": []}
    for Kc in KSWEEP["Ks"]:
        x1 = torch.randn(B, Kc, T, device="cuda", generator=g)
        w = (rng.standard_normal((M, Kc, 1)) / np.sqrt(Kc)).astype(np.float32)
        _, ms_f, cs_f = run("f32", x1, None, w, 1, 0, 64064642, 0, args.iters)
        _, ms_b, cs_b = run("bf3", x1, None, w, 1, 0, 64064642, 0, args.iters, 0)
        _, ms_h, cs_h = run("bf3", x1, None, w, 1, 0, 64064642, 0, args.iters, 1)
        sweep["points"].append({"K": Kc, "f32_us": ms_f * 1e3, "bf3_us": ms_b * 1e3, "f16_us": ms_h * 1e3, "f32_cfg": cs_f, "bf3_cfg": cs_b, "f16_cfg": cs_h})
        print(f"K sweep K={Kc:5d}: f32 {ms_f * 1e3:7.1f} us   bf3 {ms_b * 1e3:7.1f} us   f16 {ms_h * 1e3:7.1f} us", flush=True)
    p0, p1 = sweep["points"][0], sweep["points"][-1]
    dK = p1["K"] - p0["K"]
    fl_per_k = 2.0 * B * T * M
    for k in ("f32", "bf3", "f16"):
        slope_us = (p1[k + "_us"] - p0[k + "_us"]) / dK
        sweep[k + "_kloop_tflops"] = fl_per_k / (slope_us * 1e-6) / 1e12
        sweep[k + "_fixed_us"] = p0[k + "_us"] - slope_us * p0["K"]
    sweep["kloop_speedup"] = sweep["bf3_kloop_tflops"] / sweep["f32_kloop_tflops"]
    sweep["f16_kloop_speedup"] = sweep["f16_kloop_tflops"] / sweep["f32_kloop_tflops"]
    res["k_sweep"] = sweep
    res["gates"] = {
        "error_le_2x_exact_f32_everywhere": all(s["error_ratio_max"] <= 2.0 and s["error_ratio_rms"] <= 2.0 for s in res["shapes"].values()),
        "kloop_speedup_ge_1p5": sweep["kloop_speedup"] >= 1.5,
        "worst_error_ratio_max": max(s["error_ratio_max"] for s in res["shapes"].values()),
        "worst_error_ratio_rms": max(s["error_ratio_rms"] for s in res["shapes"].values()),
        "min_launch_speedup": min(s["speedup_launch"] for s in res["shapes"].values()),
        "f16_error_le_2x_exact_f32_everywhere": all(s["f16_error_ratio_max"] <= 2.0 and s["f16_error_ratio_rms"] <= 2.0 for s in res["shapes"].values()),
        "f16_kloop_speedup_ge_1p5": sweep["f16_kloop_speedup"] >= 1.5,
        "f16_worst_error_ratio_max": max(s["f16_error_ratio_max"] for s in res["shapes"].values()),
        "f16_worst_error_ratio_rms": max(s["f16_error_ratio_rms"] for s in res["shapes"].values()),
        "f16_min_launch_speedup": min(s["f16_speedup_launch"] for s in res["shapes"].values()),
    }
    print(json.dumps(res["gates"]), json.dumps({k: v for k, v in sweep.items() if k != "points"}))
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
