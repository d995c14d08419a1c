"""Headline benchmark: mel-frames/s of the 80-channel, 50-step DPM-Solver++ sampler (BASELINE.json
configs[1]: batch 16 x 512-frame utterances per GPU, unet1d denoiser, seeded random-init weights,
synthetic inputs).  One "step" = one full pass of the hot path over one batch: condition
embedding -> 50-NFE DPM-Solver++(2M) sampling -> mel [B,512,80].

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: the utterance batch shards across ranks (weak scaling: 16 utterances per GPU); the
only collectives are the RCCL scatter of the inputs before the timed region and the gather of the
results (mel; the waveform too for --workload e2e / full_tts) inside it.  Prints ONE JSON line on
rank 0.  `python bench.py --gpus N` WITHOUT a launcher (WORLD_SIZE unset, N > 1) starts the N ranks
itself -- a child `python -m torch.distributed.run ...` created before this process touches the GPU --
checks that the line it gets back says n_gpus = N and exits non-zero otherwise: an N-GPU request can
never silently measure one GPU.  --workload: sampler (configs[1] / [2]: the headline), e2e (configs[3]:
+ HiFi-VAEGAN decode, waveform gathered), full_tts (configs[4]: phones -> RoFormer generate -> codebook ->
sampler -> vocoder, 8 utterances per GPU).  At N = 1 the sampler line also carries `extra` legs (untimed
for `value`): configs[3] with the vocoder, the per-GPU share of configs[2] (UniPC, 20 NFE), the B = 1
latency of the caller north_star names (22_infer_tts.py), and `split_f16`: the same configs[1] step with
every UNet GEMM on the opt-in two-fp16-term path (csrc/conv_bf3.hip; 22-bit operands, narrower than the
reference's fp32), reported next to -- never as -- the exact-fp32 `value`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion-speech_amd"))
sys.path.insert(1, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X dense fp32 matrix peak (MI355X_MICROARCH.md, chip-level parameters)
PEAK_BF16_MFMA_TFLOPS = 2500.0   # dense bf16 matrix peak (same table); the split-bf16 path spends 6 bf16 products per fp32 product
PEAK_SPLIT_F16_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 3.0   # fp16 MFMA = the bf16 rate; the fp16-pair path spends 3 products per fp32 product
SPLIT_MODES = {      # (the lossless three-bf16 mode of round 3 was removed: it ran no faster than exact fp32)
    "split_f16": {"dtype": "2xfp16 operands (22 significand bits, weights pre-scaled per layer), 3 products, f32 accumulate", "peak": PEAK_SPLIT_F16_TFLOPS,
                  "peak_note": "2.5 PFLOP/s dense fp16 / 3 fp16 products per fp32 product"},
}
PEAK_HBM_TBS = 8.0               # HBM3E spec peak (same table; ~6.3 TB/s achievable)
FRAME_SEC = 512 / 44100.0        # one mel frame = hop 512 @ 44.1 kHz (reference configs/config.yaml:3,12)
UNET_GFLOP_PER_UTT_FWD = 40.98   # SURVEY.md 8d (T=512, M=80), algorithmic
VOCODER_GFLOP_PER_UTT = 331.9    # SURVEY.md App. B (512 frames, synthetic h)


def cpu_baseline(T, n_eval):
    """Oracle (numpy port of the reference path) timed on the host cores: `n_eval` denoiser
    evaluations of one T-frame utterance inside the DPM-Solver++ loop, scaled to 50 NFE."""
    from lds import arch, init_weights
    from oracle import schedule, solvers, unit2mel as o_u2m
    from threadpoolctl import threadpool_limits
    cores = min(16, os.cpu_count() or 1)      # the GPU box's CPU share for one GPU; more BLAS threads only slow these small GEMMs
    threadpool_limits(limits=cores)
    cfg = arch.unet_config()
    blocks = arch.unet_blocks(cfg)
    w = init_weights.init_state(arch.unet_param_shapes(cfg), 0)
    cond = init_weights.uniform("bench.cpu.cond", (1, 256, T), 1, -1, 1)
    xT = init_weights.uniform("bench.cpu.xT", (1, 80, T), 2, -1.7, 1.7)
    f = o_u2m.make_eps_fn(w, cfg, blocks, cond)
    f(xT, np.full((1,), 500.0, dtype=np.float32))           # warm the BLAS threads
    t0 = time.perf_counter()
    solvers.dpm_solver_pp_2m(f, schedule.diffusion_buffers()["betas"], xT, n_eval)
    dt = time.perf_counter() - t0
    per_nfe = dt / n_eval
    # configs[0] (plumbing, CPU): 1 utterance x 256 frames, 100-step DDPM (k_step 100, infer_speedup 1) -- a bounded sample of
    # `n0` ancestral steps (one denoiser evaluation + the posterior update each), scaled to the 100 of the configuration
    T0, n0 = 256, max(2, n_eval // 2)
    cond0 = init_weights.uniform("bench.cpu.cond0", (1, 256, T0), 1, -1, 1)
    x0 = init_weights.uniform("bench.cpu.xT0", (1, 80, T0), 2, -1.7, 1.7)
    noise0 = init_weights.uniform("bench.cpu.noise0", (n0, 1, 80, T0), 3, -1.7, 1.7)
    f0 = o_u2m.make_eps_fn(w, cfg, blocks, cond0)
    t0 = time.perf_counter()
    solvers.ddpm(f0, schedule.diffusion_buffers(), x0, n0, noise0)
    dt0 = time.perf_counter() - t0
    config0 = {"workload": "configs[0]: 1 utterance x 256 frames, 100-step DDPM (CPU plumbing case)", "seconds_per_100_steps": 100.0 * dt0 / n0,
               "mel_frames_per_sec": T0 / (100.0 * dt0 / n0), "sample": f"{n0} ancestral steps ({dt0:.1f} s) scaled to 100"}
    return {"value": T / (per_nfe * 50), "unit": "mel-frames/s", "cores": int(cores), "kind": "port", "config0_ddpm100_T256": config0,
            "sample": f"numpy oracle, 1 utterance x {T} frames, {n_eval}-NFE DPM-Solver++ run ({dt:.1f} s) scaled to 50 NFE",
            "reference_torch_cpu_8core_survey": 29.0,
            "note": "the reference's own torch-CPU path measured during the survey (8 cores, BASELINE.md 4) is ~29 mel-frames/s; "
                    "this numpy port is ~7x slower than that and is a reported baseline only"}


def _sha256(path):
    import hashlib
    try:
        return hashlib.sha256(open(path, "rb").read()).hexdigest()
    except OSError:
        return None


def lib_fingerprint():
    """what a counter summary must have been taken on to describe THIS run: the kernels' shared object"""
    return _sha256(os.path.join(ROOT, "latent-diffusion-speech_amd", "lds", "liblds.so"))


def _newest_profile_json(suffix, key=None):
    """newest profiles/*<suffix> (by name: rNN_...) that holds `key` under "kernels" -- a round commits several summaries with one suffix
    (the sampler's and the vocoder's), and only one of them knows a given kernel"""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*" + suffix)), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if key is None or key in d.get("kernels", {}):
            return d, os.path.basename(f)
    return None, None


def _pmc_key(kernel_name):
    import re
    m = re.match(r"(\w+)<BM(\d+) BN(\d+) KT(\d+) S(\d+) U(\d+) BK(\d+)(?: NST(\d+))?(?: KS\d+)?( GNF)?>", kernel_name)
    if not m:
        return None
    fam, bm, bn, kt, st, up, bk, nst, gnf = m.groups()
    # rocprofv3 prints every template argument: <BM, BN, KT, STRIDE, UPS, BK, NST, DIL, VOC, GNF> (the UNet's kernels have DIL 1, VOC false;
    # GNF = the GroupNorm-fold instantiations)
    return f"{fam}<{bm}, {bn}, {kt}, {st}, {'true' if up == '1' else 'false'}, {bk}, {nst}, 1, false, {'true' if gnf else 'false'}>" if nst else None


# source files whose change invalidates a counter summary of a kernel family
_FAMILY_SOURCES = {"conv_dma": ["conv_dma.hip", "k4p.h", "gn_chan.h"], "conv_bf3": ["conv_bf3.hip", "k8b3.h"],
                   "attention": ["attention_k4p.hip", "k4p.h"], "gn_stream": ["k4p_ops.hip", "k4p.h"]}


def pmc_lookup(kernel_name, suffix, field=None):
    """(value, provenance) for `kernel_name` from the newest committed counter summary profiles/*<suffix> (separate rocprofv3 --pmc
    passes summarised by tools/summarize_pmc.py / summarize_mfma.py and stamped by tools/stamp_profile.py).  These are NOT measured in
    this run: the provenance says which file and which commit it was taken on, and `stale` is true when the summary carries no stamp
    or a source file of the kernel's family has changed since."""
    key = _pmc_key(kernel_name)
    d, fname = _newest_profile_json(suffix, key) if key is not None else (None, None)
    if not d:
        return None, None
    v = d["kernels"][key]
    prov = {"file": "profiles/" + fname, "commit": d.get("commit")}
    srcs = d.get("src_sha256")
    fam = kernel_name.split("<")[0]
    if not srcs:
        prov["stale"] = True
    else:
        csrc = os.path.join(ROOT, "latent-diffusion-speech_amd", "csrc")
        prov["stale"] = any(srcs.get(f) != _sha256(os.path.join(csrc, f)) for f in _FAMILY_SOURCES.get(fam, sorted(srcs)))
    return (v[field] if field else v), prov


class NativeProfiler:
    """liblds's HIP-event profiler: one event pair per launch on the launch stream; for the conv_dma family the pair is bound to the
    dispatch itself (hipExtLaunchKernelGGL start / stop slots), so those durations are the kernels' own begin-to-end times."""

    def __init__(self, detail=False):
        self.detail = detail

    def run(self, fn):
        from lds import native
        native.prof_enable(2 if self.detail else 1)
        fn()
        torch.cuda.synchronize()
        prof = native.prof_summary()
        native.prof_enable(0)
        return prof


def conv_family_rate(prof):
    conv = [r for r in prof if r["name"].startswith("conv_")]
    ms = sum(r["ms"] for r in conv)
    return (sum(r["flops"] for r in conv) / (ms * 1e-3) / 1e12) if ms > 0 else 0.0, ms


def roofline_from_profile(prof):
    """`roofline` object for the dominant kernel of one instrumented step + a per-kernel breakdown."""
    prof = sorted(prof, key=lambda r: -r["ms"])
    tot_ms = sum(r["ms"] for r in prof)
    dom = prof[0]
    ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
    conv_tf, conv_ms = conv_family_rate(prof)
    roof = {
        "bound": "mfma", "kernel": dom["name"], "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
        "frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": pmc_lookup(dom["name"], "_hbm_traffic.json", "hbm_bytes_per_launch")[0],
        "traffic_source": pmc_lookup(dom["name"], "_hbm_traffic.json", "hbm_bytes_per_launch")[1],
        "algorithmic_bytes_per_launch": dom["bytes"] / dom["count"],
        "launches": dom["count"], "avg_launch_us": 1e3 * dom["ms"] / dom["count"],
        "gflop_per_launch": dom["flops"] / dom["count"] / 1e9,
        "share_of_step_time": dom["ms"] / tot_ms,
        "all_conv_gemm_tflops": conv_tf, "all_conv_gemm_frac": conv_tf / PEAK_F32_MFMA_TFLOPS,
        "all_conv_gemm_share": conv_ms / tot_ms,
        "mfma_util_pmc": pmc_lookup(dom["name"], "_mfma_util.json")[0], "mfma_util_source": pmc_lookup(dom["name"], "_mfma_util.json")[1],
        "launches_per_step": int(sum(r["count"] for r in prof)),
    }
    # the memory-bound families against the HBM roofline (algorithmic bytes / HIP-event time)
    hbm = {}
    for r in prof:
        fam = r["name"].split("<")[0]
        if r["flops"] == 0 and r["bytes"] > 0:
            a = hbm.setdefault(fam, {"ms": 0.0, "bytes": 0.0, "launches": 0})
            a["ms"] += r["ms"]; a["bytes"] += r["bytes"]; a["launches"] += r["count"]
    roof["hbm_bound_kernels"] = {k: {"ms": round(v["ms"], 3), "launches": v["launches"], "achieved_TBs": v["bytes"] / (v["ms"] * 1e-3) / 1e12,
                                     "frac_of_8TBs": v["bytes"] / (v["ms"] * 1e-3) / 1e12 / PEAK_HBM_TBS, "share_of_step_time": v["ms"] / tot_ms}
                                 for k, v in hbm.items() if v["ms"] > 0}
    return roof, {r["name"]: round(r["ms"], 3) for r in prof[:14]}


def timed_loop(step, steps, warmup, world, sync, dist=None, dev=None):
    """W untimed warm-up steps, then exactly K steps bracketed by barrier + device sync; max over ranks."""
    def barrier():
        if dist is not None:
            dist.barrier()
        sync()

    out = None
    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    return dt, out


class SamplerPipeline:
    """configs[1] / [2] (and, with a vocoder, configs[3]): units -> mel [-> wav].  inputs: units [n,T,1280] f32, spk [n,1] i64."""
    name = "sampler"

    def __init__(self, args, model, make_inputs, voc=None):
        self.args, self.model, self.voc, self._make = args, model, voc, make_inputs
        self.B, self.T = args.batch, args.frames
        if voc is not None:
            self.name = "e2e"

    def input_specs(self, n):
        return [((n, self.T, 1280), torch.float32), ((n, 1), torch.int64)]

    def make_inputs(self, n):
        return list(self._make(n, self.T))

    def step(self, units, spk):
        mel = self.model(units, None, spk_id=spk, infer=True, infer_speedup=1000 // self.args.nfe, method=self.args.method)
        out = {"mel": mel}
        if self.voc is not None:
            out["wav"] = self.voc(mel)
        return out

    def check(self, outs, n):
        assert bool(torch.isfinite(outs["mel"]).all()), "non-finite mel"
        assert tuple(outs["mel"].shape) == (n, self.T, 80), tuple(outs["mel"].shape)
        if "wav" in outs:
            assert bool(torch.isfinite(outs["wav"]).all()), "non-finite waveform"
            assert tuple(outs["wav"].shape)[0] == n, tuple(outs["wav"].shape)

    def describe(self):
        a = self.args
        return (f"configs[1]: batch={self.B}x{self.T}-frame utterances per GPU, {a.nfe}-step {a.method}, unet1d denoiser, 80-ch mel"
                + (" + HiFi-VAEGAN vocoder, waveform gathered (configs[3])" if self.voc is not None else ""))


class FullTTSPipeline:
    """configs[4] (reference 22_infer_tts.py:76-114): phones, tones -> RoFormer generate (T sampled semantic tokens) -> k-means codebook
    rows -> Unit2Mel sampler -> HiFi-VAEGAN waveform.  inputs: phones, tones [n,Lp] i64, spk [n,1] i64."""
    name = "full_tts"

    def __init__(self, args, model, voc, text2semantic, codebook_lookup, Lp=64):
        self.args, self.model, self.voc, self.t2s, self.lookup, self.Lp = args, model, voc, text2semantic, codebook_lookup, Lp
        self.B, self.T = args.batch, args.frames

    def input_specs(self, n):
        return [((n, self.Lp), torch.int64), ((n, self.Lp), torch.int64), ((n, 1), torch.int64)]

    def make_inputs(self, n):
        idx = np.arange(n * self.Lp).reshape(n, self.Lp)
        phones = torch.from_numpy((idx * 7 % 107 + 1).astype(np.int64))
        tones = torch.from_numpy((idx * 5 % 12).astype(np.int64))
        spk = torch.from_numpy((np.arange(n) * 37 % 323 + 1).astype(np.int64).reshape(n, 1))
        return [phones, tones, spk]

    def step(self, phones, tones, spk):
        tok = self.t2s(phones, tones, self.T + 1)             # seeded random weights never emit EOS: exactly T tokens per row
        assert tuple(tok.shape) == (phones.shape[0], self.T), tuple(tok.shape)
        units = self.lookup(tok)
        mel = self.model(units, None, spk_id=spk, infer=True, infer_speedup=1000 // self.args.nfe, method=self.args.method)
        return {"mel": mel, "wav": self.voc(mel)}

    check = SamplerPipeline.check

    def describe(self):
        a = self.args
        return (f"configs[4]: batch={self.B} utterances per GPU, {self.Lp} phones -> RoFormer top-k sampling of {self.T} semantic tokens -> codebook -> "
                f"{a.nfe}-step {a.method} -> HiFi-VAEGAN, waveform gathered")


def run_job(args, rank, world, dev, model=None, voc=None, dist=None, sync=None, profiler=None, make_inputs=None, on_output=None, pipeline=None):
    """The bench proper, parameterised so tests can drive the multi-rank plumbing on CPU (gloo) with a stub model.
    Returns the result dict on rank 0 (None elsewhere)."""
    from lds import shard
    pipe = pipeline if pipeline is not None else SamplerPipeline(args, model, make_inputs, voc)
    B, T = pipe.B, pipe.T
    sync = sync or (lambda: None)

    # synthetic inputs for the global batch live on rank 0 and are scattered over RCCL (untimed set-up)
    force = dist is not None      # (--rehearse-rccl: the collectives run even at N = 1)
    full = [t.to(dev) for t in pipe.make_inputs(world * B)] if rank == 0 else None
    local = []
    for k, (shape, dtype) in enumerate(pipe.input_specs(world * B)):
        src = full[k] if rank == 0 else torch.empty(0, device=dev)
        local.append(shard.scatter_batch(src, rank, world, shape=shape, dtype=dtype, device=dev, force=force))
    del full

    def step(gather=True):
        outs = pipe.step(*local)
        if gather:
            outs = {k: shard.gather_batch(v, rank, world, sizes=[B] * world, force=force) for k, v in outs.items()}
        return outs

    dt, outs = timed_loop(step, args.steps, args.warmup, world, sync, dist, dev)
    if rank == 0:
        pipe.check(outs, world * B)
        if on_output is not None:
            on_output(outs["mel"])

    frames = world * B * T * args.steps
    value = frames / dt
    res = {
        "metric": "mel_frames_per_sec", "value": value, "unit": "mel-frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": pipe.describe(), "pipeline": pipe.name, "utterances_per_gpu": B, "frames": T, "nfe": args.nfe, "method": args.method,
                   "parallelism": f"batch-shard x{world}", "gathered": sorted(outs.keys()) if rank == 0 else None},
        "x_realtime": value * FRAME_SEC, "rtf": 1.0 / (value * FRAME_SEC),
        "mel_frames_per_sec_per_gpu": value / world,
    }
    if rank != 0:
        return None
    if profiler is not None:
        # roofline leg: one extra instrumented step on rank 0 ONLY, so it must not contain a collective (gather=False)
        prof = profiler.run(lambda: step(gather=False))
        res["roofline"], res["kernel_breakdown_ms"] = roofline_from_profile(prof)
        res["unet_algorithmic_tflops"] = (UNET_GFLOP_PER_UTT_FWD * B * args.nfe * 1e9 / (dt / args.steps) / 1e12) if T == 512 else None
    return res


def split_leg(args, model, make_inputs, mode):
    """configs[1] again with every convolution / linear layer of the UNet as a split-operand GEMM on the 16-bit matrix pipe
    (csrc/conv_bf3.hip, csrc/k8b3.h).  mode "split_f16": two fp16 terms per fp32 operand (22 bits; per-layer weight scale), three products,
    fp32 accumulate.  Same parity suite, same tolerances
    (tests/test_gpu_model.py runs every whole-path test in all modes); error study: profiles/r03_split_probe_bf16x3_f16x2.json.  Reported
    next to the exact-fp32 `value`, never as it."""
    info = SPLIT_MODES[mode]
    unet = model.decoder.denoise_fn
    units, spk = make_inputs(args.batch, args.frames)
    sync = torch.cuda.synchronize
    keep = {}

    def step(x_T=None):
        keep["mel"] = model(units, None, spk_id=spk, infer=True, infer_speedup=1000 // args.nfe, method=args.method, x_T=x_T)

    def seeded(mode):                                           # one run with a given x_T, so the two paths see the same start
        unet.set_gemm_mode(mode)
        xT = torch.randn((args.batch, 1, 80, args.frames), device="cuda", generator=torch.Generator(device="cuda").manual_seed(11))
        step(xT); sync()
        return keep["mel"].clone()
    try:
        y32, y16 = seeded("f32"), seeded(mode)
        rel = float((y16 - y32).abs().max() / y32.abs().max())
        step(); sync()                                          # (the split mode stays on from here)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        dt = (time.perf_counter() - t0) / args.steps
        assert bool(torch.isfinite(keep["mel"]).all())
        prof = NativeProfiler().run(step)
        roof, breakdown = roofline_from_profile(prof)
        dom = max((r for r in prof if r["name"].startswith("conv_bf3")), key=lambda r: r["ms"])
        ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        conv_tf, _ = conv_family_rate(prof)
        fps = args.batch * args.frames / dt
        return {
            "workload": f"configs[1] with {mode} GEMMs: batch={args.batch}x{args.frames} frames, {args.nfe}-step {args.method}",
            "dtype": info["dtype"],
            "ms_per_step": 1e3 * dt, "mel_frames_per_sec": fps, "x_realtime": fps * FRAME_SEC,
            "unet_algorithmic_tflops": UNET_GFLOP_PER_UTT_FWD * (args.frames / 512.0) * args.batch * args.nfe * 1e9 / dt / 1e12,
            "max_rel_diff_vs_exact_f32_same_xT": rel,
            "roofline": {"bound": "mfma", "kernel": dom["name"], "achieved": ach, "peak": info["peak"], "unit": "TFLOP/s (fp32-equivalent)",
                         "frac": ach / info["peak"], "peak_note": info["peak_note"],
                         "launches": dom["count"], "avg_launch_us": 1e3 * dom["ms"] / dom["count"], "gflop_per_launch": dom["flops"] / dom["count"] / 1e9,
                         "all_conv_gemm_tflops": conv_tf, "all_conv_gemm_frac": conv_tf / info["peak"],
                         "all_conv_gemm_vs_f32_mfma_peak": conv_tf / PEAK_F32_MFMA_TFLOPS, "launches_per_step": roof["launches_per_step"]},
            "kernel_breakdown_ms": breakdown,
        }
    finally:
        unet.set_gemm_mode("f32")


def extra_legs(args, dev, model, make_inputs):
    """configs[3], configs[2]'s per-GPU share and the B = 1 latency (N = 1 only; none of them feeds `value`)."""
    from encoder.hifi_vaegan.hifi_vaegan import Hifi_VAEGAN
    from lds import arch, init_weights
    T = args.frames
    sync = torch.cuda.synchronize
    extra = {}
    for mode in SPLIT_MODES:
        extra[mode] = split_leg(args, model, make_inputs, mode)
    units, spk = make_inputs(args.batch, T)

    def timeit(fn, n):
        fn(); sync()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        sync()
        return (time.perf_counter() - t0) / n

    # ---- configs[3]: units -> mel (50-step DPM-Solver++) -> waveform, RTF ----
    h = arch.SYNTHETIC_VOCODER_H
    voc = Hifi_VAEGAN(None, device=dev, h=h, state=init_weights.init_state(arch.generator_param_shapes(h), 0))
    keep = {}

    def e2e():
        mel = model(units, None, spk_id=spk, infer=True, infer_speedup=1000 // args.nfe, method=args.method)
        keep["mel"] = mel
        keep["wav"] = voc(mel)
    dt = timeit(e2e, 2)
    assert bool(torch.isfinite(keep["wav"]).all()), "non-finite waveform"
    assert tuple(keep["wav"].shape) == (args.batch, 1, T * 512)
    dtv = timeit(lambda: voc(keep["mel"]), 3)
    prof = NativeProfiler(detail=True).run(lambda: voc(keep["mel"]))
    stages = {}
    for r in prof:
        stages[r["name"]] = {"ms": round(r["ms"], 4), "launches": r["count"], "tflops": r["flops"] / (r["ms"] * 1e-3) / 1e12 if r["ms"] > 0 else 0.0,
                             "algorithmic_TBs": r["bytes"] / (r["ms"] * 1e-3) / 1e12 if r["ms"] > 0 else 0.0,
                             "flop_per_byte": r["flops"] / r["bytes"] if r["bytes"] > 0 else None}
    fps = args.batch * T / dt
    extra["configs3_e2e_vocoder"] = {
        "workload": f"configs[3]: batch={args.batch}x{T} frames, {args.nfe}-step {args.method} + HiFi-VAEGAN decode to {T * 512} samples/utt",
        "ms_per_step": 1e3 * dt, "mel_frames_per_sec": fps, "x_realtime": fps * FRAME_SEC, "rtf": 1.0 / (fps * FRAME_SEC),
        "vocoder_ms": 1e3 * dtv, "vocoder_tflops": VOCODER_GFLOP_PER_UTT * (T / 512.0) * args.batch * 1e9 / dtv / 1e12,
        "vocoder_frac_of_f32_mfma_peak": VOCODER_GFLOP_PER_UTT * (T / 512.0) * args.batch * 1e9 / dtv / 1e12 / PEAK_F32_MFMA_TFLOPS,
        "vocoder_kernels": dict(sorted(stages.items(), key=lambda kv: -kv[1]["ms"])[:24]),
    }
    del voc, keep

    # ---- configs[2], one GPU's share: 16 x 512 frames, 20-step UniPC ----
    def unipc():
        return model(units, None, spk_id=spk, infer=True, infer_speedup=50, method="unipc")
    dt = timeit(unipc, 3)
    prof = NativeProfiler().run(unipc)
    conv_tf, _ = conv_family_rate(prof)
    fps = args.batch * T / dt
    extra["configs2_unipc20_per_gpu"] = {
        "workload": f"configs[2] per-GPU share: batch={args.batch}x{T} frames, 20-step unipc",
        "ms_per_step": 1e3 * dt, "mel_frames_per_sec": fps, "x_realtime": fps * FRAME_SEC,
        "unet_algorithmic_tflops": UNET_GFLOP_PER_UTT_FWD * (T / 512.0) * args.batch * 20 * 1e9 / dt / 1e12,
        "all_conv_gemm_tflops": conv_tf, "all_conv_gemm_frac": conv_tf / PEAK_F32_MFMA_TFLOPS,
    }

    # ---- B = 1 latency (the 22_infer_tts.py caller runs one utterance at a time) ----
    u1, s1 = units[:1].contiguous(), spk[:1].contiguous()

    def one():
        return model(u1, None, spk_id=s1, infer=True, infer_speedup=1000 // args.nfe, method=args.method)
    unet = model.decoder.denoise_fn

    def b1_leg():
        dt = timeit(one, 3)
        prof = NativeProfiler().run(one)
        conv_tf, _ = conv_family_rate(prof)
        return {"ms_per_utterance": 1e3 * dt, "mel_frames_per_sec": T / dt, "x_realtime": T / dt * FRAME_SEC,
                "unet_algorithmic_tflops": UNET_GFLOP_PER_UTT_FWD * (T / 512.0) * args.nfe * 1e9 / dt / 1e12,
                "all_conv_gemm_tflops": conv_tf, "all_conv_gemm_frac": conv_tf / PEAK_F32_MFMA_TFLOPS,
                "launches_per_step": int(sum(r["count"] for r in prof))}
    extra["b1_latency"] = {"workload": f"1 utterance x {T} frames, {args.nfe}-step {args.method} (latency of the 22_infer_tts.py caller)",
                           "mode": "default: tile choices at the nominal batch of 16, bit-identical with batched results", **b1_leg()}
    # the opt-in latency mode (lds_unet_set_latency_mode): tiles and cluster split-K from the actual batch; same tolerances vs the
    # oracle (tests/test_gpu_model.py::test_latency_*), not bit-identical with the default mode
    unet.set_latency_mode(True)
    try:
        extra["b1_latency_mode"] = {"workload": extra["b1_latency"]["workload"], "mode": "latency mode, exact fp32", **b1_leg()}
        unet.set_gemm_mode("split_f16")
        leg = b1_leg()
        leg.pop("all_conv_gemm_frac")
        extra["b1_latency_mode_split_f16"] = {"workload": extra["b1_latency"]["workload"], "mode": "latency mode, split-fp16 GEMMs", **leg}
    finally:
        unet.set_gemm_mode("f32")
        unet.set_latency_mode(False)
    # ---- configs[5], one GPU's share (8 of 64 utterances): phones -> RoFormer generate (512 tokens) -> units -> mel -> wav ----
    sys.path.insert(0, ROOT)
    import infer_tts
    from lds import native
    lm = infer_tts.synthetic_lm(dev)
    voc = Hifi_VAEGAN(None, device=dev, h=h, state=init_weights.init_state(arch.generator_param_shapes(h), 0))
    Bq, Lp = 8, 64
    phones = torch.from_numpy((np.arange(Bq * Lp).reshape(Bq, Lp) * 7 % 107 + 1).astype(np.int64)).to(dev)
    tones = torch.from_numpy((np.arange(Bq * Lp).reshape(Bq, Lp) * 5 % 12).astype(np.int64)).to(dev)
    codebook = torch.from_numpy(init_weights.uniform("synthetic.codebook", (4096, 1280), 5, -1.7, 1.7)).to(dev)
    keep = {}

    def lm_only():
        torch.manual_seed(1234)                                                 # the sampler's uniforms: same draws every call
        keep["tok"] = infer_tts.text2semantic(lm, phones, tones, 1, T + 1)      # random weights never emit EOS: exactly T tokens

    def full():
        lm_only()
        units = native.gather_rows(codebook, keep["tok"].clamp(max=4095))
        mel = model(units, None, spk_id=spk[:Bq], infer=True, infer_speedup=1000 // args.nfe, method=args.method)
        keep["wav"] = voc(mel)
    dt_lm = timeit(lm_only, 2)
    dt = timeit(full, 2)
    assert tuple(keep["tok"].shape) == (Bq, T) and bool(torch.isfinite(keep["wav"]).all())
    unet.set_latency_mode(True)      # 8 utterances are half the nominal batch: the tile choices of the latency mode pay here too
    try:
        dt_lat = timeit(full, 2)
    finally:
        unet.set_latency_mode(False)
    extra["configs5_full_tts_per_gpu"] = {
        "workload": f"configs[5] per-GPU share: {Bq} utterances, {Lp} phones -> RoFormer top-k sampling of {T} semantic tokens -> {args.nfe}-step "
                    f"{args.method} -> HiFi-VAEGAN ({T * 512} samples/utt)",
        "ms_per_step": 1e3 * dt, "x_realtime": Bq * T * FRAME_SEC / dt, "rtf": dt / (Bq * T * FRAME_SEC),
        "lm_ms": 1e3 * dt_lm, "lm_tokens_per_sec": Bq * T / dt_lm, "lm_us_per_decode_step": 1e6 * dt_lm / T,
        "latency_mode_ms_per_step": 1e3 * dt_lat, "latency_mode_x_realtime": Bq * T * FRAME_SEC / dt_lat,
    }
    # ---- configs[4] in small: a RAGGED batch, 16 utterances of 16 different lengths (272 .. 512 frames), tokens -> units -> mel -> wav.  Equal
    #      lengths batch; different lengths run as buckets of one (infer_tts.synthesize_ragged), sequentially or overlapped on HIP streams ----
    from diffusion.vocoder import Vocoder
    from tools.infer_tools import DiffusionSVC
    vw = Vocoder.__new__(Vocoder)
    vw.vocoder, vw.vocoder_hop_size, vw.vocoder_sample_rate, vw.dimension, vw.device = voc, h["hop_size"], h["sampling_rate"], h["inter_channels"], dev
    svc = DiffusionSVC(device=dev)
    svc.model, svc.vocoder = model, vw
    rag_rows = [torch.from_numpy((np.arange(n) * (7 + i) % 4096).astype(np.int64)).to(dev) for i, n in enumerate(range(272, 513, 16))]
    rag_frames = sum(int(r.numel()) for r in rag_rows)

    def ragged(streams):
        keep["rag"] = infer_tts.synthesize_ragged(svc, codebook, rag_rows, 1, 1000 // args.nfe, args.method, streams=streams)
    rag = {}
    for name, lat, streams in (("sequential", False, 1), ("sequential_latency_mode", True, 1), ("streams4", False, 4), ("streams4_latency_mode", True, 4)):
        unet.set_latency_mode(lat)
        try:
            rag[name + "_ms"] = 1e3 * timeit(lambda: ragged(streams), 1)
        finally:
            unet.set_latency_mode(False)
    assert all(bool(torch.isfinite(w_).all()) for _, w_ in keep["rag"])

    def ragged_masked():      # ONE padded batch with per-utterance lengths inside the kernels: the sampler AND the vocoder (lds_*_ragged)
        keep["ragm"] = infer_tts.synthesize_ragged_masked(svc, codebook, rag_rows, 1, 1000 // args.nfe, args.method, streams=4, max_batch=16)
    rag["masked_batch_ms"] = 1e3 * timeit(ragged_masked, 1)
    assert all(bool(torch.isfinite(w_).all()) for _, w_ in keep["ragm"])
    extra["ragged16_tokens_to_wav"] = {
        "workload": f"{len(rag_rows)} utterances of {len(rag_rows)} different lengths (272 .. 512 frames, {rag_frames} in all): units -> {args.nfe}-step {args.method} "
                    "-> HiFi-VAEGAN, one bucket per length (bit-identical with each utterance alone); streams4 = buckets overlapped on 4 HIP streams / host threads; "
                    "masked_batch = sampler and vocoder each as ONE padded batch with per-utterance lengths inside the kernels (lds_sampler_run_ragged, "
                    "lds_vocoder_forward_ragged; parity tolerance, not bit-identity)",
        **rag, "best_x_realtime": rag_frames * FRAME_SEC / (min(rag.values()) * 1e-3),
    }
    # ---- the 22_infer_tts.py caller itself: ONE utterance, phones -> tokens -> units -> mel -> wav ----
    p1, t1 = phones[:1].contiguous(), tones[:1].contiguous()

    def lm_one():
        torch.manual_seed(1234)
        keep["tok1"] = infer_tts.text2semantic(lm, p1, t1, 1, T + 1)

    def full_one():
        lm_one()
        units = native.gather_rows(codebook, keep["tok1"].clamp(max=4095))
        mel = model(units, None, spk_id=spk[:1], infer=True, infer_speedup=1000 // args.nfe, method=args.method)
        keep["wav1"] = voc(mel)
    dt_lm1 = timeit(lm_one, 2)
    dt1 = timeit(full_one, 2)
    assert tuple(keep["tok1"].shape) == (1, T) and bool(torch.isfinite(keep["wav1"]).all())
    unet.set_latency_mode(True)
    try:
        dt1_lat = timeit(full_one, 2)
    finally:
        unet.set_latency_mode(False)
    extra["b1_full_tts_latency"] = {
        "workload": f"1 utterance: {Lp} phones -> {T} sampled tokens -> {args.nfe}-step {args.method} -> {T * 512} samples (22_infer_tts.py, one sentence)",
        "ms_per_utterance": 1e3 * dt1, "audio_seconds": T * FRAME_SEC, "x_realtime": T * FRAME_SEC / dt1, "rtf": dt1 / (T * FRAME_SEC),
        "lm_ms": 1e3 * dt_lm1, "lm_us_per_decode_step": 1e6 * dt_lm1 / T,
        "latency_mode_ms_per_utterance": 1e3 * dt1_lat, "latency_mode_x_realtime": T * FRAME_SEC / dt1_lat,
    }
    return extra


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=["sampler", "e2e", "full_tts"], default="sampler",
                    help="sampler = configs[1]/[2] (headline); e2e = + vocoder, waveform gathered (configs[3]); full_tts = phones -> LM -> ... -> waveform (configs[4])")
    ap.add_argument("--batch", type=int, default=None, help="utterances per GPU (default 16; 8 for --workload full_tts: configs[4] is 64 on 8 GPUs)")
    ap.add_argument("--frames", type=int, default=512)
    ap.add_argument("--nfe", type=int, default=50)
    ap.add_argument("--method", default="dpm-solver")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-nfe", type=int, default=8)
    ap.add_argument("--vocoder", action="store_true", help="same as --workload e2e")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra legs (split-bf16, configs[3], UniPC-20, B=1 latency, configs[4] share) at N=1")
    ap.add_argument("--rehearse-rccl", action="store_true",
                    help="N=1 only: open a single-rank RCCL process group and issue every collective of the N>1 path (scatter, gather, barrier, all-reduce)")
    ap.add_argument("--no-profile", action="store_true", help="skip the instrumented roofline step (for rocprofv3 --pmc passes)")
    ap.add_argument("--stub-cpu", action="store_true",
                    help="tests only: gloo ranks on the CPU with stand-in models (exercises the launcher / scatter / gather / JSON plumbing without a GPU)")
    args = ap.parse_args(argv)
    if args.vocoder and args.workload == "sampler":
        args.workload = "e2e"
    if args.batch is None:
        args.batch = 8 if args.workload == "full_tts" else 16
    return args


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks as a child `python -m torch.distributed.run` (this process
    has not touched the GPU and never will), pass their output through, and insist that the one JSON line says n_gpus = N."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if proc.returncode != 0:
        raise SystemExit(f"bench.py: the {args.gpus}-rank run failed with exit code {proc.returncode}")
    if line is None:
        raise SystemExit(f"bench.py: the {args.gpus}-rank run printed no result line")
    if json.loads(line).get("n_gpus") != args.gpus:
        raise SystemExit(f"bench.py: asked for --gpus {args.gpus} but the result line says n_gpus = {json.loads(line).get('n_gpus')}")
    print(line, flush=True)
    return 0


class _StubUnit2Mel:
    """--stub-cpu: mel[b,t,:] = 2 * units[b,t,:80] + spk_id[b] (per-utterance, so the gathered result identifies every shard)"""

    def __call__(self, units, volume, spk_id=None, infer=True, infer_speedup=10, method="dpm-solver"):
        return 2.0 * units[..., :80] + spk_id.to(torch.float32)[:, :, None]


def build_pipeline(args, dev, rank=0, world=1, dist=None):
    """models + the pipeline object of --workload (seeded random-init weights: no checkpoints exist).  Multi-rank: rank 0
    generates the weights, every other rank builds its modules empty and receives them by ONE broadcast per model (lds/shard.py broadcast_state:
    RCCL on the GPU, north_star's "RCCL broadcast/gather"); every rank then packs its own copy for its own device."""
    import contextlib
    from lds import init_weights, shard
    force = dist is not None and world == 1      # --rehearse-rccl: the broadcast runs on a single-rank group too
    share = (world > 1 or force) and not args.stub_cpu
    empty = (lambda: init_weights.deferred()) if (share and rank != 0) else contextlib.nullcontext

    def bcast(state):
        return shard.broadcast_state(state, rank, world, device=dev, force=force) if share else 0

    def make_inputs(n, T):
        units = torch.from_numpy(init_weights.uniform("bench.units", (n, T, 1280), 1, -1.7, 1.7))
        spk = torch.from_numpy((np.arange(n) * 37 % 323 + 1).astype(np.int64).reshape(n, 1))
        return units.to(dev), spk.to(dev)

    if args.stub_cpu:
        model = _StubUnit2Mel()
        voc = (lambda mel: mel.sum(-1, keepdim=True).transpose(1, 2).repeat(1, 1, 4)) if args.workload != "sampler" else None      # [n,1,4T]
        if args.workload == "full_tts":
            t2s = lambda ph, tn, max_length: (ph[:, :1] + torch.arange(max_length - 1)[None]) % 4096      # noqa: E731
            lookup = lambda tok: tok.to(torch.float32)[..., None].repeat(1, 1, 1280) / 4096.0                # noqa: E731
            return model, make_inputs, FullTTSPipeline(args, model, voc, t2s, lookup)
        return model, make_inputs, SamplerPipeline(args, model, make_inputs, voc)

    from diffusion.unit2mel import Unit2Mel
    with empty():
        model = Unit2Mel(1280, 323, 80).to(dev).eval()       # build-owned seeded init, seed 0 (rank 0; the others receive it)
    moved = bcast(model.state_dict())
    voc = None
    if args.workload != "sampler":
        from encoder.hifi_vaegan.hifi_vaegan import Hifi_VAEGAN
        from lds import arch
        h = arch.SYNTHETIC_VOCODER_H
        with empty():
            vstate = {k: torch.from_numpy(v).to(dev) for k, v in init_weights.init_state(arch.generator_param_shapes(h), 0).items()}
        moved += bcast(vstate)
        voc = Hifi_VAEGAN(None, device=dev, h=h, state={k: v.cpu().numpy() for k, v in vstate.items()})
    if args.workload == "full_tts":
        sys.path.insert(0, ROOT)
        import infer_tts
        from lds import native
        with empty():
            lm = infer_tts.synthetic_lm(dev)
        if share and rank != 0:
            codebook = torch.zeros(4096, 1280, device=dev)
        else:
            codebook = torch.from_numpy(init_weights.uniform("synthetic.codebook", (4096, 1280), 5, -1.7, 1.7)).to(dev)
        moved += bcast(lm.state_dict()) + bcast({"codebook": codebook})
    args.weights_broadcast_bytes = moved
    if not args.stub_cpu:      # pack this rank's copy now (host work + upload): part of set-up, not of the first timed step
        model.decoder.denoise_fn.native()
        model._native_embed()
    if args.workload == "full_tts":

        def t2s(phones, tones, max_length):
            torch.manual_seed(1234)                           # the top-k sampler's uniforms: the same draws every step
            return infer_tts.text2semantic(lm, phones, tones, 1, max_length)
        return model, make_inputs, FullTTSPipeline(args, model, voc, t2s, lambda tok: native.gather_rows(codebook, tok.clamp(max=4095)))
    return model, make_inputs, SamplerPipeline(args, model, make_inputs, voc)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        return spawn_ranks(args, argv)                       # before anything in this process touches the GPU
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:                                   # never fall back to fewer GPUs than asked for
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    if world > 1:
        # This image's driver only supports dmabuf IPC; the run environment exports HSA_ENABLE_IPC_MODE_LEGACY=0 for
        # multi-process GPU work (without it RCCL fails with `hipIpcGetMemHandle: invalid argument`).  Keep the caller's value.
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if args.stub_cpu:
        dev = torch.device("cpu")
        sync = None
    else:
        assert torch.cuda.is_available(), "bench.py needs a HIP device"
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        sync = torch.cuda.synchronize
    if world > 1 or args.rehearse_rccl:
        import torch.distributed as dist
        if world == 1:      # one-GPU rehearsal of the N > 1 branch: a single-rank RCCL group, every collective of the job still issued
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
        if args.stub_cpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    t_setup = time.perf_counter()
    model, make_inputs, pipe = build_pipeline(args, dev, rank, world, dist)
    if sync is not None:
        sync()
    setup_s = time.perf_counter() - t_setup
    if dist is not None:      # the slowest rank's set-up is what the job waits for
        smax = torch.tensor([setup_s], device=dev, dtype=torch.float64)
        dist.all_reduce(smax, op=dist.ReduceOp.MAX)
        setup_s = float(smax.item())
    profiler = None if (args.no_profile or args.stub_cpu) else NativeProfiler()
    res = run_job(args, rank, world, dev, dist=dist, sync=sync, profiler=profiler, pipeline=pipe)
    if rank == 0:
        # set-up per rank (max over ranks): weights generated on rank 0 and broadcast, packed on every rank, uploaded; not part of `value`
        res["setup_s"] = setup_s
        res["weights_broadcast_bytes"] = getattr(args, "weights_broadcast_bytes", 0)
        if world == 1 and not args.no_extras and not args.stub_cpu and args.workload == "sampler":
            res["extra"] = extra_legs(args, dev, model, make_inputs)
        if not args.no_cpu_baseline and world == 1 and not args.stub_cpu:      # reported on rank 0 at N = 1 only
            res["cpu_baseline"] = cpu_baseline(args.frames, args.cpu_nfe)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
