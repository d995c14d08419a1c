"""Headline benchmark: mel-frames/s of the 80-channel, 50-step DPM-Solver++ sampler (BASELINE.json
configs[1]: batch 16 x 512-frame utterances per GPU, unet1d denoiser, seeded random-init weights,
synthetic inputs).  One "step" = one full pass of the hot path over one batch: condition
embedding -> 50-NFE DPM-Solver++(2M) sampling -> mel [B,512,80].

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: the utterance batch shards across ranks (weak scaling: 16 utterances per GPU); the
only collectives are the RCCL scatter of the inputs before the timed region and the gather of the
mels inside it.  Prints ONE JSON line on rank 0.  At N = 1 the same line also carries `extra` legs
(untimed for `value`): configs[3] end-to-end with the vocoder, the per-GPU share of configs[2]
(UniPC, 20 NFE) and the B = 1 latency of the caller north_star names (22_infer_tts.py).
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
    return {"value": T / (per_nfe * 50), "unit": "mel-frames/s", "cores": int(cores), "kind": "port",
            "sample": f"numpy oracle, 1 utterance x {T} frames, {n_eval}-NFE DPM-Solver++ run ({dt:.1f} s) scaled to 50 NFE",
            "reference_torch_cpu_8core_survey": 29.0,
            "note": "the reference's own torch-CPU path measured during the survey (8 cores, BASELINE.md 4) is ~29 mel-frames/s; "
                    "this numpy port is ~7x slower than that and is a reported baseline only"}


def _newest_profile_json(suffix):
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*" + suffix)))
    if not files:
        return None
    try:
        return json.load(open(files[-1]))
    except Exception:
        return None


def _pmc_key(kernel_name):
    import re
    m = re.match(r"(\w+)<BM(\d+) BN(\d+) KT(\d+) S(\d+) U(\d+) BK(\d+)(?: NST(\d+))?>", kernel_name)
    if not m:
        return None
    fam, bm, bn, kt, st, up, bk, nst = m.groups()
    # rocprofv3 prints every template argument: <BM, BN, KT, STRIDE, UPS, BK, NST, DIL, VOC> (the UNet's kernels have DIL 1, VOC false)
    return f"{fam}<{bm}, {bn}, {kt}, {st}, {'true' if up == '1' else 'false'}, {bk}, {nst}, 1, false>" if nst else None


def pmc_traffic(kernel_name):
    """HBM bytes per launch of `kernel_name` from the newest committed PMC summary (profiles/*_hbm_traffic.json, produced by
    tools/summarize_pmc.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes); None if there is none."""
    d, key = _newest_profile_json("_hbm_traffic.json"), _pmc_key(kernel_name)
    try:
        return d["kernels"][key]["hbm_bytes_per_launch"] if d and key in d["kernels"] else None
    except Exception:
        return None


def pmc_mfma_util(kernel_name):
    """MFMA-busy fraction of `kernel_name` from the newest committed counter summary (profiles/*_mfma_util.json: separate
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES pass); None if there is none."""
    d, key = _newest_profile_json("_mfma_util.json"), _pmc_key(kernel_name)
    try:
        return d["kernels"][key] if d and key in d["kernels"] else None
    except Exception:
        return None


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
        "frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": pmc_traffic(dom["name"]),
        "algorithmic_bytes_per_launch": dom["bytes"] / dom["count"],
        "launches": dom["count"], "avg_launch_us": 1e3 * dom["ms"] / dom["count"],
        "gflop_per_launch": dom["flops"] / dom["count"] / 1e9,
        "share_of_step_time": dom["ms"] / tot_ms,
        "all_conv_gemm_tflops": conv_tf, "all_conv_gemm_frac": conv_tf / PEAK_F32_MFMA_TFLOPS,
        "all_conv_gemm_share": conv_ms / tot_ms,
        "mfma_util_pmc": pmc_mfma_util(dom["name"]),
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


def run_job(args, rank, world, dev, model, voc=None, dist=None, sync=None, profiler=None, make_inputs=None, on_output=None):
    """The bench proper, parameterised so tests can drive the multi-rank plumbing on CPU (gloo) with a stub model.
    Returns the result dict on rank 0 (None elsewhere)."""
    from lds import shard
    B, T = args.batch, args.frames
    speedup = 1000 // args.nfe
    sync = sync or (lambda: None)

    # synthetic inputs for the global batch live on rank 0 and are scattered over RCCL (untimed set-up)
    if rank == 0:
        units_all, spk_all = make_inputs(world * B, T)
    else:
        units_all = spk_all = torch.empty(0, device=dev)
    force = dist is not None      # (--rehearse-rccl: the collectives run even at N = 1)
    units = shard.scatter_batch(units_all, rank, world, shape=(world * B, T, 1280), dtype=torch.float32, device=dev, force=force)
    spk = shard.scatter_batch(spk_all, rank, world, shape=(world * B, 1), dtype=torch.int64, device=dev, force=force)
    del units_all

    def step(gather=True):
        mel = model(units, None, spk_id=spk, infer=True, infer_speedup=speedup, method=args.method)
        wav = voc(mel) if voc is not None else None
        out = shard.gather_batch(mel, rank, world, sizes=[B] * world, force=force) if gather else mel
        return out, wav

    dt, (out, wav) = timed_loop(step, args.steps, args.warmup, world, sync, dist, dev)
    if out is not None:
        assert bool(torch.isfinite(out).all()), "non-finite mel"
        assert tuple(out.shape) == (world * B, T, 80), tuple(out.shape)
        if on_output is not None:
            on_output(out)
    if wav is not None:
        assert bool(torch.isfinite(wav).all()), "non-finite waveform"

    frames = world * B * T * args.steps
    value = frames / dt
    res = {
        "metric": "mel_frames_per_sec", "value": value, "unit": "mel-frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"configs[1]: batch={B}x{T}-frame utterances per GPU, {args.nfe}-step {args.method}, unet1d denoiser, 80-ch mel"
                               + (" + HiFi-VAEGAN vocoder (configs[3])" if voc is not None else ""),
                   "utterances_per_gpu": B, "frames": T, "nfe": args.nfe, "method": args.method, "parallelism": f"batch-shard x{world}"},
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


def extra_legs(args, dev, model, make_inputs):
    """configs[3], configs[2]'s per-GPU share and the B = 1 latency (N = 1 only; none of them feeds `value`)."""
    from encoder.hifi_vaegan.hifi_vaegan import Hifi_VAEGAN
    from lds import arch, init_weights
    T = args.frames
    sync = torch.cuda.synchronize
    extra = {}
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
    dt = timeit(one, 3)
    prof = NativeProfiler().run(one)
    conv_tf, _ = conv_family_rate(prof)
    extra["b1_latency"] = {
        "workload": f"1 utterance x {T} frames, {args.nfe}-step {args.method} (latency of the 22_infer_tts.py caller)",
        "ms_per_utterance": 1e3 * dt, "mel_frames_per_sec": T / dt, "x_realtime": T / dt * FRAME_SEC,
        "unet_algorithmic_tflops": UNET_GFLOP_PER_UTT_FWD * (T / 512.0) * args.nfe * 1e9 / dt / 1e12,
        "all_conv_gemm_tflops": conv_tf, "all_conv_gemm_frac": conv_tf / PEAK_F32_MFMA_TFLOPS,
        "launches_per_step": int(sum(r["count"] for r in prof)),
    }
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
        keep["tok"] = infer_tts.text2semantic(lm, phones, tones, 1, T + 1)      # random weights never emit EOS: exactly T tokens

    def full():
        lm_only()
        units = native.gather_rows(codebook, keep["tok"].clamp(max=4095))
        mel = model(units, None, spk_id=spk[:Bq], infer=True, infer_speedup=1000 // args.nfe, method=args.method)
        keep["wav"] = voc(mel)
    dt_lm = timeit(lm_only, 2)
    dt = timeit(full, 2)
    assert tuple(keep["tok"].shape) == (Bq, T) and bool(torch.isfinite(keep["wav"]).all())
    extra["configs5_full_tts_per_gpu"] = {
        "workload": f"configs[5] per-GPU share: {Bq} utterances, {Lp} phones -> RoFormer top-k sampling of {T} semantic tokens -> {args.nfe}-step "
                    f"{args.method} -> HiFi-VAEGAN ({T * 512} samples/utt)",
        "ms_per_step": 1e3 * dt, "x_realtime": Bq * T * FRAME_SEC / dt, "rtf": dt / (Bq * T * FRAME_SEC),
        "lm_ms": 1e3 * dt_lm, "lm_tokens_per_sec": Bq * T / dt_lm, "lm_us_per_decode_step": 1e6 * dt_lm / T,
    }
    # ---- the 22_infer_tts.py caller itself: ONE utterance, phones -> tokens -> units -> mel -> wav ----
    p1, t1 = phones[:1].contiguous(), tones[:1].contiguous()

    def lm_one():
        keep["tok1"] = infer_tts.text2semantic(lm, p1, t1, 1, T + 1)

    def full_one():
        lm_one()
        units = native.gather_rows(codebook, keep["tok1"].clamp(max=4095))
        mel = model(units, None, spk_id=spk[:1], infer=True, infer_speedup=1000 // args.nfe, method=args.method)
        keep["wav1"] = voc(mel)
    dt_lm1 = timeit(lm_one, 2)
    dt1 = timeit(full_one, 2)
    assert tuple(keep["tok1"].shape) == (1, T) and bool(torch.isfinite(keep["wav1"]).all())
    extra["b1_full_tts_latency"] = {
        "workload": f"1 utterance: {Lp} phones -> {T} sampled tokens -> {args.nfe}-step {args.method} -> {T * 512} samples (22_infer_tts.py, one sentence)",
        "ms_per_utterance": 1e3 * dt1, "audio_seconds": T * FRAME_SEC, "x_realtime": T * FRAME_SEC / dt1, "rtf": dt1 / (T * FRAME_SEC),
        "lm_ms": 1e3 * dt_lm1, "lm_us_per_decode_step": 1e6 * dt_lm1 / T,
    }
    return extra


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=16, help="utterances per GPU")
    ap.add_argument("--frames", type=int, default=512)
    ap.add_argument("--nfe", type=int, default=50)
    ap.add_argument("--method", default="dpm-solver")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-nfe", type=int, default=8)
    ap.add_argument("--vocoder", action="store_true", help="run the HiFi-VAEGAN decode inside the timed step (BASELINE config 4)")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra legs (configs[3], UniPC-20, B=1 latency) at N=1")
    ap.add_argument("--rehearse-rccl", action="store_true",
                    help="N=1 only: open a single-rank RCCL process group and issue every collective of the N>1 path (scatter, gather, barrier, all-reduce)")
    ap.add_argument("--no-profile", action="store_true", help="skip the instrumented roofline step (for rocprofv3 --pmc passes)")
    return ap.parse_args(argv)


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world != 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    if world > 1:
        # This image's driver only supports dmabuf IPC; the run environment exports HSA_ENABLE_IPC_MODE_LEGACY=0 for
        # multi-process GPU work (without it RCCL fails with `hipIpcGetMemHandle: invalid argument`).  Keep the caller's value.
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 or args.rehearse_rccl:
        import torch.distributed as dist
        if world == 1:      # one-GPU rehearsal of the N > 1 branch: a single-rank RCCL group, every collective of the job still issued
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from diffusion.unit2mel import Unit2Mel
    from lds import init_weights

    model = Unit2Mel(1280, 323, 80).to(dev).eval()           # build-owned seeded init, seed 0

    def make_inputs(n, T):
        units = torch.from_numpy(init_weights.uniform("bench.units", (n, T, 1280), 1, -1.7, 1.7)).to(dev)
        spk = torch.from_numpy((np.arange(n) * 37 % 323 + 1).astype(np.int64).reshape(n, 1)).to(dev)
        return units, spk

    voc = None
    if args.vocoder:
        from encoder.hifi_vaegan.hifi_vaegan import Hifi_VAEGAN
        from lds import arch
        h = arch.SYNTHETIC_VOCODER_H
        voc = Hifi_VAEGAN(None, device=dev, h=h, state=init_weights.init_state(arch.generator_param_shapes(h), 0))

    res = run_job(args, rank, world, dev, model, voc=voc, dist=dist, sync=torch.cuda.synchronize,
                  profiler=None if args.no_profile else NativeProfiler(), make_inputs=make_inputs)
    if rank == 0:
        if world == 1 and not args.no_extras:
            res["extra"] = extra_legs(args, dev, model, make_inputs)
        if not args.no_cpu_baseline and world == 1:      # reported on rank 0 at N = 1 only
            res["cpu_baseline"] = cpu_baseline(args.frames, args.cpu_nfe)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
