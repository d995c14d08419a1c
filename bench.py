"""Headline benchmark: mel-frames/s of the 80-channel, 50-step DPM-Solver++ sampler (BASELINE.json
configs[1]: batch 16 x 512-frame utterances per GPU, unet1d denoiser, seeded random-init weights,
synthetic inputs).  One "step" = one full pass of the hot path over one batch: condition
embedding -> 50-NFE DPM-Solver++(2M) sampling -> mel [B,512,80].

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: the utterance batch shards across ranks (weak scaling: 16 utterances per GPU); the
only collectives are the RCCL scatter of the inputs before the timed region and the all-gather
of the mels inside it.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion-speech_amd"))
sys.path.insert(1, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X dense fp32 matrix peak (MI355X_MICROARCH.md, chip-level parameters)
FRAME_SEC = 512 / 44100.0        # one mel frame = hop 512 @ 44.1 kHz (reference configs/config.yaml:3,12)
UNET_GFLOP_PER_UTT_FWD = 40.98   # SURVEY.md 8d (T=512, M=80), algorithmic


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
            "sample": f"numpy oracle, 1 utterance x {T} frames, {n_eval}-NFE DPM-Solver++ run ({dt:.1f} s) scaled to 50 NFE"}


def pmc_traffic(kernel_name):
    """HBM bytes per launch of `kernel_name` from the newest committed PMC summary (profiles/*_hbm_traffic.json, produced by
    tools/summarize_pmc.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes); None if there is none."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json")))
    if not files:
        return None
    m = re.match(r"(\w+)<BM(\d+) BN(\d+) KT(\d+) S(\d+) U(\d+) BK(\d+)(?: NST(\d+))?>", kernel_name)
    if not m:
        return None
    fam, bm, bn, kt, st, up, bk, nst = m.groups()
    key = f"{fam}<{bm}, {bn}, {kt}, {st}, {'true' if up == '1' else 'false'}, {bk}, {nst}>" if nst else None
    try:
        ks = json.load(open(files[-1]))["kernels"]
        return ks[key]["hbm_bytes_per_launch"] if key in ks else None
    except Exception:
        return None


def main():
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
    ap.add_argument("--vocoder", action="store_true", help="also run the HiFi-VAEGAN decode (BASELINE config 4) and report RTF")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world != 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from diffusion.unit2mel import Unit2Mel
    from lds import init_weights, native, shard

    B, T = args.batch, args.frames
    model = Unit2Mel(1280, 323, 80).to(dev).eval()           # build-owned seeded init, seed 0
    speedup = 1000 // args.nfe

    # synthetic inputs for the global batch live on rank 0 and are scattered over RCCL (untimed set-up)
    if rank == 0:
        units_all = torch.from_numpy(init_weights.uniform("bench.units", (world * B, T, 1280), 1, -1.7, 1.7)).to(dev)
        spk_all = torch.from_numpy((np.arange(world * B) * 37 % 323 + 1).astype(np.float32)).to(dev)
    else:
        units_all, spk_all = torch.empty(0, device=dev), torch.empty(0, device=dev)
    units = shard.scatter_batch(units_all, rank, world, shape=(world * B, T, 1280))
    spk = shard.scatter_batch(spk_all.reshape(-1, 1), rank, world, shape=(world * B, 1)).to(torch.int64)
    del units_all
    gen = torch.Generator(device=dev)
    gen.manual_seed(2 + rank)

    voc = None
    if args.vocoder:
        from encoder.hifi_vaegan.hifi_vaegan import Hifi_VAEGAN
        from lds import arch
        h = arch.SYNTHETIC_VOCODER_H
        voc = Hifi_VAEGAN(None, device=dev, h=h, state=init_weights.init_state(arch.generator_param_shapes(h), 0))

    def step():
        mel = model(units, None, spk_id=spk, infer=True, infer_speedup=speedup, method=args.method)
        wav = voc(mel) if voc is not None else None
        out = shard.gather_batch(mel, rank, world, sizes=[B] * world)
        return out, wav

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, wav = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    assert out is None or bool(torch.isfinite(out).all())

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

    if rank == 0:
        # ---- roofline leg: one extra instrumented step, HIP events around every launch on the launch stream ----
        native.prof_enable(True)
        step()
        torch.cuda.synchronize()
        prof = native.prof_summary()
        native.prof_enable(False)
        prof.sort(key=lambda r: -r["ms"])
        tot_ms = sum(r["ms"] for r in prof)
        dom = prof[0]
        ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        conv = [r for r in prof if r["name"].startswith("conv_")]
        conv_tf = sum(r["flops"] for r in conv) / (sum(r["ms"] for r in conv) * 1e-3) / 1e12
        res["roofline"] = {
            "bound": "mfma", "kernel": dom["name"], "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": pmc_traffic(dom["name"]),
            "algorithmic_bytes_per_launch": dom["bytes"] / dom["count"],
            "launches": dom["count"], "avg_launch_us": 1e3 * dom["ms"] / dom["count"],
            "gflop_per_launch": dom["flops"] / dom["count"] / 1e9,
            "share_of_step_time": dom["ms"] / tot_ms,
            "all_conv_gemm_tflops": conv_tf, "all_conv_gemm_frac": conv_tf / PEAK_F32_MFMA_TFLOPS,
            "all_conv_gemm_share": sum(r["ms"] for r in conv) / tot_ms,
        }
        res["kernel_breakdown_ms"] = {r["name"]: round(r["ms"], 3) for r in prof[:12]}
        res["unet_algorithmic_tflops"] = UNET_GFLOP_PER_UTT_FWD * (T / 512.0) * B * args.nfe * 1e9 / (dt / args.steps) / 1e12 if T == 512 else None
        if not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(T, args.cpu_nfe)
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
