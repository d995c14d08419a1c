"""How long does the HOST take to enqueue one sampler run (50-step DPM-Solver++), against the run's wall time?  If the two are close the
run is bound by the host's launch rate, not by the GPU.

    python tools/host_enqueue_time.py [B] [--latency] [--mode f32|split_f16]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion-speech_amd"))
import torch  # noqa: E402

from diffusion.unit2mel import Unit2Mel  # noqa: E402
from lds import init_weights  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("B", nargs="?", type=int, default=1)
ap.add_argument("--latency", action="store_true")
ap.add_argument("--mode", default="f32")
a = ap.parse_args()
T = 512
m = Unit2Mel(1280, 323, 80).to("cuda").eval()
m.decoder.denoise_fn.set_gemm_mode(a.mode)
m.decoder.denoise_fn.set_latency_mode(a.latency)
units = torch.from_numpy(init_weights.uniform("bench.units", (a.B, T, 1280), 1, -1.7, 1.7)).cuda()
spk = torch.ones(a.B, 1, dtype=torch.int64, device="cuda")
for _ in range(2):
    m(units, None, spk_id=spk, infer=True, infer_speedup=20, method="dpm-solver")
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    m(units, None, spk_id=spk, infer=True, infer_speedup=20, method="dpm-solver")
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"B={a.B} mode={a.mode} latency={int(a.latency)}: host enqueue {1e3 * (t1 - t0):7.1f} ms, wall {1e3 * (t2 - t0):7.1f} ms", flush=True)
