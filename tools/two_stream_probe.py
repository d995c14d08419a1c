"""Do two half-batches on two HIP streams beat one full batch on one stream?  (the fixed per-launch cost of one stream's kernel
could overlap the other's K loop).  Two module instances (own workspaces), one host thread per stream."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion-speech_amd"))
import torch  # noqa: E402

from diffusion.unit2mel import Unit2Mel  # noqa: E402
from lds import init_weights  # noqa: E402

B, T = 16, 512
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 2
MODE = sys.argv[2] if len(sys.argv) > 2 else "f32"      # f32 | split_bf16 | split_f16
units = torch.from_numpy(init_weights.uniform("bench.units", (B, T, 1280), 1, -1.7, 1.7)).cuda()
spk = torch.ones(B, 1, dtype=torch.int64, device="cuda")
mods = [Unit2Mel(1280, 323, 80).to("cuda").eval() for _ in range(NS)]
for m_ in mods:
    m_.decoder.denoise_fn.set_gemm_mode(MODE)
print("GEMM mode", MODE, flush=True)
streams = [torch.cuda.Stream() for _ in range(NS)]
h = B // NS


def call_full():
    return mods[0](units, None, spk_id=spk, infer=True, infer_speedup=20, method="dpm-solver")


def call_part(i):
    with torch.cuda.stream(streams[i]):
        return mods[i](units[i * h:(i + 1) * h], None, spk_id=spk[i * h:(i + 1) * h], infer=True, infer_speedup=20, method="dpm-solver")


def run_parts():
    th = [threading.Thread(target=call_part, args=(i,)) for i in range(NS)]
    for t in th:
        t.start()
    for t in th:
        t.join()


for _ in range(2):
    call_full()
    run_parts()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    call_full()
torch.cuda.synchronize()
print(f"one stream, B=16      : {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms / call", flush=True)
t0 = time.perf_counter()
for _ in range(3):
    run_parts()
torch.cuda.synchronize()
print(f"{NS} streams, B={h} each : {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms / call", flush=True)
t0 = time.perf_counter()
for _ in range(3):
    call_part(0)
torch.cuda.synchronize()
print(f"one stream, B={h}       : {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms / call", flush=True)
