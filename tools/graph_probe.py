"""Does replaying one sampler call from a HIP graph beat stream launches?  (torch.cuda.CUDAGraph around the module call)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion-speech_amd"))
import torch  # noqa: E402

from diffusion.unit2mel import Unit2Mel  # noqa: E402
from lds import init_weights  # noqa: E402

B, T = (int(sys.argv[1]) if len(sys.argv) > 1 else 16), 512
m = Unit2Mel(1280, 323, 80).to("cuda").eval()
m.decoder.denoise_fn.set_gemm_mode(os.environ.get("MODE", "f32"))          # MODE=split_f16 LAT=1: the GEMM / latency modes
m.decoder.denoise_fn.set_latency_mode(os.environ.get("LAT", "0") == "1")
units = torch.from_numpy(init_weights.uniform("bench.units", (B, T, 1280), 1, -1.7, 1.7)).cuda()
spk = torch.ones(B, 1, dtype=torch.int64, device="cuda")


def call():
    return m(units, None, spk_id=spk, infer=True, infer_speedup=20, method="dpm-solver")


for _ in range(2):
    call()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    call()
torch.cuda.synchronize()
print(f"stream: {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms / call", flush=True)

s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    call()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
t0 = time.perf_counter()
with torch.cuda.graph(g):
    out = call()
torch.cuda.synchronize()
print(f"capture+instantiate: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
print(f"graph : {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms / call", flush=True)
