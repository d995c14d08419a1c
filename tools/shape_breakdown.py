"""Per-shape time / TFLOP/s of every conv_dma launch inside one real sampler step (HIP-event profiler, level 2 = names carry shapes)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion-speech_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from diffusion.unit2mel import Unit2Mel  # noqa: E402
from lds import init_weights, native  # noqa: E402

B, T = (int(sys.argv[1]) if len(sys.argv) > 1 else 16), 512
m = Unit2Mel(1280, 323, 80).to("cuda").eval()
units = torch.from_numpy(init_weights.uniform("bench.units", (B, T, 1280), 1, -1.7, 1.7)).cuda()
spk = torch.ones(B, 1, dtype=torch.int64, device="cuda")
m(units, None, spk_id=spk, infer=True, infer_speedup=100, method="dpm-solver")
torch.cuda.synchronize()
native.prof_enable(2)
m(units, None, spk_id=spk, infer=True, infer_speedup=100, method="dpm-solver")     # 10 NFE
torch.cuda.synchronize()
prof = native.prof_summary()
native.prof_enable(0)
prof.sort(key=lambda r: -r["ms"])
tot = sum(r["ms"] for r in prof)
print(f"total {tot:.1f} ms for 10 NFE")
for r in prof[:90]:
    tf = r["flops"] / (r["ms"] * 1e-3) / 1e12 if r["flops"] else 0
    print(f"{r['name']:78s} n={r['count']:4d} {r['ms']:8.2f} ms {100 * r['ms'] / tot:5.1f}%  {1e3 * r['ms'] / r['count']:7.1f} us/launch {tf:6.1f} TF")
