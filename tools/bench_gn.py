"""Micro-benchmark of the streaming GroupNorm (gn_stream) on the UNet's shapes at B=16: back-to-back launches, HIP events."""
import ctypes as ct
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion-speech_amd"))
import torch  # noqa: E402

from lds import native  # noqa: E402

SHAPES = [(256, 0, 512), (256, 256, 512), (384, 256, 512), (384, 0, 256), (384, 384, 256), (512, 384, 256), (512, 0, 128), (512, 512, 128),
          (512, 0, 64), (512, 512, 64)]
B = 16
L = native.lib()
tot = 0.0
for C1, C2, T in SHAPES:
    ms = ct.c_float()
    native.check(L.lds_bench_gn_stream(C1, C2, T, B, 200, ct.byref(ms), ct.c_void_p(torch.cuda.current_stream().cuda_stream)))
    by = 8.0 * B * (C1 + C2) * T
    print(f"C {C1:4d}+{C2:<4d} T {T:4d}: {ms.value * 1e3:7.2f} us  {by / (ms.value * 1e-3) / 1e12:5.2f} TB/s ({by / 1e6:6.1f} MB)")
    tot += ms.value
print(f"sum {tot * 1e3:.1f} us")
