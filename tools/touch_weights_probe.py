"""Upper bound of what a weight prefetcher could give: the sum of the convolution launches' own durations (HIP events bound to the dispatch)
inside one 10-NFE sampler run, with every convolution's weights read by a small launch immediately before it (lds_debug_set_touch_weights)
and without.  The touching launches themselves are NOT counted (they would not exist in a real prefetcher, whose reads ride in an earlier
kernel).

    python tools/touch_weights_probe.py [B] [--latency]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion-speech_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from diffusion.unit2mel import Unit2Mel  # noqa: E402
from lds import init_weights, native  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("B", nargs="?", type=int, default=16)
    ap.add_argument("--latency", action="store_true")
    a = ap.parse_args()
    B, T = a.B, 512
    m = Unit2Mel(1280, 323, 80).to("cuda").eval()
    m.decoder.denoise_fn.set_latency_mode(a.latency)
    units = torch.from_numpy(init_weights.uniform("bench.units", (B, T, 1280), 1, -1.7, 1.7)).cuda()
    spk = torch.from_numpy((np.arange(B) * 37 % 323 + 1).astype(np.int64).reshape(B, 1)).cuda()

    def run():
        m(units, None, spk_id=spk, infer=True, infer_speedup=100, method="dpm-solver")
        torch.cuda.synchronize()

    for rep in range(2):
        for on in (0, 1):
            native.check(native.lib().lds_debug_set_touch_weights(on))
            run()
            native.prof_enable(1)
            run()
            prof = native.prof_summary()
            native.prof_enable(0)
            conv = sum(r["ms"] for r in prof if r["name"].startswith("conv_"))
            n = sum(r["count"] for r in prof if r["name"].startswith("conv_"))
            allk = sum(r["ms"] for r in prof)
            print(f"B = {B}{' latency mode' if a.latency else ''}: weights touched before every convolution = {on}: convolution launches {conv:.2f} ms "
                  f"({n} launches, {1e3 * conv / n:.2f} us each), all kernels {allk:.2f} ms", flush=True)
    native.check(native.lib().lds_debug_set_touch_weights(0))


if __name__ == "__main__":
    main()
