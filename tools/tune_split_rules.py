"""In-model tuning of the split-GEMM tile rules: total device time of a 10-NFE sampler run (B = 16 x 512 frames) per rule mask
(conv_bf3.hip bf3_pick, lds_debug_set_split_rule), for one GEMM mode:
    python tools/tune_split_rules.py split_f16 0,1,2,4,8,16"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion-speech_amd"))
import torch  # noqa: E402

from diffusion.unit2mel import Unit2Mel  # noqa: E402
from lds import init_weights, native  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "split_f16"
rules = [int(t) for t in (sys.argv[2] if len(sys.argv) > 2 else "0,1,2,4,8,16").split(",")]
B, T = 16, 512
m = Unit2Mel(1280, 323, 80).to("cuda").eval()
m.decoder.denoise_fn.set_gemm_mode(mode)
units = torch.from_numpy(init_weights.uniform("bench.units", (B, T, 1280), 1, -1.7, 1.7)).cuda()
spk = torch.ones(B, 1, dtype=torch.int64, device="cuda")


def run():
    return m(units, None, spk_id=spk, infer=True, infer_speedup=100, method="dpm-solver")


for rnd in range(2):
    for r in rules:
        native.lib().lds_debug_set_split_rule(r)
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            run()
        e1.record(); torch.cuda.synchronize()
        print(f"round {rnd} rule {r:3d}: {e0.elapsed_time(e1) / 3:8.2f} ms per 10-NFE run", flush=True)
native.lib().lds_debug_set_split_rule(0)
