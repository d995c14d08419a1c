"""Per-shape time / TFLOP/s of every vocoder launch (HIP-event profiler at shape detail): 16 utterances x 512 frames."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion-speech_amd"))
import torch  # noqa: E402

from encoder.hifi_vaegan.hifi_vaegan import Hifi_VAEGAN  # noqa: E402
from lds import arch, init_weights, native  # noqa: E402

B, T = 16, 512
h = arch.SYNTHETIC_VOCODER_H
voc = Hifi_VAEGAN(None, device="cuda", h=h, state=init_weights.init_state(arch.generator_param_shapes(h), 0))
mel = torch.from_numpy(init_weights.uniform("voc.mel", (B, T, 80), 5, -1, 1)).cuda()
voc(mel)
torch.cuda.synchronize()
native.prof_enable(2)
voc(mel)
torch.cuda.synchronize()
prof = native.prof_summary()
native.prof_enable(0)
prof.sort(key=lambda r: -r["ms"])
tot = sum(r["ms"] for r in prof)
print(f"total {tot:.1f} ms")
for r in prof:
    tf = r["flops"] / (r["ms"] * 1e-3) / 1e12 if r["flops"] else 0
    print(f"{r['name']:86s} n={r['count']:3d} {r['ms']:7.2f} ms {100 * r['ms'] / tot:5.1f}% {1e3 * r['ms'] / r['count']:8.1f} us {tf:6.1f} TF")

# wall time of whole decodes (events around 5 calls), fused residual steps on / off (csrc/voc_pair.hip, lds_debug_set_voc_pair)
for on in (1, 0, 1):
    native.check(native.lib().lds_debug_set_voc_pair(on))
    voc(mel)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        voc(mel)
    e1.record()
    torch.cuda.synchronize()
    print(f"decode, narrow-stage residual steps {'fused' if on else 'as two launches'}: {e0.elapsed_time(e1) / 5:.2f} ms")
native.check(native.lib().lds_debug_set_voc_pair(1))
