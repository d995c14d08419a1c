"""Decode-step time of the text2semantic RoFormer (synthetic weights): 512 sampled tokens, batch 1 and 8."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion-speech_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import infer_tts  # noqa: E402

lm = infer_tts.synthetic_lm("cuda")
for B in (1, 8):
    ph = torch.from_numpy((np.arange(B * 64).reshape(B, 64) * 7 % 107 + 1).astype(np.int64)).cuda()
    tn = torch.from_numpy((np.arange(B * 64).reshape(B, 64) * 5 % 12).astype(np.int64)).cuda()
    infer_tts.text2semantic(lm, ph, tn, 1, 65)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tok = infer_tts.text2semantic(lm, ph, tn, 1, 513)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"B {B}: {tuple(tok.shape)} {dt * 1e3:.1f} ms, {dt * 1e6 / 512:.1f} us/step, {B * 512 / dt:.0f} tokens/s")
