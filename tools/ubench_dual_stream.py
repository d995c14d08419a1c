"""Does the sampler gain from running two halves of the batch on two HIP streams?  (Every launch has ~7 us of ramp-up and tail that a
second queue's kernels could fill: DESIGN.md section 12.)  Times one B-utterance call against k concurrent calls of B/k utterances,
each on its own stream from its own host thread (the library's handles are immutable; the workspace is keyed by stream).

    python tools/ubench_dual_stream.py --B 16 --T 512 --splits 2,4

Reference call being timed: /root/reference/diffusion/unit2mel.py:124-140 (Unit2Mel.forward, infer=True)."""
import argparse
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "latent-diffusion-speech_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=16)
    ap.add_argument("--T", type=int, default=512)
    ap.add_argument("--nfe", type=int, default=50)
    ap.add_argument("--splits", default="2,4")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--mode", default="f32")
    a = ap.parse_args()
    import torch
    from diffusion.unit2mel import Unit2Mel
    from lds import init_weights
    dev = torch.device("cuda:0")
    m = Unit2Mel(1280, 323, 80).to(dev).eval()
    m.decoder.denoise_fn.set_gemm_mode(a.mode)
    B, T = a.B, a.T
    units = torch.from_numpy(init_weights.uniform("bench.units", (B, T, 1280), 1, -1.7, 1.7)).to(dev)
    spk = torch.from_numpy((np.arange(B) * 37 % 323 + 1).astype(np.int64).reshape(B, 1)).to(dev)
    x_T = torch.from_numpy(init_weights.uniform("bench.xT", (B, 1, 80, T), 2, -2, 2)).to(dev)

    def call(lo, hi):
        return m(units[lo:hi], None, spk_id=spk[lo:hi], infer=True, infer_speedup=1000 // a.nfe, method="dpm-solver++", x_T=x_T[lo:hi])

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(a.reps):
            t0 = time.perf_counter()
            out = fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        return out, min(ts), ts

    ref, t1, all1 = timed(lambda: call(0, B))
    print(f"one call, B = {B}: {t1:.2f} ms  {['%.1f' % v for v in all1]}", flush=True)
    for k in [int(v) for v in a.splits.split(",")]:
        streams = [torch.cuda.Stream(dev) for _ in range(k)]
        cuts = [B * i // k for i in range(k + 1)]
        outs = [None] * k

        def worker(i):
            with torch.cuda.stream(streams[i]):
                outs[i] = call(cuts[i], cuts[i + 1])

        def run():
            for s in streams:
                s.wait_stream(torch.cuda.current_stream(dev))
            th = [threading.Thread(target=worker, args=(i,)) for i in range(k)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            for s in streams:
                torch.cuda.current_stream(dev).wait_stream(s)
            return torch.cat(outs, 0)

        out, tk, allk = timed(run)
        print(f"{k} concurrent calls of B = {B // k} on {k} streams: {tk:.2f} ms  {['%.1f' % v for v in allk]}  "
              f"bit-equal to the one call: {bool(torch.equal(out, ref))}  max rel {float((out - ref).abs().max() / ref.abs().max()):.2e}", flush=True)


if __name__ == "__main__":
    main()
