"""Determinism of the GroupNorm-fold proj_in launch alone (include/lds_test.h lds_test_gn_fold_split): `reps` launches on the same inputs, every
repetition compared with the first; prints which (channel block, frame block) tiles differ.

    python tools/diag_gnf_repro.py --Cm 256 --T 2050 --B 1 --tile-batch 1 --fmt 1 --reps 200"""
import argparse
import ctypes as ct
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "latent-diffusion-speech_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--Cm", type=int, default=256)
    ap.add_argument("--T", type=int, default=2050)
    ap.add_argument("--B", type=int, default=1)
    ap.add_argument("--tile-batch", type=int, default=1)
    ap.add_argument("--fmt", type=int, default=1)
    ap.add_argument("--reps", type=int, default=100)
    ap.add_argument("--cfg", type=int, default=0)
    ap.add_argument("--calls", type=int, default=3)
    ap.add_argument("--rule", type=int, default=0)
    a = ap.parse_args()
    import torch
    from lds import init_weights, native
    U = lambda n, s, lo=-1.0, hi=1.0: init_weights.uniform(n, s, 7, lo, hi)
    Cm, T, B = a.Cm, a.T, a.B
    C, Co = 64, Cm
    x = U("x", (B, C, T), -2, 2)
    w1 = (U("w1", (Cm, C)) / np.float32(np.sqrt(C))).astype(np.float32)
    b1 = U("b1", (Cm,), 0.5, 1.5)
    g, be = U("g", (Cm,), 0.5, 1.5), U("b", (Cm,), -0.5, 0.5)
    w2 = (U("w2", (Co, Cm)) / np.float32(np.sqrt(Cm))).astype(np.float32)
    b2 = U("b2", (Co,), -0.5, 0.5)
    dx = torch.from_numpy(x).cuda()
    P = lambda v: ct.c_void_p(v.ctypes.data)
    st = ct.c_void_p(torch.cuda.current_stream().cuda_stream)
    total_bad = 0
    native.check(native.lib().lds_debug_set_split_rule(a.rule))
    for call in range(a.calls):
        mid = torch.zeros((B, Cm, T), dtype=torch.float32, device="cuda")
        out = torch.zeros((a.reps, B, Co, T), dtype=torch.float32, device="cuda")
        native.check(native.lib().lds_test_gn_fold_split(ct.c_void_p(dx.data_ptr()), P(w1), P(b1), P(g), P(be), ct.c_float(1e-6), 8, P(w2), P(b2),
                                                         ct.c_void_p(mid.data_ptr()), ct.c_void_p(out.data_ptr()), B, C, Cm, Co, T, a.cfg, a.tile_batch, a.fmt,
                                                         a.reps, st))
        torch.cuda.synchronize()
        o = out.cpu().numpy()
        bad = 0
        for r in range(1, a.reps):
            if not np.array_equal(o[r], o[0]):
                bad += 1
                d = np.argwhere(o[r] != o[0])
                bs = sorted({(int(b_), int(c_) // 32, int(t_) // 64) for b_, c_, t_ in d})
                mx = float(np.abs(o[r] - o[0]).max() / np.abs(o[0]).max())
                if bad <= 12:
                    print(f"call {call} rep {r}: {len(d)} elements differ, rel {mx:.2e}, (b, channel/32, frame/64) tiles {bs[:8]}")
        print(f"call {call}: {bad} of {a.reps - 1} repetitions differ from the first", flush=True)
        if (a.rule >> 8) & 64:      # the tail as every wave of every workgroup read it (last repetition)
            nwg = 8 * ((T + 63) // 64) * B
            d2 = np.zeros((1024, 2, 4, 32), dtype=np.float32)
            native.lib().lds_debug_read_gnf2(ct.c_void_p(d2.ctypes.data), d2.size)
            d2 = d2[:nwg]
            print("  last repetition differs from the first:", not np.array_equal(o[a.reps - 1], o[0]))
            for bb in range(B):
                sel = d2[d2[:, 0, 0, 24] == bb]
                ref = sel[0, 0, 0, :24]
                for i in range(len(sel)):
                    for when in range(2):
                        for w in range(4):
                            if not np.array_equal(sel[i, when, w, :24], ref):
                                dd = np.nonzero(sel[i, when, w, :24] != ref)[0]
                                print(f"    b {bb} wg#{i} when {when} wave {w}: slots {dd.tolist()} read {sel[i, when, w][dd].tolist()} expected {ref[dd].tolist()}")
        if (a.rule >> 8) & 128:      # group 4: per-lane (n, mean, m2, popcount(exec)) before and after the wave reduction, workgroups 248 .. 311
            nwg = 8 * ((T + 63) // 64) * B
            d3 = np.zeros((64, 8, 64), dtype=np.float32)
            native.lib().lds_debug_read_gnf2(ct.c_void_p(d3.ctypes.data), d3.size)
            ref = d3[0]
            names = ("n", "mean", "m2", "exec", "n'", "mean'", "m2'", "exec'")
            for i in range(min(64, nwg - 248)):
                if not np.array_equal(d3[i], ref):
                    for k, nm in enumerate(names):
                        dd = np.nonzero(d3[i, k] != ref[k])[0]
                        if len(dd):
                            print(f"    lanes wg {248 + i} {nm}: lanes {dd.tolist()[:40]} got {d3[i, k][dd][:4].tolist()} ref {ref[k][dd][:4].tolist()}")
        if (a.rule >> 8) & 32:      # the last repetition's per-workgroup statistics: (mu, var, lane 0's first partial mean) x 8 groups
            nwg = 8 * ((T + 63) // 64) * B
            dbg = np.zeros((2048, 24), dtype=np.float32)
            native.lib().lds_debug_read_gnf(ct.c_void_p(dbg.ctypes.data), 2048 * 24)
            dbg = dbg[:nwg]
            ref = dbg[0]
            for i in range(nwg):
                if not np.array_equal(dbg[i], ref):
                    d = np.nonzero(dbg[i] != ref)[0]
                    print("    stats wg", i, "slots", d.tolist(), "values", dbg[i][d].tolist(), "ref", ref[d].tolist())
    print("TOTAL differing repetitions:", total_bad)


if __name__ == "__main__":
    main()
