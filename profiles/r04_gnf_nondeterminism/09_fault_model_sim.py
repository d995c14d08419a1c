import sys, numpy as np
sys.path.insert(0,'/root/repo/latent-diffusion-speech_amd'); sys.path.insert(0,'/root/repo')
from lds import init_weights
from oracle import unet1d
U = lambda n, s, lo=-1.0, hi=1.0: init_weights.uniform(n, s, 7, lo, hi)
Cm, T, B = 256, 2114, 1
C = 64
x = U("x", (B, C, T), -2, 2)
w1 = (U("w1", (Cm, C)) / np.float32(np.sqrt(C))).astype(np.float32)
b1 = U("b1", (Cm,), 0.5, 1.5)
mid = unet1d.conv1d(x, w1[:, :, None], b1).astype(np.float64)
g = 4
grp = mid[0, 32*g:32*g+32]           # [32, T]
print("true mean/var", grp.mean(), grp.var())
nT = (T+31)//32
# partials: index pi = kk*nT + tb, kk in 0..1 (16-ch blocks), tb in 0..nT-1
parts = []
for kk in range(2):
    for tb in range(nT):
        blk = grp[16*kk:16*kk+16, tb*32:min(T, tb*32+32)]
        parts.append((blk.size, blk.mean(), ((blk-blk.mean())**2).sum()))
P = len(parts)
def combine(ps):
    n=0.; mean=0.; m2=0.
    for (nb, mb, qb) in ps:
        if nb == 0: continue
        nn = n+nb; d = mb-mean
        mean += d*nb/nn; m2 += qb + d*d*n*nb/nn; n = nn
    return n, mean, m2/n
print("all", combine(parts))
# variant A: round r (64 partials) values replaced by round r-1 values (count kept)
for r in (1,2):
    ps = list(parts)
    for lane in range(64):
        pi = r*64+lane
        if pi < P:
            src = parts[(r-1)*64+lane]
            ps[pi] = (parts[pi][0], src[1], src[2])
    print("round", r, "uses previous round's data:", combine(ps))
# variant B: round r values zero (mean 0, m2 0) with counts kept
for r in (0,1,2):
    ps = list(parts)
    for lane in range(64):
        pi = r*64+lane
        if pi < P: ps[pi] = (parts[pi][0], 0.0, 0.0)
    print("round", r, "zero data:", combine(ps))
# variant C: a round uses the OTHER group's data (g=0's partials: wave 0's e=0 group)
grp0 = mid[0, 0:32]
parts0 = []
for kk in range(2):
    for tb in range(nT):
        blk = grp0[16*kk:16*kk+16, tb*32:min(T, tb*32+32)]
        parts0.append((blk.size, blk.mean(), ((blk-blk.mean())**2).sum()))
for r in (0,1,2):
    ps = list(parts)
    for lane in range(64):
        pi = r*64+lane
        if pi < P: ps[pi] = (parts[pi][0], parts0[pi][1], parts0[pi][2])
    print("round", r, "uses group 0's data:", combine(ps))
print("group0 true", grp0.mean(), grp0.var())

# ---- emulate the wave reduction with single-fault models ----
import itertools
def lane_acc(parts):
    L = []
    for lane in range(64):
        n=0.; mean=0.; m2=0.
        for r in range(3):
            pi = r*64+lane
            if pi < len(parts):
                nb, mb, qb = parts[pi]
                nn = n+nb; d = mb-mean
                mean += d*nb/nn; m2 += qb + d*d*n*nb/nn; n = nn
        L.append([n, mean, m2])
    return L
def chan(a, b):   # a += b
    n, mean, m2 = a; nb, mb, qb = b
    nn = n+nb
    if nn <= 0: return [n, mean, m2]
    d = mb-mean
    return [nn, mean + d*nb/nn, m2 + qb + d*d*n*nb/nn]
steps = [("shr",1),("shr",2),("shr",4),("shr",8),("bc15",0),("bc31",0)]
def src_lane(step, i):
    kind, k = step
    if kind == "shr":
        return i-k if (i % 16) >= k else None
    if kind == "bc15":
        row = i // 16
        return (row*16 - 1) if row in (1,3) else None
    if kind == "bc31":
        row = i // 16
        return 31 if row in (2,3) else None
def reduce(L, fault=None):
    L = [list(x) for x in L]
    for si, st in enumerate(steps):
        new = []
        for i in range(64):
            s = src_lane(st, i)
            b = L[s] if s is not None else [0.,0.,0.]
            if fault and fault[0] == si:
                # fault model: component comp of the fetched triple replaced for lanes in set
                kind, comp, lanes = fault[1], fault[2], fault[3]
                if i in lanes:
                    b = list(b)
                    if kind == "zero": b[comp] = 0.0
                    elif kind == "own": b[comp] = L[i][comp]
            new.append(chan(L[i], b))
        L = new
    return L[63][1], L[63][2]/L[63][0]
L = lane_acc(parts)
print("emulated", reduce(L))
targets = [(0.984811, 0.5360774), (1.1004586, 0.6409454), (0.9823091, 0.5360767), (0.9809660, 0.5360749)]
found = []
for si in range(6):
    for kind in ("zero","own"):
        for comp in (0,1,2):
            for lanes in [set(range(64)), set(range(32,64)), set(range(16,32))|set(range(48,64)), set(range(48,64)), {63}, set(range(0,32))]:
                mu, var = reduce(L, (si, kind, comp, lanes))
                for t in targets:
                    if abs(mu-t[0]) < 2e-5 and abs(var-t[1]) < 2e-5:
                        found.append((t, si, kind, comp, sorted(lanes)[:2], len(lanes)))
for f in found: print("MATCH", f)
print("done", len(found))

def reduce2(L, fault=None):
    L = [list(x) for x in L]
    hist = [L]
    for si, st in enumerate(steps):
        new = []
        for i in range(64):
            s = src_lane(st, i)
            b = list(L[s]) if s is not None else [0.,0.,0.]
            if fault and fault[0] == si and s is not None and i in fault[3] and si >= 1:
                comp = fault[2]
                b[comp] = hist[si-1][s][comp]      # the source lane's value BEFORE the previous step's update
            new.append(chan(L[i], b))
        L = new
        hist.append(L)
    return L[63][1], L[63][2]/L[63][0]
found = []
for si in range(1,6):
    for comp in (0,1,2):
        for lanes in [set(range(64)), set(range(32,64)), set(range(16,32))|set(range(48,64)), set(range(48,64)), {63}]:
            mu, var = reduce2(L, (si, 'stale', comp, lanes))
            for t in targets:
                if abs(mu-t[0]) < 2e-5 and abs(var-t[1]) < 2e-5:
                    found.append((t, si, comp, len(lanes)))
print("stale-source model matches:", found)
for si in range(1,6):
    print(si, reduce2(L, (si,'stale',1,set(range(64)))))
